"""CPU oracle: the two "next" modules (sponge_layer, ColumnNudger) and the complete supercell_example loop.
The 800-step check ties Kessler + sponge + nudging to the only quantities the survey recorded for them
(value ranges of the default 2-D run at step 800) -- a consistency check, not a bitwise pin."""
import numpy as np

RANGES_STEP_800 = {   # SURVEY.md 8(d) "Value ranges of live fields ... [probed, 2-D run step 800]"
    "rho_d": (0.094, 1.11), "u": (-19.0, 17.0), "w": (-2.3, 3.5), "T": (212.7, 298.5), "rho_v_max": 1.56e-2,
    "rho_c_max": 1.134e-3, "rho_r_max": 5.59e-6,
}


def loop_setup(oracle, nx, ny, nz):
    p, _ = oracle.make_params(nx, ny, nz, 1, 1.0e5, 1.0e5, 2.0e4)
    dyc = oracle.OracleDycore(p)
    f = oracle.Fields(dyc.p)
    dyc.init("supercell", f)
    nud = oracle.ColumnNudger()
    nud.set_column(dyc.p, f)
    oracle.perturb_temperature(dyc.p, f.temp)
    return dyc, f, nud


def test_nudging_unperturbed_state_is_identity(oracle):
    p, _ = oracle.make_params(12, 10, 12, 1, 6000., 5000., 2.0e4)
    dyc = oracle.OracleDycore(p)
    f = oracle.Fields(dyc.p)
    dyc.init("supercell", f)
    nud = oracle.ColumnNudger()
    nud.set_column(dyc.p, f)
    g = f.copy()
    nud.nudge_to_column(dyc.p, f, 10.0)
    for k, a in f.as_dict().items():
        assert np.max(np.abs(a - g.as_dict()[k])) <= 1e-15 * max(1.0, np.max(np.abs(a))), k


def test_sponge_preserves_horizontal_means_and_damps_w(oracle):
    dyc, f = oracle.supercell_setup(16, 12, 24, 1, 8000., 6000., 2.0e4)
    rng = np.random.default_rng(0)
    f.wvel += rng.normal(size=f.wvel.shape)
    f.temp += rng.normal(size=f.temp.shape)
    m0 = f.temp.mean(axis=(1, 2, 3)).copy()
    w0 = np.abs(f.wvel).max(axis=(1, 2, 3)).copy()
    t14 = f.temp[:14].copy()
    oracle.sponge_layer(dyc.p, f, 5.0, 60.0)
    assert np.max(np.abs(f.temp.mean(axis=(1, 2, 3)) - m0)) <= 1e-12 * np.max(np.abs(m0))
    assert np.array_equal(f.temp[:14], t14)                    # only the top 10 levels (sponge_layer.h:19)
    assert np.all(np.abs(f.wvel).max(axis=(1, 2, 3))[15:] < w0[15:])


def test_default_2d_run_reaches_the_recorded_state_at_step_800(oracle):
    """experiments/supercell_example/inputs/input_euler3d.yaml: 100 x 1 x 40, dt = CFL; dycore -> Kessler -> sponge -> nudger."""
    dyc, f, nud = loop_setup(oracle, 100, 1, 40)
    dt = dyc.compute_time_step()
    precl = np.zeros((1, 100, 1))
    for _ in range(800):
        dyc.time_step(f, dt)
        oracle.kessler_time_step(dyc.p.zlen / dyc.p.nz, dt, f.tracers[0], f.tracers[1], f.tracers[2], f.rho_d, f.temp, precl)
        oracle.sponge_layer(dyc.p, f, dt)
        nud.nudge_to_column(dyc.p, f, dt)
    R = RANGES_STEP_800
    assert abs(f.rho_d.min() - R["rho_d"][0]) < 1e-3 and abs(f.rho_d.max() - R["rho_d"][1]) < 5e-3
    assert abs(f.temp.min() - R["T"][0]) < 0.05 and abs(f.temp.max() - R["T"][1]) < 0.05
    assert abs(f.uvel.min() - R["u"][0]) < 0.5 and abs(f.uvel.max() - R["u"][1]) < 0.5
    assert abs(f.wvel.min() - R["w"][0]) < 0.3 and abs(f.wvel.max() - R["w"][1]) < 0.3
    assert abs(f.tracers[0].max() - R["rho_v_max"]) < 1e-4
    # cloud has just formed and rain is starting, within a few steps of the recorded maxima (recorded to 3-4 digits; the
    # oracle passes through them 7-8 steps later -- see DESIGN.md section 2)
    assert 0.9 * R["rho_c_max"] < f.tracers[1].max() < 1.1 * R["rho_c_max"]
    assert 0.6 * R["rho_r_max"] < f.tracers[2].max() < 1.4 * R["rho_r_max"]
    assert f.tracers[1].min() >= 0 and f.tracers[2].min() >= 0


def test_simple_city_modules_of_the_oracle(oracle):
    """Horizontal_Sponge / Time_Averager restatements: closed-form properties."""
    dyc, f = oracle.supercell_setup(24, 22, 6, 1, 1200., 1100., 120., init_data="city", num_tracers=1, enable_gravity=False, perturb=False)
    hs, ta = oracle.HorizontalSponge(), oracle.TimeAverager()
    hs.init(dyc.p, f, 10, 1.0)
    ta.init(dyc.p, f)
    rng = np.random.default_rng(4)
    f.uvel += rng.normal(size=f.uvel.shape)
    u0 = f.uvel.copy()
    hs.apply(dyc.p, f, 0.25, 1, 1, 0, 0)
    # i = 0 and i = nx-1: weight = (cos 0 + 1)/2 * dt/tau = 0.25 ; the 4 interior columns 10..13 are untouched ; y edges off
    col = hs.column[1][:, None, :]
    assert np.allclose(f.uvel[:, :, 0, :], 0.25 * col + 0.75 * u0[:, :, 0, :], rtol=0, atol=1e-15)
    assert np.allclose(f.uvel[:, :, -1, :], 0.25 * col + 0.75 * u0[:, :, -1, :], rtol=0, atol=1e-15)
    assert np.array_equal(f.uvel[:, :, 10:14, :], u0[:, :, 10:14, :])
    assert np.array_equal(f.uvel[:, :, 9, :], u0[:, :, 9, :])                 # i = 9: cos(pi) + 1 = 0
    # running mean with unequal steps = time-weighted mean
    vals, dts = [], [0.5, 0.25, 1.0]
    for dt in dts:
        f.temp += 1.0
        vals.append(f.temp.copy())
        ta.accumulate(dyc.p, f, dt)
    want = sum(v * dt for v, dt in zip(vals, dts)) / sum(dts)
    assert np.max(np.abs(ta.avg[4] - want)) <= 1e-13 * np.max(np.abs(want)) and abs(ta.etime - 1.75) < 1e-15

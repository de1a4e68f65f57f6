"""The complete time loop of experiments/supercell_example/driver.cpp:66-79 on the GPU -- dycore -> Kessler ->
sponge_layer -> ColumnNudger -- against the CPU oracle's loop, plus the two extra modules on their own."""
import numpy as np
import pytest
import torch

from util import compare_fields, gpu_fields, host_libm_matches_restatement, oracle_sensitivity, push_fields

pytestmark = pytest.mark.gpu


def oracle_loop_setup(oracle, nx, ny, nz, nens, xlen, ylen, zlen):
    p, _ = oracle.make_params(nx, ny, nz, nens, xlen, ylen, zlen)
    dyc = oracle.OracleDycore(p)
    f = oracle.Fields(dyc.p)
    dyc.init("supercell", f)
    nud = oracle.ColumnNudger()
    nud.set_column(dyc.p, f)
    oracle.perturb_temperature(dyc.p, f.temp)
    return dyc, f, nud


def oracle_step(oracle, dyc, f, nud, dt, precl):
    dyc.time_step(f, dt)
    oracle.kessler_time_step(dyc.p.zlen / dyc.p.nz, dt, f.tracers[0], f.tracers[1], f.tracers[2], f.rho_d, f.temp, precl)
    oracle.sponge_layer(dyc.p, f, dt)
    nud.nudge_to_column(dyc.p, f, dt)


@pytest.mark.parametrize("shape", [(24, 20, 16, 1), (64, 1, 24, 1), (12, 10, 12, 2)])
def test_full_supercell_loop(mw, oracle, shape):
    from miniweatherml_amd import modules
    nx, ny, nz, nens = shape
    xlen, ylen = 500.0 * nx, (500.0 * ny if ny > 1 else 1.0e5)
    coupler, dycore, micro, nudger = modules.make_supercell(nx, ny, nz, nens, xlen, ylen, 20000., with_nudger=True)
    dyc, f, nud = oracle_loop_setup(oracle, nx, ny, nz, nens, xlen, ylen, 20000.)
    assert np.max(np.abs(nudger.column.cpu().numpy() - nud.column)) <= 1e-13 * np.max(np.abs(nud.column))
    push_fields(coupler, f)
    dt = dycore.compute_time_step(coupler)
    precl = np.zeros((ny, nx, nens))
    for _ in range(5):
        modules.supercell_step(coupler, dycore, micro, nudger, dt)
        oracle_step(oracle, dyc, f, nud, dt, precl)
    compare_fields(gpu_fields(coupler), f.as_dict(), 1e-10, "full loop %s" % (shape,))


@pytest.mark.parametrize("shape", [(24, 20, 16, 1), (64, 1, 24, 1), (12, 10, 12, 2)])
def test_strict_full_supercell_loop_is_bit_identical_to_the_oracle(mw, oracle, shape):
    """The COMPLETE time loop of supercell_example (driver.cpp:66-79: dycore -> Kessler -> sponge_layer -> ColumnNudger) in its strict
    forms -- reference operation order everywhere, glibc's pow / exp / cos (csrc/mw_glibc_pow.h), the horizontal sums in the serial
    backend's order -- against the oracle's loop: every field bit for bit after 1, 5 and 20 steps, 3-D, 2-D and two members, with the
    set-up (init, the nudger's column, perturb_temperature) made on the device."""
    from miniweatherml_amd import modules
    nx, ny, nz, nens = shape
    xlen, ylen = 500.0 * nx, (500.0 * ny if ny > 1 else 1.0e5)
    modules.set_column_strict(1)
    try:
        coupler, dycore, micro, nudger = modules.make_supercell(nx, ny, nz, nens, xlen, ylen, 20000., with_nudger=True)
        dyc, f, nud = oracle_loop_setup(oracle, nx, ny, nz, nens, xlen, ylen, 20000.)
        if host_libm_matches_restatement():
            assert np.array_equal(nudger.column.cpu().numpy().ravel(), np.asarray(nud.column).ravel())
        compare_fields(gpu_fields(coupler), f.as_dict(), 0.0, "full loop set-up %s strict mode 1" % (shape,))     # no push: the device's own state
        dycore.set_strict(1)
        micro.set_strict(1)
        dt = dycore.compute_time_step(coupler)
        precl = np.zeros((ny, nx, nens))
        for step in range(1, 21):
            modules.supercell_step(coupler, dycore, micro, nudger, dt)
            oracle_step(oracle, dyc, f, nud, dt, precl)
            if step in (1, 5, 20):
                compare_fields(gpu_fields(coupler), f.as_dict(), 0.0, "full loop %s strict mode 1, %d steps" % (shape, step))
    finally:
        modules.set_column_strict(0)
        modules.Microphysics_Kessler().set_strict(0)
        from miniweatherml_amd import capi
        capi.check(capi.lib().mw_kessler_set_strict(0))


def test_sponge_layer_alone(mw, oracle):
    from miniweatherml_amd import modules
    coupler, dycore, micro = modules.make_supercell(20, 16, 24, 2, 10000., 8000., 20000.)
    dyc, f = oracle.supercell_setup(20, 16, 24, 2, 10000., 8000., 20000.)
    rng = np.random.default_rng(2)
    for a in [f.uvel, f.vvel, f.wvel, f.temp] + f.tracers:
        a += rng.normal(size=a.shape) * 0.01 * max(1e-6, np.max(np.abs(a)))
    push_fields(coupler, f)
    modules.sponge_layer(coupler, 0.7, 60.0)
    oracle.sponge_layer(dyc.p, f, 0.7, 60.0)
    compare_fields(gpu_fields(coupler), f.as_dict(), 1e-13, "sponge")
    g = gpu_fields(coupler)
    assert np.array_equal(g["temp"][:14], f.temp[:14])                  # only the top 10 levels are touched (:19,:47)


def test_column_nudger_alone(mw, oracle):
    from miniweatherml_amd import modules
    coupler, dycore, micro, nudger = modules.make_supercell(18, 14, 12, 1, 9000., 7000., 20000., with_nudger=True)
    dyc, f, nud = oracle_loop_setup(oracle, 18, 14, 12, 1, 9000., 7000., 20000.)
    push_fields(coupler, f)
    nudger.nudge_to_column(coupler, 5.0)
    nud.nudge_to_column(dyc.p, f, 5.0)
    compare_fields(gpu_fields(coupler), f.as_dict(), 1e-13, "nudger")
    # the temperature bubble raised the column mean, so nudging lowers temp everywhere on the bubble's levels (:62-65)
    assert np.all(gpu_fields(coupler)["wvel"] == 0.0)                   # wvel is not one of the five nudged fields


def test_horizontal_sums_are_deterministic(mw):
    from miniweatherml_amd import modules
    coupler, dycore, micro, nudger = modules.make_supercell(200, 150, 12, 1, 100000., 75000., 20000., with_nudger=True)
    nudger.set_column(coupler)
    c0 = nudger.column.clone()
    for _ in range(3):
        nudger.set_column(coupler)
        assert torch.equal(nudger.column, c0)                           # no atomics: bitwise reproducible


def test_default_2d_run_reaches_the_recorded_state_at_step_800(mw):
    """The reference's default input (experiments/supercell_example/inputs/input_euler3d.yaml: 100 x 1 x 40, CFL dt), complete
    loop on the GPU for 800 steps: the value ranges SURVEY.md 8(d) recorded from the reference at that step (the same check the
    CPU oracle passes in tests/test_oracle_full_loop.py).  Long-run evidence for dycore + Kessler + sponge + nudger together."""
    from miniweatherml_amd import modules
    from test_oracle_full_loop import RANGES_STEP_800 as R
    coupler, dycore, micro, nudger = modules.make_supercell(100, 1, 40, 1, 1.0e5, 1.0e5, 2.0e4, with_nudger=True)
    for _ in range(800):
        modules.supercell_step(coupler, dycore, micro, nudger)
    f = gpu_fields(coupler)
    assert abs(f["density_dry"].min() - R["rho_d"][0]) < 1e-3 and abs(f["density_dry"].max() - R["rho_d"][1]) < 5e-3
    assert abs(f["temp"].min() - R["T"][0]) < 0.05 and abs(f["temp"].max() - R["T"][1]) < 0.05
    assert abs(f["uvel"].min() - R["u"][0]) < 0.5 and abs(f["uvel"].max() - R["u"][1]) < 0.5
    assert abs(f["wvel"].min() - R["w"][0]) < 0.3 and abs(f["wvel"].max() - R["w"][1]) < 0.3
    assert abs(f["tracer0"].max() - R["rho_v_max"]) < 1e-4
    assert 0.9 * R["rho_c_max"] < f["tracer1"].max() < 1.1 * R["rho_c_max"]
    assert 0.6 * R["rho_r_max"] < f["tracer2"].max() < 1.4 * R["rho_r_max"]
    assert f["tracer1"].min() >= 0 and f["tracer2"].min() >= 0


def test_zero_row_maps_through_a_developing_storm(mw):
    """The zero-row maps (option zero_rows, mw_march.h: k_zero_rows) under the real thing: the complete supercell loop in 3-D -- dycore,
    Kessler, sponge, nudger -- from the cloud-free initial state through the first condensation to a storm with cloud and rain, two handles in
    lockstep, one with the maps and one without.  Every field equal at every 25th step and at the end (the maps are a superset of where
    cloud and rain are, whatever Kessler creates between two dycore steps); on the way the map-driven lean form must have been in use
    (cloud-free rows exist throughout) and the storm must have arrived (cloud and rain non-zero in part of the domain)."""
    from miniweatherml_amd import modules
    from util import launched_kernels
    names = ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "cloud_liquid", "precip_liquid")
    runs = []
    for rows in (1, 0):
        coupler, dycore, micro, nudger = modules.make_supercell(100, 40, 40, 1, 1.0e5, 4.0e4, 2.0e4, with_nudger=True)
        dycore.set_option("zero_rows", rows)
        dycore.set_option("zero_verify", rows)                       # (the maps' claims against the data, all 800 steps)
        runs.append((coupler, dycore, micro, nudger))
    launched_kernels(reset=True)
    first_cloud = None
    for n in range(1, 801):
        for coupler, dycore, micro, nudger in runs:
            modules.supercell_step(coupler, dycore, micro, nudger)
        if n % 25 == 0 or n == 800:
            a, b = (r[0].get_data_manager_readonly() for r in runs)
            for name in names:
                assert torch.equal(a.get(name, True), b.get(name, True)), (name, n)
            if first_cloud is None and float(a.get("cloud_liquid", True).max()) > 0.0:
                first_cloud = n
    assert any("k_zero_rows" in k for k in launched_kernels())
    assert runs[0][1].zero_violations() == (0, [0, 0, 0, 0])         # every claim held (and the check did run: not -1)
    dm = runs[0][0].get_data_manager_readonly()
    cl, pr = dm.get("cloud_liquid", True), dm.get("precip_liquid", True)
    assert first_cloud is not None and first_cloud < 800
    assert 0.0 < float((cl != 0).double().mean()) < 0.5 and 0.0 < float((pr != 0).double().mean()) < 0.5
    assert float(cl.max()) > 1.0e-4 and float(pr.max()) > 1.0e-5


def _loop(modules, shape, steps, defer, strict=0, peek_every=0, nranks_note=None):
    nx, ny, nz, nens = shape
    xlen, ylen = 500.0 * nx, (500.0 * ny if ny > 1 else 1.0e5)
    coupler, dycore, micro, nudger = modules.make_supercell(nx, ny, nz, nens, xlen, ylen, 20000., with_nudger=True)
    dycore.set_strict(strict)
    dt = dycore.compute_time_step(coupler)
    peeks = []
    for n in range(1, steps + 1):
        modules.supercell_step(coupler, dycore, micro, nudger, dt * (2.3 if n == 3 else 1.0), defer_nudge=defer)
        if peek_every and n % peek_every == 0:
            peeks.append(gpu_fields(coupler))                       # (through DataManager.get: parked increments are applied first)
    return coupler, dycore, gpu_fields(coupler), peeks


@pytest.mark.parametrize("shape", [(40, 36, 24, 1), (130, 44, 12, 1)])
def test_deferred_nudge_rides_on_the_next_conversion_bit_for_bit(mw, shape):
    """ColumnNudger.nudge_to_column(defer_to=dycore) (mw_nudge_to_column_deferred, round 6): the increments are parked in the dycore handle and the
    next time_step adds them while its converting y launch loads the coupler's fields -- no second pass over five fields.  Ten iterations of the
    complete supercell loop (one of them sub-cycled), deferred against eager: every field bit for bit equal at the end AND whenever somebody looks
    in between (the DataManager applies parked increments in front of every access).  The lazy form really ran (the handle counts it)."""
    from miniweatherml_amd import modules
    _, _, eager, peeks_e = _loop(modules, shape, 10, False, peek_every=4)
    coupler, dycore, lazy, peeks_l = _loop(modules, shape, 10, True, peek_every=4)
    for k in eager:
        assert np.array_equal(eager[k], lazy[k]), k
    for a, b in zip(peeks_e, peeks_l):
        for k in a:
            assert np.array_equal(a[k], b[k]), ("peek", k)
    parked, (rode, passes) = dycore.pending()
    assert not parked                                               # gpu_fields looked at the fields: nothing is parked any more
    assert rode >= 6 and passes >= 2, (rode, passes)                # 9 consumed by conversions minus the peeks; each peek + the final look: a pass
    assert " conv_in_y " in (dycore.path() + " ")


def test_deferred_nudge_and_the_raw_arrays(mw):
    """What the contract says about the raw arrays: between the deferred call and the next time step they do NOT hold the nudged values (that is
    the saving); mw_dycore_flush_pending -- what DataManager.get calls -- makes them whole; a second deferred call with increments still parked
    applies the old ones first (they count in the new averages)."""
    from miniweatherml_amd import modules
    coupler, dycore, micro, nudger = modules.make_supercell(40, 36, 16, 1, 20000., 18000., 20000., with_nudger=True)
    c2, d2, m2, n2 = modules.make_supercell(40, 36, 16, 1, 20000., 18000., 20000., with_nudger=True)
    dt = dycore.compute_time_step(coupler)
    for _ in range(2):
        modules.supercell_step(coupler, dycore, micro, nudger, dt)
        modules.supercell_step(c2, d2, m2, n2, dt)
    T = coupler.dm.entries["temp"]["data"]                          # the raw tensor, NOT through get()
    before = T.clone()
    nudger.nudge_to_column(coupler, 7.0, defer_to=dycore)
    n2.nudge_to_column(c2, 7.0)
    assert torch.equal(T, before) and dycore.pending()[0]           # parked, not applied
    assert not torch.equal(T, c2.dm.entries["temp"]["data"])
    nudger.nudge_to_column(coupler, 3.0, defer_to=dycore)           # old increments applied first, new ones parked
    n2.nudge_to_column(c2, 3.0)
    assert dycore.pending()[0]
    dycore.flush_pending()
    assert not dycore.pending()[0]
    for n in ("density_dry", "uvel", "vvel", "temp", "water_vapor"):
        assert torch.equal(coupler.dm.entries[n]["data"], c2.dm.entries[n]["data"]), n


@pytest.mark.parametrize("case", ["strict", "2d", "members", "weno3", "run_time_switches"])
def test_deferred_nudge_on_paths_that_apply_it_with_a_pass(mw, case):
    """Paths whose conversion does not take increments along -- the strict / general kernels, 2-D, ensemble members, the run-time configuration
    (K = 0: water vapour counts as a tracer that can vanish, its rows are scanned from the coupler's arrays) -- apply parked increments with a
    pass at the start of the time step; WENO-3 on the folded configuration takes them along.  Same bits as the eager loop either way."""
    from miniweatherml_amd import modules
    shape = {"2d": (64, 1, 24, 1), "members": (24, 20, 12, 2)}.get(case, (40, 36, 16, 1))
    res = []
    for defer in (False, True):
        nx, ny, nz, nens = shape
        xlen, ylen = 500.0 * nx, (500.0 * ny if ny > 1 else 1.0e5)
        coupler, dycore, micro, nudger = modules.make_supercell(nx, ny, nz, nens, xlen, ylen, 20000., with_nudger=True, ord=3 if case == "weno3" else 5)
        if case == "strict":
            dycore.set_strict(1)
        if case == "run_time_switches":
            dycore.set_option("spec", 0)
        dt = dycore.compute_time_step(coupler)
        for _ in range(5):
            modules.supercell_step(coupler, dycore, micro, nudger, dt, defer_nudge=defer)
        res.append((gpu_fields(coupler), dycore.pending()[1]))
    for k in res[0][0]:
        assert np.array_equal(res[0][0][k], res[1][0][k]), (case, k)
    rode, passes = res[1][1]
    if case == "weno3":
        assert rode == 4 and passes == 1, (rode, passes)
    else:
        assert rode == 0 and passes == 5, (case, rode, passes)

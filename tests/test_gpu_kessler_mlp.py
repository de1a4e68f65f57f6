"""GPU parity of Microphysics_Kessler::time_step and of the surrogate MLP against the CPU oracle."""
import numpy as np
import pytest
import torch

from util import compare_fields, gpu_fields, push_fields, host_libm_matches_restatement

pytestmark = pytest.mark.gpu


def rainy_state(oracle, nx, ny, nz, heavy):
    """An oracle supercell state pushed into cloud/rain so that every branch of kessler() runs; `heavy` forces
    rainsplit > 1 (fall speed * dt > 0.8 dz)."""
    dyc, f = oracle.supercell_setup(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny if ny > 1 else 1e5, 20000.)
    rng = np.random.default_rng(11)
    shp = f.rho_d.shape
    qc = rng.uniform(0, 3e-3, shp) * (rng.uniform(size=shp) > 0.4)
    qr = rng.uniform(0, (2e-2 if heavy else 5e-4), shp) * (rng.uniform(size=shp) > 0.5)
    f.tracers[1][...] = qc * f.rho_d
    f.tracers[2][...] = qr * f.rho_d
    f.tracers[0][...] *= rng.uniform(0.6, 1.3, shp)          # sub- and super-saturated columns
    return dyc, f


@pytest.mark.parametrize("heavy", [False, True])
@pytest.mark.parametrize("shape", [(16, 12, 20), (40, 1, 40)])
def test_kessler_matches_oracle(mw, oracle, shape, heavy):
    from miniweatherml_amd import modules
    nx, ny, nz = shape
    dyc, f = rainy_state(oracle, nx, ny, nz, heavy)
    coupler, dycore, micro = modules.make_supercell(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny if ny > 1 else 1e5, 20000.)
    push_fields(coupler, f)
    dt = 90.0 if heavy else dycore.compute_time_step(coupler)
    precl = np.zeros((ny, nx, 1))
    rs_ref = oracle.kessler_time_step(coupler.get_dz(), dt, f.tracers[0], f.tracers[1], f.tracers[2], f.rho_d, f.temp, precl)
    rs = micro.time_step(coupler, dt, return_rainsplit=True)
    assert rs == rs_ref
    assert (rs > 1) == heavy
    compare_fields(gpu_fields(coupler), f.as_dict(), 1e-12, "kessler heavy=%s" % heavy)
    got_precl = coupler.get_data_manager_readonly().get("precl", True).cpu().numpy()
    assert np.max(np.abs(got_precl - precl)) <= 1e-12 * max(np.max(np.abs(precl)), 1e-300)
    assert np.max(precl) > 0


@pytest.mark.parametrize("heavy", [False, True])
@pytest.mark.parametrize("shape", [(16, 12, 20), (40, 1, 40)])
def test_strict_kessler_is_bit_identical_to_the_oracle(mw, oracle, shape, heavy):
    """The strict Kessler path (mw_kessler_set_strict(1): reference operation order, theta form, IEEE divisions, glibc's pow and exp,
    csrc/mw_glibc_pow.h) against the oracle's restatement of microphysics_kessler.h:99-162, :234-339: every field and the precipitation
    rate bit for bit, with rainsplit 1 and > 1, over two consecutive calls."""
    from miniweatherml_amd import modules
    nx, ny, nz = shape
    dyc, f = rainy_state(oracle, nx, ny, nz, heavy)
    coupler, dycore, micro = modules.make_supercell(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny if ny > 1 else 1e5, 20000.)
    push_fields(coupler, f)
    micro.set_strict(1)
    dt = 90.0 if heavy else dycore.compute_time_step(coupler)
    for call in range(2):
        precl = np.zeros((ny, nx, 1))
        rs_ref = oracle.kessler_time_step(coupler.get_dz(), dt, f.tracers[0], f.tracers[1], f.tracers[2], f.rho_d, f.temp, precl)
        rs = micro.time_step(coupler, dt, return_rainsplit=True)
        assert rs == rs_ref and (rs > 1) == heavy
        compare_fields(gpu_fields(coupler), f.as_dict(), 0.0, "kessler strict mode 1 heavy=%s call %d" % (heavy, call))
        got_precl = coupler.get_data_manager_readonly().get("precl", True).cpu().numpy()
        if host_libm_matches_restatement():
            assert np.array_equal(got_precl, precl)
        else:
            assert np.max(np.abs(got_precl - precl)) <= 1e-13 * max(np.max(np.abs(precl)), 1e-300)
    micro.set_strict(0)
    micro.time_step(coupler, dt)                                  # (and the process-wide switch is back on the production kernels)


def test_kessler_dry_state_is_identity_on_vapor_free_air(mw, oracle):
    from miniweatherml_amd import modules
    coupler, dycore, micro = modules.make_supercell(12, 12, 10, 1, 6000., 6000., 20000.)
    dm = coupler.get_data_manager_readwrite()
    for n in ("water_vapor", "cloud_liquid", "precip_liquid"):
        dm.get(n).zero_()
    t0 = dm.get("temp").clone()
    micro.time_step(coupler, 1.0)
    assert float(((dm.get("temp") - t0) / t0).abs().max()) <= 4e-16     # temp -> theta * exner round trip (:143, :160)
    for n in ("water_vapor", "cloud_liquid", "precip_liquid", "precl"):
        assert float(dm.get(n).abs().max()) == 0.0


def test_kessler_rejects_nonpositive_dt(mw):
    from miniweatherml_amd import modules
    from miniweatherml_amd.capi import MWError
    coupler, dycore, micro = modules.make_supercell(8, 8, 8, 1, 4000., 4000., 20000.)
    with pytest.raises(MWError, match="nonpositive dt"):
        micro.time_step(coupler, 0.0)                                # kessler(): endrun("... nonpositive dt"), :242


def mlp_tol(scl_out, n):
    return 1e-5 * (scl_out[n, 1] - scl_out[n, 0])                   # 1e-5 on the fp32 network output


def test_mlp_matches_oracle_on_supercell_state(mw, oracle):
    from miniweatherml_amd import modules
    dyc, f = rainy_state(oracle, 20, 16, 24, False)
    W1, b1, W2, b2, si, so = modules.load_surrogate_weights()
    ref = oracle.mlp_forward(f.temp, f.rho_d, f.tracers[0], f.tracers[1], f.tracers[2], W1, b1, W2, b2, si, so)
    t = [torch.from_numpy(a).cuda() for a in (f.temp, f.rho_d, f.tracers[0], f.tracers[1], f.tracers[2])]
    outs = modules.mlp_forward(*t, W1, b1, W2, b2, si, so)
    for n, (o, r) in enumerate(zip(outs, ref)):
        assert np.max(np.abs(o.cpu().numpy() - r)) <= mlp_tol(so, n), n
    for o in outs[1:]:
        assert float(o.min()) >= 0.0                                 # densities clipped at 0 (:199-201)


def test_strict_mlp_is_bit_identical_to_the_oracle(mw, oracle):
    """mw_mlp_set_strict(1): fp32 accumulation in index order, no contraction -- the CPU restatement's order -- on a rainy supercell
    state and on inputs drawn over (and beyond) the scaling ranges; the MFMA kernels stay within 1e-5 of it."""
    from miniweatherml_amd import modules
    W1, b1, W2, b2, si, so = modules.load_surrogate_weights()
    dyc, f = rainy_state(oracle, 20, 16, 24, False)
    rng = np.random.default_rng(3)
    wide = [rng.uniform(si[i, 0] - 0.2 * (si[i, 1] - si[i, 0]), si[i, 1] + 0.2 * (si[i, 1] - si[i, 0]), 50_001) for i in range(5)]
    for ins in ([f.temp, f.rho_d, f.tracers[0], f.tracers[1], f.tracers[2]], wide):
        ref = oracle.mlp_forward(*ins, W1, b1, W2, b2, si, so)
        t = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in ins]
        outs = modules.mlp_forward(*t, W1, b1, W2, b2, si, so, strict=1)
        for n, (o, r) in enumerate(zip(outs, ref)):
            assert np.array_equal(o.cpu().numpy(), r), n
        fast = modules.mlp_forward(*t, W1, b1, W2, b2, si, so)              # (and the switch is back on the MFMA kernels)
        for n, (o, r) in enumerate(zip(fast, ref)):
            assert np.max(np.abs(o.cpu().numpy() - r)) <= mlp_tol(so, n), n


@pytest.mark.parametrize("ncells", [1, 15, 16, 17, 63, 64, 65, 1000, 4097])
def test_mlp_ragged_sizes(mw, oracle, ncells):
    from miniweatherml_amd import modules
    W1, b1, W2, b2, si, so = modules.load_surrogate_weights()
    rng = np.random.default_rng(ncells)
    ins = [rng.uniform(si[i, 0], si[i, 1], ncells) for i in range(5)]
    ref = oracle.mlp_forward(*ins, W1, b1, W2, b2, si, so)
    outs = modules.mlp_forward(*[torch.from_numpy(a).cuda() for a in ins], W1, b1, W2, b2, si, so)
    for n, (o, r) in enumerate(zip(outs, ref)):
        assert o.shape == (ncells,)
        assert np.max(np.abs(o.cpu().numpy() - r)) <= mlp_tol(so, n)


def test_mlp_operand_layout_with_one_hot_weights(mw, oracle):
    """Every (input i -> hidden u -> output n) path alone: catches any row/column permutation slip in the MFMA operand
    images (asymmetric data, exact small integers)."""
    from miniweatherml_amd import modules
    rng = np.random.default_rng(0)
    sid = np.ascontiguousarray(np.array([[0., 1.]] * 5))
    sod = np.ascontiguousarray(np.array([[0., 1.]] * 4))
    n = 200
    ins = [rng.integers(1, 9, n).astype(np.float64) / 8.0 for _ in range(5)]
    tin = [torch.from_numpy(a).cuda() for a in ins]
    for i in range(5):
        for u in range(10):
            for o in range(4):
                W1 = np.zeros((5, 10), np.float32); W1[i, u] = 1 + i
                W2 = np.zeros((10, 4), np.float32); W2[u, o] = 1 + o
                b1 = np.zeros(10, np.float32); b2 = np.zeros(4, np.float32); b2[o] = 0.25
                ref = oracle.mlp_forward(*ins, W1, b1, W2, b2, sid, sod)
                outs = modules.mlp_forward(*tin, W1, b1, W2, b2, sid, sod)
                for k in range(4):
                    assert np.array_equal(outs[k].cpu().numpy(), ref[k]), (i, u, o, k)


def test_surrogate_module_runs_beside_kessler(mw, oracle):
    """custom_modules::Microphysics_Kessler of the surrogate experiment: NN inference + true Kessler, NN not written back."""
    from miniweatherml_amd import modules
    micro = modules.Microphysics_Kessler_Surrogate()
    coupler, dycore, micro = modules.make_supercell(16, 16, 12, 1, 8000., 8000., 20000., micro=micro)
    dyc, f = oracle.supercell_setup(16, 16, 12, 1, 8000., 8000., 20000.)
    dt = dycore.compute_time_step(coupler)
    nn = micro.time_step(coupler, dt)
    precl = np.zeros((16, 16, 1))
    oracle.kessler_time_step(coupler.get_dz(), dt, f.tracers[0], f.tracers[1], f.tracers[2], f.rho_d, f.temp, precl)
    compare_fields(gpu_fields(coupler), f.as_dict(), 1e-12, "surrogate module leaves Kessler result in the coupler")
    d = micro.mean_diffs(coupler)                                # the "Relative diff" prints (:266-269): mean(NN - Kessler), on the device
    assert set(d) == {"rho_v", "rho_c", "rho_r", "temp"} and all(np.isfinite(v) for v in d.values())
    g = gpu_fields(coupler)
    for key, x, name in (("temp", nn[0], "temp"), ("rho_v", nn[1], "tracer0"), ("rho_c", nn[2], "tracer1"), ("rho_r", nn[3], "tracer2")):
        ref = float(np.mean(x.cpu().numpy() - g[name]))
        assert abs(d[key] - ref) <= 1e-12 * max(abs(ref), float(np.max(np.abs(g[name]))))
    assert len(nn) == 4


def _oracle_online_step(oracle, dyc, f, nud, dt, precl, W):
    """One iteration of inference_ponni.cpp:69-82 with microphysics_kessler_ponni.h:273-276 un-commented, on the CPU oracle: dycore, the
    network on the state IN FRONT of the Kessler step (:176-202), the true Kessler step (:204-262; its result only feeds the printed
    differences), the four deep_copy_to lines -- the network's (temp, rho_v, rho_c, rho_r) overwrite Kessler's -- sponge layer, nudger."""
    dyc.time_step(f, dt)
    nn = oracle.mlp_forward(f.temp, f.rho_d, f.tracers[0], f.tracers[1], f.tracers[2], *W)
    oracle.kessler_time_step(dyc.p.zlen / dyc.p.nz, dt, f.tracers[0], f.tracers[1], f.tracers[2], f.rho_d, f.temp, precl)
    diffs = [(float(np.mean(nn[k] - a)), float(np.mean(np.abs(nn[k] - a)))) for k, a in ((1, f.tracers[0]), (2, f.tracers[1]), (3, f.tracers[2]), (0, f.temp))]
    f.temp[...] = nn[0]
    for t in range(3):
        f.tracers[t][...] = nn[1 + t]
    oracle.sponge_layer(dyc.p, f, dt)
    nud.nudge_to_column(dyc.p, f, dt)
    return nn, diffs


@pytest.mark.parametrize("shape", [(20, 16, 12, 1), (48, 1, 20, 1)])
def test_online_surrogate_step_equals_the_oracles_mlp_then_overwrite(mw, oracle, shape):
    """`online = True` (modules.Microphysics_Kessler_Surrogate; C++ twin: host/mw_ponni.h `online`): the reference's four commented-out
    deep_copy_to lines, microphysics_kessler_ponni.h:273-276, switched on -- the network's output replaces Kessler's in the coupler and the
    run is steered by the network from then on.  The complete surrogate loop, three steps, in the strict forms (dycore, Kessler, the
    horizontal sums, the thread-per-cell MLP in index order): every coupler field BIT-identical to the oracle's MLP-then-overwrite loop
    after every step, the returned network output equal to the oracle's, the four printed mean differences computed BEFORE the overwrite."""
    from test_gpu_full_loop import oracle_loop_setup
    from miniweatherml_amd import capi, modules
    nx, ny, nz, nens = shape
    xlen, ylen = 500.0 * nx, (500.0 * ny if ny > 1 else 1.0e5)
    modules.set_column_strict(1)
    try:
        micro = modules.Microphysics_Kessler_Surrogate()
        micro.online, micro.mlp_strict = True, 1
        coupler, dycore, micro, nudger = modules.make_supercell(nx, ny, nz, nens, xlen, ylen, 20000., micro=micro, with_nudger=True)
        dyc, f, nud = oracle_loop_setup(oracle, nx, ny, nz, nens, xlen, ylen, 20000.)
        push_fields(coupler, f)
        nudger.column.copy_(torch.from_numpy(np.ascontiguousarray(nud.column)).reshape(nudger.column.shape))
        dycore.set_strict(1)
        micro.set_strict(1)
        W = (micro.W1, micro.b1, micro.W2, micro.b2, micro.scl_in, micro.scl_out)
        dt = dycore.compute_time_step(coupler)
        precl = np.zeros((ny, nx, nens))
        for step in range(1, 4):
            dycore.time_step(coupler, dt)
            nn_gpu = micro.time_step(coupler, dt)
            after_micro = gpu_fields(coupler)
            modules.sponge_layer(coupler, dt)
            nudger.nudge_to_column(coupler, dt)
            nn, diffs = _oracle_online_step(oracle, dyc, f, nud, dt, precl, W)
            for k in range(4):
                assert np.array_equal(nn_gpu[k].cpu().numpy(), nn[k]), (step, k)
            # right behind micro.time_step the coupler holds the NETWORK's values, not Kessler's
            for k, name in enumerate(("temp", "tracer0", "tracer1", "tracer2")):
                assert np.array_equal(after_micro[name], nn[k]), (step, name)
            d = micro._diffs                                      # (:266-269) mean(NN - Kessler), taken before the overwrite
            for key, (ref, scale) in zip(("rho_v", "rho_c", "rho_r", "temp"), diffs):      # (another summation order: relative to the mean |difference|)
                assert abs(d[key] - ref) <= 1e-12 * scale + 1e-300, (step, key, d[key], ref)
            compare_fields(gpu_fields(coupler), f.as_dict(), 0.0, "online surrogate loop %s, %d steps (strict forms)" % (shape, step))
        assert float(np.max(np.abs(f.tracers[1]))) > 0.0          # the network does produce cloud: the overwrite is not a no-op
    finally:
        modules.set_column_strict(0)
        capi.check(capi.lib().mw_kessler_set_strict(0))
        capi.check(capi.lib().mw_mlp_set_strict(0))


def test_online_surrogate_step_production_kernels(mw, oracle):
    """The same switch on the production kernels (MFMA MLP, fast Kessler): one micro step from identical inputs -- the coupler holds the
    network's output, within 1e-5 of the output scaling's range of the oracle's (the MLP tolerance of this suite)."""
    from miniweatherml_amd import modules
    micro = modules.Microphysics_Kessler_Surrogate()
    micro.online = True
    coupler, dycore, micro = modules.make_supercell(16, 16, 12, 1, 8000., 8000., 20000., micro=micro)
    dyc, f = oracle.supercell_setup(16, 16, 12, 1, 8000., 8000., 20000.)
    push_fields(coupler, f)
    dt = dycore.compute_time_step(coupler)
    nn_gpu = micro.time_step(coupler, dt)
    nn = oracle.mlp_forward(f.temp, f.rho_d, f.tracers[0], f.tracers[1], f.tracers[2], micro.W1, micro.b1, micro.W2, micro.b2, micro.scl_in, micro.scl_out)
    g = gpu_fields(coupler)
    for k, name in enumerate(("temp", "tracer0", "tracer1", "tracer2")):
        tol = 1e-5 * (micro.scl_out[k, 1] - micro.scl_out[k, 0])
        assert np.max(np.abs(g[name] - nn[k])) <= tol, name
        assert np.array_equal(g[name], nn_gpu[k].cpu().numpy()), name          # the coupler holds exactly what the module returned
    assert np.array_equal(g["density_dry"], f.rho_d)


def test_kessler_math_against_host_libm(mw):
    """The module's own log / exp / sqrt / reciprocal (mw_kessler.hip: short forms for positive finite arguments) against
    numpy's: <= 4 ulp-ish (4.5e-16 relative), log(0) = -inf, exp(-inf) = 0, tiny and huge arguments included."""
    import ctypes as C
    from miniweatherml_amd import capi
    rng = np.random.default_rng(3)
    cases = {
        0: (np.concatenate([np.exp(rng.uniform(-300, 5, 200000)), rng.uniform(0.5, 2.0, 200000), [1.0, 2.0, 0.5, 1e-310, 5e-324, 1e300]]), np.log),
        1: (np.concatenate([rng.uniform(-700, 50, 200000), rng.uniform(-1, 1, 100000), [0.0, -745.0, 709.0]]), np.exp),
        2: (np.exp(rng.uniform(-7, 7, 200000)), np.sqrt),
        3: (np.exp(rng.uniform(-40, 40, 200000)) * np.sign(rng.normal(size=200000)), lambda v: 1.0 / v),
    }
    for fn, (x, ref) in cases.items():
        xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        yd = torch.empty_like(xd)
        capi.check(capi.lib().mw_kessler_math_probe(xd.numel(), C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), fn,
                                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        y, r = yd.cpu().numpy(), ref(x)
        err = np.abs(y - r) / np.maximum(np.abs(r), 1e-300)
        err[(r == 0) & (y == 0)] = 0.0
        assert np.all(np.isfinite(y)), fn
        assert err.max() <= 4.5e-16, (fn, err.max(), x[np.argmax(err)])
    z = torch.tensor([0.0, -np.inf, np.inf], dtype=torch.float64).cuda()
    o = torch.empty_like(z)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    capi.check(capi.lib().mw_kessler_math_probe(1, C.c_void_p(z.data_ptr()), C.c_void_p(o.data_ptr()), 0, st))
    assert float(o[0]) == -np.inf                                  # log(0): so that exp(a log 0) = 0 = pow(0, a)
    capi.check(capi.lib().mw_kessler_math_probe(3, C.c_void_p(z.data_ptr()), C.c_void_p(o.data_ptr()), 1, st))
    assert o.cpu().tolist() == [1.0, 0.0, np.inf]

"""SURVEY.md 8(f) rank 2 + 3 on the GPU: Horizontal_Sponge, Time_Averager, the simple_city loop
(experiments/simple_city/driver.cpp:66-79) and the CDF-5 file output, against the oracle."""
import os

import numpy as np
import pytest

import cdf
from util import compare_fields, gpu_fields, host_libm_matches_restatement, push_fields

pytestmark = pytest.mark.gpu


def oracle_city(oracle, nx, ny, nz, nens, xlen, ylen, zlen, init):
    odyc, of = oracle.supercell_setup(nx, ny, nz, nens, xlen, ylen, zlen, init_data=init, num_tracers=1, enable_gravity=False,
                                      perturb=False)
    of.tracers[0][...] = 0.0
    hs, ta = oracle.HorizontalSponge(), oracle.TimeAverager()
    hs.init(odyc.p, of, 10, 1.0)
    ta.init(odyc.p, of)
    return odyc, of, hs, ta


@pytest.mark.parametrize("edges", [(1, 1, 1, 1), (1, 0, 0, 1), (0, 1, 1, 0)])
@pytest.mark.parametrize("shape", [(24, 22, 6, 1), (13, 9, 5, 2)])
def test_horizontal_sponge_and_time_averager(mw, oracle, shape, edges):
    from miniweatherml_amd import modules
    nx, ny, nz, nens = shape
    coupler, dycore, hs, ta = modules.make_simple_city(nx, ny, nz, nens, 50. * nx, 50. * ny, 120., "city")
    odyc, of, ohs, ota = oracle_city(oracle, nx, ny, nz, nens, 50. * nx, 50. * ny, 120., "city")
    rng = np.random.default_rng(2)
    for a in (of.rho_d, of.uvel, of.vvel, of.wvel, of.temp, of.tracers[0]):
        a += 0.01 * np.abs(a).max() * rng.normal(size=a.shape) + 1e-3 * rng.normal(size=a.shape)
    push_fields(coupler, of)
    hs.override_uvel(7.0)
    ohs.column[1][...] = 7.0
    assert np.array_equal(hs.column.cpu().numpy()[[0, 2, 3, 4, 5]], ohs.column[[0, 2, 3, 4, 5]])     # init took cell (k,0,0,e) before the noise
    for step, dt in enumerate((0.3, 0.05, 0.2)):
        hs.apply(coupler, dt, *[bool(e) for e in edges])
        ohs.apply(odyc.p, of, dt, *edges)
        ta.accumulate(coupler, dt)
        ota.accumulate(odyc.p, of, dt)
        got = gpu_fields(coupler)
        exact = host_libm_matches_restatement()       # the weight's cos has glibc's bits (csrc/mw_glibc_pow.h): bit-identical
        for k, a in of.as_dict().items():
            assert np.array_equal(got[k], a) if exact else np.max(np.abs(got[k] - a)) <= 4e-16 * max(1.0, np.max(np.abs(a))), (k, step)
        dm = coupler.get_data_manager_readonly()
        for n, a in zip(("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor"), ota.avg):
            g = dm.get("time_avg_" + n).cpu().numpy()
            assert np.array_equal(g, a) if exact else np.max(np.abs(g - a)) <= 4e-16 * max(1.0, np.max(np.abs(a))), (n, step)
    assert abs(ta.etime - 0.55) < 1e-15


@pytest.mark.parametrize("case", [("city", 48, 48, 12, 2400., 2400., 120.), ("building", 40, 40, 16, 200., 200., 80.)])
def test_strict_simple_city_loop_is_bit_identical_to_the_oracle(mw, oracle, case):
    """The complete simple_city loop (driver.cpp:66-79: horiz_sponge.apply -> dycore.time_step -> sponge_layer(dt, 1) ->
    time_averager.accumulate) in its strict forms, from the device's own initial state: fields and running time averages bit for
    bit after every one of 8 steps."""
    from miniweatherml_amd import modules
    init, nx, ny, nz, xlen, ylen, zlen = case
    coupler, dycore, hs, ta = modules.make_simple_city(nx, ny, nz, 1, xlen, ylen, zlen, init)
    odyc, of, ohs, ota = oracle_city(oracle, nx, ny, nz, 1, xlen, ylen, zlen, init)
    modules.set_column_strict(1)
    try:
        dycore.set_strict(1)
        compare_fields(gpu_fields(coupler), of.as_dict(), 0.0, "%s set-up strict mode 1" % init)
        for step in range(8):
            dt = modules.simple_city_step(coupler, dycore, hs, ta)
            ohs.apply(odyc.p, of, dt, 1, 1, 0, 0)
            odyc.time_step(of, dt)
            oracle.sponge_layer(odyc.p, of, dt, 1.0)
            ota.accumulate(odyc.p, of, dt)
            compare_fields(gpu_fields(coupler), of.as_dict(), 0.0, "%s loop strict mode 1, step %d" % (init, step + 1))
            if host_libm_matches_restatement():
                dm = coupler.get_data_manager_readonly()
                for n, a in zip(("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor"), ota.avg):
                    assert np.array_equal(dm.get("time_avg_" + n).cpu().numpy(), a), (n, step)
    finally:
        modules.set_column_strict(0)


@pytest.mark.parametrize("case", [("city", 48, 48, 12, 2400., 2400., 120.), ("building", 40, 40, 16, 200., 200., 80.)])
def test_simple_city_loop_and_output(mw, oracle, tmp_path, case):
    """horiz_sponge.apply -> dycore.time_step -> sponge_layer(dt, 1) -> time_averager.accumulate, 5 steps, then both files."""
    from miniweatherml_amd import modules
    init, nx, ny, nz, xlen, ylen, zlen = case
    prefix = str(tmp_path / "city")
    coupler, dycore, hs, ta = modules.make_simple_city(nx, ny, nz, 1, xlen, ylen, zlen, init, out_prefix=prefix)
    odyc, of, ohs, ota = oracle_city(oracle, nx, ny, nz, 1, xlen, ylen, zlen, init)
    compare_fields(gpu_fields(coupler), of.as_dict(), 1e-13, "init")
    push_fields(coupler, of)
    dycore.output(coupler, 0.0)
    snaps = [gpu_fields(coupler)]
    etime = 0.0
    for step in range(5):
        dt = modules.simple_city_step(coupler, dycore, hs, ta)
        assert dt == odyc.compute_time_step()
        ohs.apply(odyc.p, of, dt, 1, 1, 0, 0)
        odyc.time_step(of, dt)
        oracle.sponge_layer(odyc.p, of, dt, 1.0)
        ota.accumulate(odyc.p, of, dt)
        etime += dt
        compare_fields(gpu_fields(coupler), of.as_dict(), 1e-11 if step == 0 else 1e-9, "%s loop step %d" % (init, step + 1))
        if step in (1, 4):
            dycore.output(coupler, etime)
            snaps.append(gpu_fields(coupler))
    ta.finalize(coupler, str(tmp_path / "time_averaged_fields.nc"))
    # --- the running output file: CDF-5, dims/vars/order of :2114-2131, one record per output() call, member 0
    r = cdf.Reader(prefix + ".nc")
    assert r.version == 5 and r.numrecs == 3
    assert r.dims == [("x", nx), ("y", ny), ("z", nz), ("t", 0)]
    assert [v["name"] for v in r.vars] == ["x", "y", "z", "t", "density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor"]
    assert np.array_equal(r.get("x"), (np.arange(nx) + 0.5) * (xlen / nx)) and np.array_equal(r.get("z"), (np.arange(nz) + 0.5) * (zlen / nz))
    assert r.get("t")[0] == 0.0 and r.get("t")[2] == etime
    for rec, snap in enumerate(snaps):
        for k, name in (("density_dry",) * 2, ("uvel",) * 2, ("vvel",) * 2, ("wvel",) * 2, ("temp",) * 2, ("tracer0", "water_vapor")):
            assert np.array_equal(r.get(name)[rec], snap[k][..., 0]), (rec, name)        # bit-exact copy of member 0
    assert min(v["begin"] for v in r.vars) == 1 << 20                                     # the reference's 1 MiB header hint
    # --- the time-averaged file (time_averager.h:104-120): no record dimension
    r2 = cdf.Reader(str(tmp_path / "time_averaged_fields.nc"))
    assert r2.version == 5 and r2.dims == [("x", nx), ("y", ny), ("z", nz)]
    assert [v["name"] for v in r2.vars] == ["x", "y", "z", "density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor"]
    for name, a in zip(("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor"), ota.avg):
        g = r2.get(name)
        assert np.max(np.abs(g - a[..., 0])) <= 1e-9 * max(1e-30, np.max(np.abs(a))) + 1e-12, name


def test_output_from_two_ranks_into_one_file(mw, oracle, tmp_path):
    """Two ranks (two handles in this process) write their y-blocks into the one shared file, like write1_all (:2184)."""
    from miniweatherml_amd import modules
    prefix = str(tmp_path / "shared")
    ranks = [modules.make_supercell(16, 12, 6, 2, 8000., 6000., 20000., nranks=2, myrank=r) for r in range(2)]
    single = modules.make_supercell(16, 12, 6, 2, 8000., 6000., 20000.)
    for c, _, _ in ranks:
        c.set_option("out_prefix", prefix)
    import threading

    def run(etime):
        # the ranks' collective call: two threads sharing a barrier (the processes of a real run share dist.barrier())
        bar = threading.Barrier(2)
        errs = []

        def one(r):
            try:
                c, d, _ = ranks[r]
                d.output(c, etime, barrier=bar.wait)
            except Exception as e:                        # pragma: no cover
                errs.append(e)
                bar.abort()
        th = [threading.Thread(target=one, args=(r,)) for r in (1, 0)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
    run(0.0)
    run(12.5)
    r = cdf.Reader(prefix + ".nc")
    assert r.numrecs == 2 and r.dims[:2] == [("x", 16), ("y", 12)]
    ref = gpu_fields(single[0])
    for name, key in (("density_dry", "density_dry"), ("temp", "temp"), ("water_vapor", "tracer0")):
        for rec in range(2):
            assert np.array_equal(r.get(name)[rec], ref[key][..., 0]), name
    assert np.array_equal(r.get("y"), (np.arange(12) + 0.5) * 500.0) and list(r.get("t")) == [0.0, 12.5]


def test_file_per_process_output(mw, tmp_path):
    """The file_per_process branch (:2038-2090): one file per rank with local sizes and the rank's own coordinates."""
    from miniweatherml_amd import modules
    prefix = str(tmp_path / "fpp")
    ranks = [modules.make_supercell(16, 12, 6, 1, 8000., 6000., 20000., nranks=2, myrank=r) for r in range(2)]
    for r, (c, d, _) in enumerate(ranks):
        c.set_option("out_prefix", prefix)
        c.set_option("file_per_process", True)
        d.output(c, 0.0)
        d.output(c, 3.5)
        rd = cdf.Reader("%s_%08d.nc" % (prefix, r))
        assert rd.numrecs == 2 and rd.dims == [("x", 16), ("y", 6), ("z", 6), ("t", 0)] and list(rd.get("t")) == [0.0, 3.5]
        assert np.array_equal(rd.get("y"), (np.arange(6) + 6 * r + 0.5) * 500.0)
        assert np.array_equal(rd.get("temp")[1], gpu_fields(c)["temp"][..., 0])

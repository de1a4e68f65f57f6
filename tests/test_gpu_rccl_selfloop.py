"""The asynchronous halo exchange inside RESULT-CHECKED time steps on one GPU (round 5).

1. The real RCCL transport (mw_rccl.cpp: pack kernels -> ev_ready -> ncclGroup of sends and receives on the side stream -> ev_done ->
   unpack kernels; the pipelined and the two-stream schedules of mw_dycore.hip around it).  A periodic domain tiled 2 x 2 or 4 x 2
   from copies of ONE block has neighbours whose strips are bit-identical to the block's own, so rank 0 of that decomposition with every
   peer mapped to itself on a 1-rank communicator (mw_dycore_use_rccl_self) exchanges exactly the messages the real job would -- and
   must equal the one-rank run of the block BITWISE.  (halo_exchange, dynamics_euler_stratified_wenofv.h:641-723; neighbour matrix
   core/coupler.h:169-179.)
2. The in-process transport between R handles made stream-ordered (tests/util.py: StreamExchanger -- no host synchronisation inside
   the callback) with seeded delay fuzz: 50 seeds x (2 x 2, 4 x 2), bitwise against one rank.
"""
import ctypes as C
import threading

import numpy as np
import pytest
import torch

from util import StreamExchanger, gpu_fields, set_options

pytestmark = pytest.mark.gpu

FIELDS = ("density_dry", "uvel", "vvel", "wvel", "temp")


def _copy_state(src, dst):
    s, d = src.get_data_manager_readonly(), dst.get_data_manager_readwrite()
    for n in FIELDS + tuple(src.get_tracer_names()):
        d.get(n).copy_(s.get(n, True))


def _assert_equal(got, ref, what):
    for k in ref:
        assert np.array_equal(got[k], ref[k]), "%s: field %s differs from the one-rank run (max|diff| %.3e)" % (what, k, float(np.abs(got[k] - ref[k]).max()))


def _tiled_supercell(tiles, nx, ny, nz, nens, ord=5, with_nudger=False):
    """(block of a periodic domain tiled from `tiles` = 4 | 8 copies of one nx x ny block, the one-rank handle of that block)."""
    from miniweatherml_amd import modules
    tx, ty = (2, 2) if tiles == 4 else (4, 2)
    xlen, ylen = 500.0 * nx, 500.0 * ny
    ref = modules.make_supercell(nx, ny, nz, nens, xlen, ylen, 20000., ord=ord, with_nudger=with_nudger)
    til = modules.make_supercell(nx * tx, ny * ty, nz, nens, xlen * tx, ylen * ty, 20000., nranks=tiles, myrank=0, ord=ord, with_nudger=with_nudger)
    g = til[0].grid
    assert (g.nproc_x, g.nproc_y, g.nx, g.ny, g.px, g.py) == (tx, ty, nx, ny, 0, 0)
    return til, ref


@pytest.mark.parametrize("tiles", [4, 8])
@pytest.mark.parametrize("schedule", ["pipelined", "two_stream"])
@pytest.mark.parametrize("case", ["supercell", "supercell_nens4", "supercell_ord3", "supercell_nens3"])
def test_rccl_self_loop_equals_one_rank(mw, monkeypatch, case, schedule, tiles):
    """K = 1 (supercell), member-major with the members-in-one-workgroup launches (4 members) and the plain member-major pass (3 members),
    WENO-3; the pipelined schedule with the split y launch (36-row blocks) and the two-stream schedule (two lanes on one communicator)."""
    from miniweatherml_amd import modules
    if schedule == "two_stream":
        set_options(monkeypatch, pipe=0)
    nens = 4 if case.endswith("nens4") else 3 if case.endswith("nens3") else 1
    ord = 3 if case.endswith("ord3") else 5
    nx, ny, nz = (70, 36, 10) if nens == 1 else (66, 36, 8)
    (tc, td, _), (rc, rd, _) = _tiled_supercell(tiles, nx, ny, nz, nens, ord)
    if nens > 1:                                               # members differ
        rc.get_data_manager_readwrite().get("temp").add_(0.05 * torch.arange(nens, device=rc.device, dtype=torch.float64))
    rc.get_data_manager_readwrite().get("cloud_liquid").fill_(2.0e-4)
    _copy_state(rc, tc)
    modules.use_rccl_self_exchange(td, tc)
    assert td.rccl_info()[0] == 1                              # a 1-rank communicator plays all ranks
    dt = rd.compute_time_step(rc)
    assert dt == td.compute_time_step(tc)
    for n in range(3):
        f = 2.3 if n == 1 else 1.0                             # step 1 is sub-cycled (three cycles: the exchange pattern of `!last`)
        rd.time_step(rc, dt * f)
        td.time_step(tc, dt * f)
    code = td.schedule()["code"]
    assert (code & 3) == (2 if schedule == "pipelined" else 1) and not (code & 8), td.schedule()
    assert (rd.schedule()["code"] & 3) == 0
    _assert_equal(gpu_fields(tc), gpu_fields(rc), "rccl self-loop %s %s %d tiles" % (case, schedule, tiles))
    assert float(np.abs(gpu_fields(rc)["vvel"]).max()) > 0.0


@pytest.mark.parametrize("tiles", [4, 8])
@pytest.mark.parametrize("schedule", ["pipelined", "two_stream"])
def test_rccl_self_loop_simple_city(mw, monkeypatch, schedule, tiles):
    """K = 2 (simple_city: immersed buildings, gravity off, water vapour only) -- the building block's immersed proportion is copied too."""
    from miniweatherml_amd import modules
    if schedule == "two_stream":
        set_options(monkeypatch, pipe=0)
    tx, ty = (2, 2) if tiles == 4 else (4, 2)
    nx, ny, nz = 64, 40, 12
    xlen, ylen, zlen = 5.0 * nx, 5.0 * ny, 5.0 * nz
    rc, rd, _, _ = modules.make_simple_city(nx, ny, nz, 1, xlen, ylen, zlen, "building")
    tc, td, _, _ = modules.make_simple_city(nx * tx, ny * ty, nz, 1, xlen * tx, ylen * ty, zlen, "building", nranks=tiles, myrank=0)
    _copy_state(rc, tc)
    imm = rd.immersed_proportion(rc)
    assert float(imm.max()) == 1.0
    td.immersed_proportion(tc).copy_(imm)
    modules.use_rccl_self_exchange(td, tc)
    dt = rd.compute_time_step(rc)
    for n in range(3):
        rd.time_step(rc, dt)
        td.time_step(tc, dt)
    assert (td.schedule()["code"] & 3) == (2 if schedule == "pipelined" else 1)
    _assert_equal(gpu_fields(tc), gpu_fields(rc), "rccl self-loop simple_city %s %d tiles" % (schedule, tiles))


@pytest.mark.parametrize("tiles", [4, 8])
def test_rccl_self_loop_full_supercell_loop(mw, tiles):
    """The complete supercell_example loop (driver.cpp:66-79: dycore, Kessler, sponge_layer, ColumnNudger) on the tiled block: halo
    exchange over RCCL point-to-point and the column modules' MPI_Allreduce over the SAME communicator (mw_dycore_rccl_allreduce_sum;
    sponge_layer.h:53-63, column_nudging.h:89-99), 6 steps, bitwise against one rank."""
    from miniweatherml_amd import modules
    (tc, td, tm, tn), (rc, rd, rm, rn) = _tiled_supercell(tiles, 70, 36, 24, 1, with_nudger=True)
    dm = rc.get_data_manager_readwrite()
    dm.get("cloud_liquid").fill_(3.0e-4).mul_(dm.get("density_dry"))          # Kessler has work from the first step
    dm.get("precip_liquid").fill_(1.0e-4).mul_(dm.get("density_dry"))
    _copy_state(rc, tc)
    modules.use_rccl_self_exchange(td, tc)
    rn.set_column(rc)
    tn.set_column(tc)                                          # the all-reduce of the 4 / 8 identical blocks' sums, over RCCL
    assert torch.equal(tn.column, rn.column)
    for _ in range(6):
        modules.supercell_step(rc, rd, rm, rn)
        modules.supercell_step(tc, td, tm, tn)
    _assert_equal(gpu_fields(tc), gpu_fields(rc), "rccl self-loop full loop %d tiles" % tiles)


@pytest.mark.parametrize("seed", range(1, 13))
def test_rccl_self_loop_with_delay_fuzz(mw, monkeypatch, seed):
    """Option xchg_fuzz: spin kernels of seeded random length (0-255 us) on the side stream in front of the ncclGroup and between it and
    ev_done -- the strips leave late and are reported late, stage after stage, while the compute stream runs on.  A kernel that read a halo
    without waiting for its exchange, or a pack that overwrote a strip still in flight, changes bits."""
    from miniweatherml_amd import modules
    tiles = 4 if seed % 2 else 8
    set_options(monkeypatch, xchg_fuzz=seed, pipe=0 if seed % 3 == 0 else 1, rccl_lanes=1 if seed % 4 == 0 else 0,
                rccl_inline=seed % 2)                          # even seeds: the group on the transport's side stream (event hand-overs both ways)
    (tc, td, _), (rc, rd, _) = _tiled_supercell(tiles, 70, 36, 10, 1)
    rc.get_data_manager_readwrite().get("cloud_liquid").fill_(2.0e-4)
    _copy_state(rc, tc)
    modules.use_rccl_self_exchange(td, tc)
    dt = rd.compute_time_step(rc)
    for n in range(4):
        f = 2.3 if n == 2 else 1.0
        rd.time_step(rc, dt * f)
        td.time_step(tc, dt * f)
    _assert_equal(gpu_fields(tc), gpu_fields(rc), "rccl self-loop fuzz seed %d" % seed)


def test_self_loop_negative_control(mw):
    """The comparison has teeth: the tiled block stepped WITHOUT its neighbours' strips refreshed (a transport that delivers nothing)
    does not reproduce the one-rank run."""
    from miniweatherml_amd import capi, modules
    (tc, td, _), (rc, rd, _) = _tiled_supercell(4, 70, 36, 10, 1)
    _copy_state(rc, tc)
    cb = capi.EXCHANGE_FN(lambda *a: 0)
    capi.check(capi.lib().mw_dycore_set_exchange(td.h, cb, None))
    dt = rd.compute_time_step(rc)
    for _ in range(2):
        rd.time_step(rc, dt)
        td.time_step(tc, dt)
    a, b = gpu_fields(tc), gpu_fields(rc)
    assert not np.array_equal(a["uvel"], b["uvel"])


# ---------------------------------------------------------------------------------------------------------------------
# R handles in R threads, stream-ordered in-process transport with delay fuzz
# ---------------------------------------------------------------------------------------------------------------------
_REF = {}


def _one_rank_reference(nxg, nyg, nz, nsteps, pipe):
    key = (nxg, nyg, nz, nsteps)
    if key not in _REF:
        from miniweatherml_amd import modules
        coupler, dycore, _ = modules.make_supercell(nxg, nyg, nz, 1, 500.0 * nxg, 500.0 * nyg, 20000.)
        coupler.get_data_manager_readwrite().get("cloud_liquid").fill_(2.0e-4)
        dt = dycore.compute_time_step(coupler)
        for n in range(nsteps):
            dycore.time_step(coupler, dt * (2.3 if n == 1 else 1.0))
        _REF[key] = gpu_fields(coupler)
    return _REF[key]


@pytest.mark.parametrize("seed", range(1, 51))
@pytest.mark.parametrize("layout", [(4, 48, 64), (8, 96, 40)])
def test_stream_ordered_exchange_with_delay_fuzz(mw, monkeypatch, layout, seed):
    """2 x 2 and 4 x 2 ranks as handles in threads of this process; strips travel by hipMemcpyAsync on per-rank side streams behind the
    peers' post-pack events with random delays in front of and behind the copies; the gathered result equals the one-rank run bitwise.
    Every third seed runs the two-stream schedule, every fifth the pipelined one with its edge strips inline."""
    from miniweatherml_amd import capi, modules
    nranks, nxg, nyg = layout
    nz, nsteps = 8, 3
    pipe = 0 if seed % 3 == 0 else 1
    set_options(monkeypatch, pipe=pipe, pipe_edge_inline=1 if seed % 5 == 0 else 0)
    ex = StreamExchanger(nranks, fuzz_seed=seed)
    results, keep = [None] * nranks, []

    def worker(rank):
        try:
            coupler, dycore, _ = modules.make_supercell(nxg, nyg, nz, 1, 500.0 * nxg, 500.0 * nyg, 20000., nranks=nranks, myrank=rank)
            coupler.get_data_manager_readwrite().get("cloud_liquid").fill_(2.0e-4)
            cb = ex.make_cb(rank, coupler.grid)
            keep.append(cb)
            capi.check(capi.lib().mw_dycore_set_exchange(dycore.h, cb, None))
            dt = dycore.compute_time_step(coupler)
            for n in range(nsteps):
                dycore.time_step(coupler, dt * (2.3 if n == 1 else 1.0))
            torch.cuda.synchronize()
            results[rank] = (coupler.grid.i_beg, coupler.grid.j_beg, gpu_fields(coupler), dycore.schedule()["code"])
        except Exception as e:                                          # pragma: no cover
            ex.errors.append("rank %d: %r" % (rank, e))
            ex.bar.abort()

    ths = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    assert not ex.errors, ex.errors
    torch.cuda.synchronize()
    ex.close()
    ref = _one_rank_reference(nxg, nyg, nz, nsteps, pipe)
    for ib, jb, blk, code in results:
        assert (code & 3) == (2 if pipe else 1)
        for k, a in blk.items():
            ny, nx = a.shape[1], a.shape[2]
            assert np.array_equal(a, ref[k][:, jb:jb + ny, ib:ib + nx]), (k, ib, jb, seed)

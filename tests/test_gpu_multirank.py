"""The slab-decomposed path on ONE GPU: R ranks = R handles in R host threads, each owning its block of the 2-D
decomposition; the halo-exchange callback (mw_dycore_set_exchange) moves the packed 3-cell strips between the
handles' device buffers.  This exercises pack/unpack, the rank grid, the neighbour matrix and the two-stream
pipeline exactly as the RCCL transport does (only the transport differs), and the gathered result must equal the
single-rank GPU run BITWISE (decomposition invariance)."""
import ctypes as C
import threading

import numpy as np
import pytest
import torch

from util import Exchanger, OracleExchanger, compare_fields, gpu_fields, push_fields, set_options

pytestmark = pytest.mark.gpu


def run_ranks(nranks, nxg, nyg, nz, nens, nsteps, ord=5):
    from miniweatherml_amd import capi, modules
    xlen, ylen = 500.0 * nxg, (500.0 * nyg if nyg > 1 else 1.0e5)
    ex = Exchanger(nranks)
    results = [None] * nranks
    keep = []

    def worker(rank):
        try:
            coupler, dycore, _ = modules.make_supercell(nxg, nyg, nz, nens, xlen, ylen, 20000., nranks=nranks, myrank=rank, ord=ord)
            cb = ex.make_cb(rank, coupler.grid)
            keep.append(cb)
            capi.check(capi.lib().mw_dycore_set_exchange(dycore.h, cb, None))
            dt = dycore.compute_time_step(coupler)
            for _ in range(nsteps):
                dycore.time_step(coupler, dt)
            torch.cuda.synchronize()
            results[rank] = (coupler.grid.i_beg, coupler.grid.j_beg, gpu_fields(coupler))
        except Exception as e:                                          # pragma: no cover
            ex.errors.append("rank %d: %r" % (rank, e))
            ex.bar.abort()

    ths = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    assert not ex.errors, ex.errors
    # single-rank reference on the same GPU
    coupler, dycore, _ = modules.make_supercell(nxg, nyg, nz, nens, xlen, ylen, 20000., ord=ord)
    dt = dycore.compute_time_step(coupler)
    for _ in range(nsteps):
        dycore.time_step(coupler, dt)
    ref = gpu_fields(coupler)
    for ib, jb, blk in results:
        for k, a in blk.items():
            ny, nx = a.shape[1], a.shape[2]
            assert np.array_equal(a, ref[k][:, jb:jb + ny, ib:ib + nx]), (k, ib, jb)


def test_two_ranks_3d(mw):
    run_ranks(2, 24, 32, 12, 1, 3)          # 1x2: south == north peer


def test_four_ranks_2x2(mw):
    run_ranks(4, 32, 32, 10, 1, 2)


@pytest.mark.parametrize("nranks", [2, 4])
def test_blocks_of_15_and_16_rows_take_the_same_first_stage(mw, nranks):
    """ny_glob = 31 on two y ranks: blocks of 16 and 15 rows, either side of the 4 * MW_Y_EDGE threshold that decides whether the first
    stage of the pipelined schedule splits its exchange (maps_early: 4 callbacks instead of 3).  The decision must be rank-uniform (the
    smallest block decides, as for the zero-row maps) -- ranks that post different exchange sequences would hang or corrupt halos."""
    run_ranks(nranks, 40, 31, 10, 1, 3)


def test_eight_ranks_4x2_nens2(mw):
    run_ranks(8, 48, 24, 8, 2, 2)


def test_four_ranks_2x2_nens4(mw):
    """config 4's shape in small: four members per block, the blocks of a 2 x 2 decomposition exchanging strips (the members of a tile
    share a workgroup in the last stage's kernels -- MemberOff, mw_march.h -- here with filled halos instead of the index wrap)."""
    run_ranks(4, 140, 72, 9, 4, 2)           # (36-row blocks: the pipelined schedule splits the y launch)


@pytest.mark.parametrize("ord", [3, 7, 9])
def test_four_ranks_other_weno_orders(mw, ord):
    """Orders 7 / 9 exchange 4- / 5-cell strips (hs + 1); the exchange is installed AFTER the order is set here, and the
    buffers follow the halo width either way (mw_dycore_set_order re-allocates them)."""
    run_ranks(4, 24, 20, 10, 1, 2, ord=ord)


def test_four_ranks_two_stream_schedule(mw, monkeypatch):
    """The default with an exchange is the pipelined one-stream schedule (rk_stage_pipe: k_y_all on the rows that read no halo row while
    the strips travel, then the two 4-row edge strips).  Option pipe = 0 selects the two-stream schedule (state | tracer pipelines,
    k_y_state + k_y_tracers): same bits."""
    set_options(monkeypatch, pipe=0)
    run_ranks(4, 32, 72, 10, 1, 2)


def test_four_ranks_pipelined_with_edge_strips(mw):
    """Blocks of 36 rows: the y launch is split into the inner rows and the two MW_Y_EDGE-row strips (ny >= 4 * MW_Y_EDGE)."""
    run_ranks(4, 32, 72, 10, 1, 3)


def test_four_ranks_simple_city(mw):
    """BASELINE.json configs[4] in small: the simple_city configuration (immersed buildings, gravity off, water vapour only -- the
    folded configuration K = 2) on a 2 x 2 decomposition, pipelined schedule with the split y launch, bitwise against one rank."""
    from miniweatherml_amd import capi, modules
    nranks, nxg, nyg, nz, nsteps = 4, 64, 72, 12, 3
    xlen, ylen, zlen = 50.0 * nxg, 50.0 * nyg, 120.0
    ex = Exchanger(nranks)
    results = [None] * nranks
    keep = []

    def worker(rank):
        try:
            coupler, dycore, _, _ = modules.make_simple_city(nxg, nyg, nz, 1, xlen, ylen, zlen, "city", nranks=nranks, myrank=rank)
            cb = ex.make_cb(rank, coupler.grid)
            keep.append(cb)
            capi.check(capi.lib().mw_dycore_set_exchange(dycore.h, cb, None))
            dt = dycore.compute_time_step(coupler)
            for _ in range(nsteps):
                dycore.time_step(coupler, dt)
            torch.cuda.synchronize()
            results[rank] = (coupler.grid.i_beg, coupler.grid.j_beg, gpu_fields(coupler))
        except Exception as e:                                          # pragma: no cover
            ex.errors.append("rank %d: %r" % (rank, e))
            ex.bar.abort()

    ths = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    assert not ex.errors, ex.errors
    coupler, dycore, _, _ = modules.make_simple_city(nxg, nyg, nz, 1, xlen, ylen, zlen, "city")
    dt = dycore.compute_time_step(coupler)
    for _ in range(nsteps):
        dycore.time_step(coupler, dt)
    ref = gpu_fields(coupler)
    assert float(np.abs(ref["uvel"]).max()) > 1.0                      # the flow is there
    for ib, jb, blk in results:
        for k, a in blk.items():
            ny, nx = a.shape[1], a.shape[2]
            assert np.array_equal(a, ref[k][:, jb:jb + ny, ib:ib + nx]), (k, ib, jb)


def test_one_rank_with_exchange_installed_three_members(mw):
    """A 1 x 1 decomposition WITH a transport installed (both index wraps on, pipelined schedule selected): three members have no
    members-in-one-workgroup launch, so the handle must take the full conversion pass instead of the converting k_y_all (round 3's
    advisor finding: the K = 2 kernel was launched for this K = 0 configuration).  Bitwise against the handle without a transport."""
    run_ranks(1, 40, 24, 8, 3, 2)


def test_one_rank_with_exchange_installed_four_members(mw):
    """... and four members (members-in-one-workgroup form exists: D1 inside the first k_y_all of the pipelined schedule)."""
    run_ranks(1, 64, 40, 8, 4, 2)


def test_two_ranks_2d(mw):
    run_ranks(2, 64, 1, 16, 1, 3)           # 2x1: west == east peer


def test_rccl_transport_selftest(mw, monkeypatch):
    """The RCCL transport itself (mw_rccl.cpp) on one GPU: a 1-rank communicator sends the four strips to itself through the
    exchange's own ncclGroup / side-stream / event sequence (receives posted E,W,N,S against sends W,E,S,N)."""
    import torch
    from miniweatherml_amd import capi
    with torch.cuda.device(0):
        st = torch.cuda.current_stream().cuda_stream
        capi.check(capi.lib().mw_rccl_selftest(3 * 100 * 400 * 5, C.c_void_p(st)))          # one state strip of config 2
        capi.check(capi.lib().mw_rccl_selftest(7, C.c_void_p(st)))
        # both lanes -- the state pipeline's and the tracer pipeline's side stream + event pair -- were driven, from two caller streams:
        # on the handle's one communicator (default) ...
        assert capi.lib().mw_rccl_selftest_lanes() == 21
        # ... and with a communicator per lane (opt-in; torch's RCCL 2.26 provides ncclCommSplit)
        capi.check(capi.lib().mw_rccl_selftest_config(2, 1))
        try:
            capi.check(capi.lib().mw_rccl_selftest(4096, C.c_void_p(st)))
            assert capi.lib().mw_rccl_selftest_lanes() == 22
            capi.check(capi.lib().mw_rccl_selftest_config(1, 0))      # one lane: a third caller stream shares it
            capi.check(capi.lib().mw_rccl_selftest(4096, C.c_void_p(st)))
            assert capi.lib().mw_rccl_selftest_lanes() == 11
        finally:
            capi.check(capi.lib().mw_rccl_selftest_config(0, 0))


@pytest.mark.parametrize("layout", [(2, 20, 24, 10), (4, 32, 28, 8)])
def test_ranks_with_a_busy_limiter_match_the_multi_rank_oracle(mw, oracle, layout):
    """FCT across rank boundaries: a face on a rank edge is only ever scaled by a donor cell of the SAME rank (the reference
    never communicates multipliers), so a decomposed run legitimately differs from the single-rank run once the limiter is
    active there.  The decomposed GPU run must equal the equally decomposed oracle run."""
    from miniweatherml_amd import capi, modules
    nranks, nxg, nyg, nz = layout
    xlen, ylen = 500.0 * nxg, 500.0 * nyg
    # one global rough state, cut into the ranks' blocks
    rng = np.random.default_rng(77)
    glob = {}
    shape = (nz, nyg, nxg, 1)
    glob["u"] = 20.0 * rng.uniform(-1, 1, shape); glob["v"] = 20.0 * rng.uniform(-1, 1, shape); glob["w"] = 5.0 * rng.uniform(-1, 1, shape)
    for t in (1, 2):
        blob = rng.uniform(size=shape)
        glob["tr%d" % t] = np.where(blob > 0.7, 2e-3 * rng.uniform(size=shape), 0.0)
    oranks, plans = [], []
    for r in range(nranks):
        odyc, of = oracle.supercell_setup(nxg, nyg, nz, 1, xlen, ylen, 20000., nranks=nranks, rank=r)
        p = odyc.p
        sl = (slice(None), slice(p.j_beg, p.j_beg + p.ny), slice(p.i_beg, p.i_beg + p.nx))
        of.uvel += glob["u"][sl]; of.vvel += glob["v"][sl]; of.wvel += glob["w"][sl]
        of.tracers[1][...] = glob["tr1"][sl]; of.tracers[2][...] = glob["tr2"][sl]
        g = capi.Grid()
        capi.check(capi.lib().mw_decompose(nranks, r, nxg, nyg, C.byref(g)))
        peers, so, ro, act = (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)()
        capi.check(capi.lib().mw_exchange_plan(C.byref(g), peers, so, ro, act))
        oranks.append((odyc, of))
        plans.append((list(peers), list(act)))
    oex = OracleExchanger(nranks, plans)
    gex = Exchanger(nranks)
    nsteps = 2
    gpu_out, errors, keep = [None] * nranks, [], []

    def oracle_worker(r):
        try:
            odyc, of = oranks[r]
            odyc.set_exchange(oex.make_cb(r))
            dt = odyc.compute_time_step()
            for _ in range(nsteps):
                odyc.time_step(of, dt)
        except Exception as e:                                  # pragma: no cover
            errors.append("oracle rank %d: %r" % (r, e)); oex.bar.abort()

    def gpu_worker(r):
        try:
            coupler, dycore, _ = modules.make_supercell(nxg, nyg, nz, 1, xlen, ylen, 20000., nranks=nranks, myrank=r)
            odyc, of0 = oracle.supercell_setup(nxg, nyg, nz, 1, xlen, ylen, 20000., nranks=nranks, rank=r)
            p = odyc.p
            sl = (slice(None), slice(p.j_beg, p.j_beg + p.ny), slice(p.i_beg, p.i_beg + p.nx))
            of0.uvel += glob["u"][sl]; of0.vvel += glob["v"][sl]; of0.wvel += glob["w"][sl]
            of0.tracers[1][...] = glob["tr1"][sl]; of0.tracers[2][...] = glob["tr2"][sl]
            push_fields(coupler, of0)
            cb = gex.make_cb(r, coupler.grid)
            keep.append(cb)
            capi.check(capi.lib().mw_dycore_set_exchange(dycore.h, cb, None))
            dt = dycore.compute_time_step(coupler)
            for _ in range(nsteps):
                dycore.time_step(coupler, dt)
            torch.cuda.synchronize()
            gpu_out[r] = gpu_fields(coupler)
        except Exception as e:                                  # pragma: no cover
            errors.append("gpu rank %d: %r" % (r, e)); gex.bar.abort()

    for worker in (oracle_worker, gpu_worker):
        ths = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
        [t.start() for t in ths]
        [t.join(300) for t in ths]
        assert not errors and not gex.errors, (errors, gex.errors)
    for r in range(nranks):
        compare_fields(gpu_out[r], oranks[r][1].as_dict(), 1e-10, "rank %d of %d, busy limiter" % (r, nranks))


def _run_decomposed_gpu_and_oracle(oracle, nranks, nxg, nyg, nz, init, bc, mode, nsteps, xlen, ylen, zlen):
    """R product handles (threads, one GPU) and R oracle ranks (threads) on the same decomposition, boundary types and inputs."""
    from miniweatherml_amd import capi, modules
    oranks, plans = [], []
    for r in range(nranks):
        odyc, of = oracle.supercell_setup(nxg, nyg, nz, 1, xlen, ylen, zlen, init_data=init, perturb=(init == "supercell"),
                                          nranks=nranks, rank=r)
        odyc.p.bc_x, odyc.p.bc_y, odyc.p.bc_z = bc
        g = capi.Grid()
        capi.check(capi.lib().mw_decompose(nranks, r, nxg, nyg, C.byref(g)))
        peers, so, ro, act = (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)()
        capi.check(capi.lib().mw_exchange_plan(C.byref(g), peers, so, ro, act))
        oranks.append((odyc, of))
        plans.append((list(peers), list(act)))
    oex, gex = OracleExchanger(nranks, plans), Exchanger(nranks)
    gpu_out, errors, keep = [None] * nranks, [], []
    inputs = [of.copy() for _, of in oranks]

    def oracle_worker(r):
        try:
            odyc, of = oranks[r]
            odyc.set_exchange(oex.make_cb(r))
            dt = odyc.compute_time_step()
            for _ in range(nsteps):
                odyc.time_step(of, dt)
        except Exception as e:                                  # pragma: no cover
            errors.append("oracle rank %d: %r" % (r, e)); oex.bar.abort()

    def gpu_worker(r):
        try:
            coupler, dycore, _ = modules.make_supercell(nxg, nyg, nz, 1, xlen, ylen, zlen, init, nranks=nranks, myrank=r,
                                                        perturb=(init == "supercell"))
            push_fields(coupler, inputs[r])
            dycore.set_strict(mode)
            dycore.set_bc(coupler, *bc)
            cb = gex.make_cb(r, coupler.grid)
            keep.append(cb)
            capi.check(capi.lib().mw_dycore_set_exchange(dycore.h, cb, None))
            dt = dycore.compute_time_step(coupler)
            for _ in range(nsteps):
                dycore.time_step(coupler, dt)
            torch.cuda.synchronize()
            gpu_out[r] = gpu_fields(coupler)
        except Exception as e:                                  # pragma: no cover
            errors.append("gpu rank %d: %r" % (r, e)); gex.bar.abort()

    for worker in (oracle_worker, gpu_worker):
        ths = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
        [t.start() for t in ths]
        [t.join(300) for t in ths]
        assert not errors and not gex.errors, (errors, gex.errors)
    return gpu_out, [of.as_dict() for _, of in oranks]


@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("bc", [(2, 2, 2), (1, 1, 2), (2, 0, 2), (0, 1, 1)])
@pytest.mark.parametrize("nranks", [2, 4])
def test_wall_and_open_boundaries_on_several_ranks(mw, oracle, nranks, bc, mode):
    """Wall (2) / open (1) x and y boundaries with MORE THAN ONE rank in that direction (:782-825 halo rule on the `px == 0` /
    `px == nproc_x-1` ranks only, :1040-1081 edge rule; no `else if` quirk once nproc > 1): every rank of the decomposed GPU run
    must match the same rank of the equally decomposed oracle run.  (2 ranks = 1x2, 4 ranks = 2x2; thermal-bubble case as in the
    single-rank boundary test, hence the allow-listed sensitivity yardstick.)"""
    from util import oracle_sensitivity
    make = lambda: oracle.supercell_setup(16, 16, 16, 1, 20000., 20000., 10000., init_data="thermal", perturb=False)   # noqa: E731
    sens = oracle_sensitivity(oracle, "thermal3d_16x16x16", make, (3,))
    got, want = _run_decomposed_gpu_and_oracle(oracle, nranks, 16, 16, 16, "thermal", bc, mode, 3, 20000., 20000., 10000.)
    for r in range(nranks):
        compare_fields(got[r], want[r], 1e-10, "bc %s mode %d, rank %d of %d" % (bc, mode, r, nranks), sens[3])


@pytest.mark.parametrize("layout", [(2, 48, 1, 12, (2, 0, 2)), (2, 24, 20, 8, (0, 2, 2)), (4, 24, 20, 8, (1, 2, 2)), (8, 48, 24, 8, (2, 2, 1))])
def test_boundaries_on_several_ranks_supercell(mw, oracle, layout):
    """The same on the supercell case (plain 1e-10, no sensitivity fallback): 2-D with two ranks in x, 3-D 1x2, 2x2 and 4x2."""
    nranks, nxg, nyg, nz, bc = layout
    xlen, ylen = 500.0 * nxg, (500.0 * nyg if nyg > 1 else 1.0e5)
    got, want = _run_decomposed_gpu_and_oracle(oracle, nranks, nxg, nyg, nz, "supercell", bc, 0, 2, xlen, ylen, 20000.)
    for r in range(nranks):
        compare_fields(got[r], want[r], 1e-10, "supercell bc %s, rank %d of %d" % (bc, r, nranks))


def _specks(coupler, nxg, nyg, nz, extra=()):
    """Single non-zero cells of cloud water / rain at GLOBAL positions: in the domain's corners, and one or two cells either side of the
    seams of a 2 x 2 / 4 x 2 / 1 x 2 decomposition -- what a block's tracer kernel sees of them arrives through its halos."""
    dm = coupler.get_data_manager_readwrite()
    rho = dm.get("density_dry")
    ib, jb = coupler.grid.i_beg, coupler.grid.j_beg
    ny, nx = rho.shape[1], rho.shape[2]
    cl, pr = torch.zeros_like(rho), torch.zeros_like(rho)
    cloud = [(0, 0, 0), (nz - 1, nyg - 1, nxg - 1), (3, nyg // 2 - 1, nxg // 2 - 1), (4, nyg // 2, nxg // 2 + 1), (2, nyg // 2 + 2, 5),
             (nz - 2, 7, nxg // 2), (nz // 2, nyg - 1, nxg // 4), (1, nyg // 2 - 2, nxg - 1)]
    rain = [(nz // 2, nyg // 2, nxg // 4 - 1), (0, nyg - 2, nxg // 2 - 2), (nz - 1, 1, 3 * nxg // 4)]
    cloud += [e[:3] for e in extra if e[3] == 0]
    rain += [e[:3] for e in extra if e[3] == 1]
    for lst, fld, val in ((cloud, cl, 2.0e-4), (rain, pr, 1.0e-4)):
        for (k, j, i) in lst:
            if jb <= j < jb + ny and ib <= i < ib + nx:
                fld[k, j - jb, i - ib] = val
    dm.get("cloud_liquid").copy_(cl * rho)
    dm.get("precip_liquid").copy_(pr * rho)


def _specks_blocks(layout, zero_rows, fuzz, nz=12, extra=(), factors=(1.0, 2.3, 1.0)):
    from miniweatherml_amd import capi, modules
    from util import StreamExchanger
    nranks, nxg, nyg = layout
    nsteps = len(factors)
    ex = StreamExchanger(nranks, fuzz_seed=fuzz)
    results, keep = [None] * nranks, []

    def worker(rank):
        try:
            coupler, dycore, _ = modules.make_supercell(nxg, nyg, nz, 1, 500.0 * nxg, 500.0 * nyg, 20000., nranks=nranks, myrank=rank)
            _specks(coupler, nxg, nyg, nz, extra)
            dycore.set_option("zero_rows", zero_rows)
            dycore.set_option("zero_verify", zero_rows)              # (every claim of the maps against the data, on every block)
            cb = ex.make_cb(rank, coupler.grid)
            keep.append(cb)
            capi.check(capi.lib().mw_dycore_set_exchange(dycore.h, cb, None))
            dt = dycore.compute_time_step(coupler)
            for n in range(nsteps):
                dycore.time_step(coupler, dt * factors[n])
            torch.cuda.synchronize()
            results[rank] = (coupler.grid.i_beg, coupler.grid.j_beg, gpu_fields(coupler), dycore.path())
            if zero_rows:
                nviol, kinds = dycore.zero_violations()
                if nviol != 0:
                    ex.errors.append("rank %d: zero_verify counted %r (total %d)" % (rank, kinds, nviol))
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
    return results


@pytest.mark.parametrize("seed", range(12))
def test_zero_row_maps_on_random_decompositions(mw, seed):
    """Seeded sweep: 2, 4 or 8 blocks of random (uneven) extents, random levels, cloud / rain cells at random global positions on top of the
    seam cells, random sub-cycling, random transport delays -- the maps on against the maps off on the same decomposition: same bits."""
    rng = np.random.default_rng(500 + seed)
    nranks = int(rng.choice([2, 4, 8]))
    nxg, nyg, nz = int(rng.integers(76, 170)), int(rng.integers(40, 100)), int(rng.integers(6, 20))
    extra = [(int(rng.integers(0, nz)), int(rng.integers(0, nyg)), int(rng.integers(0, nxg)), int(rng.integers(0, 2))) for _ in range(int(rng.integers(0, 12)))]
    factors = (1.0, float(rng.choice([1.0, 2.3])), 1.0)
    on = _specks_blocks((nranks, nxg, nyg), 1, 11 + seed, nz=nz, extra=extra, factors=factors)
    off = _specks_blocks((nranks, nxg, nyg), 0, 0, nz=nz, extra=extra, factors=factors)
    for (ib, jb, blk, path), (ib0, jb0, ref, _) in zip(on, off):
        assert (ib, jb) == (ib0, jb0)
        for k, a in blk.items():
            assert np.array_equal(a, ref[k]), (k, ib, jb, seed, nranks, nxg, nyg, nz)


@pytest.mark.parametrize("layout", [(4, 96, 64), (8, 160, 40), (2, 40, 72), (2, 40, 31), (4, 96, 31)])
def test_zero_row_maps_on_decomposed_blocks(mw, layout):
    """The zero-row maps (mw_march.h: k_zero_rows) of a decomposed domain: every block ORs its neighbours' maps into its own (k_zero_merge:
    west / east, whole rows; k_zero_halo: the south / north neighbours' edge rows, two small messages per sub-cycle through the halo
    transport).  Cloud and rain are single cells next to the seams and in the corners, so whether a block's wave may skip a row is decided
    by what the NEIGHBOUR holds; the transport delays its copies at random.  Same bits as the same decomposition WITHOUT the maps; three
    steps, one of them sub-cycled.  (Not compared with the one-rank run: single cells keep the positivity limiter busy, and a flux that
    enters a block from its halo is never scaled -- the reference's FCT loops over a rank's own cells, dynamics_euler_stratified_wenofv.h
    :496-514 -- so with an active limiter the result depends on where the seams are, there as here.)"""
    from util import launched_kernels
    launched_kernels(reset=True)
    on = _specks_blocks(layout, 1, 7)
    assert any("k_zero_merge" in k for k in launched_kernels(reset=True))
    off = _specks_blocks(layout, 0, 0)
    assert not any("k_zero_rows" in k for k in launched_kernels(reset=False))
    nonzero = cells = 0
    for (ib, jb, blk, path), (ib0, jb0, ref, _) in zip(on, off):
        assert " pipe " in path and (ib, jb) == (ib0, jb0)
        for k, a in blk.items():
            assert np.array_equal(a, ref[k]), (k, ib, jb)
        nonzero += int((ref["tracer1"] != 0).sum()); cells += ref["tracer1"].size
    assert 0 < nonzero < (0.5 if layout[2] >= 40 else 1.0) * cells      # (the 31-row layouts are small: the seam cells' cloud reaches most of them)

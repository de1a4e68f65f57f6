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

from util import compare_fields, gpu_fields, push_fields

pytestmark = pytest.mark.gpu


class Exchanger:
    """All ranks live in this process; strips are copied device-to-device after a barrier."""

    def __init__(self, nranks):
        self.n = nranks
        self.bar = threading.Barrier(nranks)
        self.send = [None] * nranks
        self.errors = []

    def make_cb(self, rank, grid):
        from miniweatherml_amd import capi
        peers, so, ro, act = (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)()
        capi.check(capi.lib().mw_exchange_plan(C.byref(grid), peers, so, ro, act))
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]

        def cb(ctx, sW, sE, sS, sN, rW, rE, rS, rN, nWE, nSN, stream):
            try:
                hip.hipStreamSynchronize(stream)                          # my strips are packed
                self.send[rank] = (sW, sE, sS, sN)
                self.bar.wait(timeout=60)
                cnt = [nWE, nWE, nSN, nSN]
                recv = [rW, rE, rS, rN]
                for d in range(4):                                      # my halo d comes from peer[d]'s opposite strip
                    if recv[d] and cnt[d] and act[d]:
                        src = self.send[peers[d]][d ^ 1]
                        assert hip.hipMemcpy(recv[d], src, cnt[d] * 8, 3) == 0      # hipMemcpyDeviceToDevice
                # a device-to-device hipMemcpy is not synchronous with the host, and the library's streams do not synchronise
                # with the null stream it runs on: finish the copies before the unpack kernels are enqueued
                assert hip.hipStreamSynchronize(None) == 0
                self.bar.wait(timeout=60)
                return 0
            except Exception as e:                                      # pragma: no cover
                self.errors.append(repr(e))
                self.bar.abort()
                return 1
        return capi.EXCHANGE_FN(cb)


def run_ranks(nranks, nxg, nyg, nz, nens, nsteps):
    from miniweatherml_amd import capi, modules
    xlen, ylen = 500.0 * nxg, (500.0 * nyg if nyg > 1 else 1.0e5)
    ex = Exchanger(nranks)
    results = [None] * nranks
    keep = []

    def worker(rank):
        try:
            coupler, dycore, _ = modules.make_supercell(nxg, nyg, nz, nens, xlen, ylen, 20000., nranks=nranks, myrank=rank)
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
    coupler, dycore, _ = modules.make_supercell(nxg, nyg, nz, nens, xlen, ylen, 20000.)
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


def test_eight_ranks_4x2_nens2(mw):
    run_ranks(8, 48, 24, 8, 2, 2)


def test_two_ranks_2d(mw):
    run_ranks(2, 64, 1, 16, 1, 3)           # 2x1: west == east peer


def test_rccl_transport_selftest(mw):
    """The RCCL transport itself (mw_rccl.cpp) on one GPU: a 1-rank communicator sends the four strips to itself through the
    exchange's own ncclGroup / side-stream / event sequence (receives posted E,W,N,S against sends W,E,S,N)."""
    import torch
    from miniweatherml_amd import capi
    with torch.cuda.device(0):
        st = torch.cuda.current_stream().cuda_stream
        capi.check(capi.lib().mw_rccl_selftest(3 * 100 * 400 * 5, C.c_void_p(st)))          # one state strip of config 2
        capi.check(capi.lib().mw_rccl_selftest(7, C.c_void_p(st)))


class OracleExchanger:
    """The oracle's halo/edge exchange callback between oracle ranks living in threads of this process."""

    def __init__(self, nranks, plans):
        self.n, self.plans = nranks, plans
        self.bar = threading.Barrier(nranks)
        self.send = [None] * nranks

    def make_cb(self, rank):
        peers, act = self.plans[rank]

        def cb(ctx, kind, sW, sE, sS, sN, rW, rE, rS, rN, nWE, nSN):
            cnt = [nWE, nWE, nSN, nSN]
            sb, rb = [sW, sE, sS, sN], [rW, rE, rS, rN]
            self.send[rank] = [np.ctypeslib.as_array(sb[d], shape=(cnt[d],)).copy() if cnt[d] else None for d in range(4)]
            self.bar.wait(timeout=120)
            for d in range(4):
                if cnt[d]:
                    src = self.send[peers[d]][d ^ 1] if act[d] else self.send[rank][d ^ 1]      # single rank in a direction: self wrap
                    np.ctypeslib.as_array(rb[d], shape=(cnt[d],))[:] = src
            self.bar.wait(timeout=120)
        return cb


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

"""Cost of the halo-exchange machinery on ONE GPU: rank 0's block of a 2- or 4-rank decomposition, 400x400x100 cells, with a
loop-back transport (every strip this rank sends is copied, device to device and stream-ordered, into its own opposite halo --
what a periodic neighbour with identical data would send).  Pack, unpack, the copies and the two-stream schedule are all there;
only the xGMI transfer itself is missing.  Compared with the plain single-rank run of the same block (index wrap, one stream).
Round 5: next to the memcpy loop-back the REAL transport -- mw_dycore_use_rccl_self: ncclSend / ncclRecv groups on the side stream of a
1-rank communicator, every peer this rank itself -- on the block's own state tiled periodically (physically the one-rank run).
    python tools/exchange_overhead.py [--steps 20] [--pipe 0|1]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from miniweatherml_amd import capi, modules

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--nx", type=int, default=400); ap.add_argument("--ny", type=int, default=400); ap.add_argument("--nz", type=int, default=100)
ap.add_argument("--pipe", type=int, default=1, help="0: the two-stream schedule instead of the pipelined one")
ap.add_argument("--opt", action="append", default=[], help="key=value handle option (repeatable), e.g. --opt rccl_prio=0")
a = ap.parse_args()
modules.DEFAULT_OPTIONS["pipe"] = a.pipe
for kv in a.opt:
    modules.DEFAULT_OPTIONS[kv.split("=")[0]] = int(kv.split("=")[1])
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]


def loopback():
    def cb(ctx, sW, sE, sS, sN, rW, rE, rS, rN, nWE, nSN, stream):
        for src, dst, n in ((sW, rE, nWE), (sE, rW, nWE), (sS, rN, nSN), (sN, rS, nSN)):
            if src and dst and n:
                if hip.hipMemcpyAsync(dst, src, n * 8, 3, stream) != 0:
                    return 1
        return 0
    return capi.EXCHANGE_FN(cb)


def timed(dycore, coupler, dt, steps):
    for _ in range(3):
        dycore.time_step(coupler, dt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        dycore.time_step(coupler, dt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


res = {"schedule": "pipelined" if a.pipe else "two streams", "options": dict(modules.DEFAULT_OPTIONS)}
coupler, dycore, _ = modules.make_supercell(a.nx, a.ny, a.nz, 1, 500.0 * a.nx, 500.0 * a.ny, 20000.)
dt = dycore.compute_time_step(coupler)
start = {n: coupler.get_data_manager_readonly().get(n, True).clone() for n in ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "cloud_liquid", "precip_liquid")}
res["single_rank_ms"] = timed(dycore, coupler, dt, a.steps)
keep = []
for nranks in (2, 4, 8):
    g = capi.Grid()
    capi.check(capi.lib().mw_decompose(nranks, 0, a.nx * nranks, a.ny * nranks, C.byref(g)))
    npx, npy = g.nproc_x, g.nproc_y
    coupler, dycore, _ = modules.make_supercell(a.nx * npx, a.ny * npy, a.nz, 1, 500.0 * a.nx * npx, 500.0 * a.ny * npy, 20000.,
                                                nranks=nranks, myrank=0)
    assert coupler.get_nx() == a.nx and coupler.get_ny() == a.ny
    cb = loopback(); keep.append(cb)
    capi.check(capi.lib().mw_dycore_set_exchange(dycore.h, cb, None))
    ms = timed(dycore, coupler, dt, a.steps)
    res["%d_ranks_%dx%d_ms" % (nranks, npx, npy)] = ms
    res["%d_ranks_overhead" % nranks] = ms / res["single_rank_ms"] - 1.0
    # the real RCCL transport, every peer = this rank, on the one-rank block's own initial state (a periodic tiling of it)
    coupler, dycore, _ = modules.make_supercell(a.nx * npx, a.ny * npy, a.nz, 1, 500.0 * a.nx * npx, 500.0 * a.ny * npy, 20000.,
                                                nranks=nranks, myrank=0)
    for n, t in start.items():
        coupler.get_data_manager_readwrite().get(n).copy_(t)
    modules.use_rccl_self_exchange(dycore, coupler)
    ms = timed(dycore, coupler, dt, a.steps)
    res["%d_ranks_%dx%d_rccl_self_ms" % (nranks, npx, npy)] = ms
    res["%d_ranks_rccl_self_overhead" % nranks] = ms / res["single_rank_ms"] - 1.0
    del coupler, dycore
    torch.cuda.empty_cache()
print(json.dumps(res))

import os, sys, threading
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import test_gpu_multirank as T
from util import StreamExchanger, gpu_fields
from miniweatherml_amd import capi, modules

def run(layout, nsteps, opts, which="specks"):
    nranks, nxg, nyg = layout
    nz = 12
    ex = StreamExchanger(nranks, fuzz_seed=0)
    results, keep = [None] * nranks, []
    def init(coupler):
        if which == "specks": T._specks(coupler, nxg, nyg, nz)
    def worker(rank):
        try:
            coupler, dycore, _ = modules.make_supercell(nxg, nyg, nz, 1, 500.0 * nxg, 500.0 * nyg, 20000., nranks=nranks, myrank=rank)
            init(coupler)
            for k, v in opts.items(): dycore.set_option(k, v)
            cb = ex.make_cb(rank, coupler.grid); keep.append(cb)
            capi.check(capi.lib().mw_dycore_set_exchange(dycore.h, cb, None))
            dt = dycore.compute_time_step(coupler)
            for n in range(nsteps): dycore.time_step(coupler, dt)
            torch.cuda.synchronize()
            results[rank] = (coupler.grid.i_beg, coupler.grid.j_beg, gpu_fields(coupler), dycore.path())
        except Exception as e:
            ex.errors.append("rank %d: %r" % (rank, e)); ex.bar.abort()
    ths = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
    [t.start() for t in ths]; [t.join(300) for t in ths]
    assert not ex.errors, ex.errors
    torch.cuda.synchronize(); ex.close()
    coupler, dycore, _ = modules.make_supercell(nxg, nyg, nz, 1, 500.0 * nxg, 500.0 * nyg, 20000.)
    init(coupler)
    dycore.set_option("zero_rows", 0)
    for k, v in opts.items():
        if k in ("zero_skip",): dycore.set_option(k, v)
    dt = dycore.compute_time_step(coupler)
    for n in range(nsteps): dycore.time_step(coupler, dt)
    ref = gpu_fields(coupler)
    out = []
    for ib, jb, blk, path in results:
        for k, a in blk.items():
            ny, nx = a.shape[1], a.shape[2]
            r = ref[k][:, jb:jb + ny, ib:ib + nx]
            if not np.array_equal(a, r):
                w = np.argwhere(a != r)
                out.append((k, ib, jb, len(w), [x.tolist()[:3] for x in w[:4]], float(np.abs(a - r).max())))
    print(layout, nsteps, opts, results[0][3], "->", "EQUAL" if not out else "", flush=True)
    for o in out[:12]: print("    ", o, flush=True)

L = (4, 96, 64)
run(L, 1, {"zero_rows": 0})
run(L, 1, {"zero_rows": 0, "zero_skip": 0})
run(L, 1, {"zero_rows": 0, "pipe": 0})
run(L, 1, {"zero_rows": 0, "pipe_convert": 0})
run(L, 1, {"zero_rows": 0, "pipe_split_edges": 0})

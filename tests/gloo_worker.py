"""Worker of tests/test_multirank_cpu.py: one rank of a world_size-N gloo job (CPU).
Runs the CPU oracle on this rank's block of the 2-D decomposition; the oracle's halo/edge exchange callback moves the
strips with torch.distributed point-to-point calls posted in the order of the PRODUCT's exchange plan
(mw_exchange_plan, include/mw_cdna4.h) -- the same plan the RCCL path uses on GPUs.  Rank 0 gathers the blocks and
compares them bitwise with a single-rank oracle run."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mw_oracle as O  # noqa: E402
from miniweatherml_amd import capi  # noqa: E402


def threaded_reference(world, nxg, nyg, nz, nsteps, bc, xlen, ylen, names):
    """The same decomposed oracle run with all ranks as threads of this process (tests/util.py: OracleExchanger)."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import OracleExchanger
    ranks, plans = [], []
    for r in range(world):
        d, f = O.supercell_setup(nxg, nyg, nz, 1, xlen, ylen, 20000., nranks=world, rank=r)
        d.p.bc_x, d.p.bc_y, d.p.bc_z = bc
        g = capi.Grid()
        capi.check(capi.lib().mw_decompose(world, r, nxg, nyg, C.byref(g)))
        peers, so, ro, act = (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)()
        capi.check(capi.lib().mw_exchange_plan(C.byref(g), peers, so, ro, act))
        ranks.append((d, f)); plans.append((list(peers), list(act)))
    ex = OracleExchanger(world, plans)
    errs = []

    def work(r):
        try:
            d, f = ranks[r]
            d.set_exchange(ex.make_cb(r))
            dt = d.compute_time_step()
            for _ in range(nsteps):
                d.time_step(f, dt)
        except Exception as e:                                  # pragma: no cover
            errs.append(repr(e)); ex.bar.abort()
    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ths]
    [t.join(200) for t in ths]
    assert not errs, errs
    full = np.zeros((len(names), nz, nyg, nxg, 1))
    for d, f in ranks:
        q = d.p
        full[:, :, q.j_beg:q.j_beg + q.ny, q.i_beg:q.i_beg + q.nx] = np.stack([f.as_dict()[k] for k in names])
    return full


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    nxg, nyg, nz, nsteps = [int(v) for v in sys.argv[4:8]]
    bc = tuple(int(v) for v in sys.argv[8].split(",")) if len(sys.argv) > 8 else None      # optional bc_x,bc_y,bc_z
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    xlen, ylen = 500.0 * nxg, (500.0 * nyg if nyg > 1 else 1.0e5)
    dyc, f = O.supercell_setup(nxg, nyg, nz, 1, xlen, ylen, 20000., nranks=world, rank=rank)
    p = dyc.p
    if bc:
        p.bc_x, p.bc_y, p.bc_z = bc
    # the product's plan for this rank
    g = capi.Grid()
    capi.check(capi.lib().mw_decompose(world, rank, nxg, nyg, C.byref(g)))
    peers, so, ro, act = (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)()
    capi.check(capi.lib().mw_exchange_plan(C.byref(g), peers, so, ro, act))
    assert (g.nx, g.ny, g.i_beg, g.j_beg) == (p.nx, p.ny, p.i_beg, p.j_beg)

    def xchg(ctx, kind, sW, sE, sS, sN, rW, rE, rS, rN, nWE, nSN):
        sb, rb, cnt = [sW, sE, sS, sN], [rW, rE, rS, rN], [nWE, nWE, nSN, nSN]
        send_t = [torch.from_numpy(np.ctypeslib.as_array(sb[d], shape=(cnt[d],)).copy()) if cnt[d] else None for d in range(4)]
        recv_t = [torch.empty(cnt[d], dtype=torch.float64) if cnt[d] else None for d in range(4)]
        reqs = []
        for o in range(4):
            d = so[o]
            if cnt[d] == 0:
                continue
            if act[d]:
                reqs.append(dist.isend(send_t[d], peers[d]))
        for o in range(4):
            d = ro[o]
            if cnt[d] == 0:
                continue
            if act[d]:
                reqs.append(dist.irecv(recv_t[d], peers[d]))
            else:                     # single rank in this direction: periodic self-wrap (W strip -> own E halo ...)
                recv_t[d] = send_t[d ^ 1]
        for r in reqs:
            r.wait()
        for d in range(4):
            if cnt[d]:
                np.ctypeslib.as_array(rb[d], shape=(cnt[d],))[:] = recv_t[d].numpy()

    dyc.set_exchange(xchg)
    dt = dyc.compute_time_step()
    for _ in range(nsteps):
        dyc.time_step(f, dt)
    # gather on rank 0
    names = sorted(f.as_dict())
    mine = np.stack([f.as_dict()[k] for k in names])
    meta = torch.tensor([p.i_beg, p.j_beg, p.nx, p.ny], dtype=torch.int64)
    metas = [torch.zeros(4, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(metas, meta)
    ok = 1
    if rank == 0:
        full = np.zeros((len(names), nz, nyg, nxg, 1))
        full[:, :, p.j_beg:p.j_beg + p.ny, p.i_beg:p.i_beg + p.nx] = mine
        for r in range(1, world):
            ib, jb, nx, ny = [int(v) for v in metas[r]]
            buf = torch.empty((len(names), nz, ny, nx, 1), dtype=torch.float64)
            dist.recv(buf, r)
            full[:, :, jb:jb + ny, ib:ib + nx] = buf.numpy()
        if bc and (bc[0] != 0 and p.nproc_x > 1 or bc[1] != 0 and p.nproc_y > 1):
            # a wall / open direction that is decomposed: the single-rank run is NOT the reference (its `else if` quirk applies
            # the high-side edge rule differently) -- compare with the same decomposition run in threads
            ref = threaded_reference(world, nxg, nyg, nz, nsteps, bc, xlen, ylen, names)
        else:
            d1, f1 = O.supercell_setup(nxg, nyg, nz, 1, xlen, ylen, 20000.)
            if bc:
                d1.p.bc_x, d1.p.bc_y, d1.p.bc_z = bc
            for _ in range(nsteps):
                d1.time_step(f1, dt)
            ref = np.stack([f1.as_dict()[k] for k in names])
        if not np.array_equal(full, ref):
            ok = 0
            print("MISMATCH max|diff| per field:", {k: float(np.max(np.abs(full[i] - ref[i]))) for i, k in enumerate(names)})
    else:
        dist.send(torch.from_numpy(mine.copy()), 0)
    flag = torch.tensor([ok])
    dist.broadcast(flag, 0)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()

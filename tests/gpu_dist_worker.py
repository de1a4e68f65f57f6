"""Worker of tests/test_gpu_multiprocess.py: one rank of a multi-PROCESS job on one GPU (gloo rendezvous, host-staged
strips).  Each rank owns its block through the product path (C ABI + HIP kernels) and exchanges halos through the
torch.distributed transport of miniweatherml_amd.modules; rank 0 gathers and compares with a single-rank GPU run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from miniweatherml_amd import modules  # noqa: E402
from util import gpu_fields  # noqa: E402


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    nxg, nyg, nz, nsteps = [int(v) for v in sys.argv[4:8]]
    mode = sys.argv[8] if len(sys.argv) > 8 else "dycore"
    full = mode in ("full", "rccl_full", "full_defer")      # the complete driver loop (Kessler, sponge, nudger with all-reduce)
    defer = mode == "full_defer"                            # ... with the nudger's increments parked in the dycore handle (a decomposed block applies them with a pass at entry)
    rccl = mode.startswith("rccl")                          # one rank per GPU, the built-in RCCL transport (mw_rccl.cpp) over xGMI
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    device = "cuda:%d" % (rank if rccl else 0)
    if rccl:
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    xlen, ylen = 500.0 * nxg, (500.0 * nyg if nyg > 1 else 1.0e5)
    coupler, dycore, micro, nudger = modules.make_supercell(nxg, nyg, nz, 1, xlen, ylen, 20000., "supercell", device, nranks=world,
                                                            myrank=rank, with_nudger=True)
    if rccl:
        import ctypes as C
        from miniweatherml_amd import capi
        modules.use_rccl_exchange(dycore, coupler)
        ver = C.c_int(0)
        path = capi.lib().mw_rccl_library_path(C.byref(ver)).decode()
        torch_rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        # ONE RCCL in the process: the library's entry points come from the librccl torch.distributed uses
        assert os.path.realpath(path) == os.path.realpath(torch_rccl), (path, torch_rccl)
        mapped = [ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln]
        assert len(set(os.path.realpath(m) for m in mapped)) == 1, sorted(set(mapped))
    else:
        modules.use_torch_distributed_exchange(dycore, coupler, host_staged=True)
    dt = dycore.compute_time_step(coupler)
    for _ in range(nsteps):
        if full:
            modules.supercell_step(coupler, dycore, micro, nudger, dt, defer_nudge=defer)
        else:
            dycore.time_step(coupler, dt)
    torch.cuda.synchronize()
    if defer:
        parked, (rode, passes) = dycore.pending()
        assert parked and rode == 0 and passes == nsteps - 1, (parked, rode, passes)   # the last call's increments are still parked; gpu_fields applies them
    g = gpu_fields(coupler)
    names = sorted(g)
    mine = np.stack([g[k] for k in names])
    cdev = device if rccl else "cpu"                        # nccl groups move device tensors
    meta = torch.tensor([coupler.grid.i_beg, coupler.grid.j_beg, coupler.get_nx(), coupler.get_ny()], dtype=torch.int64, device=cdev)
    metas = [torch.zeros(4, dtype=torch.int64, device=cdev) for _ in range(world)]
    dist.all_gather(metas, meta)
    ok = 1
    if rank == 0:
        gathered = np.zeros((len(names), nz, nyg, nxg, 1))
        for r in range(world):
            ib, jb, nx, ny = [int(v) for v in metas[r]]
            if r == 0:
                blk = mine
            else:
                buf = torch.empty((len(names), nz, ny, nx, 1), dtype=torch.float64, device=cdev)
                dist.recv(buf, r)
                blk = buf.cpu().numpy()
            gathered[:, :, jb:jb + ny, ib:ib + nx] = blk
        c1, d1, m1, n1 = modules.make_supercell(nxg, nyg, nz, 1, xlen, ylen, 20000., "supercell", device, with_nudger=True)
        for _ in range(nsteps):
            if full:
                modules.supercell_step(c1, d1, m1, n1, dt)
            else:
                d1.time_step(c1, dt)
        ref = gpu_fields(c1)
        for i, k in enumerate(names):
            if full:      # the cross-rank sum order of the horizontal means differs from the single-rank order (and the
                          # Kessler rainsplit is rank-local in the reference, quirk 5): rounding-level agreement
                err, scale = float(np.max(np.abs(gathered[i] - ref[k]))), float(np.max(np.abs(ref[k])))
                if err > 1e-9 * scale + 1e-9:
                    ok = 0
                    print("MISMATCH", k, err, scale)
            elif not np.array_equal(gathered[i], ref[k]):
                ok = 0
                print("MISMATCH", k, float(np.max(np.abs(gathered[i] - ref[k]))))
    else:
        dist.send(torch.from_numpy(mine.copy()).to(cdev), 0)
    flag = torch.tensor([ok], device=cdev)
    dist.broadcast(flag, 0)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()

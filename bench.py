#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native dycore.

    python bench.py --gpus N --steps K --warmup W

metric  : cell-updates/s (whole job) of Dynamics_Euler_Stratified_WenoFV::time_step on the supercell grid.
          One cell-update = one (k,j,i,iens) cell advanced through one dycore sub-cycle (3 SSPRK stages, all
          V = 8 prognostic variables) -- SURVEY.md 8(d).  A "step" = one time_step(coupler, dt_CFL) call = 1 sub-cycle.
workload: BASELINE.json configs[1]: supercell 400x400x100, nens 1, fp64, 3 Kessler tracers advected, dycore only,
          dx = dy = 500 m, dz = 200 m, out_freq = -1, dt_phys = CFL step (community_benchmark/driver.cpp:66-82 timed
          region).  N > 1: weak scaling, every GPU keeps a 400x400x100 block of a (400*nproc_x) x (400*nproc_y) x 100
          grid (2-D x/y decomposition of coupler.h:127-179), 3-cell halos exchanged over RCCL once per RK stage and group.
timing  : W untimed warm-up steps, then exactly K steps bracketed by barrier + torch.cuda.synchronize(); max over ranks.
roofline: dominant kernel k_xz_state (x/z WENO reconstruction + Riemann + complete state tendencies + SSPRK3 combine, one
          launch per RK stage).  achieved = algorithmic bytes per launch / average launch duration, the duration from
          hipEvents recorded on the kernel's own stream inside the timed region (mw_dycore_profile).  Algorithmic bytes
          per cell and launch: read 5 state + 5 y-tendencies (+ 5 q^n in stages 2,3), write 5 state
          + 2 face mass fluxes + 2 selector bytes = 138 B (stage 1) / 178 B (stages 2,3), 164.7 B on average.
          peak 8 TB/s HBM3E spec.  traffic = measured HBM bytes per launch from profiles/ (2 x FETCH_SIZE + WRITE_SIZE,
          calibrated with mw_calib_copy).  SURVEY.md 8(d)'s own figures are reported beside it: roofline.pipeline (512 B per
          cell-update) and roofline.flux_stencil_stage (256 B per cell and stage, against one third of the step).  The kernel is fp64-VALU bound: see roofline.valu_busy_frac.
cpu_baseline: the CPU oracle (a port: the reference itself is unbuildable here) timed on one host core
          on BASELINE.json configs[0] (supercell 200x200x50), rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nx", type=int, default=400, help="local block, x")
    ap.add_argument("--ny", type=int, default=400, help="local block, y")
    ap.add_argument("--nz", type=int, default=100)
    ap.add_argument("--nens", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-loop", action="store_true", help="time the whole supercell_example step (dycore, Kessler, sponge, nudger) "
                    "instead of the dycore alone (SURVEY 8(d): report both); not the headline metric")
    ap.add_argument("--cpu-sample", type=str, default="200x200x50", help="oracle sample grid nx x ny x nz")
    ap.add_argument("--strict", type=int, default=0)
    ap.add_argument("--transport", choices=["rccl", "torch"], default="rccl", help="halo-exchange transport for N > 1")
    return ap.parse_args()


def cpu_baseline(sample):
    """Oracle (port of the reference's serial path) on a bounded sample of the same workload, one host core."""
    from oracle import mw_oracle as O
    nx, ny, nz = [int(v) for v in sample.split("x")]
    dyc, f = O.supercell_setup(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny, 20000.0)
    dt = dyc.compute_time_step()
    nsteps = 2
    t0 = time.perf_counter()
    for _ in range(nsteps):
        dyc.time_step(f, dt)
    el = time.perf_counter() - t0
    return {"value": nsteps * nx * ny * nz / el, "unit": "cell-updates/s", "cores": 1, "kind": "port",
            "sample": "2 dycore time_steps (3 RK stages each) of supercell %dx%dx%d nens=1, 3 tracers, CPU oracle "
                      "(oracle/mw_oracle.cpp, -O2 -ffp-contract=off), %.1f s on 1 of %d host cores" % (nx, ny, nz, el, os.cpu_count())}


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks ourselves, one per GPU, as a
    `torch.distributed.run` CHILD process -- this parent never touches the GPU (no HIP call before or after; counting devices
    does not initialise it), relays the child's output and exits with its code."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    if ndev < a.gpus:
        sys.exit("bench.py: --gpus %d requested but only %d GPU(s) are visible on this node" % (a.gpus, ndev))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC: RCCL between processes needs it on this pool
    sys.exit(subprocess.call(cmd, env=env))


def main():
    a = parse()
    if a.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a)                                           # does not return
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:                                          # a launcher started a different number of ranks than asked for
        sys.exit("bench.py: --gpus %d but the launcher set WORLD_SIZE=%d" % (a.gpus, world))
    if torch.cuda.device_count() <= local_rank:
        sys.exit("bench.py: rank %d (LOCAL_RANK %d) has no GPU: %d visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))

    from miniweatherml_amd import capi, modules
    import ctypes as C
    L = capi.lib()

    # global grid: every rank keeps an (nx, ny, nz) block (weak scaling)
    g0 = capi.Grid()
    capi.check(L.mw_decompose(world, rank, a.nx * world, a.ny * world if a.ny > 1 else 1, C.byref(g0)))   # only to learn nproc_x/y
    npx, npy = g0.nproc_x, g0.nproc_y
    nx_glob, ny_glob = a.nx * npx, (a.ny * npy if a.ny > 1 else 1)
    xlen, ylen, zlen = 500.0 * nx_glob, 500.0 * max(ny_glob, 1) if ny_glob > 1 else 500.0 * a.ny, 20000.0
    nudger = None
    if a.full_loop:
        coupler, dycore, micro, nudger = modules.make_supercell(nx_glob, ny_glob, a.nz, a.nens, xlen, ylen, zlen, "supercell", device,
                                                                nranks=world, myrank=rank, with_nudger=True)
    else:
        coupler, dycore, micro = modules.make_supercell(nx_glob, ny_glob, a.nz, a.nens, xlen, ylen, zlen, "supercell", device,
                                                        nranks=world, myrank=rank)
    assert coupler.get_nx() == a.nx and (coupler.get_ny() == a.ny or ny_glob == 1)
    dycore.set_strict(a.strict)
    transport = modules.install_exchange(dycore, coupler, a.transport) if world > 1 else "none"   # all ranks agree on one

    dt = dycore.compute_time_step(coupler)
    V = 5 + coupler.get_num_tracers()
    ncells_local = a.nx * coupler.get_ny() * a.nz * a.nens

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def step():
        dycore.time_step(coupler, dt)
        if nudger is not None:                                   # experiments/supercell_example/driver.cpp:74-76
            micro.time_step(coupler, dt)
            modules.sponge_layer(coupler, dt)
            nudger.nudge_to_column(coupler, dt)

    for _ in range(a.warmup):
        step()
    sync()
    dycore.profile(2)                                            # hipEvents around the dominant kernel only (on its stream)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    el = time.perf_counter() - t0
    KNAMES = ["xz_state", "tracer_patch", "tracer_update_unfused", "halo", "convert", "y_state", "y_tracers", "tracers_fused"]
    prof_dom = dycore.profile_get(0)                             # the roofline's duration: live, over the timed region
    # Outside the timed region: every kernel class bracketed by events (their markers cost ~2 % of the step), same schedule
    dycore.profile(1)
    for _ in range(3):
        step()
    prof = {n: dycore.profile_get(i) for i, n in enumerate(KNAMES)}
    dycore.profile(0)
    # Outside the timed region: the same kernels with the two pipelines serialised, so that each kernel's duration is
    # exclusive.  (With N > 1 -- or MW_OVERLAP=1 -- the state and tracer pipelines of the timed region run on two streams and
    # share the chip; on one rank the default schedule is one stream and the two sets of numbers agree.)
    prof_excl = None
    if not a.strict:
        os.environ["MW_NO_OVERLAP"] = "1"
        dycore.time_step(coupler, dt)
        torch.cuda.synchronize()
        dycore.profile(1)
        for _ in range(3):
            dycore.time_step(coupler, dt)
        prof_excl = {n: dycore.profile_get(i) for i, n in enumerate(KNAMES)}
        dycore.profile(0)
        del os.environ["MW_NO_OVERLAP"]
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # sanity: the run must still be physical (no NaN) -- a blown-up run would be an invalid measurement
    wmax = float(coupler.get_data_manager_readonly().get("wvel", True).abs().max())
    assert wmax == wmax and wmax < 100.0, "unphysical state after the timed region (max|w| = %r)" % wmax

    if rank == 0:
        ncycles = 1
        total_updates = float(ncells_local) * world * ncycles * a.steps
        value = total_updates / el
        flux_ms, flux_n = prof_dom
        avg_flux_s = flux_ms / 1e3 / max(1, flux_n)
        # k_xz_state per cell and launch: read 5 (state) + 5 (y tendencies) [+ 5 q^n in stages 2,3], write 5 + 2 doubles + 2 bytes
        alg_bytes = ((10 + 15 + 15) / 3.0 + 7) * 8.0 * ncells_local + 2.0 * ncells_local
        if a.strict:                                             # general path: k_flux reads V, writes 3V doubles per cell
            alg_bytes = 32.0 * V * ncells_local
        achieved = alg_bytes / avg_flux_s / 1e9 if flux_n else None
        traffic, valu_busy, valu_instr = None, None, None
        pmc = os.path.join(ROOT, "profiles", "latest_summary.json")
        if os.path.exists(pmc) and not a.strict:
            try:
                pj = json.load(open(pmc))
                if int(pj.get("cells_per_launch", 0)) == ncells_local:
                    ks = [v for k, v in pj["kernels"].items() if k.startswith("k_xz_state")]
                    traffic = sum(k["hbm_read_bytes"] + k["hbm_write_bytes"] for k in ks) / len(ks)
                    valu_busy = sum(k["valu_busy_frac"] for k in ks) / len(ks)
                    valu_instr = sum(k["valu_instr_per_cell"] for k in ks) / len(ks)
            except Exception:
                traffic = None
        out = {
            "metric": "cell-updates/s" + (" (full supercell_example loop)" if a.full_loop else ""), "value": value, "unit": "cell-updates/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "supercell %dx%dx%d nens=%d per GPU (global %dx%dx%d), WENO-FV dycore only, 3 tracers, "
                                   "CFL dt" % (a.nx, coupler.get_ny(), a.nz, a.nens, nx_glob, ny_glob, a.nz),
                       "parallelism": "%dx%d slab" % (npx, npy), "halo_transport": transport, "V": V, "strict": a.strict,
                       "schedule": ("two streams (state | tracers, tracer stream at high priority)" if (os.environ.get("MW_OVERLAP", "1" if world > 1 else "0") != "0" and not a.strict) else "one stream"),
                       "alg_bytes_per_cell_update": 64 * V,
                       "hbm_frac_cell_update": value * 64 * V / 8.0e12 / world},
            "roofline": {"bound": "hbm", "kernel": "k_flux" if a.strict else "k_xz_state", "achieved": achieved, "peak": 8000.0,
                         "unit": "GB/s", "frac": (achieved / 8000.0) if achieved else None, "traffic": traffic,
                         "avg_launch_ms": avg_flux_s * 1e3, "launches": flux_n, "alg_bytes_per_launch": alg_bytes,
                         "avg_launch_ms_exclusive": (prof_excl["xz_state"][0] / max(1, prof_excl["xz_state"][1])) if prof_excl else None,
                         "achieved_exclusive": (alg_bytes / (prof_excl["xz_state"][0] / max(1, prof_excl["xz_state"][1]) * 1e-3) / 1e9)
                         if prof_excl else None,
                         "valu_busy_frac": valu_busy, "valu_instr_per_cell": valu_instr,
                         "note": "fp64-VALU bound kernel (SURVEY.md 8(d)): valu_busy_frac / valu_instr_per_cell from the committed "
                                 "rocprofv3 PMC summary profiles/latest_summary.json; duration measured live with hipEvents"},
            "kernel_ms_per_step": {k: v[0] / 3.0 for k, v in prof.items()},
            "kernel_ms_per_step_exclusive": ({k: v[0] / 3.0 for k, v in prof_excl.items()} if prof_excl else None),
        }
        # SURVEY.md 8(d)'s own per-unit figures, next to the dominant kernel's: whole pipeline = 64 V B per cell-update
        # ("roofline.achieved = cell_updates_per_s x B_alg / 8.0e12"); flux stencil of one stage = 32 V B per cell, priced
        # against one third of the step (everything a stage does, per GPU)
        per_gpu = value / world
        out["roofline"]["pipeline"] = {"alg_bytes_per_cell_update": 64 * V, "achieved": per_gpu * 64 * V / 1e9,
                                       "frac": per_gpu * 64 * V / 8.0e12}
        out["roofline"]["flux_stencil_stage"] = {"alg_bytes_per_cell": 32 * V, "achieved": 3.0 * per_gpu * 32 * V / 1e9,
                                                 "frac": 3.0 * per_gpu * 32 * V / 8.0e12}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_sample)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

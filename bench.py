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
roofline: SURVEY.md 8(d)'s flux-stencil figure: 32 V = 256 B per cell and RK stage (read V, write 3 V doubles), priced against
          everything one stage launches (k_y_all + k_xz_state + k_tracers_fused + k_tracer_patch -- two streams: k_y_state + k_y_tracers instead of k_y_all; the reference's
          D6 + D9 stencil split by direction and variable group, with D10-D12 fused in).  achieved = 256 B x cells / average stage
          duration, the duration from hipEvents recorded on the handle's stream around each stage inside the timed region
          (mw_dycore_profile class 8; N > 1 runs two streams: one third of the step instead).  roofline.dominant_kernel carries
          k_xz_state's own numbers (its algorithmic bytes: read 5 state + 5 y-tendencies (+ 5 q^n in stages 2,3), write 5 state + 2
          face mass fluxes + 2 selector bytes = 138 / 178 B per cell, 164.7 B on average), roofline.pipeline the 512 B per
          cell-update figure.  peak 8 TB/s HBM3E spec.  traffic / valu_* are counted LIVE at N = 1: before this process touches the GPU,
          three child processes `rocprofv3 --pmc <counters> -- python3 bench.py --pmc-worker` run the same workload for three steps
          (2 x FETCH_SIZE + WRITE_SIZE calibrated with mw_calib_copy; roofline.pmc_provenance.source says "live"); when rocprofv3 is
          absent or fails the committed summary profiles/latest_summary.json is quoted instead, and only while the kernel sources still
          hash to what it was taken from; otherwise null.
          roofline.calibration / roofline.floors / roofline.fp64_valu.peak_measured (round 5): the ceilings MEASURED in this run -- sustained
          v_fma_f64 issue rate and the clock it held (mw_calib_fma64), the stage's bare arithmetic on registers (mw_calib_stage_arith: the
          time no schedule of this arithmetic can beat), the streaming copy rate -- and roofline.bound says what they imply: the stage is
          co-limited by fp64 VALU issue and HBM traffic, and the arithmetic alone caps the algorithmic HBM fraction below the 0.60 target.
micro   : after the timed region (the headline is untouched): sustained / value_sustained (round 6: the headline's own step after 1200 back-to-back
          steps, i.e. at the board's sustained power limit -- the timed region is the first 0.1 s after an idle start, at boost); Kessler (two states) and the surrogate MLP on the same grid, 72 B per
          cell each, and the dycore step on a state with cloud and rain (FCT limiter + y-face correction pass active):
          developed_ms_per_step (a seeded stress state: rims everywhere) and storm.ms_per_step / value_storm, mature.ms_per_step /
          value_mature (the dycore steps of the complete supercell_example loop's own last 100 iterations in front of --storm-steps and
          --mature-steps, hipEvents around each: the states the simulation really passes through).  simulation_loop / value_simulation_loop: the wall clock of that whole
          loop -- the reference's own timed region (community_benchmark/driver.cpp:66-82), the storm developing inside it; value_storm and
          value_simulation_loop, not the cloud-free `value`, are the regression metrics.  transport_self_loop: rank 0's block of the
          1 x 2 / 2 x 2 / 4 x 2 decompositions with the built-in RCCL transport in its self-loop form (every peer is this rank on a 1-rank
          communicator) -- what the exchange costs beside the stencils with the real send / receive groups, on one GPU; not a multi-GPU number.
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
    ap.add_argument("--workload", choices=["config2", "config3", "config4", "config5"], default="config2",
                    help="config2 (default): BASELINE.json configs[1], 400x400x100 nens 1 per GPU, dx 500 m.  config3: configs[2], the same grid with the "
                         "complete surrogate loop of inference_ponni.cpp:69-82 (dycore, ponni MLP inference beside the true Kessler step, sponge_layer, "
                         "ColumnNudger); not the headline metric.  config4: configs[3]'s per-GPU "
                         "block 256x512x128 nens 4, dx 800 m (weak-scaling series 256x512 ... 1024x1024 global for 1 ... 8 GPUs).  config5: configs[4], "
                         "simple_city 512x512x256 per GPU (immersed buildings, gravity off, water vapour only: V = 6), dx = dy = dz = 5 m; "
                         "--full-loop adds Horizontal_Sponge, sponge_layer and Time_Averager (experiments/simple_city/driver.cpp:66-79)")
    ap.add_argument("--ord", type=int, default=5, choices=[3, 5, 7, 9], help="WENO order (the reference's -DMW_ORD; 5 = its default and the headline; "
                    "3 = the order its GPU benchmark environment builds, build/machines/aws/aws_a100_gpu.env:21)")
    ap.add_argument("--storm-steps", type=int, default=2600, help="steps of the complete supercell loop before the 'storm' dycore timing of the "
                    "micro section (0 = skip)")
    ap.add_argument("--mature-steps", type=int, default=12900, help="total steps of the same loop before the 'mature' dycore timing (12900 CFL steps = 3600 s "
                    "simulated on the 400 x 400 x 100 grid; the reference's supercell_example runs 7200 s, input_euler3d.yaml:3); 0 or <= --storm-steps = skip")
    ap.add_argument("--sustained-steps", type=int, default=1200, help="micro section: back-to-back steps in front of the sustained-power measurement of the headline state (0 = skip)")
    ap.add_argument("--no-micro", action="store_true", help="skip the Kessler / MLP / developed-state section after the timed region")
    ap.add_argument("--no-pmc", action="store_true", help="do not start the rocprofv3 --pmc child processes that count HBM bytes / VALU "
                    "instructions of this very run's kernels (roofline.traffic, fp64_valu); the committed summary is quoted instead")
    ap.add_argument("--no-selfloop", action="store_true", help="skip the micro section's run of the RCCL self-loop transport (rank 0's block of 2 / 4 / 8 ranks)")
    ap.add_argument("--no-calib", action="store_true", help="skip the fp64 FMA ceiling / arithmetic floor / streaming-copy calibration after the timed region")
    ap.add_argument("--pmc-worker", action="store_true", help=argparse.SUPPRESS)      # the child's mode: a few dycore steps, nothing else
    ap.add_argument("--timeout-s", type=float, default=float(os.environ.get("MW_BENCH_TIMEOUT_S", "900")),
                    help="wall-clock limit of a run: a rank that has not finished by then prints where it is and exits non-zero "
                         "(a first contact between GPUs over RCCL must fail fast, not hang the caller's lease)")
    a = ap.parse_args()
    if a.workload == "config3":
        a.full_loop = True
    if a.workload == "config4":
        a.nx, a.ny, a.nz, a.nens = 256, 512, 128, 4
    if a.workload == "config5":
        a.nx, a.ny, a.nz, a.nens = 512, 512, 256, 1
        a.no_cpu_baseline = True                                 # (the CPU sample is a supercell run)
    return a




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


# what the dycore kernels are compiled from
KERNEL_SOURCES = ("mw_march.h", "mw_weno.h", "mw_weno79.h", "mw_common.h", "mw_calib.h", "mw_dycore.hip")


def kernel_source_hash():
    """sha256 (16 hex digits) over the HIP sources of the dycore kernels: ties counter values copied from a committed
    rocprofv3 summary to the code they were measured on (tools/summarize_profiles.py records the same hash)."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "miniweatherml_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


PMC_PASSES = (("f", ["FETCH_SIZE"]), ("w", ["WRITE_SIZE"]), ("sq", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_WAVES"]))


def pmc_worker(a):
    """The counted child (bench.py --pmc-worker, started under rocprofv3 --pmc by pmc_children): the bench's own handle and state, one
    warm-up step and two steps of the timed loop's step(), nothing else."""
    import torch
    from miniweatherml_amd import modules
    dxy = 800.0 if a.workload == "config4" else 500.0
    coupler, dycore, _ = modules.make_supercell(a.nx, a.ny, a.nz, a.nens, dxy * a.nx, dxy * a.ny, 20000.0, "supercell", "cuda:0", ord=a.ord)
    dt = dycore.compute_time_step(coupler)
    for _ in range(3):
        dycore.time_step(coupler, dt)
    torch.cuda.synchronize()


def pmc_children(a):
    """LIVE hardware counters of this run's kernels: before this process touches the GPU, three fresh child processes -- `rocprofv3 --pmc
    <counters> -- python3 bench.py --pmc-worker` (the program directly behind `--`, counters in their own passes, no tracing: the pool's
    rules) -- run the bench workload for three steps each; their CSVs give, per kernel and launch, HBM bytes (2 x FETCH_SIZE + WRITE_SIZE
    in KiB: the gfx950 correction calibrated with mw_calib_copy, tools/calib_pmc.py) and VALU instructions / busy cycles.
    -> (summary dict in the layout of profiles/*_summary.json, None) or (None, reason)."""
    import collections
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 is not on PATH"
    tmp = tempfile.mkdtemp(prefix="mw_bench_pmc_")
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("MW_BENCH_PROGRESS_DIR", None)
    t0 = time.time()
    try:
        for tag, ctrs in PMC_PASSES:
            cmd = [exe, "--pmc"] + ctrs + ["--output-format", "csv", "-d", os.path.join(tmp, tag), "--", sys.executable, os.path.abspath(__file__),
                                           "--pmc-worker", "--nx", str(a.nx), "--ny", str(a.ny), "--nz", str(a.nz), "--nens", str(a.nens),
                                           "--ord", str(a.ord), "--workload", a.workload if a.workload in ("config2", "config4") else "config2"]
            # (its own process group: on a timeout the whole group goes -- rocprofv3 AND the `python --pmc-worker` grandchild, which would
            #  otherwise keep running on the GPU during the timed region; three passes may take a quarter of the run's budget together)
            child = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
            try:
                _, err = child.communicate(timeout=max(40.0, a.timeout_s / 12))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(child.pid, signal.SIGKILL)
                except OSError:
                    pass
                child.communicate()
                return None, "rocprofv3 --pmc %s did not finish within %.0f s (its process group was ended)" % (" ".join(ctrs), max(40.0, a.timeout_s / 12))
            if child.returncode != 0:
                return None, "rocprofv3 --pmc %s exited with %d: %s" % (" ".join(ctrs), child.returncode, (err or b"").decode(errors="replace")[-300:])

        def short(n):
            return n.replace("void ", "").replace("mw::", "").split("(")[0]

        def load(tag):
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            dur = collections.defaultdict(dict)
            for f in glob.glob(os.path.join(tmp, tag, "*", "*counter_collection.csv")):
                for row in csv.DictReader(open(f)):
                    if "mw::" not in row["Kernel_Name"]:
                        continue
                    k = short(row["Kernel_Name"])
                    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    try:
                        dur[k][row["Dispatch_Id"]] = (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-3
                    except (KeyError, ValueError):
                        pass
            return ({k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()},
                    {k: len(next(iter(cs.values()))) for k, cs in acc.items()}, {k: (sum(d.values()) / len(d)) for k, d in dur.items() if d})
        (F, nF, _), (W, _, _), (SQ, _, usSQ) = load("f"), load("w"), load("sq")
        if not F or not W or not SQ:
            return None, "the rocprofv3 passes produced no rows for the library's kernels"
        cells = float(a.nx * a.ny * a.nz * a.nens)
        kernels = {}
        for k in F:
            e = {"calls": nF[k]}
            if k in W:
                e["hbm_read_bytes"] = 2.0 * F[k]["FETCH_SIZE"] * 1024
                e["hbm_write_bytes"] = W[k]["WRITE_SIZE"] * 1024
            m = SQ.get(k, {})
            if "SQ_INSTS_VALU" in m:
                e["valu_instr_per_cell"] = m["SQ_INSTS_VALU"] * 64 / cells
                if m.get("GRBM_GUI_ACTIVE"):
                    cyc = m["GRBM_GUI_ACTIVE"] / 8
                    e["valu_busy_frac"] = m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc
                    if k in usSQ and usSQ[k] > 0:
                        e["avg_us"] = usSQ[k]                   # (duration under the SQ pass: the counters' own run)
                        e["clock_GHz_under_profile"] = cyc / (usSQ[k] * 1e-6) / 1e9
            kernels[k] = e
        return {"tag": "live", "cells_per_launch": cells, "kernel_sources_sha16": kernel_source_hash(), "kernels": kernels,
                "seconds": time.time() - t0, "note": "FETCH_SIZE x2 (gfx950: 8 B/lane streaming reads report exactly 1/2, calibrated with "
                "mw_calib_copy), WRITE_SIZE x1; KiB -> bytes; three rocprofv3 --pmc child processes of this bench run"}, None
    except Exception as e:                                       # a profiler problem must never cost the measurement
        return None, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def calibration(torch, device, stage_cells):
    """The measured ceilings this run's fractions are quoted against (mw_calib_*; SURVEY.md 8(d) "calibrate with an FMA microbenchmark"):
    sustained v_fma_f64 issue at 2 and 8 wavefronts per SIMD and the clock it held, the stage's bare arithmetic (24 WENO-5 + 3 Riemann
    per cell on registers) for one stage of this block on smooth and on rough data, and the streaming copy rate (8 B per lane)."""
    import ctypes as C
    from miniweatherml_amd import calib, capi
    out = {"fma64": [calib.fma64(w, 0.4, device) for w in (2, 8)],
           "stage_arith": {k: calib.stage_arith(k, stage_cells, 25, device) for k in ("smooth", "rough", "cloud_free")}}
    n = 1 << 27                                                  # 1 GiB each way
    src = torch.ones(n, dtype=torch.float64, device=device)
    dst = torch.empty_like(src)
    L = capi.lib()
    st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    for _ in range(2):
        capi.check(L.mw_calib_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), n, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        capi.check(L.mw_calib_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), n, st))
    e1.record()
    torch.cuda.synchronize()
    out["stream_copy_TBps"] = 5 * 16.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e12
    del src, dst
    torch.cuda.empty_cache()
    return out


class PowerSampler:
    """Package power and shader clock of THIS process's card while a region runs (round 6: the stage runs against the board's power cap, and what a
    state costs is set by the clock the part can hold on it -- profiles/r06_storm_vs_initial.json).  A thread reads the amdgpu hwmon files
    (power1_average | power1_input in uW, freq1_input in Hz) of the card whose PCI address hipDeviceGetPCIBusId reports, every 10 ms; the step loop
    spends its time inside ctypes calls, which release the GIL.  All failures are silent: the figures are evidence, not the measurement."""

    def __init__(self, device_index=0, period=0.010):
        import ctypes
        import glob
        import threading
        self.files, self.rows, self.on, self.period, self.thread = {}, [], False, period, None
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) != 0:
                return
            want = buf.value.decode().lower()
            for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
                if want not in os.path.realpath(os.path.join(h, "..", "..")).lower():
                    continue
                for key, names in (("power_uW", ("power1_average", "power1_input")), ("sclk_Hz", ("freq1_input",))):
                    for n in names:
                        f = os.path.join(h, n)
                        if key not in self.files and os.path.exists(f):
                            try:
                                float(open(f).read()); self.files[key] = f
                            except (OSError, ValueError):
                                pass
        except Exception:
            self.files = {}
        self._threading = threading

    def _run(self):
        while self.on:
            r = {}
            for k, f in self.files.items():
                try:
                    r[k] = float(open(f).read())
                except (OSError, ValueError):
                    pass
            self.rows.append(r)
            time.sleep(self.period)

    def start(self):
        self.rows = []
        if self.files:
            self.on = True
            self.thread = self._threading.Thread(target=self._run, daemon=True)
            self.thread.start()
        return self

    def stop(self, window=None):
        """window = (first, last) sample indices (negative: from the end) to summarise instead of all samples."""
        if self.thread is not None:
            self.on = False
            self.thread.join()
            self.thread = None
        return self.stats(window)

    def stats(self, window=None):
        rows = self.rows if window is None else self.rows[window[0]:window[1]]
        out = {"samples": len(rows)}
        for k, scale, name in (("power_uW", 1e-6, "power_W"), ("sclk_Hz", 1e-6, "sclk_MHz")):
            v = sorted(r[k] * scale for r in rows if k in r)
            if v:
                out[name] = {"mean": sum(v) / len(v), "min": v[0], "max": v[-1]}
        return out if len(out) > 1 else None


def rows_full_form(dycore):
    """Fraction of (level, row) words of the last sub-cycle's stage maps Q1..Q3 (mw_debug_zero_maps) in which cloud or rain may be non-zero: the
    rows whose iterations of the fused tracer kernel take the FULL form (the others neither load nor compute those two tracers).  None: no maps."""
    import ctypes as C
    import numpy as np
    from miniweatherml_amd import capi
    L = capi.lib()
    dims = (C.c_int * 2)()
    n = L.mw_debug_zero_maps(dycore.h, None, 0, dims)
    if n <= 0:
        return None
    buf = np.empty(n, np.uint32)
    if L.mw_debug_zero_maps(dycore.h, buf.ctypes.data_as(C.c_void_p), n, dims) != n:
        return None
    maps = buf.reshape(10, dims[0], dims[1])[:, :, 9:-9]
    return float(((maps[1:4] & 0x6) != 0).mean())


def tiles_full_form(dycore, nx, tile=58, halo=4):
    """The same per 58-cell x tile of a row (round 6: the maps carry x segments, bits 4 .. 29 of a word, L = ceil(nx / 26) cells each): the share of
    (level, row, tile) iterations of the fused tracer kernel that take the FULL form -- a tile is full when a segment its lanes overlap is set."""
    import ctypes as C
    import numpy as np
    from miniweatherml_amd import capi
    L = capi.lib()
    dims = (C.c_int * 2)()
    n = L.mw_debug_zero_maps(dycore.h, None, 0, dims)
    if n <= 0:
        return None
    buf = np.empty(n, np.uint32)
    if L.mw_debug_zero_maps(dycore.h, buf.ctypes.data_as(C.c_void_p), n, dims) != n:
        return None
    q = buf.reshape(10, dims[0], dims[1])[1:4, :, 9:-9]
    Ls = (nx + 25) // 26
    full, tiles = 0.0, 0
    for t0 in range(0, nx, tile):
        lo, hi = t0 - halo, t0 + tile + halo - 1
        segs = set(((x % nx) // Ls) for x in range(lo, hi + 1))
        m = np.uint32(sum(1 << (4 + sg) for sg in segs))
        full += float(((q & m) != 0).mean()); tiles += 1
    return full / tiles


def cloud_extent(torch, dm, tile=58):
    """Diagnostics of where cloud / rain are (what a finer map granularity could skip): the share of cells that hold any, and -- after growing that
    set by the 9 cells per direction a sub-cycle can move a tracer -- the share of x rows and of 58-cell x tiles (one wavefront of the marching
    kernels) that it touches."""
    m = ((dm.get("cloud_liquid", True) != 0) | (dm.get("precip_liquid", True) != 0))[..., 0].to(torch.float32)
    cells = float(m.mean())
    g = m[None, None]
    for ax, k in ((2, (19, 1, 1)), (3, (1, 19, 1)), (4, (1, 1, 19))):
        g = torch.nn.functional.max_pool3d(g, kernel_size=k, stride=1, padding=tuple(9 if kk == 19 else 0 for kk in k))
    g = g[0, 0]
    nz, ny, nx = g.shape
    rows = float((g.amax(dim=2) > 0).to(torch.float32).mean())
    pad = (-nx) % tile
    gt = torch.nn.functional.pad(g, (0, pad)).reshape(nz, ny, -1, tile)
    tiles = float((gt.amax(dim=3) > 0).to(torch.float32).mean())
    return {"cells_with_cloud_or_rain": cells, "rows_touched_grown_by_9": rows, "tiles58_touched_grown_by_9": tiles}


def micro_section(torch, modules, coupler, dycore, micro, dt, a):
    """After the timed region: Kessler and the surrogate MLP on the bench grid (72 algorithmic bytes per cell each: 5 fields read, 4
    written) and the dycore on a state that HAS cloud and rain (the headline state has none, so its FCT limiter and y-face
    correction pass idle).  Durations from events on torch's current stream, which is the stream these entry points launch on."""
    dm = coupler.get_data_manager_readwrite()
    rho_d = dm.get("density_dry")
    nz, ny, nx, nens = rho_d.shape
    ncell = rho_d.numel()

    def timed(fn, iters=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    g = torch.Generator(device=rho_d.device).manual_seed(1)
    rnd = lambda: torch.rand(rho_d.shape, generator=g, device=rho_d.device, dtype=torch.float64)        # noqa: E731
    qc = rnd() * 2e-3 * (rnd() > 0.6)
    qr = rnd() * 3e-4 * (rnd() > 0.6)
    box = torch.zeros_like(rho_d)
    box[: int(0.6 * nz), ny // 4: ny // 2, nx // 4: nx // 2] = 1.0
    base = {n: dm.get(n).clone() for n in ("water_vapor", "cloud_liquid", "precip_liquid", "temp")}
    states = {"one_storm": "cloud and rain inside one storm (1/16 of the columns, lower 60 % of the levels): rain-free wavefronts take "
                           "the short cut", "scattered": "cloud / rain scattered at random over 40 % of the cells (no rain-free wavefront)"}
    res = {"kessler": {"alg_bytes_per_cell": 72, "cells": ncell, "states": {}}}
    saved = dict(base)

    def restore():
        for n, t in saved.items():
            dm.get(n).copy_(t)

    def kessler_once():
        restore()
        micro.time_step(coupler, dt)

    for name, desc in states.items():
        m = box if name == "one_storm" else 1.0
        saved["cloud_liquid"], saved["precip_liquid"] = qc * rho_d * m, qr * rho_d * m
        ms = timed(kessler_once) - timed(restore)
        res["kessler"]["states"][name] = {"state": desc, "ms_per_call": ms, "cells_per_s": ncell / ms * 1e3,
                                          "achieved_GBps": ncell * 72 / ms / 1e6, "frac": ncell * 72 / ms / 1e6 / 8000.0}
    res["kessler"]["kernels"] = "k_kessler_prep + k_kessler_sweep (z chunks when rainsplit == 1, whole columns otherwise): two launches per call"
    restore()
    W1, b1, W2, b2, si, so = modules.load_surrogate_weights()
    ins = [dm.get(n) for n in ("temp", "density_dry", "water_vapor", "cloud_liquid", "precip_liquid")]
    outs = [torch.empty_like(ins[0]) for _ in range(4)]
    ms = timed(lambda: modules.mlp_forward(*ins, W1, b1, W2, b2, si, so, outs))
    res["mlp"] = {"kernel": "k_mlp (v_mfma_f32_16x16x4_f32, 5 MFMAs per 16 cells)", "alg_bytes_per_cell": 72, "cells": ncell, "ms_per_call": ms,
                  "cells_per_s": ncell / ms * 1e3, "achieved_GBps": ncell * 72 / ms / 1e6, "frac": ncell * 72 / ms / 1e6 / 8000.0,
                  "fp32_gflops_nominal": ncell * 208 / ms / 1e6, "bound": "hbm (2.9 flop/B: MFMA utilisation is necessarily << 1 %)"}
    # ---- the dycore on a state with cloud and rain: smooth blobs with sharp rims in the sheared wind -> FCT multipliers < 1
    k = torch.arange(nz, device=rho_d.device, dtype=torch.float64).view(nz, 1, 1, 1)
    j = torch.arange(ny, device=rho_d.device, dtype=torch.float64).view(1, ny, 1, 1)
    i = torch.arange(nx, device=rho_d.device, dtype=torch.float64).view(1, 1, nx, 1)
    blob = ((torch.sin(i * 0.11) * torch.cos(j * 0.07)) > 0.3).to(torch.float64)
    for n, t in base.items():
        dm.get(n).copy_(t)
    dm.get("cloud_liquid").copy_(2.0e-3 * blob * ((k > 0.15 * nz) & (k < 0.45 * nz)) * (0.5 + 0.5 * torch.sin(0.3 * k + 0.05 * i) ** 2) * rho_d)
    dm.get("precip_liquid").copy_(4.0e-4 * blob * (k < 0.3 * nz) * (0.5 + 0.5 * torch.cos(0.2 * k + 0.03 * j) ** 2) * rho_d)
    for _ in range(2):
        dycore.time_step(coupler, dt)
    dev_ms = timed(lambda: dycore.time_step(coupler, dt), 6)       # (no per-kernel events here: they cost ~2 % of the step)
    dycore.profile(1)
    for _ in range(2):
        dycore.time_step(coupler, dt)
    patch_ms, patch_n = dycore.profile_get(1)
    dycore.profile(0)
    res["developed_ms_per_step"] = dev_ms
    res["developed_rows_full_form"] = rows_full_form(dycore)
    res["developed_tiles_full_form"] = tiles_full_form(dycore, nx)
    res["value_developed"] = ncell / dev_ms * 1e3                # cell-updates/s on the developed state: read it next to "value"
    res["developed_state"] = {"state": "the bench state after the timed region + seeded cloud (2e-3) and rain (4e-4) blobs with sharp rims: "
                                       "FCT multipliers < 1 and a busy y-face correction pass", "cell_updates_per_s": ncell / dev_ms * 1e3,
                              "tracer_patch_ms_per_launch": patch_ms / max(1, patch_n),
                              "cloud_max": float(dm.get("cloud_liquid").max()), "rain_max": float(dm.get("precip_liquid").max())}
    # ---- ... and on a REAL developed storm: the complete supercell_example loop (dycore, Kessler, sponge layer, column nudger,
    # experiments/supercell_example/driver.cpp:66-79) run for a.storm_steps steps from the initial state (2600 steps of the
    # 400 x 400 x 100 configuration: 18 m/s updraft, cloud and rain confined to the storm), then the dycore step alone.
    if a.storm_steps > 0 and nens == 1:
        xlen, ylen = float(coupler.get_xlen()), float(coupler.get_ylen())
        c2, d2, m2, n2 = modules.make_supercell(nx, ny, nz, nens, xlen, ylen, 20000.0, "supercell", rho_d.device, with_nudger=True, ord=a.ord)
        dt2 = d2.compute_time_step(c2)
        # the REFERENCE'S OWN timed region (experiments/community_benchmark/driver.cpp:66-82: the whole `while (etime < sim_time)` loop,
        # init excluded, out_freq = -1, CFL dt): wall clock around all a.storm_steps iterations of dycore + Kessler + sponge + nudger,
        # the storm developing inside it
        # (round 6: the nudger's increments ride on the next dycore step's conversion instead of a pass of their own -- defer_nudge, same bits;
        #  the eager form is timed beside it below)
        for _ in range(3):
            modules.supercell_step(c2, d2, m2, n2, dt2, defer_nudge=True)
        torch.cuda.synchronize()
        ps_loop = PowerSampler(rho_d.device.index or 0, period=0.05).start()      # (20 samples per second: nothing the loop would notice)
        t_loop = time.perf_counter()
        tail_n = min(100, max(0, a.storm_steps - 3 - 1))         # the loop's last iterations carry ONE hipEvent pair per dycore step (profile 3)
        for _ in range(a.storm_steps - 3 - tail_n):
            modules.supercell_step(c2, d2, m2, n2, dt2, defer_nudge=True)
        d2.profile(3)
        for _ in range(tail_n):
            modules.supercell_step(c2, d2, m2, n2, dt2, defer_nudge=True)
        storm_in = d2.profile_get(9)                             # (total ms, steps): the dycore steps of the loop's last iterations, inside the loop
        d2.profile(0)
        d2.flush_pending()                                       # (inside the timed region: the loop ends with every field whole)
        torch.cuda.synchronize()
        loop_s = time.perf_counter() - t_loop
        ps_loop.stop()
        storm_maps = (rows_full_form(d2), tiles_full_form(d2, nx))   # the maps of the loop's last dycore step: the state as the loop left it
        # (the same loop, the same process: its first seconds run on the cloud-free state, its last on the storm -- 50 ms samples, 2 s windows)
        res["power_loop"] = {"early_cloud_free": ps_loop.stats((20, 60)), "late_storm": ps_loop.stats((-40, None)),
                             "what": "50 ms samples: seconds 1-3 and the last 2 s of the simulation loop (and of its continuation to the mature state)"}
        res["simulation_loop"] = {"what": "wall clock of the whole simulation loop the reference's benchmark driver times (community_benchmark/"
                                          "driver.cpp:66-82): %d iterations of dycore.time_step + micro.time_step + sponge_layer + "
                                          "nudge_to_column from the initial state, the storm developing inside it" % (a.storm_steps - 3),
                                  "steps": a.storm_steps - 3, "seconds": loop_s, "ms_per_step": loop_s / (a.storm_steps - 3) * 1e3,
                                  "simulated_seconds": dt2 * (a.storm_steps - 3), "cell_updates_per_s": ncell * (a.storm_steps - 3) / loop_s}
        res["value_simulation_loop"] = ncell * (a.storm_steps - 3) / loop_s
        res["simulation_loop"]["nudger"] = "deferred (mw_nudge_to_column_deferred: the increments are added by the next time step's conversion; flushed at the end, inside the timed region)"
        f2 = c2.get_data_manager_readonly()
        storm_extent = cloud_extent(torch, f2)
        storm_fields = {"max_abs_w": float(f2.get("wvel").abs().max()), "cloud_max": float(f2.get("cloud_liquid").max()), "rain_max": float(f2.get("precip_liquid").max())}
        # Round 6 (late): the dycore step ON THE STORM is the one the loop itself runs there -- hipEvents around every mw_dycore_time_step of the
        # loop's last iterations.  Rounds 4-6 timed ten back-to-back dycore steps BEHIND the loop instead; without Kessler between them (which
        # evaporates the tiny amounts of cloud and rain that the advection stencil spreads into clear air, back to exact zeros) the set of non-zero
        # cells grows by up to 9 cells per step in every direction, the zero-row maps fill up and every further step is slower: that figure is kept as
        # `isolated_after` -- it measures a state the simulation never visits.
        storm_iso = timed(lambda: d2.time_step(c2, dt2), 10)
        storm_ms = (storm_in[0] / storm_in[1]) if storm_in[1] else storm_iso
        res["storm"] = {"state": "the last %d iterations of the complete supercell_example loop in front of iteration %d from the initial state" % (storm_in[1], a.storm_steps),
                        "ms_per_step": storm_ms, "cell_updates_per_s": ncell / storm_ms * 1e3,
                        "timing": "hipEvents around each mw_dycore_time_step inside the loop (the step includes the nudger's parked increments riding on its conversion)",
                        "rows_full_form": storm_maps[0], "tiles_full_form": storm_maps[1], "extent": storm_extent,
                        "isolated_after": {"ms_per_step": storm_iso, "cell_updates_per_s": ncell / storm_iso * 1e3, "rows_full_form": rows_full_form(d2),
                                           "tiles_full_form": tiles_full_form(d2, nx),
                                           "what": "12 back-to-back dycore steps behind the loop (rounds 4-6's figure): no Kessler in between, the non-zero set of cloud and rain spreads"}}
        res["storm"].update(storm_fields)
        res["value_storm"] = ncell / storm_ms * 1e3
        # the same loop iteration with the nudger's own second pass (the reference's structure), interleaved with the deferred form, at this state
        ab = {"eager": [], "deferred": []}
        for _ in range(2):
            ab["eager"].append(timed(lambda: modules.supercell_step(c2, d2, m2, n2, dt2), 30))
            ab["deferred"].append(timed(lambda: modules.supercell_step(c2, d2, m2, n2, dt2, defer_nudge=True), 30))
        d2.flush_pending()
        res["simulation_loop"]["nudger_ab_ms_per_iteration"] = {k: min(v) for k, v in ab.items()}
        # ---- ... and on the MATURE storm: the same loop continued to a.mature_steps steps (3600 s simulated by default; the reference's
        # supercell_example runs 7200 s): cloud and anvil cover a large share of the rows, where the data-dependent short-cuts stop helping
        if a.mature_steps > a.storm_steps:
            more = a.mature_steps - a.storm_steps
            torch.cuda.synchronize()
            ps_loop = PowerSampler(rho_d.device.index or 0, period=0.05).start()
            t_loop = time.perf_counter()
            tail_n = min(100, more - 1)
            for _ in range(more - tail_n):
                modules.supercell_step(c2, d2, m2, n2, dt2, defer_nudge=True)
            d2.profile(3)
            for _ in range(tail_n):
                modules.supercell_step(c2, d2, m2, n2, dt2, defer_nudge=True)
            mature_in = d2.profile_get(9)
            d2.profile(0)
            d2.flush_pending()
            torch.cuda.synchronize()
            loop2_s = time.perf_counter() - t_loop
            ps_loop.stop()
            mature_maps = (rows_full_form(d2), tiles_full_form(d2, nx))
            res["power_loop"]["late_mature"] = ps_loop.stats((-40, None))
            mature_extent = cloud_extent(torch, f2)
            mature_fields = {"max_abs_w": float(f2.get("wvel").abs().max()), "cloud_max": float(f2.get("cloud_liquid").max()), "rain_max": float(f2.get("precip_liquid").max())}
            mature_iso = timed(lambda: d2.time_step(c2, dt2), 10)
            mature_ms = (mature_in[0] / mature_in[1]) if mature_in[1] else mature_iso
            res["mature"] = {"state": "the last %d iterations in front of iteration %d (%.0f s simulated) of the complete supercell_example loop from the initial state" % (mature_in[1], a.mature_steps, a.mature_steps * dt2),
                             "ms_per_step": mature_ms, "cell_updates_per_s": ncell / mature_ms * 1e3,
                             "timing": "hipEvents around each mw_dycore_time_step inside the loop (see storm)",
                             "rows_full_form": mature_maps[0], "tiles_full_form": mature_maps[1], "extent": mature_extent,
                             "isolated_after": {"ms_per_step": mature_iso, "cell_updates_per_s": ncell / mature_iso * 1e3, "rows_full_form": rows_full_form(d2),
                                                "tiles_full_form": tiles_full_form(d2, nx), "what": "12 back-to-back dycore steps behind the loop (see storm)"},
                             "loop_ms_per_step_storm_to_mature": loop2_s / more * 1e3, "loop_cell_updates_per_s_storm_to_mature": ncell * more / loop2_s}
            res["mature"].update(mature_fields)
            res["value_mature"] = ncell / mature_ms * 1e3
            res["simulation_loop"]["to_mature"] = {"steps": a.mature_steps - 3, "seconds": loop_s + loop2_s,
                                                   "ms_per_step": (loop_s + loop2_s) / (a.mature_steps - 3) * 1e3,
                                                   "cell_updates_per_s": ncell * (a.mature_steps - 3) / (loop_s + loop2_s)}
    # ---- what the strip exchange costs beside the stencils, with the REAL transport, on this one GPU: rank 0's block of the 1 x 2, 2 x 2
    # and 4 x 2 decompositions the driver's N = 2, 4, 8 runs use, the built-in RCCL transport in its self-loop form (every peer is this rank
    # on a 1-rank communicator: mw_dycore_use_rccl_self; the block's own state tiled periodically, so the run is physical and -- tests --
    # bitwise the one-rank run).  The strips move through local HBM instead of xGMI; pack / group / unpack, the side streams, the pipelined
    # schedule and the contention with the compute kernels are the real ones.  NOT a multi-GPU measurement.
    if nens == 1 and not getattr(a, "no_selfloop", False):
        try:
            import ctypes as C
            from miniweatherml_amd import capi
            base_ms = timed(lambda: dycore.time_step(coupler, dt), 10)
            emu = {"what": "rank 0's block of an N-rank decomposition with the built-in RCCL transport in self-loop form on ONE GPU (mw_dycore_use_rccl_self); "
                           "ratio = one-rank ms / N-rank-block ms: what the weak-scaling efficiency of the dycore step would be if xGMI behaved like the "
                           "self-loop.  Not a multi-GPU measurement.", "one_rank_ms": base_ms, "ranks": {}}
            for nr in (2, 4, 8):
                g = capi.Grid()
                capi.check(capi.lib().mw_decompose(nr, 0, nx * nr, ny * nr, C.byref(g)))
                npx, npy = g.nproc_x, g.nproc_y
                xl, yl = float(coupler.get_xlen()), float(coupler.get_ylen())
                c3, d3, _ = modules.make_supercell(nx * npx, ny * npy, nz, 1, xl * npx, yl * npy, 20000.0, "supercell", rho_d.device, nranks=nr, myrank=0, ord=a.ord)
                for n in ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "cloud_liquid", "precip_liquid"):
                    c3.get_data_manager_readwrite().get(n).copy_(dm.get(n))
                modules.use_rccl_self_exchange(d3, c3)
                # (the part slows down by several per cent over a bench run: the one-rank step is timed again right before and right after
                #  every block, and the ratio uses their mean -- a single one-rank figure in front of all blocks biased the ratios low)
                b0 = timed(lambda: dycore.time_step(coupler, dt), 6)
                ms = timed(lambda: d3.time_step(c3, dt), 10)
                b1 = timed(lambda: dycore.time_step(coupler, dt), 6)
                emu["ranks"][str(nr)] = {"rank_grid": "%dx%d" % (npx, npy), "ms_per_step": ms, "one_rank_ms_around": [b0, b1], "ratio": 0.5 * (b0 + b1) / ms, "path": d3.path()}
                del c3, d3
                torch.cuda.empty_cache()
            res["transport_self_loop"] = emu
        except Exception as e:                                   # evidence, not the measurement
            res["transport_self_loop"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # ---- the SUSTAINED step on the headline's own state (round 6, late): the timed region above is the first 0.1 s of GPU work after an idle
    # start -- the package-power average has not reached the board's 1400 W limit yet and the clock is at its boost.  A run that keeps stepping
    # settles on the limit within ~1.5 s (tools: --warmup 300 / 1000), and every step of a simulation runs there.  Same cloud-free initial state
    # on a fresh handle, a.sustained_steps steps back to back, the last 200 timed (cloud and rain stay exactly zero without microphysics).
    if a.sustained_steps > 0:
        xlen, ylen = float(coupler.get_xlen()), float(coupler.get_ylen())
        c4, d4, _m4 = modules.make_supercell(nx, ny, nz, nens, xlen, ylen, 20000.0, "supercell", rho_d.device, ord=a.ord)
        dt4 = d4.compute_time_step(c4)
        ps4 = PowerSampler(rho_d.device.index or 0, period=0.05).start()
        for _ in range(a.sustained_steps):
            d4.time_step(c4, dt4)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            d4.time_step(c4, dt4)
        e1.record()
        torch.cuda.synchronize()
        ps4.stop()
        sus_ms = e0.elapsed_time(e1) / 200
        res["sustained"] = {"what": "the headline's workload and state (cloud-free initial state, dycore only) after %d back-to-back steps: 200 steps timed with the card at "
                                    "its sustained power limit; `value` is the same step in the first 0.1 s after an idle start" % a.sustained_steps,
                            "ms_per_step": sus_ms, "cell_updates_per_s": ncell / sus_ms * 1e3, "power_last_second": ps4.stats((-20, None))}
        res["value_sustained"] = ncell / sus_ms * 1e3
        del c4, d4, _m4
        torch.cuda.empty_cache()
    return res


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
    # the ranks report their progress into one small file each; if the job does not end within the limit this GPU-less parent ends
    # the child's whole process group, says which rank stopped where, and exits non-zero
    import signal
    import tempfile
    pdir = tempfile.mkdtemp(prefix="mw_bench_progress_")
    env["MW_BENCH_PROGRESS_DIR"] = pdir
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        rc = child.wait(timeout=a.timeout_s + 60.0)              # (every rank has its own watchdog at timeout_s: this is the backstop)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(child.pid, signal.SIGKILL)
        except OSError:
            pass
        child.wait()
        print("bench.py: the %d-rank job did not finish within %.0f s and was ended; last progress per rank:" % (a.gpus, a.timeout_s + 60.0), file=sys.stderr)
        for r in range(a.gpus):
            try:
                last = open(os.path.join(pdir, "rank%d" % r)).read().strip().splitlines()[-1]
            except (OSError, IndexError):
                last = "(nothing reported)"
            print("  rank %d: %s" % (r, last), file=sys.stderr)
        rc = 124
    sys.exit(rc)


def main():
    a = parse()
    if a.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if a.pmc_worker:
        pmc_worker(a)
        return
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a)                                           # does not return
    import threading
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # (before anything initialises the GPU: the host driver only supports dmabuf IPC)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    t_start = time.time()
    where = ["start"]

    def progress(msg):
        """One line per milestone: to stderr on multi-rank runs, and into the parent's progress file when there is one."""
        where[0] = msg
        line = "[bench rank %d/%d +%.1fs] %s" % (rank, world, time.time() - t_start, msg)
        if world > 1:
            print(line, file=sys.stderr, flush=True)
        pdir = os.environ.get("MW_BENCH_PROGRESS_DIR")
        if pdir:
            try:
                with open(os.path.join(pdir, "rank%d" % rank), "a") as f:
                    f.write(line + "\n")
            except OSError:
                pass

    def watchdog():
        # fires only if the run is still going after --timeout-s: the main thread may be blocked inside RCCL / a HIP call (the GIL is
        # released there), so this thread reports and ends the process; the launcher then ends the other ranks
        print("bench.py: rank %d did not finish within %.0f s; last milestone: %s" % (rank, a.timeout_s, where[0]), file=sys.stderr, flush=True)
        os._exit(3)
    # live counters first: fresh child processes under rocprofv3 --pmc, BEFORE this process imports torch or touches the GPU.  (They carry
    # their own time limits and are ended as a process group; the run's watchdog is armed behind them, with the whole budget.)
    live_pmc, live_pmc_why = None, "not attempted"
    if a.gpus == 1 and "WORLD_SIZE" not in os.environ and not a.no_pmc and not a.strict and not a.full_loop and a.workload in ("config2", "config4"):
        progress("rocprofv3 --pmc child processes (live counters of this run's kernels) ...")
        live_pmc, live_pmc_why = pmc_children(a)
        progress("live counters: %s" % ("ok" if live_pmc else live_pmc_why))
    wd = threading.Timer(a.timeout_s, watchdog)
    wd.daemon = True
    wd.start()
    import torch
    import torch.distributed as dist
    if world != a.gpus:                                          # a launcher started a different number of ranks than asked for
        sys.exit("bench.py: --gpus %d but the launcher set WORLD_SIZE=%d" % (a.gpus, world))
    if torch.cuda.device_count() <= local_rank:
        sys.exit("bench.py: rank %d (LOCAL_RANK %d) has no GPU: %d visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        progress("init_process_group(nccl) on %s ..." % device)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        progress("init_process_group ok")

    from miniweatherml_amd import capi, modules
    import ctypes as C
    L = capi.lib()

    # global grid: every rank keeps an (nx, ny, nz) block (weak scaling)
    g0 = capi.Grid()
    capi.check(L.mw_decompose(world, rank, a.nx * world, a.ny * world if a.ny > 1 else 1, C.byref(g0)))   # only to learn nproc_x/y
    npx, npy = g0.nproc_x, g0.nproc_y
    nx_glob, ny_glob = a.nx * npx, (a.ny * npy if a.ny > 1 else 1)
    dxy = 800.0 if a.workload == "config4" else 5.0 if a.workload == "config5" else 500.0   # input_euler3d_1024x1024x100.yaml: xlen = 819200 / 1024 cells
    xlen, ylen, zlen = dxy * nx_glob, dxy * max(ny_glob, 1) if ny_glob > 1 else dxy * a.ny, 20000.0
    nudger = None
    city = None
    if a.workload == "config5":
        zlen = 5.0 * a.nz
        coupler, dycore, hs_, ta_ = modules.make_simple_city(nx_glob, ny_glob, a.nz, 1, xlen, ylen, zlen, "city", device, nranks=world, myrank=rank)
        micro, city = None, (hs_, ta_)
    elif a.full_loop:
        coupler, dycore, micro, nudger = modules.make_supercell(nx_glob, ny_glob, a.nz, a.nens, xlen, ylen, zlen, "supercell", device,
                                                                nranks=world, myrank=rank, with_nudger=True, ord=a.ord,
                                                                micro=modules.Microphysics_Kessler_Surrogate() if a.workload == "config3" else None)
    else:
        coupler, dycore, micro = modules.make_supercell(nx_glob, ny_glob, a.nz, a.nens, xlen, ylen, zlen, "supercell", device,
                                                        nranks=world, myrank=rank, ord=a.ord)
    assert coupler.get_nx() == a.nx and (coupler.get_ny() == a.ny or ny_glob == 1)
    dycore.set_strict(a.strict)
    progress("model set up (%d x %d x %d x %d per rank); installing the halo-exchange transport ..." % (a.nx, coupler.get_ny(), a.nz, a.nens))
    transport = modules.install_exchange(dycore, coupler, a.transport) if world > 1 else "none"   # all ranks agree on one
    rccl = None
    if transport == "rccl":
        ver = C.c_int(0)
        path = (L.mw_rccl_library_path(C.byref(ver)) or b"").decode()
        n_, r_, lanes_ = dycore.rccl_info()
        rccl = {"comm_ranks": n_, "comm_rank": r_, "lanes": lanes_, "library": path, "version": ver.value}
        progress("ncclCommInitRank ok: communicator of %d ranks, this is rank %d, %s (version %d)" % (n_, r_, path, ver.value))
        if n_ != world:
            sys.exit("bench.py: the RCCL communicator reports %d ranks, the job has %d" % (n_, world))

    dt = dycore.compute_time_step(coupler)
    V = 5 + coupler.get_num_tracers()
    ncells_local = a.nx * coupler.get_ny() * a.nz * a.nens

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def step():
        if city is not None and a.full_loop:                      # experiments/simple_city/driver.cpp:66-79
            modules.simple_city_step(coupler, dycore, city[0], city[1])
            return
        dycore.time_step(coupler, dt)
        if nudger is not None:                                   # experiments/supercell_example/driver.cpp:74-76
            micro.time_step(coupler, dt)
            modules.sponge_layer(coupler, dt)
            nudger.nudge_to_column(coupler, dt)

    if world > 1:
        step()
        sync()
        progress("first time_step (first halo exchanges between the GPUs) ok")
    for _ in range(a.warmup - (1 if world > 1 else 0)):
        step()
    sync()
    progress("warm-up done")
    power = PowerSampler(local_rank).start()
    dycore.profile(3)                                            # ONE hipEvent pair per time_step on the handle's stream (pairs around every stage and the dominant kernel cost 1.5 % of the step: tools/event_overhead.py)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    el = time.perf_counter() - t0
    power_timed = power.stop()
    progress("timed region done")
    sched = dycore.schedule()                                    # what the timed time_steps ran (mw_dycore_schedule), not re-derived here
    KNAMES = ["xz_state", "tracer_patch", "tracer_update_unfused", "halo", "convert", "y_state", "y_tracers", "tracers_fused"]
    prof_step = dycore.profile_get(9)                            # every time_step, first launch to the join: live, over the timed region
    prof_stage = (prof_step[0], 3 * prof_step[1])                # = its three RK stages (the bench's dt is one sub-cycle)
    # Outside the timed region: every kernel class bracketed by events (their markers cost ~2 % of the step), same schedule
    dycore.profile(1)
    for _ in range(3):
        step()
    prof = {n: dycore.profile_get(i) for i, n in enumerate(KNAMES)}
    prof_waits = (dycore.profile_get(10), dycore.profile_get(11))  # pipelined schedule of a decomposed block: the compute stream's waits for the state / tracer strips
    prof_dom = prof["xz_state"]                                  # the dominant kernel on its own: live, the three steps behind the timed region
    dycore.profile(0)
    # Outside the timed region: the same kernels with the two pipelines serialised, so that each kernel's duration is
    # exclusive.  (With N > 1 -- or option overlap = 1 -- the state and tracer pipelines of the timed region run on two streams and
    # share the chip; on one rank the default schedule is one stream and the two sets of numbers agree.)
    prof_excl = None
    if not a.strict:
        dycore.set_option("overlap", 0)                          # (a handle option: nothing in this process reads the environment per launch)
        dycore.time_step(coupler, dt)
        torch.cuda.synchronize()
        dycore.profile(1)
        for _ in range(3):
            dycore.time_step(coupler, dt)
        prof_excl = {n: dycore.profile_get(i) for i, n in enumerate(KNAMES)}
        dycore.profile(0)
        dycore.set_option("overlap", -1)
    el_local = el
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # sanity: the run must still be physical -- a blown-up run would be an invalid measurement.  DataManager::validate_all
    # (DataManager.h:385-387, one device pass per entry here): NaN / inf in any field, negative values in the positive-definite tracers
    coupler.get_data_manager_readonly().validate_all(die_on_failed_check=True)
    # ... and bounded: a run that blew up but stayed finite is not a measurement either (the supercell's updraft reaches 20-50 m/s, the
    # city's flow 20 m/s; the reference has no such check -- its validators only look for NaN / inf / negative values)
    w_max = float(coupler.get_data_manager_readonly().get("wvel", True).abs().max())
    assert w_max < 100.0, "max|w| = %g m/s after the timed region: the run is not physical, the measurement is invalid" % w_max
    # the timed region is over: from here on a slow host section (micro, CPU baseline) must not cost the finished measurement -- the
    # watchdog is re-armed and, should it fire, prints the headline line that is complete by then and exits 0
    wd.cancel()
    pending = [None]

    def watchdog_late():
        if pending[0] is not None:
            pending[0]["note"] = "the optional sections after the timed region did not finish within %.0f s (last: %s); headline complete" % (a.timeout_s, where[0])
            print(json.dumps(pending[0]), flush=True)
            os._exit(0)
        watchdog()
    wd = threading.Timer(a.timeout_s, watchdog_late)
    wd.daemon = True
    wd.start()
    if world > 1:                                                # every rank's library path / communicator view, gathered for the JSON line
        infos = [None] * world
        dist.all_gather_object(infos, rccl)
    else:
        infos = [rccl]

    # ---- self-diagnosis of a launched (torchrun) job, so that ONE run on an 8-GPU node says what the scaling is made of: every rank's
    # own time per step, what the exchange costs on the critical path (the compute stream's waits for the strips, per RK stage), and the
    # SAME block as a one-rank periodic domain timed on the same GPUs right behind the run -- efficiency computed inside the run
    multi = None
    if "WORLD_SIZE" in os.environ:
        progress("multi-GPU diagnostics (per-rank times, strip waits, the one-rank block) ...")
        mine = {"rank": rank, "device": torch.cuda.get_device_name(local_rank), "ms_per_step": el_local / a.steps * 1e3,
                "wait_state_strips_ms_per_stage": (prof_waits[0][0] / prof_waits[0][1]) if prof_waits[0][1] else 0.0,
                "wait_tracer_strips_ms_per_stage": (prof_waits[1][0] / prof_waits[1][1]) if prof_waits[1][1] else 0.0,
                "waits_recorded": int(prof_waits[0][1] + prof_waits[1][1]), "one_rank_block_ms_per_step": None}
        if city is None and not a.full_loop and not a.strict:
            try:
                ny_l = coupler.get_ny()
                c1, d1, _ = modules.make_supercell(a.nx, ny_l, a.nz, a.nens, dxy * a.nx, dxy * ny_l if ny_l > 1 else ylen, zlen, "supercell", device, ord=a.ord)
                dt1 = d1.compute_time_step(c1)
                for _ in range(max(1, a.warmup)):
                    d1.time_step(c1, dt1)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.steps):
                    d1.time_step(c1, dt1)
                e1.record()
                torch.cuda.synchronize()
                mine["one_rank_block_ms_per_step"] = e0.elapsed_time(e1) / a.steps
                mine["one_rank_block_path"] = d1.path()
                del c1, d1
                torch.cuda.empty_cache()
            except Exception as e:                               # evidence, not the measurement
                mine["one_rank_block_error"] = "%s: %s" % (type(e).__name__, e)
        if world > 1:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
        else:
            per_rank = [mine]
        blocks = [r["one_rank_block_ms_per_step"] for r in per_rank if r.get("one_rank_block_ms_per_step")]
        multi = {"world": world, "rank_grid": "%dx%d" % (npx, npy), "halo_transport": transport,
                 "rccl_ranks_seen": (infos[0] or {}).get("comm_ranks") if infos and infos[0] else None,
                 "per_rank": per_rank,
                 "ms_per_step_max_over_ranks": el / a.steps * 1e3,
                 "one_rank_block_ms_per_step_mean": (sum(blocks) / len(blocks)) if blocks else None,
                 "efficiency_in_run": ((sum(blocks) / len(blocks)) / (el / a.steps * 1e3)) if blocks else None,
                 "what": "per_rank: every rank's own wall time per step over the timed region, the time its compute stream sat waiting for the state / "
                         "tracer strips per RK stage (profile classes 10 / 11 of the three profiled steps behind the timed region; 0 without an exchange), "
                         "and the same local block as a ONE-RANK periodic domain timed on that rank's GPU right after the run; efficiency_in_run = "
                         "mean one-rank block time / the job's max-over-ranks time per step"}
        progress("multi-GPU diagnostics done")

    two_streams = (sched["code"] & 3) == 1                       # (decided inside the library per time_step, not re-derived here)
    if rank == 0:
        ncycles = 1
        total_updates = float(ncells_local) * world * ncycles * a.steps
        value = total_updates / el
        ms_per_step = el / a.steps * 1e3
        per_gpu = value / world
        # ---- the flux stencil of one RK stage = 32 V B per cell, against everything that stage launches
        stage_ms_live = (prof_stage[0] / prof_stage[1]) if (prof_stage[1] and not two_streams and not a.strict) else None
        stage_ms = stage_ms_live if stage_ms_live else ms_per_step / 3.0
        stage_bytes = 32.0 * V * ncells_local
        achieved = stage_bytes / (stage_ms * 1e-3) / 1e9
        # ---- the dominant kernel on its own
        flux_ms, flux_n = prof_dom
        avg_flux_s = flux_ms / 1e3 / max(1, flux_n)
        # k_xz_state per cell and launch: read 5 (state) + 5 (y tendencies) [+ 5 q^n in stages 2,3], write 5 + 2 doubles + 2 bytes
        dom_bytes = ((10 + 15 + 15) / 3.0 + 7) * 8.0 * ncells_local + 2.0 * ncells_local
        if a.strict:                                             # general path: k_flux reads V, writes 3V doubles per cell
            dom_bytes = 32.0 * V * ncells_local
        dom_achieved = dom_bytes / avg_flux_s / 1e9 if flux_n else None
        # ---- counters from the committed rocprofv3 summary: only while the kernel sources are the profiled ones
        traffic, dom_traffic, valu_busy, valu_instr, prov, valu_side, derived = None, None, None, None, None, None, None
        pmc = os.path.join(ROOT, "profiles", "latest_summary.json")
        if (live_pmc is not None or os.path.exists(pmc)) and not a.strict:
            try:
                now = kernel_source_hash()
                if live_pmc is not None:                       # counted on THIS box by this run's own child processes
                    pj = live_pmc
                    prov = {"file": None, "source": "live: rocprofv3 --pmc child processes of this bench run (%.0f s)" % pj["seconds"], "tag": "live",
                            "profiled_sources_sha16": now, "current_sources_sha16": now, "valid": int(pj["cells_per_launch"]) == ncells_local}
                else:
                    pj = json.load(open(pmc))
                    prov = {"file": "profiles/latest_summary.json", "source": "replayed from the committed summary (live counters: %s)" % live_pmc_why,
                            "tag": pj.get("tag"), "profiled_sources_sha16": pj.get("kernel_sources_sha16"),
                            "current_sources_sha16": now, "valid": bool(pj.get("kernel_sources_sha16") == now and
                                                                        int(pj.get("cells_per_launch", 0)) == ncells_local)}
                    if a.ord != 5:
                        prov["valid"] = False                  # the committed counters are the WENO-5 kernels'
                if prov["valid"]:
                    K = pj["kernels"]
                    ks = [v for k, v in K.items() if k.startswith("k_xz_state")]
                    dom_traffic = sum(k["hbm_read_bytes"] + k["hbm_write_bytes"] for k in ks) / len(ks)
                    valu_busy = sum(k["valu_busy_frac"] for k in ks) / len(ks)
                    valu_instr = sum(k["valu_instr_per_cell"] for k in ks) / len(ks)
                    stage_k = [v for k, v in K.items() if k.startswith(("k_xz_state", "k_y_all", "k_y_state", "k_y_tracers", "k_tracers_fused", "k_tracer_patch"))]
                    nstages = sum(k["calls"] for k in ks)      # one k_xz_state launch per RK stage
                    traffic = sum((k["hbm_read_bytes"] + k["hbm_write_bytes"]) * k["calls"] for k in stage_k if "hbm_read_bytes" in k) / nstages
                    # per-kernel counted bandwidth of the profiled run (bytes / that run's average duration) and VALU-busy range
                    big = [k for k in stage_k if "hbm_read_bytes" in k and k.get("avg_us", 0) > 50]
                    kbw = [(k["hbm_read_bytes"] + k["hbm_write_bytes"]) / (k["avg_us"] * 1e-6) / 1e12 for k in big]
                    kvb = [k["valu_busy_frac"] for k in big if "valu_busy_frac" in k]
                    derived = {"kernel_TBps_min": min(kbw), "kernel_TBps_max": max(kbw), "valu_busy_min": min(kvb), "valu_busy_max": max(kvb),
                               "traffic_over_algorithmic": traffic / (32.0 * V * ncells_local)}
                    # fp64-VALU side of the roofline (the binding one): counted VALU instructions of one cell-update against the
                    # chip's issue rate (1024 SIMDs, one wave64 instruction per 4 cycles, 2.4 GHz peak engine clock)
                    instr_cu = sum(k["valu_instr_per_cell"] * k["calls"] for k in stage_k if "valu_instr_per_cell" in k) / nstages * 3.0
                    valu_side = {"instr_per_cell_update": instr_cu, "achieved_wave_instr_per_s": per_gpu * instr_cu / 64.0,
                                 "peak_wave_instr_per_s": 1024 * 2.4e9 / 4.0, "frac": per_gpu * instr_cu / 64.0 / (1024 * 2.4e9 / 4.0),
                                 "source": "SQ_INSTS_VALU of %s x live cell-updates/s" % (prov["file"] or "this run's rocprofv3 --pmc children")}
            except Exception:
                traffic = dom_traffic = valu_side = derived = None
        # ---- measured ceilings (mw_calib_*): the fp64 issue rate this chip sustains, the stage's bare arithmetic, the streaming copy rate
        cal, floors, bound = None, None, "hbm"
        if world == 1 and not a.no_calib and not a.strict:
            try:
                cal = calibration(torch, device, ncells_local)
                peak_meas = max(c["wave_instr_per_s"] for c in cal["fma64"])
                fl_s, fl_r = cal["stage_arith"]["smooth"]["ms_per_stage_of_requested_cells"], cal["stage_arith"]["rough"]["ms_per_stage_of_requested_cells"]
                fl_c = cal["stage_arith"]["cloud_free"]["ms_per_stage_of_requested_cells"]
                hbm_floor = (traffic / (cal["stream_copy_TBps"] * 1e12) * 1e3) if traffic else None
                alg_floor = stage_bytes / (cal["stream_copy_TBps"] * 1e12) * 1e3
                top = max(fl_s, hbm_floor or 0.0)
                floors = {"arith_floor_ms_per_stage": fl_s, "arith_floor_ms_per_stage_rough_data": fl_r,
                          "arith_floor_ms_per_stage_cloud_free_data": fl_c, "stage_over_arith_floor_cloud_free": stage_ms / fl_c,
                          "hbm_floor_ms_per_stage_counted_traffic": hbm_floor, "hbm_floor_ms_per_stage_algorithmic_bytes": alg_floor,
                          "stream_copy_TBps": cal["stream_copy_TBps"], "stage_ms": stage_ms, "stage_over_arith_floor": stage_ms / fl_s,
                          "stage_over_max_floor": stage_ms / top,
                          "frac_ceiling_if_arith_bound": stage_bytes / (fl_s * 1e-3) / 8.0e12,
                          "what": "arith floor = mw_calib_stage_arith: 24 WENO-5 + 3 Riemann solves + passive fluxes per cell on registers, no HBM "
                                  "traffic, k_xz_state's workgroup shape; hbm floor = bytes / the measured streaming copy rate"}
                if valu_side:
                    valu_side["peak_measured"] = peak_meas
                    valu_side["frac_of_measured"] = valu_side["achieved_wave_instr_per_s"] / peak_meas
                    valu_side["clock_GHz_under_fma_load"] = [c["clock_GHz_in_kernel"] for c in cal["fma64"]]
                # what the numbers say: the stage cannot beat its bare arithmetic, which alone caps the algorithmic HBM fraction
                cap = floors["frac_ceiling_if_arith_bound"]
                bound = ("fp64-valu + hbm co-limited: the stage's bare arithmetic alone takes %.2f ms (%.2f ms on this cloud-free state, where the zero "
                         "short-cut leaves 18 of 24 reconstructions; frac <= %.2f / %.2f at any schedule), its counted traffic at streaming speed %s; "
                         "0.60 of 8 TB/s is out of reach in fp64 with the reference's limiter"
                         % (fl_s, fl_c, cap, stage_bytes / (fl_c * 1e-3) / 8.0e12, ("%.2f ms" % hbm_floor) if hbm_floor else "n/a"))
            except Exception as e:                               # calibration is evidence, not the measurement
                cal = {"error": "%s: %s" % (type(e).__name__, e)}
        what = "complete supercell_example loop: WENO-FV dycore + Kessler + sponge_layer + ColumnNudger" if a.full_loop else "WENO-FV dycore only"
        if a.workload == "config3":
            what = "complete supercell_kessler_surrogate loop: WENO-FV dycore + ponni MLP inference (MFMA) beside Kessler + sponge_layer + ColumnNudger"
        if city is not None:
            what = "complete simple_city loop: Horizontal_Sponge + WENO-FV dycore + sponge_layer + Time_Averager" if a.full_loop else "WENO-FV dycore only"
        out = {
            "metric": "cell-updates/s" + (" (full supercell_kessler_surrogate loop)" if a.workload == "config3" else " (full supercell_example loop)" if a.full_loop else "") + (" (MW_ORD = %d)" % a.ord if a.ord != 5 else ""), "value": value, "unit": "cell-updates/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s %dx%dx%d nens=%d per GPU (global %dx%dx%d, dx = dy = %g m), %s, %d tracer%s, "
                                   "CFL dt" % ("simple_city" if city is not None else "supercell", a.nx, coupler.get_ny(), a.nz, a.nens, nx_glob, ny_glob, a.nz, dxy, what,
                                               V - 5, "" if V == 6 else "s"),
                       "baseline_config": "configs[3] per-GPU block" if a.workload == "config4" else "configs[4] per-GPU block" if city is not None else
                                          "configs[2]" if a.workload == "config3" else "configs[1]",
                       "parallelism": "%dx%d slab" % (npx, npy), "halo_transport": transport, "V": V, "strict": a.strict, "weno_order": a.ord,
                       "schedule": sched["streams"], "schedule_code": sched["code"], "y_faces_in_one_launch": sched["y_all"],
                       "dispatcher_path": dycore.path(),
                       "zero_tracer_shortcut": bool(dycore.get_option("zero_skip")),   # (bit-neutral: tracers that are exactly 0 over a wavefront's stencil are not reconstructed)
                       "zero_stores_skipped": bool(dycore.get_option("zero_skip") and dycore.get_option("zero_rows") and dycore.get_option("zero_stores")),   # (... nor are zeros stored over rows that hold zeros already)
                       "zero_row_maps": bool(dycore.get_option("zero_skip") and dycore.get_option("zero_rows")),   # (... and x rows of a tracer that are known to be 0 are neither loaded nor computed by the fused tracer kernel)
                       "rccl_ranks": (infos[0] or {}).get("comm_ranks") if world > 1 else None,
                       "rccl_per_rank": infos if world > 1 else None,
                       "alg_bytes_per_cell_update": 64 * V,
                       "hbm_frac_cell_update": value * 64 * V / 8.0e12 / world,
                       # through the Coupler boundary, which is what is timed: + 32 V for D1 and D13, + 24 with immersed boundaries
                       "alg_bytes_per_cell_update_boundary": 96 * V + (24 if city is not None else 0),
                       "hbm_frac_cell_update_boundary": value * (96 * V + (24 if city is not None else 0)) / 8.0e12 / world},
            # bound: the resource that binds the stage: its COUNTED HBM traffic -- 2.1 x the algorithmic bytes: y tendencies,
            # face mass fluxes, tracer y fluxes written by one launch and read by the next -- moving at 4.6-5.6 TB/s, with the fp64
            # instruction stream at 0.80-0.87 VALU busy right under it (the WENO-3 build, half the arithmetic, takes 93 % of the time).
            # achieved / peak / frac are SURVEY.md 8(d)'s algorithmic figure (32 V B per cell against 8 TB/s: what the north star's 60 %
            # target is quoted in); traffic / traffic_frac the counted bytes; roofline.fp64_valu the instruction side.
            "roofline": {"bound": bound,
                         "floors": floors,
                         "binding_resource": ("HBM traffic of the stage's three launches (counted bytes: roofline.traffic = %.2f x algorithmic; "
                                              "%.2f-%.2f TB/s per kernel in the profiled run, a streaming copy reaches 5.0-5.5 on this part) with "
                                              "fp64 VALU issue co-limiting at %.2f-%.2f busy; numbers from roofline.pmc_provenance.file"
                                              % (derived["traffic_over_algorithmic"], derived["kernel_TBps_min"], derived["kernel_TBps_max"],
                                                 derived["valu_busy_min"], derived["valu_busy_max"])) if derived else
                                             "HBM traffic of the stage's launches with fp64 VALU issue co-limiting (DESIGN.md 0b / 0c); no counter "
                                             "summary matches the current kernel sources, so no figures are quoted here",
                         "binding_resource_derived": derived,
                         "kernel": "one RK stage = k_y_all (y faces of all variables) + k_xz_state + k_tracers_fused + k_tracer_patch "
                                   "(SURVEY.md 8(d) flux stencil, 32 V B per cell)" if not a.strict else "one RK stage (general path)",
                         "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                         "traffic_frac": (traffic / (stage_ms * 1e-3) / 8.0e12) if traffic else None,
                         "alg_bytes_per_launch": stage_bytes, "avg_launch_ms": stage_ms,
                         "avg_launch_ms_source": "hipEvents around every time_step on the handle's stream over the timed region, / 3 RK stages (a pair per stage costs 1.5 % of what it times)" if stage_ms_live
                                                 else "ms_per_step / 3 (two-stream schedule: a stage's launches overlap the next stage's)",
                         "launches": prof_stage[1] if stage_ms_live else 3 * a.steps,
                         "pmc_provenance": prov,
                         "dominant_kernel": {"kernel": "k_flux" if a.strict else "k_xz_state", "achieved": dom_achieved,
                                             "frac": (dom_achieved / 8000.0) if dom_achieved else None, "traffic": dom_traffic,
                                             "avg_launch_ms": avg_flux_s * 1e3, "launches": flux_n, "alg_bytes_per_launch": dom_bytes,
                                             "avg_launch_ms_source": "hipEvents around every launch on its stream, the three steps behind the timed region",
                                             "avg_launch_ms_exclusive": (prof_excl["xz_state"][0] / max(1, prof_excl["xz_state"][1])) if prof_excl else None,
                                             "valu_busy_frac": valu_busy, "valu_instr_per_cell": valu_instr},
                         "fp64_valu": valu_side,
                         "calibration": cal,
                         "pipeline": {"alg_bytes_per_cell_update": 64 * V, "achieved": per_gpu * 64 * V / 1e9, "frac": per_gpu * 64 * V / 8.0e12}},
            "multi_gpu": multi,
            "power": {"timed_region": power_timed, "what": "package power (hwmon power1_*) and shader clock (freq1_input) of this rank's card, sampled every 10 ms: "
                      "timed_region = the K timed steps (short: the package-power reading is still ramping up from idle there); simulation_loop = 2 s windows of the whole loop "
                      "of the micro section, early (cloud-free) against late (storm, mature) -- the steady figures; board limit 1400 W"},
            "kernel_ms_per_step": {k: v[0] / 3.0 for k, v in prof.items()},
            "kernel_ms_per_step_exclusive": ({k: v[0] / 3.0 for k, v in prof_excl.items()} if prof_excl else None),
        }
        pending[0] = out
        if world == 1 and not a.no_micro and not a.strict and city is None:
            progress("micro section (Kessler, MLP, developed state, the whole simulation loop) ...")
            out.update(micro_section(torch, modules, coupler, dycore, micro, dt, a))
            out["config"]["value_simulation_loop"] = out.get("value_simulation_loop")
            # the headline state is the benign one (cloud-free initial field): the same step on a developed storm, next to `value`
            out["config"]["value_storm"] = out.get("value_storm")
            out["config"]["value_developed"] = out.get("value_developed")
            out["config"]["value_mature"] = out.get("value_mature")
            out["config"]["value_sustained"] = out.get("value_sustained")
            out["power"]["simulation_loop"] = out.pop("power_loop", None)
            # ---- the roofline figure, state by state (the headline state is the best case: cloud and rain identically zero).  frac = SURVEY.md
            # 8(d)'s 32 V B per cell and stage with V = 8; frac_moved = the same with the variables that actually move: the six that are
            # never skipped plus cloud and rain in the share of rows whose tracer iterations take the full form (the zero-row maps)
            def _state(ms, rows, tiles=None):
                if not ms:
                    return None
                st_s = ms / 3.0 * 1e-3
                e = {"ms_per_step": ms, "cell_updates_per_s": ncells_local / ms * 1e3, "frac": 32.0 * V * ncells_local / st_s / 8.0e12, "rows_full_form": rows,
                     "tiles_full_form": tiles}
                share = tiles if tiles is not None else rows          # (the share of the tracer kernel's iterations that move cloud and rain)
                if share is not None and V == 8:
                    e["V_moved"] = 6.0 + 2.0 * share
                    e["frac_moved"] = 32.0 * (6.0 + 2.0 * share) * ncells_local / st_s / 8.0e12
                return e
            out["roofline"]["frac_by_state"] = {
                "cloud_free": _state(stage_ms * 3.0, 0.0 if out["config"]["zero_row_maps"] else 1.0, 0.0 if out["config"]["zero_row_maps"] else 1.0),
                "cloud_free_sustained": _state((out.get("sustained") or {}).get("ms_per_step"), 0.0 if out["config"]["zero_row_maps"] else 1.0, 0.0 if out["config"]["zero_row_maps"] else 1.0),
                "storm": _state((out.get("storm") or {}).get("ms_per_step"), (out.get("storm") or {}).get("rows_full_form"), (out.get("storm") or {}).get("tiles_full_form")),
                "developed": _state(out.get("developed_ms_per_step"), out.get("developed_rows_full_form"), out.get("developed_tiles_full_form")),
                "mature": _state((out.get("mature") or {}).get("ms_per_step"), (out.get("mature") or {}).get("rows_full_form"), (out.get("mature") or {}).get("tiles_full_form")),
                "what": "frac = 32 V B (V = 8) x cells / (ms_per_step / 3) / 8 TB/s per state; rows_full_form = share of (level, row) words of the stage maps in "
                        "which cloud or rain may be non-zero; tiles_full_form = the same per 58-cell x tile (the x segments of the maps: the fused tracer kernel's FULL "
                        "iterations); frac_moved prices only the variables that move: 6 + 2 x tiles_full_form.  cloud_free = the headline state (this line's roofline.frac), timed over the first 0.1 s after an idle start; cloud_free_sustained = the same state and step "
                        "after 1200 back-to-back steps, the card at its 1400 W limit (every state below runs there); storm = 725 s into the run, "
                        "mature = 3600 s into the run: the dycore steps of the loop's own last 100 iterations in front of those times, hipEvents around each "
                        "(storm.isolated_after / mature.isolated_after = rounds 4-6's figure, back-to-back dycore steps behind the loop: a state the simulation never visits); "
                        "developed = a seeded stress state with cloud / rain rims everywhere"}
            out["roofline"]["state"] = "cloud_free (headline; see frac_by_state for the storm / developed / mature states)"
        if world == 1 and not a.no_cpu_baseline:
            progress("cpu baseline ...")
            out["cpu_baseline"] = cpu_baseline(a.cpu_sample)
        pending[0] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    wd.cancel()


if __name__ == "__main__":
    main()

"""Sweep of the marching kernels' chunk lengths (options chunk_y / chunk_z / chunk_f; 0 = the chunk model) on config 2, one process, interleaved
repetitions: cloud-free state and the developed storm (--file from tools/storm_state.py save).  One JSON line."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miniweatherml_amd import modules
ap = argparse.ArgumentParser(); ap.add_argument("--file", default="/tmp/storm.pt"); ap.add_argument("--steps", type=int, default=15); ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
nx, ny, nz = 400, 400, 100
c, d, _ = modules.make_supercell(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny, 20000.0)
dt = d.compute_time_step(c)
NAMES = ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "cloud_liquid", "precip_liquid")
dm = c.get_data_manager_readwrite()
states = {"cloud_free": {k: dm.get(k).clone() for k in NAMES}}
if os.path.exists(a.file):
    states["storm"] = {k: v.to(dm.get(k).device) for k, v in torch.load(a.file).items()}
VARIANTS = [("default", {}), ("f20", {"chunk_f": 20}), ("f34", {"chunk_f": 34}), ("f50", {"chunk_f": 50}), ("f100", {"chunk_f": 100}),
            ("z20", {"chunk_z": 20}), ("z34", {"chunk_z": 34}), ("z50", {"chunk_z": 50}),
            ("y40", {"chunk_y": 40}), ("y67", {"chunk_y": 67}), ("y80", {"chunk_y": 80}), ("y100", {"chunk_y": 100}), ("y134", {"chunk_y": 134})]


def timed(opts):
    for k in ("chunk_y", "chunk_z", "chunk_f"):
        d.set_option(k, opts.get(k, 0))
    for _ in range(3):
        d.time_step(c, dt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        d.time_step(c, dt)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.steps


res = {}
for sname, st in states.items():
    t = {n: [] for n, _ in VARIANTS}
    for _ in range(a.reps):
        for n, o in VARIANTS:
            for k in NAMES:
                dm.get(k).copy_(st[k])
            t[n].append(timed(o))
    res[sname] = {n: round(min(v), 4) for n, v in t.items()}
    res[sname + "_ratio"] = {n: round(min(v) / min(t["default"]), 4) for n, v in t.items()}
print(json.dumps(res))

"""Round 6, VERDICT item 4 -- MEASURE the slab-stepped stage (option slab_rows): k_y_all -> k_xz_state -> k_tracers_fused y slab by y slab, so that the
hand-off arrays (y tendencies 40 B, face mass fluxes + selectors 18 B, tracer y fluxes <= 24 B per cell) of a slab are still in the 256 MB
Infinity Cache when the next launch reads them.  Same-process, interleaved A/B on config 2 (400 x 400 x 100), cloud-free state and the seeded
'developed' state; prints one JSON line.   python tools/slab_ab.py [--reps 3] [--steps 20]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miniweatherml_amd import modules

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=3); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--slabs", default="0,16,24,32,48,64,100,200"); a = ap.parse_args()
nx, ny, nz = 400, 400, 100
c, d, _ = modules.make_supercell(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny, 20000.0)
dt = d.compute_time_step(c)
NAMES = ["xz_state", "patch", "upd", "halo", "convert", "y_all", "y_tracers", "fused"]


def timed(slab):
    d.set_option("slab_rows", slab)
    for _ in range(3):
        d.time_step(c, dt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        d.time_step(c, dt)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.steps


def classes(slab):
    d.set_option("slab_rows", slab)
    d.time_step(c, dt)
    d.profile(1)
    for _ in range(3):
        d.time_step(c, dt)
    out = {n: round(d.profile_get(i)[0] / 3, 3) for i, n in enumerate(NAMES)}
    d.profile(0)
    return out


slabs = [int(v) for v in a.slabs.split(",")]
res = {"grid": "400x400x100", "steps": a.steps, "cells_per_slab_row": nx * nz,
       "handoff_bytes_per_cell": 40 + 18 + 8, "mall_bytes": 256 << 20, "states": {}}
for state in ("cloud_free", "developed"):
    if state == "developed":
        dm = c.get_data_manager_readwrite(); rho_d = dm.get("density_dry")
        k = torch.arange(nz, device=rho_d.device, dtype=torch.float64).view(nz, 1, 1, 1)
        j = torch.arange(ny, device=rho_d.device, dtype=torch.float64).view(1, ny, 1, 1)
        i = torch.arange(nx, device=rho_d.device, dtype=torch.float64).view(1, 1, nx, 1)
        blob = ((torch.sin(i * 0.11) * torch.cos(j * 0.07)) > 0.3).to(torch.float64)
        dm.get("cloud_liquid").copy_(2.0e-3 * blob * ((k > 0.15 * nz) & (k < 0.45 * nz)) * (0.5 + 0.5 * torch.sin(0.3 * k + 0.05 * i) ** 2) * rho_d)
        dm.get("precip_liquid").copy_(4.0e-4 * blob * (k < 0.3 * nz) * (0.5 + 0.5 * torch.cos(0.2 * k + 0.03 * j) ** 2) * rho_d)
    t = {s: [] for s in slabs}
    for _ in range(a.reps):
        for s in slabs:
            t[s].append(timed(s))
    res["states"][state] = {"ms_per_step": {str(s): [round(v, 4) for v in t[s]] for s in slabs},
                            "best_ms": {str(s): round(min(t[s]), 4) for s in slabs},
                            "ratio_to_whole_block": {str(s): round(min(t[s]) / min(t[0]), 4) for s in slabs},
                            "classes_ms_per_step": {str(s): classes(s) for s in (0, 24, 48, 100)}}
d.set_option("slab_rows", 0)
print(json.dumps(res))

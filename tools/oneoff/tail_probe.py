"""Probe of the launch-tail (wave-slot quantisation) effect: per-class kernel time per cell for several ny on 400 x ny x 100, and for
several chunk_z / chunk_f options on the headline grid.  python tools/tail_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miniweatherml_amd import modules

NAMES = ["xz_state", "patch", "upd", "halo", "convert", "y_state", "y_tracers", "fused"]


def run(ny, nz=100, nx=400, env=None):
    for k, v in (env or {}).items(): modules.DEFAULT_OPTIONS[k] = int(v)
    coupler, dycore, _ = modules.make_supercell(nx, ny, nz, 1, 500. * nx, 500. * ny, 2e4)
    dt = dycore.compute_time_step(coupler)
    for _ in range(3): dycore.time_step(coupler, dt)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): dycore.time_step(coupler, dt)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    dycore.profile(1)
    for _ in range(5): dycore.time_step(coupler, dt)
    cls = {n: dycore.profile_get(i)[0] / 5 for i, n in enumerate(NAMES)}
    dycore.profile(0)
    cells = nx * ny * nz
    tiles = (nx + 57) // 58
    for k in (env or {}): del modules.DEFAULT_OPTIONS[k]
    print("ny %4d nz %3d %-28s step %.3f ms  ns/cell: step %.4f xz %.4f fused %.4f y_state %.4f y_tr %.4f   xz rounds(4 chunks) %.2f" % (
        ny, nz, str(env or ""), ms, ms * 1e6 / cells, cls["xz_state"] * 1e6 / cells, cls["fused"] * 1e6 / cells, cls["y_state"] * 1e6 / cells,
        cls["y_tracers"] * 1e6 / cells, tiles * ny * 4 / 2048.), flush=True)
    del coupler, dycore
    torch.cuda.empty_cache()


what = sys.argv[1] if len(sys.argv) > 1 else "xz"
if what == "xz":
    for ny in (292, 293, 300, 340, 365, 366, 380, 400, 420, 438, 439, 512):
        run(ny)
    for cz in (13, 17, 20, 25, 34, 50, 100):
        run(400, env={"chunk_z": cz, "chunk_f": cz})
elif what == "y":                                               # y chunk sizes on the headline grid
    for cy in (25, 29, 34, 40, 45, 50, 58, 67, 80, 100, 134, 200):
        run(400, env={"chunk_y": cy, "chunk_yt": cy})
elif what == "small":                                           # z chunk counts on small grids
    for n in (200, 100):
        for cz in (5, 7, 9, 10, 13, 17, 25, 50):
            run(n, nz=50, nx=n, env={"chunk_z": cz, "chunk_f": cz})
        run(n, nz=50, nx=n)

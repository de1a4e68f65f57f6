"""Kernel classes of a dycore step on the developed storm (2600 steps of the complete loop) under different options, same process."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from miniweatherml_amd import modules
KN = ["xz_state", "tracer_patch", "tracer_update_unfused", "halo", "convert", "y_state", "y_tracers", "tracers_fused"]
nx, ny, nz = 400, 400, 100
c, d, m, n = modules.make_supercell(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny, 20000.0, with_nudger=True)
dt = d.compute_time_step(c)
def classes(tag):
    for _ in range(3): d.time_step(c, dt)
    d.profile(True)
    for _ in range(10): d.time_step(c, dt)
    torch.cuda.synchronize()
    r = {k: round(d.profile_get(i)[0] / 10.0, 3) for i, k in enumerate(KN)}
    d.profile(False)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); ev0.record()
    for _ in range(10): d.time_step(c, dt)
    ev1.record(); torch.cuda.synchronize()
    print("%-28s ms/step %.3f" % (tag, ev0.elapsed_time(ev1) / 10), {k: v for k, v in r.items() if v > 0.05}, flush=True)
while d.etime < 2600 * dt - 1e-9:
    modules.supercell_step(c, d, m, n, dt)
keep = {k: c.get_data_manager_readwrite().get(k).clone() for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "cloud_liquid", "precip_liquid")}
def reset():
    dm = c.get_data_manager_readwrite()
    for k, v in keep.items(): dm.get(k).copy_(v)
for rep in range(2):
    for opts in ({}, {"chunk_f": 13}, {"chunk_f": 17}, {"chunk_f": 9}, {"tf_rows4": 0}, {"zero_rows": 0}):
        reset()
        for k in ("chunk_f", "tf_rows4", "zero_rows"): d.set_option(k, {"chunk_f": 0, "tf_rows4": 1, "zero_rows": 1}[k])
        for k, v in opts.items(): d.set_option(k, v)
        classes(str(opts))

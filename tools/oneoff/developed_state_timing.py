"""Dycore / Kessler time per call on config 2 at the start of a supercell run and in the developed storm (quoted in DESIGN.md)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miniweatherml_amd import modules
coupler, dycore, micro, nudger = modules.make_supercell(400, 400, 100, 1, 2e5, 2e5, 2e4, with_nudger=True)
def timed(fn, n=10):
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
dt = dycore.compute_time_step(coupler)
def report(tag):
    import copy
    f = coupler.get_data_manager_readonly()
    print(tag, "max|w| %.2f cloud %.2e rain %.2e" % (float(f.get("wvel", True).abs().max()), float(f.get("cloud_liquid", True).max()), float(f.get("precip_liquid", True).max())), flush=True)
for _ in range(3): modules.supercell_step(coupler, dycore, micro, nudger)
report("start")
def classes():
    dycore.set_option("overlap", 0)
    dycore.time_step(coupler, dt); torch.cuda.synchronize()
    dycore.profile(1)
    for _ in range(5): dycore.time_step(coupler, dt)
    names = ["xz_state", "patch", "upd", "halo", "convert", "y_state", "y_tracers", "fused"]
    out = {n: round(dycore.profile_get(i)[0] / 5, 3) for i, n in enumerate(names)}
    dycore.profile(0); dycore.set_option("overlap", -1)
    return out
print("dycore ms", timed(lambda: dycore.time_step(coupler, dt)), "kessler ms", timed(lambda: micro.time_step(coupler, dt)), classes(), flush=True)
for s in range(2600):
    modules.supercell_step(coupler, dycore, micro, nudger)
report("developed")
print("dycore ms", timed(lambda: dycore.time_step(coupler, dt)), "kessler ms", timed(lambda: micro.time_step(coupler, dt)), classes(), flush=True)

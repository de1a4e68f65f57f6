"""Per-call time of the per-step modules around the dycore on config 2 (quoted in DESIGN.md section 5)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miniweatherml_amd import modules
coupler, dycore, micro, nudger = modules.make_supercell(400, 400, 100, 1, 2e5, 2e5, 2e4, with_nudger=True)
dt = dycore.compute_time_step(coupler)
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("kessler ms", timed(lambda: micro.time_step(coupler, dt)))
print("sponge ms", timed(lambda: modules.sponge_layer(coupler, dt)))
print("nudger ms", timed(lambda: nudger.nudge_to_column(coupler, dt)))

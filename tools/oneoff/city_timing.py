"""BASELINE.json configs[4] on one GPU: simple_city 512 x 512 x 256 (one tracer, immersed boundaries, gravity off)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miniweatherml_amd import modules
coupler, dycore, hs, ta = modules.make_simple_city(512, 512, 256, 1, 2560., 2560., 1280., "city")
dt = dycore.compute_time_step(coupler)
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
ms = timed(lambda: dycore.time_step(coupler, dt))
print("city 512x512x256 V=6 dycore ms", ms, "cell-updates/s %.3e" % (512*512*256/ms*1e3))
ms2 = timed(lambda: modules.simple_city_step(coupler, dycore, hs, ta))
print("full simple_city step ms", ms2, "%.3e" % (512*512*256/ms2*1e3))

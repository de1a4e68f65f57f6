"""What hipEvent pairs inside a time step cost (mw_dycore_profile modes 3, 2, 1): the same 40 steps with and without them, interleaved.  (Measured, round 5:
mode 2 -- bench.py's timed region until then -- 1.5 % of the step, mode 1 5 %; the timed region now runs mode 3.  The step time itself climbs by 10 % over
the 600 steps of this script: the part throttles.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from miniweatherml_amd import modules
c, d, _ = modules.make_supercell(400, 400, 100, 1, 200000., 200000., 20000.)
dt = d.compute_time_step(c)
for _ in range(5): d.time_step(c, dt)
def run(mode, n=40):
    d.profile(mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): d.time_step(c, dt)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / n * 1e3
    d.profile(0)
    return el
for rep in range(4):
    print("rep %d: no events %.3f ms/step | mode 3 (a pair per time step) %.3f | mode 2 (a pair per stage and per k_xz_state) %.3f | mode 1 (every kernel class) %.3f"
          % (rep, run(0), run(3), run(2), run(1)), flush=True)

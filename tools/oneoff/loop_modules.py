"""Where a step of the complete supercell loop goes: dycore | Kessler | sponge | nudger, hipEvent-timed per module over 20 steps, on the
cloud-free state after 50 steps and on the developed storm (2600 steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from miniweatherml_amd import modules
nx, ny, nz = 400, 400, 100
c, d, m, n = modules.make_supercell(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny, 20000.0, with_nudger=True)
dt = d.compute_time_step(c)
def measure(tag):
    names = ["dycore", "kessler", "sponge", "nudger"]
    fns = [lambda: d.time_step(c, dt), lambda: m.time_step(c, dt), lambda: modules.sponge_layer(c, dt), lambda: n.nudge_to_column(c, dt)]
    tot = [0.0] * 4
    for _ in range(20):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        evs[0].record()
        for i, f in enumerate(fns):
            f(); evs[i + 1].record()
        torch.cuda.synchronize()
        for i in range(4): tot[i] += evs[i].elapsed_time(evs[i + 1])
    print(tag, {k: round(v / 20, 3) for k, v in zip(names, tot)}, "sum %.3f" % (sum(tot) / 20), flush=True)
for _ in range(50): modules.supercell_step(c, d, m, n, dt)
measure("step 50  ")
while d.etime < 2600 * dt - 1e-9: modules.supercell_step(c, d, m, n, dt)
measure("step 2600")

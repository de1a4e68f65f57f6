"""rocprofv3 --kernel-trace --stats -- python3 tools/oneoff/hsum_probe.py : 50 eager nudges + sponges on config 2 (k_hsum_partial's average)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from miniweatherml_amd import modules
c, d, m, n = modules.make_supercell(400, 400, 100, 1, 2e5, 2e5, 2e4, with_nudger=True)
dt = d.compute_time_step(c)
for _ in range(50):
    modules.sponge_layer(c, dt)
    n.nudge_to_column(c, dt)
torch.cuda.synchronize()

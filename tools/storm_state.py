"""save: spin the complete supercell loop up for --steps steps and save the fields; run: load them and take --n dycore steps (for rocprofv3)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miniweatherml_amd import modules
ap = argparse.ArgumentParser(); ap.add_argument("mode"); ap.add_argument("--steps", type=int, default=2600); ap.add_argument("--n", type=int, default=6)
ap.add_argument("--file", default="/tmp/storm.pt"); a = ap.parse_args()
NAMES = ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "cloud_liquid", "precip_liquid")
nx, ny, nz = 400, 400, 100
c, d, m, n = modules.make_supercell(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny, 20000.0, with_nudger=True)
dt = d.compute_time_step(c)
if a.mode == "save":
    for _ in range(a.steps):
        modules.supercell_step(c, d, m, n, dt)
    dm = c.get_data_manager_readonly()
    torch.save({k: dm.get(k, True).cpu() for k in NAMES}, a.file)
else:
    if a.mode == "run":
        st = torch.load(a.file)
        dm = c.get_data_manager_readwrite()
        for k in NAMES:
            dm.get(k).copy_(st[k].to(dm.get(k).device))
    for _ in range(a.n):
        d.time_step(c, dt)
    torch.cuda.synchronize()

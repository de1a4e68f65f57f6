"""save: spin the complete supercell loop up for --steps steps and save the fields; run: load them and take --n dycore steps (for rocprofv3);
initial: the steps on the cloud-free start; developed: on bench.py's seeded stress state (cloud and rain blobs with sharp rims everywhere)."""
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
    if a.mode == "developed":                                        # bench.py's developed_state, spelled the same way
        dm = c.get_data_manager_readwrite()
        rho_d = dm.get("density_dry")
        k = torch.arange(nz, device=rho_d.device, dtype=torch.float64).view(nz, 1, 1, 1)
        j = torch.arange(ny, device=rho_d.device, dtype=torch.float64).view(1, ny, 1, 1)
        i = torch.arange(nx, device=rho_d.device, dtype=torch.float64).view(1, 1, nx, 1)
        blob = ((torch.sin(i * 0.11) * torch.cos(j * 0.07)) > 0.3).to(torch.float64)
        dm.get("cloud_liquid").copy_(2.0e-3 * blob * ((k > 0.15 * nz) & (k < 0.45 * nz)) * (0.5 + 0.5 * torch.sin(0.3 * k + 0.05 * i) ** 2) * rho_d)
        dm.get("precip_liquid").copy_(4.0e-4 * blob * (k < 0.3 * nz) * (0.5 + 0.5 * torch.cos(0.2 * k + 0.03 * j) ** 2) * rho_d)
    for _ in range(a.n):
        d.time_step(c, dt)
    torch.cuda.synchronize()

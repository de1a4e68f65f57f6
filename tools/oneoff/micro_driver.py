"""Runs ONE micro workload repeatedly so that rocprofv3 attributes its kernels cleanly (tools/profile_micro.sh):
    python3 tools/micro_driver.py kessler_scattered|kessler_storm|kessler_dry|mlp [--iters 20] [--nx 400 --ny 400 --nz 100]
Same states as bench.py's micro section (72 algorithmic bytes per cell: 5 fields read, 4 written)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from miniweatherml_amd import modules

ap = argparse.ArgumentParser()
ap.add_argument("what", choices=["kessler_scattered", "kessler_storm", "kessler_dry", "mlp"])
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--nx", type=int, default=400); ap.add_argument("--ny", type=int, default=400); ap.add_argument("--nz", type=int, default=100)
a = ap.parse_args()
coupler, dycore, micro = modules.make_supercell(a.nx, a.ny, a.nz, 1, 500.0 * a.nx, 500.0 * a.ny, 20000.)
dm = coupler.get_data_manager_readwrite()
rho_d = dm.get("density_dry")
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda: torch.rand(rho_d.shape, generator=g, device="cuda", dtype=torch.float64)      # noqa: E731
qc = rnd() * 2e-3 * (rnd() > 0.6)
qr = rnd() * 3e-4 * (rnd() > 0.6)
box = torch.zeros_like(rho_d)
box[: int(0.6 * a.nz), a.ny // 4: a.ny // 2, a.nx // 4: a.nx // 2] = 1.0
m = {"kessler_scattered": 1.0, "kessler_storm": box, "kessler_dry": 0.0, "mlp": 1.0}[a.what]
dm.get("cloud_liquid").copy_(qc * rho_d * m); dm.get("precip_liquid").copy_(qr * rho_d * m)
saved = {n: dm.get(n).clone() for n in ("water_vapor", "cloud_liquid", "precip_liquid", "temp")}
dt = dycore.compute_time_step(coupler)
if a.what == "mlp":
    W1, b1, W2, b2, si, so = modules.load_surrogate_weights()
    ins = [dm.get(n) for n in ("temp", "density_dry", "water_vapor", "cloud_liquid", "precip_liquid")]
    outs = [torch.empty_like(ins[0]) for _ in range(4)]
    for _ in range(a.iters):
        modules.mlp_forward(*ins, W1, b1, W2, b2, si, so, outs)
else:
    for _ in range(a.iters):
        for n, t in saved.items():
            dm.get(n).copy_(t)
        micro.time_step(coupler, dt)
torch.cuda.synchronize()
print("done", a.what, a.iters)

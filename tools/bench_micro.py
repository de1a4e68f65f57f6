"""Kessler and surrogate-MLP throughput on the supercell 400x400x100 grid (BASELINE.json configs[1]/[2] sizes), GPU only
(the CPU oracle is test infrastructure: tests/bench_oracle_micro.py times it).  Prints one JSON object; results are quoted in DESIGN.md section 5.
    python tools/bench_micro.py [--nx 400 --ny 400 --nz 100 --iters 20]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miniweatherml_amd import modules

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=400); ap.add_argument("--ny", type=int, default=400); ap.add_argument("--nz", type=int, default=100)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--no-cpu", action="store_true", help="(kept for old command lines; the CPU oracle is timed by tests/bench_oracle_micro.py)")
a = ap.parse_args()

coupler, dycore, micro = modules.make_supercell(a.nx, a.ny, a.nz, 1, 500.0 * a.nx, 500.0 * a.ny, 20000.)
dm = coupler.get_data_manager_readwrite()
ncell = a.nx * a.ny * a.nz
# a rainy state so that every Kessler branch runs (about 40 % of the cells active, generate_micro_surrogate_data.h:47-49)
g = torch.Generator(device="cuda").manual_seed(1)
rho_d = dm.get("density_dry")
qc = torch.rand(rho_d.shape, generator=g, device="cuda", dtype=torch.float64) * 2e-3 * (torch.rand(rho_d.shape, generator=g, device="cuda") > 0.6)
qr = torch.rand(rho_d.shape, generator=g, device="cuda", dtype=torch.float64) * 3e-4 * (torch.rand(rho_d.shape, generator=g, device="cuda") > 0.6)
dm.get("cloud_liquid").copy_(qc * rho_d); dm.get("precip_liquid").copy_(qr * rho_d)
saved = {n: dm.get(n).clone() for n in ("water_vapor", "cloud_liquid", "precip_liquid", "temp")}
dt = dycore.compute_time_step(coupler)


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def kessler_once():
    for n, t in saved.items():
        dm.get(n).copy_(t)
    micro.time_step(coupler, dt)


def restore_only():
    for n, t in saved.items():
        dm.get(n).copy_(t)


t_k = timed(kessler_once, a.iters) - timed(restore_only, a.iters)
rs = micro.time_step(coupler, dt, return_rainsplit=True)
# the same with spatially coherent cloud/rain (one storm: 1/16 of the columns, lower 60 % of the levels) -- rain-free
# wavefronts take the short cuts of mw_kessler.hip
scattered = {n: t.clone() for n, t in saved.items()}
box = torch.zeros_like(rho_d)
box[: int(0.6 * a.nz), a.ny // 4: a.ny // 2, a.nx // 4: a.nx // 2] = 1.0
for n in ("cloud_liquid", "precip_liquid"):
    saved[n] = scattered[n] * box
t_kc = timed(kessler_once, a.iters) - timed(restore_only, a.iters)
for n in scattered:
    saved[n] = scattered[n]
restore_only()
W1, b1, W2, b2, si, so = modules.load_surrogate_weights()
ins = [dm.get(n) for n in ("temp", "density_dry", "water_vapor", "cloud_liquid", "precip_liquid")]
outs = [torch.empty_like(ins[0]) for _ in range(4)]
t_m = timed(lambda: modules.mlp_forward(*ins, W1, b1, W2, b2, si, so, outs), a.iters)
res = {"grid": [a.nx, a.ny, a.nz], "cells": ncell,
       "kessler": {"s_per_call": t_k, "cells_per_s": ncell / t_k, "rainsplit": rs, "alg_bytes_per_cell": 72 + 8.0 / a.nz,
                   "hbm_GBps_alg": ncell * (72 + 8.0 / a.nz) / t_k / 1e9, "frac_of_8TBps": ncell * (72 + 8.0 / a.nz) / t_k / 8e12,
                   "state": "cloud/rain scattered at random over 40 % of the cells (worst case: no rain-free wavefront)",
                   "one_storm": {"s_per_call": t_kc, "cells_per_s": ncell / t_kc, "frac_of_8TBps": ncell * (72 + 8.0 / a.nz) / t_kc / 8e12}},
       "mlp": {"s_per_call": t_m, "cells_per_s": ncell / t_m, "alg_bytes_per_cell": 72, "hbm_GBps_alg": ncell * 72 / t_m / 1e9,
               "frac_of_8TBps": ncell * 72 / t_m / 8e12, "fp32_gflops_nominal": ncell * 208 / t_m / 1e9}}
print(json.dumps(res))

"""CPU oracle throughput of Kessler and the surrogate MLP on one host core (bounded sample), the CPU figures quoted next to
tools/bench_micro.py's GPU numbers in DESIGN.md.  Lives under tests/ because it uses the oracle.
    python tests/bench_oracle_micro.py [--nz 100]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mw_oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nz", type=int, default=100)
a = ap.parse_args()
nxs, nys = 100, 100
dyc, f = O.supercell_setup(nxs, nys, a.nz, 1, 500.0 * nxs, 500.0 * nys, 20000.)
dt = dyc.compute_time_step()
rng = np.random.default_rng(1)
f.tracers[1][...] = rng.uniform(0, 2e-3, f.rho_d.shape) * (rng.uniform(size=f.rho_d.shape) > 0.6) * f.rho_d
f.tracers[2][...] = rng.uniform(0, 3e-4, f.rho_d.shape) * (rng.uniform(size=f.rho_d.shape) > 0.6) * f.rho_d
precl = np.zeros((nys, nxs, 1))
t0 = time.perf_counter()
for _ in range(5):
    O.kessler_time_step(20000. / a.nz, dt, f.tracers[0], f.tracers[1], f.tracers[2], f.rho_d, f.temp, precl)
tk = (time.perf_counter() - t0) / 5
data = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "miniweatherml_amd", "data")
w = np.loadtxt(os.path.join(data, "kessler_surrogate_weights.txt"), comments="#").astype(np.float32)
W1, b1, W2, b2 = w[:50].reshape(5, 10).copy(), w[50:60].copy(), w[60:100].reshape(10, 4).copy(), w[100:104].copy()
si = np.ascontiguousarray(np.loadtxt(os.path.join(data, "kessler_surrogate_input_scaling.txt")).reshape(5, 2))
so = np.ascontiguousarray(np.loadtxt(os.path.join(data, "kessler_surrogate_output_scaling.txt")).reshape(4, 2))
t0 = time.perf_counter()
for _ in range(5):
    O.mlp_forward(f.temp, f.rho_d, f.tracers[0], f.tracers[1], f.tracers[2], W1, b1, W2, b2, si, so)
tm = (time.perf_counter() - t0) / 5
n = nxs * nys * a.nz
print(json.dumps({"cpu_oracle_1core": {"sample_cells": n, "kessler_cells_per_s": n / tk, "mlp_cells_per_s": n / tm}}))

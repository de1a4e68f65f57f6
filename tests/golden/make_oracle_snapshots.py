"""Regenerates tests/golden/oracle_snapshots.json: per-field statistics of the CPU oracle after a few steps on the
small configurations the GPU parity tests use.  These are ORACLE SELF-SNAPSHOTS (regression guard for
oracle/mw_oracle.cpp), not reference vectors -- the reference-run vectors are baseline_known_answers.json.

    python tests/golden/make_oracle_snapshots.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mw_oracle as O  # noqa: E402

CASES = {
    # name: (nx, ny, nz, nens, xlen, ylen, zlen, init_data, num_tracers, enable_gravity, nsteps)
    "supercell3d_16x16x8": (16, 16, 8, 1, 16000., 16000., 20000., "supercell", 3, True, 3),
    "supercell2d_64x1x32": (64, 1, 32, 1, 100000., 100000., 20000., "supercell", 3, True, 3),
    "supercell3d_nens2_12x10x8": (12, 10, 8, 2, 6000., 5000., 20000., "supercell", 3, True, 2),
    "thermal3d_16x16x16": (16, 16, 16, 1, 20000., 20000., 10000., "thermal", 3, True, 3),
    "building_40x40x16_nograv": (40, 40, 16, 1, 200., 200., 80., "building", 1, False, 3),
    "city_48x48x12_nograv": (48, 48, 12, 1, 2400., 2400., 120., "city", 1, False, 2),
}


def stats(a):
    a = np.asarray(a, dtype=np.float64)
    return {"min": float(a.min()), "max": float(a.max()), "sum": float(np.sum(a.ravel().tolist()) if a.size < 1 else
                                                                         __import__("math").fsum(a.ravel().tolist())),
            "sumsq": float(__import__("math").fsum((a.ravel() ** 2).tolist()))}


def run(case):
    nx, ny, nz, nens, xlen, ylen, zlen, init, nt, grav, nsteps = case
    dyc, f = O.supercell_setup(nx, ny, nz, nens, xlen, ylen, zlen, init_data=init, num_tracers=nt, enable_gravity=grav,
                               perturb=(init == "supercell"))
    dt = dyc.compute_time_step()
    for _ in range(nsteps):
        dyc.time_step(f, dt)
    out = {k: stats(v) for k, v in f.as_dict().items()}
    out["immersed_cells"] = float(dyc.immersed_proportion().sum())
    return out


if __name__ == "__main__":
    res = {name: run(case) for name, case in CASES.items()}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_snapshots.json"), "w") as fh:
        json.dump({"cases": {k: list(v) for k, v in CASES.items()}, "stats": res}, fh, indent=1, sort_keys=True)
    print("wrote", len(res), "cases")

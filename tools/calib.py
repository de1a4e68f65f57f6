"""The measured fp64 ceiling and the arithmetic floor of a stage on this GPU (round 5):
    python tools/calib.py [seconds per FMA run]
prints one JSON object: v_fma_f64 issue rate at 1 / 2 / 4 / 8 wavefronts per SIMD (and the clock it ran at), and the time of the
stage's bare arithmetic (24 WENO-5 + 3 Riemann per cell) for config 2's 1.6e7 cells on smooth and on rough data."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miniweatherml_amd import calib

sec = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
out = {"fma64": [calib.fma64(w, sec) for w in (1, 2, 4, 8)],
       "stage_arith": [calib.stage_arith(kind, levels=lv) for kind in ("smooth", "cloud_free", "rough") for lv in (25, 100)]}
print(json.dumps(out))

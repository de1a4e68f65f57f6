"""One-off: the seeded zero-row-map sweeps of tests/ over many more seeds (not part of the suite)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import miniweatherml_amd as mw
import test_gpu_options as TO
import test_gpu_multirank as TM
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(a, b):
    try:
        TO.test_zero_row_maps_on_random_configurations(mw, seed)
    except AssertionError as e:
        bad.append(("one-rank", seed, str(e)[:200]))
    except Exception as e:
        if "grid too small" not in str(e): bad.append(("one-rank", seed, repr(e)[:200]))
print("one-rank seeds %d..%d: %d failures" % (a, b, len(bad)), bad[:5], flush=True)
bad2 = []
for seed in range(a, a + (b - a) // 8):
    try:
        TM.test_zero_row_maps_on_random_decompositions(mw, seed)
    except Exception as e:
        bad2.append((seed, repr(e)[:300]))
print("decompositions seeds %d..%d: %d failures" % (a, a + (b - a) // 8, len(bad2)), bad2[:5], flush=True)

"""One-off: the complete supercell loop at the benchmark's size, two handles in lockstep with and without the zero-row maps, every field compared
every 100 steps (the suite's test_zero_row_maps_through_a_developing_storm does this on 100 x 40 x 40)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from miniweatherml_amd import modules
names = ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "cloud_liquid", "precip_liquid")
nx, ny, nz, steps = 400, 400, 100, int(sys.argv[1]) if len(sys.argv) > 1 else 2600
runs = []
for rows in (1, 0):
    c, d, m, n = modules.make_supercell(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny, 20000.0, with_nudger=True)
    d.set_option("zero_rows", rows)
    runs.append((c, d, m, n))
dt = runs[0][1].compute_time_step(runs[0][0])
bad = 0
for s in range(1, steps + 1):
    for c, d, m, n in runs:
        modules.supercell_step(c, d, m, n, dt)
    if s % 100 == 0 or s == steps:
        a, b = (r[0].get_data_manager_readonly() for r in runs)
        ne = [k for k in names if not torch.equal(a.get(k, True), b.get(k, True))]
        bad += len(ne)
        print("step %d: %s   cloud cells %d  rain cells %d  max|w| %.2f" % (s, "EQUAL" if not ne else "DIFFERENT " + str(ne), int((a.get("cloud_liquid", True) != 0).sum()),
              int((a.get("precip_liquid", True) != 0).sum()), float(a.get("wvel", True).abs().max())), flush=True)
print("fields that differed at some check:", bad)

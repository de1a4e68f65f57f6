"""The complete supercell loop at the benchmark's size, two handles in lockstep: A = the production settings of round 6 (zero-row maps with x
segments, zero_verify on, the nudger's increments riding on the conversion), B = the plain forms (maps off, the nudger's own second pass).
Every field compared bit for bit every 100 steps, from the cloud-free start through the first cloud and rain to the storm; at the end the
violation counters of zero_verify and the deferred-nudge counters.  python tools/storm_lockstep.py [steps] -> log + one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miniweatherml_amd import modules
names = ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "cloud_liquid", "precip_liquid")
nx, ny, nz, steps = 400, 400, 100, int(sys.argv[1]) if len(sys.argv) > 1 else 2600
runs = []
for prod in (True, False):
    c, d, m, n = modules.make_supercell(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny, 20000.0, with_nudger=True)
    d.set_option("zero_rows", 1 if prod else 0)
    d.set_option("zero_verify", 1 if prod else 0)
    runs.append((c, d, m, n, prod))
dt = runs[0][1].compute_time_step(runs[0][0])
bad, checks = 0, 0
for s in range(1, steps + 1):
    for c, d, m, n, prod in runs:
        modules.supercell_step(c, d, m, n, dt, defer_nudge=prod)
    if s % 100 == 0 or s == steps:
        a, b = (r[0].get_data_manager_readonly() for r in runs)      # (get applies parked increments first)
        ne = [k for k in names if not torch.equal(a.get(k, True), b.get(k, True))]
        bad += len(ne); checks += 1
        print("step %d: %s   cloud cells %d  rain cells %d  max|w| %.2f" % (s, "EQUAL" if not ne else "DIFFERENT " + str(ne), int((a.get("cloud_liquid", True) != 0).sum()),
              int((a.get("precip_liquid", True) != 0).sum()), float(a.get("wvel", True).abs().max())), flush=True)
d = runs[0][1]
viol = d.zero_violations()
print(json.dumps({"steps": steps, "checks": checks, "fields_that_differed": bad, "zero_verify_violations": viol[0], "zero_verify_kinds": viol[1],
                  "deferred_nudge_rode_on_conversion_/_applied_by_pass": d.pending()[1], "path": d.path()}))

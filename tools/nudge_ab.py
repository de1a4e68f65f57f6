"""Round 6: the complete supercell loop with the nudger's second pass (eager) against increments that ride on the next dycore step's conversion
(mw_nudge_to_column_deferred): same-process interleaved A/B on config 2, cloud-free state, plus per-module times.  One JSON line."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miniweatherml_amd import modules
ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=3); ap.add_argument("--steps", type=int, default=40); a = ap.parse_args()
nx, ny, nz = 400, 400, 100
c, d, m, n = modules.make_supercell(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny, 20000.0, with_nudger=True)
dt = d.compute_time_step(c)


def timed(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


res = {"eager": [], "deferred": []}
for _ in range(a.reps):
    res["eager"].append(timed(lambda: modules.supercell_step(c, d, m, n, dt), a.steps))
    res["deferred"].append(timed(lambda: modules.supercell_step(c, d, m, n, dt, defer_nudge=True), a.steps))
d.flush_pending()
res["modules_ms"] = {"dycore": timed(lambda: d.time_step(c, dt), 20), "kessler": timed(lambda: m.time_step(c, dt), 20),
                     "sponge": timed(lambda: modules.sponge_layer(c, dt), 20), "nudger_eager": timed(lambda: n.nudge_to_column(c, dt), 20),
                     "nudger_deferred_sums_only": timed(lambda: n.nudge_to_column(c, dt, defer_to=d), 20)}
d.flush_pending()
res["pending_counters"] = d.pending()[1]
res["best"] = {k: min(v) for k, v in res.items() if isinstance(v, list)}
print(json.dumps(res))

"""A/B of two builds of the library on the complete supercell loop (MW_LIB_PATH picks the build): sha256 of all fields at checkpoints on the
way to the mature storm; in front of the checkpoints at 2600 and `steps` the loop iteration (100 iterations, events) and the tracer-patch
launches (profile class 1 over 20 iterations) timed INSIDE the loop, on the states the loop really passes through.  One JSON line.
    MW_LIB_PATH=... python tools/patch_ab.py [steps]"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miniweatherml_amd import modules, capi
names = ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "cloud_liquid", "precip_liquid")
nx, ny, nz, steps = 400, 400, 100, int(sys.argv[1]) if len(sys.argv) > 1 else 12900
c, d, m, n = modules.make_supercell(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny, 20000.0, with_nudger=True)
dt = d.compute_time_step(c)
out = {"lib": capi.LIB_PATH, "steps": steps, "hashes": {}, "timed": {}}
timed_at = [t for t in (2600, steps) if t <= steps]
s = 0
def run(k):
    global s
    for _ in range(k):
        modules.supercell_step(c, d, m, n, dt, defer_nudge=True)
    s += k
while s < steps:
    nxt = min([t for t in timed_at + [800, 6000, 9000, steps] if t > s])
    if nxt in timed_at and nxt - s > 120:
        run(nxt - s - 120)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(100); e1.record(); torch.cuda.synchronize()
        d.profile(1); run(20); pm, pc = d.profile_get(1); d.profile(0)
        out["timed"][str(nxt)] = {"ms_per_loop_iteration": e0.elapsed_time(e1) / 100, "tracer_patch_ms_per_step": pm / 20, "patch_launches": pc}
    else:
        run(nxt - s)
    dm = c.get_data_manager_readonly()
    h = hashlib.sha256()
    for k in names:
        h.update(dm.get(k, True).cpu().numpy().tobytes())
    out["hashes"][str(s)] = h.hexdigest()[:24]
print(json.dumps(out))

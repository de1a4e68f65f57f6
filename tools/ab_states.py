"""Same-box A/B of two builds of the library (MW_LIB_PATH) on two states of config 2: ms per dycore step on the cloud-free initial state and on
bench.py's seeded stress state, plus the fused tracer kernel's share (profile class 7).  One JSON line.  MW_LIB_PATH=... python tools/ab_states.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miniweatherml_amd import modules, capi
nx, ny, nz = 400, 400, 100
c, d, m = modules.make_supercell(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny, 20000.0)
dt = d.compute_time_step(c)
def timed(n, warm):
    for _ in range(warm):
        d.time_step(c, dt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        d.time_step(c, dt)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    d.profile(1)
    for _ in range(3):
        d.time_step(c, dt)
    tr = d.profile_get(7); pa = d.profile_get(1)
    d.profile(0)
    return {"ms_per_step": ms, "tracers_fused_ms_per_step": tr[0] / 3, "tracer_patch_ms_per_step": pa[0] / 3}
out = {"lib": os.path.basename(capi.LIB_PATH), "cloud_free": timed(20, 5)}
dm = c.get_data_manager_readwrite()
rho_d = dm.get("density_dry")
k = torch.arange(nz, device=rho_d.device, dtype=torch.float64).view(nz, 1, 1, 1)
j = torch.arange(ny, device=rho_d.device, dtype=torch.float64).view(1, ny, 1, 1)
i = torch.arange(nx, device=rho_d.device, dtype=torch.float64).view(1, 1, nx, 1)
blob = ((torch.sin(i * 0.11) * torch.cos(j * 0.07)) > 0.3).to(torch.float64)
dm.get("cloud_liquid").copy_(2.0e-3 * blob * ((k > 0.15 * nz) & (k < 0.45 * nz)) * (0.5 + 0.5 * torch.sin(0.3 * k + 0.05 * i) ** 2) * rho_d)
dm.get("precip_liquid").copy_(4.0e-4 * blob * (k < 0.3 * nz) * (0.5 + 0.5 * torch.cos(0.2 * k + 0.03 * j) ** 2) * rho_d)
out["developed"] = timed(6, 2)
import hashlib
h = hashlib.sha256()
for n in ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "cloud_liquid", "precip_liquid"):
    h.update(c.get_data_manager_readonly().get(n, True).cpu().numpy().tobytes())
out["sha"] = h.hexdigest()[:16]
print(json.dumps(out))

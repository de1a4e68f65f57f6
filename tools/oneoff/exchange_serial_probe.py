"""What the strip exchange costs when NOTHING runs beside it (option overlap = 0: one stream, the exchange in line): per step, the halo
class of mw_dycore_profile (pack kernels + transport + unpack kernels, events on the stream) for rank 0's 400 x 400 x 100 block of a 2 x 2
tiling with the RCCL self-loop transport.  Compared with the pipelined schedule's step time this says how much of the chain is hidden.
    python tools/exchange_serial_probe.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from miniweatherml_amd import modules

NAMES = ["xz_state", "patch", "upd", "halo", "convert", "y_state", "y_tracers", "fused"]
nx, ny, nz = 400, 400, 100
ref = modules.make_supercell(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny, 20000.)
start = {n: ref[0].get_data_manager_readonly().get(n, True).clone() for n in ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "cloud_liquid", "precip_liquid")}
out = {}
for label, opts in (("serial (overlap=0)", {"overlap": 0}), ("pipelined", {}), ("two streams", {"pipe": 0})):
    modules.DEFAULT_OPTIONS.clear(); modules.DEFAULT_OPTIONS.update(opts)
    coupler, dycore, _ = modules.make_supercell(2 * nx, 2 * ny, nz, 1, 1000.0 * nx, 1000.0 * ny, 20000., nranks=4, myrank=0)
    for n, t in start.items():
        coupler.get_data_manager_readwrite().get(n).copy_(t)
    modules.use_rccl_self_exchange(dycore, coupler)
    dt = dycore.compute_time_step(coupler)
    for _ in range(3):
        dycore.time_step(coupler, dt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        dycore.time_step(coupler, dt)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    dycore.profile(1)
    for _ in range(5):
        dycore.time_step(coupler, dt)
    cls = {n: round(dycore.profile_get(i)[0] / 5, 3) for i, n in enumerate(NAMES)}
    dycore.profile(0)
    out[label] = {"ms_per_step": round(ms, 3), "classes_ms_per_step": cls, "path": dycore.path()}
    del coupler, dycore
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))

"""A short run of rank 0's block of a 2 x 2 decomposition with the RCCL self-loop transport, for `rocprofv3 --kernel-trace`: the timeline of
the compute stream and the exchange stream of the pipelined schedule (tools/r05 analysis: where the compute stream waits).
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/selfloop_trace.py [steps] [key=value options ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from miniweatherml_amd import modules

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for kv in sys.argv[2:]:
    modules.DEFAULT_OPTIONS[kv.split("=")[0]] = int(kv.split("=")[1])
nx, ny, nz = 400, 400, 100
ref = modules.make_supercell(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny, 20000.)
coupler, dycore, _ = modules.make_supercell(2 * nx, 2 * ny, nz, 1, 1000.0 * nx, 1000.0 * ny, 20000., nranks=4, myrank=0)
for n in ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "cloud_liquid", "precip_liquid"):
    coupler.get_data_manager_readwrite().get(n).copy_(ref[0].get_data_manager_readonly().get(n, True))
modules.use_rccl_self_exchange(dycore, coupler)
dt = dycore.compute_time_step(coupler)
for _ in range(steps):
    dycore.time_step(coupler, dt)
torch.cuda.synchronize()
print("done", dycore.path())

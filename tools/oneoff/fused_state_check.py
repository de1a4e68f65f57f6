"""Field-by-field comparison of the fused state-stage kernel (option fused_state = 4, csrc/mw_fused.h; a -DMW_EXPERIMENTS build: MW_LIB_PATH=miniweatherml_amd/ab/libmw_exp.so) with the production schedule on one GPU:
    python tools/fused_state_check.py supercell|city <steps> [chunk_z] [rough]
prints max |difference|, where it sits and how many cells differ (0 everywhere = bitwise equal).  `rough` adds a random perturbation of
every field first (an indexing error then shows as an O(1) difference; a rounding-level one only on smooth states)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from miniweatherml_amd import modules
from util import gpu_fields
case = sys.argv[1]; nsteps = int(sys.argv[2]); chunk = sys.argv[3] if len(sys.argv) > 3 else None
res = {}
for fused in ("0", "1"):
    modules.DEFAULT_OPTIONS["fused_state"] = 4 if fused == "1" else 0
    if chunk: modules.DEFAULT_OPTIONS["chunk_z"] = int(chunk)
    if case == "supercell":
        coupler, dycore, _ = modules.make_supercell(130, 24, 26, 1, 65000., 12000., 20000.)
    else:
        coupler, dycore, _, _ = modules.make_simple_city(96, 48, 16, 1, 480., 240., 80., "building")
    if len(sys.argv) > 4:
        import torch
        g = torch.Generator(device="cuda").manual_seed(5)
        dm = coupler.get_data_manager_readwrite()
        for nme, amp in (("temp", 0.01), ("density_dry", 0.01)):
            t = dm.get(nme); t.mul_(1.0 + amp * (torch.rand(t.shape, generator=g, device="cuda", dtype=torch.float64) - 0.5))
        for nme, amp in (("uvel", 5.0), ("vvel", 5.0), ("wvel", 1.0)):
            t = dm.get(nme); t.add_(amp * (torch.rand(t.shape, generator=g, device="cuda", dtype=torch.float64) - 0.5))
    dt = dycore.compute_time_step(coupler)
    for n in range(nsteps):
        dycore.time_step(coupler, dt * (2.2 if n == 1 else 1.0))
    res[fused] = gpu_fields(coupler)
for k in res["0"]:
    d = np.abs(res["0"][k] - res["1"][k])
    idx = np.unravel_index(np.argmax(d), d.shape)
    print(case, k, "max diff %.3e at %s  scale %.3e  ndiff %d of %d" % (d.max(), idx, np.abs(res["0"][k]).max(), int((d > 0).sum()), d.size))

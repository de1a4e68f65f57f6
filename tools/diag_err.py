import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from miniweatherml_amd import modules
from oracle import mw_oracle as O
from util import gpu_fields, push_fields
nx, ny, nz = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16, 16, 8))]
nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
xl = float(os.environ.get("XL", 500.0 * nx))
INIT = os.environ.get("INIT", "supercell"); ZL = float(os.environ.get("ZL", 20000.))
odyc, of = O.supercell_setup(nx, ny, nz, 1, xl, xl if ny > 1 else 1e5, ZL, init_data=INIT, perturb=(INIT == "supercell"))
refs = []
f = of.copy()
dt = odyc.compute_time_step()
for s in range(nsteps):
    odyc.time_step(f, dt); refs.append(f.copy())
for mode in (1, 2, 0):
    coupler, dycore, micro = modules.make_supercell(nx, ny, nz, 1, xl, xl if ny > 1 else 1e5, ZL, INIT, perturb=(INIT == "supercell"))
    push_fields(coupler, of)
    dycore.set_strict(mode)
    out = []
    for s in range(nsteps):
        dycore.time_step(coupler, dt)
        g = gpu_fields(coupler); r = refs[s].as_dict()
        out.append(max(np.max(np.abs(g[k] - r[k])) / max(np.max(np.abs(r[k])), 1e-300) for k in ("density_dry", "temp", "tracer0")))
        outw = max(np.max(np.abs(g[k] - r[k])) for k in ("uvel", "vvel", "wvel"))
        out[-1] = (out[-1], outw)
    print("mode", mode, " ".join("%.1e/%.1e" % o for o in out))

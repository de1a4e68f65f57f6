"""Which device results are bit-identical to the CPU oracle?  (diagnostic; GPU box)"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from oracle import mw_oracle as O
from miniweatherml_amd import modules
from util import gpu_fields
import test_gpu_dycore_parity as T
for name in sorted(T.SNAP['cases']):
    case = T.SNAP['cases'][name]
    nx, ny, nz, nens, xlen, ylen, zlen, init, nt, grav, nsteps = case
    for perturb in (False, True):
        micro = None
        if nt == 1:
            class OneTracer(modules.Microphysics_Kessler):
                def init(self, coupler):
                    coupler.add_tracer("water_vapor", "Water Vapor", True, True)
            micro = OneTracer()
        coupler, dycore, _ = modules.make_supercell(nx, ny, nz, nens, xlen, ylen, zlen, init, micro=micro, enable_gravity=grav, perturb=perturb)
        odyc, of = O.supercell_setup(nx, ny, nz, nens, xlen, ylen, zlen, init_data=init, num_tracers=nt, enable_gravity=grav, perturb=perturb)
        g, o = gpu_fields(coupler), of.as_dict()
        bad = {k: float(np.max(np.abs(g[k] - o[k]))) for k in o if not np.array_equal(g[k], o[k])}
        print(name, 'perturb' if perturb else 'init only', 'BITWISE' if not bad else bad)

"""Soak run: the complete supercell loop for many steps with periodic file output; reports device-memory drift and field sanity.
    python tools/soak.py [nsteps]"""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miniweatherml_amd import modules

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
coupler, dycore, micro, nudger = modules.make_supercell(200, 200, 50, 1, 1e5, 1e5, 2e4, with_nudger=True)
tmp = tempfile.mkdtemp()
coupler.set_option("out_prefix", os.path.join(tmp, "soak"))
dycore.output(coupler, 0.0)
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
t = 0.0
for s in range(n):
    t += modules.supercell_step(coupler, dycore, micro, nudger)
    if (s + 1) % 100 == 0:
        dycore.output(coupler, t)
        torch.cuda.synchronize()
        f = coupler.get_data_manager_readonly()
        w = float(f.get("wvel", True).abs().max())
        qc = float(f.get("cloud_liquid", True).max())
        print("step %5d  t %8.1f s  max|w| %7.3f  max cloud %.3e  free-mem drift %+d MiB" %
              (s + 1, t, w, qc, (torch.cuda.mem_get_info()[0] - free0) // (1 << 20)), flush=True)
        assert w == w and w < 100.0
print("ok")

import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch, ctypes as C
from oracle import mw_oracle as O
from miniweatherml_amd import capi, modules
from util import gpu_fields, push_fields
import test_gpu_dycore_parity as T
libm_pow, restated = O.powcheck()
rng = np.random.default_rng(2)
n = 1_000_000
x = np.concatenate([rng.uniform(50, 400, n), rng.uniform(1e-5, 1e-1, n), np.exp(rng.uniform(-20, 20, n))])
y = np.concatenate([np.full(n, 1003/716.), np.full(n, 716/1003.), rng.uniform(-8, 8, n)])
xd, yd = torch.tensor(x, device='cuda'), torch.tensor(y, device='cuda')
out = torch.empty_like(xd); mp = torch.empty(xd.numel(), dtype=torch.uint8, device='cuda')
capi.check(capi.lib().mw_strict_pow(xd.numel(), xd.data_ptr(), yd.data_ptr(), out.data_ptr(), mp.data_ptr(), None))
torch.cuda.synchronize()
ref = libm_pow(x, y)
print('device pow vs libm: main', int(mp.sum()), 'mismatch', int(np.sum(out.cpu().numpy().view(np.uint64) != ref.view(np.uint64))))
for name in sorted(T.SNAP['cases']):
    coupler, dycore, odyc, of = T.setup_case(O, T.SNAP['cases'][name])
    push_fields(coupler, of)
    dycore.set_strict(1)
    dt = dycore.compute_time_step(coupler)
    for step in range(10):
        dycore.time_step(coupler, dt); odyc.time_step(of, dt)
        if step in (0, 9):
            g, o = gpu_fields(coupler), of.as_dict()
            diffs = {k: float(np.max(np.abs(g[k] - o[k]))) for k in o}
            print(name, 'step', step + 1, 'bitwise' if all(np.array_equal(g[k], o[k]) for k in o) else diffs)

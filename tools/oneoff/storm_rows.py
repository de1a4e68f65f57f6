"""How much of a developed storm the zero-row maps leave to the FULL form of the tracer kernel, and what x tiles instead of whole x rows
would leave (a what-if for DESIGN.md; torch arithmetic on the coupler's arrays, no library internals).
    python tools/storm_rows.py [--steps 2600]"""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from miniweatherml_amd import modules

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=2600); a = ap.parse_args()
nx, ny, nz = 400, 400, 100
c, d, m, n = modules.make_supercell(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny, 20000.0, with_nudger=True)
dt = d.compute_time_step(c)
out = {}
for target in sorted({a.steps // 2, a.steps}):
    while d.etime < target * dt - 1e-9:
        modules.supercell_step(c, d, m, n, dt)
    dm = c.get_data_manager_readonly()
    nzc = ((dm.get("cloud_liquid", True) != 0) | (dm.get("precip_liquid", True) != 0))[..., 0]      # (nz, ny, nx)

    def dil(x, dim, lo, hi, periodic):
        r = x.clone()
        for s in range(lo, hi + 1):
            if s == 0:
                continue
            if periodic:
                r |= torch.roll(x, shifts=-s, dims=dim)
            else:
                idx = (torch.arange(x.shape[dim], device=x.device) + s).clamp_(0, x.shape[dim] - 1)
                r |= x.index_select(dim, idx)
        return r
    rows = nzc.any(dim=2)                                                                           # M0[k][j]
    res = {"cells_nonzero": float(nzc.double().mean()), "rows_nonzero": float(rows.double().mean())}
    for s in (1, 2, 3):
        q = dil(dil(rows, 1, -3 * s, 3 * s, True), 0, -3 * s - 5, 3 * s + 1, False)
        res["Q%d_rows_full" % s] = float(q.double().mean())
    T = 58
    nt = (nx + T - 1) // T
    pad = torch.zeros(nz, ny, nt * T, dtype=torch.bool, device=nzc.device); pad[:, :, :nx] = nzc
    tiles = pad.view(nz, ny, nt, T).any(dim=3)                                                      # [k][j][tile]
    for s in (1, 2, 3):
        q = dil(dil(dil(tiles, 2, -1, 1, True), 1, -3 * s, 3 * s, True), 0, -3 * s - 5, 3 * s + 1, False)
        res["Q%d_tiles_full" % s] = float(q.double().mean())
    out["step_%d" % target] = res
print(json.dumps(out, indent=1))

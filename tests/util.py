"""Shared helpers for the parity tests (tests only)."""
import numpy as np


def rel_err(a, b):
    """max|a-b| / max|b|  (the tolerance definition of BASELINE.md section 4)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.max(np.abs(b))
    d = np.max(np.abs(a - b))
    return d / scale if scale > 0 else d


def gpu_fields(coupler):
    """coupler fields -> dict of numpy arrays, oracle naming (tracer0..)."""
    dm = coupler.get_data_manager_readonly()
    out = {n: dm.get(n, True).cpu().numpy() for n in ("density_dry", "uvel", "vvel", "wvel", "temp")}
    for t, n in enumerate(coupler.get_tracer_names()):
        out["tracer%d" % t] = dm.get(n, True).cpu().numpy()
    return out


def push_fields(coupler, f):
    """oracle Fields -> coupler tensors (identical inputs on both sides)."""
    import torch
    dm = coupler.get_data_manager_readwrite()
    for n, a in (("density_dry", f.rho_d), ("uvel", f.uvel), ("vvel", f.vvel), ("wvel", f.wvel), ("temp", f.temp)):
        dm.get(n).copy_(torch.from_numpy(a))
    for t, n in enumerate(coupler.get_tracer_names()):
        dm.get(n).copy_(torch.from_numpy(f.tracers[t]))


# absolute floor for fields that are identically ~0 in the oracle (e.g. vvel early in a symmetric run)
ABS_FLOOR = {"vvel": 1e-11, "wvel": 1e-11, "uvel": 1e-11, "tracer1": 1e-14, "tracer2": 1e-14}


def compare_fields(got, ref, tol, what=""):
    worst = {}
    for k in ref:
        scale = np.max(np.abs(ref[k]))
        d = np.max(np.abs(got[k] - ref[k]))
        lim = tol * scale + ABS_FLOOR.get(k, 0.0) * (tol / 1e-11)
        worst[k] = (d, scale)
        assert np.all(np.isfinite(got[k])), "%s: non-finite values in %s" % (what, k)
        assert d <= lim, "%s: field %s max|diff| %.3e > %.3e (scale %.3e)" % (what, k, d, lim, scale)
    return worst

"""GPU against the COMMITTED vectors of tests/golden/oracle_vectors.json (no oracle at run time): device WENO-5 (strict path
bitwise, production arithmetic to 1e-13), hydrostatic columns (bitwise: host code), Kessler incl. rainsplit > 1 (1e-12), MLP (1e-5)."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
VEC = json.load(open(os.path.join(HERE, "golden", "oracle_vectors.json")))


def test_device_weno5_on_the_golden_stencils(mw):
    from miniweatherml_amd import capi
    st = torch.tensor([v["stencil"] for v in VEC["weno5"]], dtype=torch.float64, device="cuda")
    want = np.array([v["gll"] for v in VEC["weno5"]])
    out = torch.empty((st.shape[0], 2), dtype=torch.float64, device="cuda")
    for strict in (1, 0):
        capi.check(capi.lib().mw_weno5_edges(st.shape[0], C.c_void_p(st.data_ptr()), C.c_void_p(out.data_ptr()), strict, None))
        got = out.cpu().numpy()
        if strict:
            assert np.array_equal(got, want)                           # same operation order, contraction off: bit for bit
        else:
            scale = np.maximum(np.abs(np.array([v["stencil"] for v in VEC["weno5"]])).max(axis=1, keepdims=True), 1e-300)
            assert np.max(np.abs(got - want) / scale) <= 1e-13         # re-associated production arithmetic


def test_hydrostatic_columns_match_the_golden_vectors(mw):
    from miniweatherml_amd import modules
    for nz, cols in VEC["hydrostatic"].items():
        coupler, dycore, _ = modules.make_supercell(8, 1, int(nz), 1, 8000., 1.0e5, 20000., perturb=False)
        for k, a in cols.items():
            assert getattr(dycore, k)[:, 0].tolist() == a, (nz, k)


@pytest.mark.parametrize("name", sorted(VEC["kessler"]))
def test_kessler_on_the_golden_states(mw, name):
    from miniweatherml_amd import capi
    v = VEC["kessler"][name]
    nz, ncol = v["nz"], v["ncol"]
    t = {k: torch.tensor(x, dtype=torch.float64, device="cuda").reshape(nz, ncol).contiguous() for k, x in v["before"].items()}
    precl = torch.zeros(ncol, dtype=torch.float64, device="cuda")
    L = capi.lib()
    ws = torch.empty(L.mw_kessler_workspace_bytes(nz, ncol) // 8 + 1, dtype=torch.float64, device="cuda")
    rs = C.c_int(0)
    capi.check(L.mw_kessler_time_step(nz, ncol, v["dz"], v["dt"], *[C.c_void_p(t[k].data_ptr()) for k in ("rho_v", "rho_c", "rho_r", "rho_d", "temp")],
                                      C.c_void_p(precl.data_ptr()), C.c_void_p(ws.data_ptr()), C.byref(rs), None))
    assert rs.value == v["rainsplit"]
    for k in ("temp", "rho_v", "rho_c", "rho_r"):
        want = np.array(v["after"][k])
        got = t[k].cpu().numpy().ravel()
        assert np.max(np.abs(got - want)) <= 1e-12 * max(np.max(np.abs(want)), 1e-30) + 1e-18, (name, k)
    want = np.array(v["after"]["precl"])
    assert np.max(np.abs(precl.cpu().numpy() - want)) <= 1e-12 * max(np.max(np.abs(want)), 1e-30) + 1e-20


def test_mlp_on_the_golden_rows(mw):
    from miniweatherml_amd import modules
    W1, b1, W2, b2, si, so = modules.load_surrogate_weights()
    ins = [torch.tensor(a, dtype=torch.float64, device="cuda") for a in VEC["mlp"]["inputs"]]
    outs = modules.mlp_forward(*ins, W1, b1, W2, b2, si, so)
    for i, (got, want) in enumerate(zip(outs, VEC["mlp"]["outputs"])):
        rng = so[i, 1] - so[i, 0]
        assert np.max(np.abs(got.cpu().numpy() - np.array(want))) <= 1e-5 * abs(rng), i


@pytest.mark.parametrize("name", sorted(VEC["tendencies"]))
def test_stage_tendencies_and_fluxes_against_the_golden_statistics(mw, name):
    """mw_dycore_compute_tendencies + the six public flux arrays on the strict path: sum, sum of squares, min and max of every
    variable's tendency and flux array against the committed oracle statistics (no oracle at run time)."""
    import math
    from miniweatherml_amd import modules
    v = VEC["tendencies"][name]
    nx, ny, nz = v["grid"]
    coupler, dycore, _ = modules.make_supercell(nx, ny, nz, 1, 500.0 * nx, 500.0 * max(ny, 2) if ny > 1 else 1.0e5, 20000.)
    dycore.set_strict(1)
    dt = dycore.compute_time_step(coupler)
    st, tt = dycore.compute_tendencies(coupler, dt)
    fl = {k: a.cpu().numpy() for k, a in dycore.fluxes(coupler).items()}

    def check(arr, want, what, scale):
        a = np.asarray(arr, dtype=np.float64).ravel()
        got = {"min": float(a.min()), "max": float(a.max()), "sum": math.fsum(a.tolist()), "sumsq": math.fsum((a * a).tolist())}
        n = a.size
        assert abs(got["min"] - want["min"]) <= 1e-11 * scale and abs(got["max"] - want["max"]) <= 1e-11 * scale, what
        assert abs(got["sum"] - want["sum"]) <= 1e-11 * scale * n and abs(got["sumsq"] - want["sumsq"]) <= 1e-10 * scale * scale * n, what
    for grp, arrs, wants in (("state_tend", st.cpu().numpy(), v["state_tend"]), ("tracers_tend", tt.cpu().numpy(), v["tracers_tend"])):
        for i, w in enumerate(wants):
            check(arrs[i], w, (grp, i), max(abs(w["min"]), abs(w["max"]), 1e-30))
    for k, wants in v["fluxes"].items():
        for i, w in enumerate(wants):
            scale = max(max(abs(x["min"]), abs(x["max"])) for d in "xyz" for x in [v["fluxes"][k[:-1] + d][i]])
            check(fl[k][i], w, (k, i), max(scale, 1e-30))

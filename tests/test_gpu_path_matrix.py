"""The dispatcher's path space as a test (round 5).  mw_dycore_time_step picks among {general-strict, general-fast, marching} x {folded
configuration K 0 / 1 / 2} x {WENO 3 / 5 (7 / 9 general)} x {nens 1, members fused, member-major, members-in-one-workgroup} x {one stream,
two streams, pipelined} x {k_y_all, split y launches} x {D1 inside the first y launch, pipelined partial conversion, conversion pass} x
{fused / unfused tracer stage} x {2-D, 3-D} x {no transport, halo exchange installed}.  tests/util.py states its rules in Python
(reachable_paths); every combination they allow is REALISED here -- configuration + options that make the library choose exactly that path
(asserted through mw_dycore_path) -- and its result compared with the CPU oracle.  Combinations with a transport run the real RCCL
self-loop transport on the block of a periodic tiling (mw_dycore_use_rccl_self), whose result equals the one-rank domain's.
A kernel launched for a handle it was not built for -- round 3's K = 2 kernel on a K = 0 handle -- fails here mechanically."""
import numpy as np
import pytest
import torch

from util import compare_fields, gpu_fields, path_string, push_fields, reachable_paths, set_options

pytestmark = pytest.mark.gpu

PATHS = reachable_paths()


def _smooth(of, three_d):
    """A smooth, block-periodic disturbance so that all three directions (and every tracer) carry signal."""
    nz, ny, nx, nens = of.uvel.shape
    i = np.arange(nx).reshape(1, 1, nx, 1) * (2 * np.pi / nx)
    j = np.arange(ny).reshape(1, ny, 1, 1) * (2 * np.pi / max(ny, 1))
    k = np.arange(nz).reshape(nz, 1, 1, 1) * (np.pi / nz)
    e = np.arange(nens).reshape(1, 1, 1, nens)
    of.uvel += 3.0 * np.sin(i) * np.cos(j) * np.sin(k) + 0.3 * e
    if three_d:
        of.vvel += 2.0 * np.cos(i) * np.sin(j) * np.sin(k)
    of.wvel += 0.5 * np.sin(i + j) * np.sin(k)
    of.temp += 0.3 * np.cos(i - j) + 0.05 * e


@pytest.mark.parametrize("c", PATHS, ids=[path_string(c).replace(" ", "-") for c in PATHS])
def test_path(mw, oracle, monkeypatch, c):
    from miniweatherml_amd import modules
    march = c["family"] == "march"
    order = c["ord"]
    O = oracle if order == 5 else oracle.with_order(order)
    three_d = (not march) or c["dim"] == "3d"
    K = c.get("K", 0)
    lay = c["layout"]
    nens = {"nens1": 1, "fused_members": 2, "member_major": 3, "mm_direct": 4 if order == 5 else 2}[lay]
    tp = c["transport"]
    opts = {}
    if march:
        if lay == "fused_members":
            opts["member_major"] = 0
        if K == 0 and three_d:
            opts["spec"] = 0
        if c["tracers"] == "tracers_unfused":
            opts["fused_tracers"] = 0
        if not tp:
            if c["sched"] == "two_stream":
                opts["overlap"] = 1
        else:
            if c["sched"] == "one_stream":
                opts["overlap"] = 0
            elif c["sched"] == "two_stream":
                opts["pipe"] = 0
        would_y_all = c["sched"] != "two_stream" and c["tracers"] == "tracers_fused" and three_d
        if c["y"] == "y_split" and would_y_all:
            opts["y_all"] = 0
        if c["conv"] == "conv_pass":
            opts["fused_convert"] = 0
            opts["pipe_convert"] = 0
    set_options(monkeypatch, **opts)
    # ---- the block (one-rank domain of the oracle) and, with a transport, rank 0 of its periodic 2 x 2 (2-D: 2 x 1) tiling
    nx, ny, nz = (64 if order <= 5 else 24), (20 if three_d else 1), (8 if three_d else 10)
    city = K == 2
    if city:
        xlen, ylen, zlen = 5.0 * nx, 5.0 * ny, 5.0 * nz
    else:
        xlen, ylen, zlen = 500.0 * nx, 500.0 * max(ny, 2), 20000.0
    tx, ty = (2, 2) if three_d else (2, 1)
    nranks = tx * ty if tp else 1
    gx, gy = (nx * tx, ny * ty) if tp else (nx, ny)
    gxl, gyl = (xlen * tx, ylen * (ty if three_d else 1)) if tp else (xlen, ylen)
    if city:
        coupler, dycore, _, _ = modules.make_simple_city(gx, gy, nz, nens, gxl, gyl, zlen, "building", nranks=nranks, myrank=0, ord=order)
        odyc, of = O.supercell_setup(nx, ny, nz, nens, xlen, ylen, zlen, init_data="building", num_tracers=1, enable_gravity=False, perturb=False)
    else:
        coupler, dycore, _ = modules.make_supercell(gx, gy, nz, nens, gxl, gyl, zlen, nranks=nranks, myrank=0, ord=order)
        odyc, of = O.supercell_setup(nx, ny, nz, nens, xlen, ylen, zlen)
        of.tracers[1][...] = 2.0e-4 * of.rho_d
        of.tracers[2][...] = 5.0e-5 * of.rho_d
    assert (coupler.get_nx(), coupler.get_ny()) == (nx, ny)
    _smooth(of, three_d)
    push_fields(coupler, of)
    if city:
        dycore.immersed_proportion(coupler).copy_(torch.from_numpy(odyc.immersed_proportion()))
    if not march:
        dycore.set_strict(1 if c["family"] == "general-strict" else 2)
    if tp:
        modules.use_rccl_self_exchange(dycore, coupler)
    dt = dycore.compute_time_step(coupler)
    assert dt == odyc.compute_time_step()
    for n in range(2):
        f = 2.3 if (n == 1 and not march and order == 5) else 1.0      # (general kernels: one sub-cycled step -- k_update<3, 0>)
        dycore.time_step(coupler, dt * f)
        odyc.time_step(of, dt * f)
    assert dycore.path() == path_string(c), (dycore.path(), path_string(c), opts)
    what = "path matrix [%s]%s, 2 steps" % (path_string(c), " strict arithmetic: mode 1" if c["family"] == "general-strict" else "")
    compare_fields(gpu_fields(coupler), of.as_dict(), 0.0 if c["family"] == "general-strict" else 1e-10, what)


# ---------------------------------------------------------------------------------------------------------------------
# Instantiation sweep: the kernel templates carry the tracer count (1-4) and stage / mode pairs that the paths above reach with the
# shipped tracer sets only.  Every tracer count x layout x WENO order x tracer-stage form on the K = 0 kernels, with mixed positivity /
# mass flags, one sub-cycled step (the last stage of a cycle that is NOT the last: <3, 0> forms), then the six public flux arrays
# (member-major: k_member_to_fused) and one compute_tendencies call (k_update<1, 2>) -- each compared with the oracle.
# tests/conftest.py lists, at session end, every compiled instantiation that no passed comparison exercised.
# ---------------------------------------------------------------------------------------------------------------------
INST = [(o, nt, lay, tr) for o in (5, 3) for nt in (1, 2, 3, 4) for lay in ("nens1", "fused_members", "member_major", "mm_direct")
        for tr in ("tracers_fused", "tracers_unfused")
        if not (lay == "fused_members" and o != 5) and not (tr == "tracers_unfused" and (o == 3 or lay in ("member_major", "mm_direct")))]


@pytest.mark.parametrize("order,nt,lay,tr", INST)
def test_tracer_count_and_layout_instantiations(mw, oracle, monkeypatch, order, nt, lay, tr):
    from miniweatherml_amd import modules
    from test_gpu_dycore_parity import check_fluxes
    O = oracle if order == 5 else oracle.with_order(order)
    nens = {"nens1": 1, "fused_members": 2, "member_major": 3, "mm_direct": 2}[lay]
    opts = {"chunk_z": 5, "chunk_f": 5, "chunk_y": 9}
    if lay == "fused_members":
        opts["member_major"] = 0
    if tr == "tracers_unfused":
        opts["fused_tracers"] = 0
    set_options(monkeypatch, **opts)
    pos, adds = [1, 0, 1, 1][:nt], [1, 1, 0, 1][:nt]
    nx, ny, nz = 62, 22, 11
    xlen, ylen = 500.0 * nx, 500.0 * ny

    class Micro(modules.Microphysics_Kessler):
        def init(self, coupler):
            coupler.add_tracer("water_vapor", "Water Vapor", True, True)
            for t in range(1, nt):
                coupler.add_tracer("tr%d" % t, "", bool(pos[t]), bool(adds[t]))
    coupler, dycore, _ = modules.make_supercell(nx, ny, nz, nens, xlen, ylen, 20000., micro=Micro(), perturb=False, ord=order)
    p, _ = O.make_params(nx, ny, nz, nens, xlen, ylen, 20000., num_tracers=nt)
    odyc = O.OracleDycore(p, tracer_positive=pos, tracer_adds_mass=adds)
    of = O.Fields(odyc.p)
    odyc.init("supercell", of)
    for t in range(1, nt):
        of.tracers[t][...] = (1.0e-4 * t) * of.rho_d
    _smooth(of, True)
    push_fields(coupler, of)
    dt = dycore.compute_time_step(coupler)
    for n in range(3):
        f = 2.3 if n == 1 else 1.0                             # step 1: three sub-cycles
        dycore.time_step(coupler, dt * f)
        odyc.time_step(of, dt * f)
    assert "march ord%d K0 %s" % (order, lay) in dycore.path() and tr in dycore.path(), dycore.path()
    what = "instantiation sweep ord %d, %d tracers, %s, %s" % (order, nt, lay, tr)
    compare_fields(gpu_fields(coupler), of.as_dict(), 1e-10, what + ", 3 steps (one sub-cycled)")
    check_fluxes(dycore, coupler, odyc, 1e-10)
    st, tt = dycore.compute_tendencies(coupler, dt)
    ost, ott = odyc.stage_tendencies(of, dt)
    compare_fields({"state_tend": st.cpu().numpy(), "tracers_tend": tt.cpu().numpy()}, {"state_tend": ost, "tracers_tend": ott}, 1e-9,
                   what + ", compute_tendencies")

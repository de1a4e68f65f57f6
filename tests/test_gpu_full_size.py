"""Size-independent properties at BASELINE.json's full single-GPU size (configs[1]: supercell 400x400x100, fp64),
where the CPU oracle would need minutes per step: total mass conserved to round-off (periodic x/y + wall z),
positivity of the FCT-limited tracers, finite fields, ensemble members identical, 2-D runs keep v == 0, and a slab of
the big run matches an oracle run of a *smaller* periodic problem that contains the same physics (x-y translation
invariance of the initial column away from the bubble is NOT assumed; we compare whole small domains instead)."""
import numpy as np
import pytest
import torch

from util import set_options

pytestmark = pytest.mark.gpu


def total_mass(coupler):
    dm = coupler.get_data_manager_readonly()
    m = dm.get("density_dry", True).sum(dtype=torch.float64)
    for n, t in zip(coupler.get_tracer_names(), coupler.tracers):
        if t["adds_mass"]:
            m = m + dm.get(n, True).sum(dtype=torch.float64)
    return float(m)


def test_config2_mass_positivity_finite(mw):
    from miniweatherml_amd import modules
    coupler, dycore, _ = modules.make_supercell(400, 400, 100, 1, 200000., 200000., 20000.)
    dt = dycore.compute_time_step(coupler)
    assert abs(dt - 0.6 * 200.0 / 430.0) < 1e-15
    m0 = total_mass(coupler)
    for _ in range(5):
        dycore.time_step(coupler, dt)
    m1 = total_mass(coupler)
    assert abs(m1 - m0) <= 1e-11 * m0
    dm = coupler.get_data_manager_readonly()
    for n in ("density_dry", "uvel", "vvel", "wvel", "temp") + tuple(coupler.get_tracer_names()):
        assert bool(torch.isfinite(dm.get(n, True)).all()), n
    for n in coupler.get_tracer_names():
        assert float(dm.get(n, True).min()) >= 0.0, n
    w = dm.get("wvel", True)
    assert 0.1 < float(w.abs().max()) < 20.0                      # the bubble is rising, nothing blew up
    # mirror symmetry in y of the symmetric set-up (approximate: the upwind tie-break is not mirror-symmetric)
    assert float((w - w.flip(1)).abs().max()) <= 1e-6 * float(w.abs().max())


def test_nens4_members_identical(mw):
    from miniweatherml_amd import modules
    coupler, dycore, _ = modules.make_supercell(128, 96, 64, 4, 64000., 48000., 20000.)
    dt = dycore.compute_time_step(coupler)
    for _ in range(3):
        dycore.time_step(coupler, dt)
    dm = coupler.get_data_manager_readonly()
    for n in ("density_dry", "uvel", "wvel", "temp", "water_vapor"):
        a = dm.get(n, True)
        for e in range(1, 4):
            assert torch.equal(a[..., 0], a[..., e]), (n, e)


def test_two_d_keeps_v_zero(mw):
    from miniweatherml_amd import modules
    coupler, dycore, _ = modules.make_supercell(1000, 1, 200, 1, 100000., 100000., 20000.)
    dt = dycore.compute_time_step(coupler)
    m0 = total_mass(coupler)
    for _ in range(5):
        dycore.time_step(coupler, dt)
    assert float(coupler.get_data_manager_readonly().get("vvel", True).abs().max()) == 0.0
    assert abs(total_mass(coupler) - m0) <= 1e-11 * m0
    fl = dycore.fluxes(coupler)
    assert float(fl["state_flux_y"].abs().max()) == 0.0 and float(fl["tracers_flux_y"].abs().max()) == 0.0


@pytest.mark.parametrize("order", [5, 3])
def test_production_path_equals_general_path_at_size(mw, order):
    """The marching production kernels and the general flux-materialising kernels (same fast arithmetic) are two
    independent implementations of one stage; they must agree to rounding on a mid-size grid after 5 steps.  Order 3 = the
    reference's GPU-benchmark build (-DMW_ORD=3): 3-level windows, 60 cells per wave."""
    from miniweatherml_amd import modules
    out = []
    for mode in (0, 2):
        coupler, dycore, _ = modules.make_supercell(160, 120, 60, 1, 80000., 60000., 20000., ord=order)
        dycore.set_strict(mode)
        dt = dycore.compute_time_step(coupler)
        for _ in range(5):
            dycore.time_step(coupler, dt)
        dm = coupler.get_data_manager_readonly()
        out.append({n: dm.get(n, True).clone() for n in ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor")})
    for n in out[0]:
        scale = float(out[1][n].abs().max())
        floor = 1e-11 if n in ("uvel", "vvel", "wvel") else 0.0
        assert float((out[0][n] - out[1][n]).abs().max()) <= 1e-10 * scale + floor, n


@pytest.mark.parametrize("case", ["supercell", "supercell_ord3", "supercell_nens2", "city"])
def test_folded_configurations_are_bitwise_the_run_time_switches(mw, monkeypatch, case):
    """Cf<K>: the marching kernels with the shipped configurations' switches folded at compile time (K = 1 supercell, K = 2
    simple_city) execute the same arithmetic as with every switch at run time (K = 0, option spec = 0) -- the folded terms are exact
    no-ops (fcor = 0, bcmode = 0, mask bits set) -- so the results must be bit-identical, 5 steps, orders 5 and 3, member-major too."""
    from miniweatherml_amd import modules
    out = []
    for nospec in (False, True):
        set_options(monkeypatch, spec=0 if nospec else 1)
        if case == "city":
            coupler, dycore, hs, ta = modules.make_simple_city(96, 80, 24, 1, 480., 400., 120., "city")
        else:
            coupler, dycore, _ = modules.make_supercell(120, 88, 40, 2 if case.endswith("nens2") else 1, 60000., 44000., 20000.,
                                                        ord=3 if case.endswith("ord3") else 5)
        dt = dycore.compute_time_step(coupler)
        for _ in range(5):
            dycore.time_step(coupler, dt)
        dm = coupler.get_data_manager_readonly()
        out.append({n: dm.get(n, True).clone() for n in ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor")})
    for n in out[0]:
        if case == "supercell_ord3":
            # WENO-3: the two instantiations of the contracted arithmetic may fuse a different multiply-add pair (the compiler's choice
            # depends on the surrounding code -- k_tracers_fused's loop body exists in two forms since the zero-row maps of round 5), so
            # the tracers' edge values and, through D13 and the next step's conversion, every field agree to rounding only (measured
            # 1e-14 of the field's scale after 5 steps; v itself is a cancellation residue of the y-symmetric set-up, 1e-13 m/s).
            # Every field at WENO-5 is bit-identical.
            scale = float(out[0]["uvel" if n in ("uvel", "vvel", "wvel") else n].abs().max())
            d = float((out[0][n] - out[1][n]).abs().max())
            assert d <= 1e-12 * scale, (n, d)
        else:
            assert torch.equal(out[0][n], out[1][n]), n


def test_two_stream_schedule_is_bitwise_the_one_stream_schedule(mw, monkeypatch):
    """Option overlap = 1 runs the state and tracer pipelines on two streams (the default with a neighbour exchange), overlap = 0 on
    one (the default on one rank): the same kernels on the same data, so any difference would be a missing stream dependency."""
    import torch
    from miniweatherml_amd import modules
    out = []
    for ov in ("0", "1"):
        set_options(monkeypatch, overlap=ov)
        coupler, dycore, _ = modules.make_supercell(160, 120, 60, 1, 80000., 60000., 20000.)
        modules.perturb_temperature(coupler)
        dt = dycore.compute_time_step(coupler)
        for _ in range(6):
            dycore.time_step(coupler, dt)
        dm = coupler.get_data_manager_readonly()
        out.append({n: dm.get(n, True).clone() for n in ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor")})
    for n in out[0]:
        assert torch.equal(out[0][n], out[1][n]), n


def test_index_wrap_is_bitwise_the_halo_fill(mw, monkeypatch):
    """Periodic x/y on one rank: the marching kernels wrap their indices and no halo is filled (default); option wrap = 0 fills the
    x/y halos and reads them.  Same values reach the same arithmetic, so the fields must be identical -- also with 3 members."""
    import torch
    from miniweatherml_amd import modules
    for nens in (1, 3):
        out = []
        for nowrap in (None, "1"):
            set_options(monkeypatch, wrap=0 if nowrap else 1)
            coupler, dycore, _ = modules.make_supercell(100, 61, 24, nens, 50000., 30500., 20000.)
            modules.perturb_temperature(coupler)
            dt = dycore.compute_time_step(coupler)
            for _ in range(4):
                dycore.time_step(coupler, dt)
            fl = dycore.fluxes(coupler)                          # on-demand rebuild fills the skipped halos first
            dm = coupler.get_data_manager_readonly()
            out.append({n: dm.get(n, True).clone() for n in ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor")})
            out[-1].update({k: v.clone() for k, v in fl.items()})
        for n in out[0]:
            assert torch.equal(out[0][n], out[1][n]), (nens, n)


def test_conversion_inside_y_state_is_bitwise_the_conversion_pass(mw, monkeypatch):
    """One rank, periodic x/y, one stream: the first k_y_state converts the coupler rows it loads and fills the slab (default);
    option fused_convert = 0 runs k_coupler_to_state_fast first.  Same device function on the same inputs: identical fields, also
    with sub-cycling (only the first cycle converts) and with a developed, perturbed state."""
    import torch
    from miniweatherml_amd import modules
    out = []
    for nofuse in (None, "1"):
        set_options(monkeypatch, fused_convert=0 if nofuse else 1)
        coupler, dycore, _ = modules.make_supercell(90, 70, 30, 1, 45000., 35000., 20000.)
        modules.perturb_temperature(coupler)
        dt = dycore.compute_time_step(coupler)
        for n in range(4):
            dycore.time_step(coupler, dt * (2.5 if n == 2 else 1.0))          # step 2: three sub-cycles
        dm = coupler.get_data_manager_readonly()
        out.append({n: dm.get(n, True).clone() for n in ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "cloud_liquid")})
    for n in out[0]:
        assert torch.equal(out[0][n], out[1][n]), n


def test_config4_block_per_gpu(mw):
    """BASELINE.json configs[3]: the per-GPU block of the 8-GPU supercell run, 256 x 512 x 128 with 4 ensemble members
    (6.7e7 cells, 4.3 GB per slab: byte offsets far beyond 2^32).  Members identical, mass conserved, tracers positive."""
    from miniweatherml_amd import modules
    coupler, dycore, _ = modules.make_supercell(256, 512, 128, 4, 256 * 800., 512 * 800., 20000.)
    dt = dycore.compute_time_step(coupler)
    m0 = total_mass(coupler)
    for _ in range(2):
        dycore.time_step(coupler, dt)
    assert abs(total_mass(coupler) - m0) <= 1e-11 * m0
    dm = coupler.get_data_manager_readonly()
    for n in ("density_dry", "uvel", "wvel", "temp") + tuple(coupler.get_tracer_names()):
        f = dm.get(n, True)
        assert bool(torch.isfinite(f).all()), n
        assert bool((f[..., 0:1] == f).all()), n                  # all four members bit-identical
    assert float(dm.get("water_vapor", True).min()) >= 0.0
    del coupler, dycore
    torch.cuda.empty_cache()


def test_config5_city_block(mw):
    """BASELINE.json configs[4]: simple_city immersed-boundary flow 512 x 512 x 256 (one tracer, gravity off), 5 m spacing."""
    from miniweatherml_amd import modules
    coupler, dycore, hs, ta = modules.make_simple_city(512, 512, 256, 1, 2560., 2560., 1280., "city")
    imm = dycore.immersed_proportion(coupler)
    frac = float(imm.mean())
    assert 0.001 < frac < 0.2 and float(imm.max()) == 1.0          # buildings are there (45 x 45 blocks, :1432-1447)
    for _ in range(2):
        modules.simple_city_step(coupler, dycore, hs, ta)
    dm = coupler.get_data_manager_readonly()
    for n in ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "time_avg_uvel"):
        assert bool(torch.isfinite(dm.get(n, True)).all()), n
    u = dm.get("uvel", True)
    assert 15.0 < float(u.max()) < 40.0 and float(u[imm > 0.5].abs().max()) < float(u.max())   # the flow is being braked inside the buildings
    del coupler, dycore, hs, ta
    torch.cuda.empty_cache()


def test_member_major_layout_agrees_with_the_member_fastest_kernels(mw, monkeypatch):
    """nens > 1: the default keeps the handle's arrays member after member and runs the nens = 1 kernels per member (DPP shifts);
    Option member_major = 0 keeps the coupler's member-fastest order inside and fetches the x neighbours by loads.  The stencil
    arithmetic is the same; the conversions at the coupler boundary are not (the member-major handle stores q / rho in its slab and
    multiplies back in k_member_to_coupler, the other writes rho q straight from the last stage): agreement to a few ulp, with
    members that differ from each other and chunks that do not divide nz."""
    import torch
    from miniweatherml_amd import modules
    set_options(monkeypatch, chunk_z=7, chunk_f=9)
    out = []
    for legacy in (None, "1"):
        set_options(monkeypatch, member_major=0 if legacy else 1)
        coupler, dycore, _ = modules.make_supercell(70, 45, 26, 3, 35000., 22500., 20000.)
        modules.perturb_temperature(coupler)
        dm = coupler.get_data_manager_readwrite()
        t = dm.get("temp", True)
        t += 0.05 * torch.arange(3, device=t.device, dtype=t.dtype)          # members differ
        dt = dycore.compute_time_step(coupler)
        for n in range(3):
            dycore.time_step(coupler, dt * (2.2 if n == 1 else 1.0))            # step 1: three sub-cycles
        fl = dycore.fluxes(coupler)
        dmr = coupler.get_data_manager_readonly()
        out.append({n: dmr.get(n, True).clone() for n in ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "cloud_liquid")})
        out[-1].update({k: v.clone() for k, v in fl.items()})
    for n in out[0]:
        scale = float(out[1][n].abs().max())
        floor = 1e-11 if n in ("uvel", "vvel", "wvel") else 1e-18          # (v starts at 0 and grows from rounding-level seeds)
        assert float((out[0][n] - out[1][n]).abs().max()) <= 1e-12 * scale + floor, n
    a = out[0]["temp"]
    assert not torch.equal(a[..., 0], a[..., 1])


@pytest.mark.parametrize("nens,nx,ny,nz,order", [(4, 70, 45, 26, 5), (2, 131, 9, 26, 5), (4, 24, 6, 12, 5), (2, 58, 4, 14, 5), (4, 70, 11, 26, 3), (2, 125, 7, 13, 3)])
def test_members_in_one_workgroup_agree_with_the_conversion_passes_and_the_oracle(mw, oracle, monkeypatch, nens, nx, ny, nz, order):
    """Member-major handles with 2 or 4 members read D1 in the first k_y_state and write D13 from the last stage's kernels, with the
    members of the same cells in ONE workgroup (MemberOff / k_y_state<.., MM = 2>, mw_march.h) so that their accesses to the coupler's
    member-fastest arrays meet in L1 / L2.  Option mm_direct = 0 keeps the k_member_to_coupler pass (q / rho in the slab, multiplied
    back: a few ulp apart) and the fused-lane D1 launch.  Members that differ, several tiles per row with a ragged last one, an odd row
    count (nens = 2: a workgroup's second row does not exist), chunks that do not divide nz, a sub-cycled step.  Then the same path
    against the CPU oracle (tolerance of BASELINE.md section 4); WENO-5 and WENO-3."""
    import torch
    from miniweatherml_amd import modules
    from util import compare_fields, gpu_fields, push_fields
    set_options(monkeypatch, chunk_z=7, chunk_f=9)
    out = []
    for pass13 in (None, "1"):
        set_options(monkeypatch, mm_direct=0 if pass13 else 1)
        coupler, dycore, _ = modules.make_supercell(nx, ny, nz, nens, 500. * nx, 500. * ny, 20000., ord=order)
        modules.perturb_temperature(coupler)
        dm = coupler.get_data_manager_readwrite()
        t = dm.get("temp", True)
        t += 0.05 * torch.arange(nens, device=t.device, dtype=t.dtype)          # members differ
        dm.get("cloud_liquid", True).fill_(2.0e-4)                            # (all D13 outputs non-trivial)
        dt = dycore.compute_time_step(coupler)
        for n in range(3):
            dycore.time_step(coupler, dt * (2.2 if n == 1 else 1.0))            # step 1: three sub-cycles
        dmr = coupler.get_data_manager_readonly()
        out.append({n: dmr.get(n, True).clone() for n in ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "cloud_liquid", "precip_liquid")})
    for n in out[0]:
        scale = float(out[1][n].abs().max())
        floor = 1e-11 if n in ("uvel", "vvel", "wvel") else 1e-18
        assert float((out[0][n] - out[1][n]).abs().max()) <= 1e-12 * scale + floor, n
    a = out[0]["temp"]
    assert not torch.equal(a[..., 0], a[..., 1])
    # --- against the oracle, production arithmetic
    set_options(monkeypatch, mm_direct=1)
    coupler, dycore, _ = modules.make_supercell(nx, ny, nz, nens, 500. * nx, 500. * ny, 20000., ord=order)
    odyc, of = (oracle if order == 5 else oracle.with_order(order)).supercell_setup(nx, ny, nz, nens, 500. * nx, 500. * ny, 20000.)
    of.temp += 0.05 * np.arange(nens)
    push_fields(coupler, of)
    dt = dycore.compute_time_step(coupler)
    for _ in range(2):
        dycore.time_step(coupler, dt)
        odyc.time_step(of, dt)
    compare_fields(gpu_fields(coupler), of.as_dict(), 1e-11, "members in one workgroup (D1 / D13), nens %d %dx%dx%d WENO-%d, 2 steps" % (nens, nx, ny, nz, order))


def test_members_in_one_workgroup_with_a_chunk_longer_than_the_lds_tables_allow(mw, oracle, monkeypatch):
    """The D13 launch of k_xz_state with the members of a tile in one workgroup keeps one background table per wave in LDS:
    (chunk + 2) x 256 B on top of ~20.5 KB of static LDS.  A chunk of 200 levels (option chunk_z, or the chunk rule on a grid with many
    wavefronts and a tall column) would exceed the 64 KB a workgroup may have and the launch would fail (round 3's advisor finding):
    that launch now cuts its own chunks at 170 levels.  nens = 4, nz = 200, against the oracle."""
    from miniweatherml_amd import modules
    from util import compare_fields, gpu_fields, push_fields
    set_options(monkeypatch, chunk_z=200, chunk_f=200)
    nx, ny, nz, nens = 24, 8, 200, 4
    coupler, dycore, _ = modules.make_supercell(nx, ny, nz, nens, 500. * nx, 500. * ny, 20000.)
    odyc, of = oracle.supercell_setup(nx, ny, nz, nens, 500. * nx, 500. * ny, 20000.)
    of.temp += 0.05 * np.arange(nens)
    push_fields(coupler, of)
    dt = dycore.compute_time_step(coupler)
    for _ in range(2):
        dycore.time_step(coupler, dt)
        odyc.time_step(of, dt)
    compare_fields(gpu_fields(coupler), of.as_dict(), 1e-11, "members in one workgroup, one 200-level chunk, 2 steps")


@pytest.mark.parametrize("nx,ny,nz,order", [(61, 23, 17, 5), (130, 9, 26, 5), (64, 37, 12, 3)])
def test_y_faces_of_all_variables_in_one_launch_equal_the_two_launches(mw, oracle, monkeypatch, nx, ny, nz, order):
    """k_y_all (state variables and tracers in one y march; the converting first stage included) against k_y_state + k_y_tracers
    (option y_all = 0): the same arithmetic, so the same bits -- odd sizes, y chunks that do not divide ny, a sub-cycled step, cloud
    and rain present.  Then against the CPU oracle (tolerance of BASELINE.md section 4)."""
    from miniweatherml_amd import modules
    from util import compare_fields, gpu_fields, push_fields
    set_options(monkeypatch, chunk_y=7)
    out = []
    for two in (None, "1"):
        set_options(monkeypatch, y_all=0 if two else 1)
        coupler, dycore, _ = modules.make_supercell(nx, ny, nz, 1, 500. * nx, 500. * ny, 20000., ord=order)
        dm = coupler.get_data_manager_readwrite()
        dm.get("cloud_liquid", True).fill_(3.0e-4); dm.get("precip_liquid", True).fill_(1.0e-4)
        dt = dycore.compute_time_step(coupler)
        for n in range(3):
            dycore.time_step(coupler, dt * (2.2 if n == 1 else 1.0))
        out.append(gpu_fields(coupler))
    for n in out[0]:
        assert np.array_equal(out[0][n], out[1][n]), n
    set_options(monkeypatch, y_all=1)
    O = oracle if order == 5 else oracle.with_order(order)
    coupler, dycore, _ = modules.make_supercell(nx, ny, nz, 1, 500. * nx, 500. * ny, 20000., ord=order)
    odyc, of = O.supercell_setup(nx, ny, nz, 1, 500. * nx, 500. * ny, 20000.)
    of.tracers[1][...] = 3.0e-4 * of.rho_d; of.tracers[2][...] = 1.0e-4 * of.rho_d
    push_fields(coupler, of)
    dt = dycore.compute_time_step(coupler)
    for _ in range(2):
        dycore.time_step(coupler, dt)
        odyc.time_step(of, dt)
    compare_fields(gpu_fields(coupler), of.as_dict(), 1e-11, "k_y_all %dx%dx%d WENO-%d, 2 steps" % (nx, ny, nz, order))


@pytest.mark.parametrize("schedule", ["pipelined", "two_stream"])
def test_config4_8_rank_block_over_the_rccl_self_loop(mw, monkeypatch, schedule):
    """BASELINE.json configs[3] in the form the 8-GPU job runs it, at FULL size on one GPU: rank 0's 256 x 512 x 128 x 4 block of the
    4 x 2 decomposition with the built-in RCCL transport (33.6 / 16.8 MB strips per direction and stage through ncclSend / ncclRecv groups
    on the side stream, mw_dycore_use_rccl_self: every peer is this rank, i.e. the 1024 x 1024 domain is the periodic tiling of the block)
    against the one-rank run of the block: bit for bit, members differing, after two steps."""
    from miniweatherml_amd import modules
    if schedule == "two_stream":
        set_options(monkeypatch, pipe=0)
    nx, ny, nz, nens = 256, 512, 128, 4
    ref = modules.make_supercell(nx, ny, nz, nens, 800.0 * nx, 800.0 * ny, 20000.)
    til = modules.make_supercell(4 * nx, 2 * ny, nz, nens, 3200.0 * nx, 1600.0 * ny, 20000., nranks=8, myrank=0)
    (rc, rd, _), (tc, td, _) = ref, til
    assert (tc.grid.nproc_x, tc.grid.nproc_y, tc.get_nx(), tc.get_ny()) == (4, 2, nx, ny)
    rc.get_data_manager_readwrite().get("temp").add_(0.05 * torch.arange(nens, device=rc.device, dtype=torch.float64))
    for n in ("density_dry", "uvel", "vvel", "wvel", "temp") + tuple(rc.get_tracer_names()):
        tc.get_data_manager_readwrite().get(n).copy_(rc.get_data_manager_readonly().get(n, True))
    modules.use_rccl_self_exchange(td, tc)
    dt = rd.compute_time_step(rc)
    for _ in range(2):
        rd.time_step(rc, dt)
        td.time_step(tc, dt)
    assert "mm_direct" in td.path() and ("pipe" if schedule == "pipelined" else "two_stream") in td.path() and "transport" in td.path(), td.path()
    for n in ("density_dry", "uvel", "vvel", "wvel", "temp") + tuple(rc.get_tracer_names()):
        assert torch.equal(tc.get_data_manager_readonly().get(n, True), rc.get_data_manager_readonly().get(n, True)), n
    t = rc.get_data_manager_readonly().get("temp", True)
    assert not torch.equal(t[..., 0], t[..., 1])
    del ref, til, rc, rd, tc, td
    torch.cuda.empty_cache()


def test_config5_8_rank_block_over_the_rccl_self_loop(mw):
    """BASELINE.json configs[4]'s weak-scaling block (128 x 256 x 256 of the 4 x 2 decomposition, simple_city: immersed buildings, V = 6) with
    the built-in RCCL transport in self-loop form against the one-rank run of the block: bit for bit after two steps."""
    from miniweatherml_amd import modules
    nx, ny, nz = 128, 256, 256
    rc, rd, _, _ = modules.make_simple_city(nx, ny, nz, 1, 5.0 * nx, 5.0 * ny, 5.0 * nz, "building")
    tc, td, _, _ = modules.make_simple_city(4 * nx, 2 * ny, nz, 1, 20.0 * nx, 10.0 * ny, 5.0 * nz, "building", nranks=8, myrank=0)
    for n in ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor"):
        tc.get_data_manager_readwrite().get(n).copy_(rc.get_data_manager_readonly().get(n, True))
    imm = rd.immersed_proportion(rc)
    assert float(imm.max()) == 1.0
    td.immersed_proportion(tc).copy_(imm)
    modules.use_rccl_self_exchange(td, tc)
    dt = rd.compute_time_step(rc)
    for _ in range(2):
        rd.time_step(rc, dt)
        td.time_step(tc, dt)
    assert "K2" in td.path() and "pipe" in td.path(), td.path()
    for n in ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor"):
        assert torch.equal(tc.get_data_manager_readonly().get(n, True), rc.get_data_manager_readonly().get(n, True)), n
    del rc, rd, tc, td
    torch.cuda.empty_cache()

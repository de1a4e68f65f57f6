"""mw_dycore_set_option / mw_dycore_get_option (include/mw_cdna4.h): the typed run-time options that replaced the MW_* environment switches of
rounds 1-4 -- defaults, round trips, error behaviour, and that the library really stopped reading the environment per call."""
import ctypes as C
import os

import numpy as np
import pytest

from util import gpu_fields

pytestmark = pytest.mark.gpu

DEFAULTS = {"overlap": -1, "pipe": 1, "pipe_edge_inline": 0, "pipe_convert": 1, "pipe_split_edges": 1, "spec": 1, "wrap": 1, "y_all": 1, "y_all_conv": 1,
            "member_major": 1, "mm_direct": 1, "mm_conv": 1, "fused_convert": 1, "fused_convert_mm": 1, "fused_tracers": 1, "chunk_y": 0, "chunk_yt": 0,
            "chunk_z": 0, "chunk_f": 0, "chunk_model": 1, "tf_rows4": 1, "zero_skip": 1, "zero_rows": 1, "zero_stores": 1, "zero_verify": 0, "pipe_maps_early": 1, "rccl_lanes": 0, "rccl_two_comms": -1, "rccl_prio": 1, "rccl_inline": 1, "xchg_fuzz": 0, "debug_no_patch": 0}


def test_defaults_round_trips_and_errors(mw):
    from miniweatherml_amd import capi, modules
    from miniweatherml_amd.capi import MWError
    coupler, dycore, _ = modules.make_supercell(24, 20, 10, 1, 12000., 10000., 20000.)
    for k, v in DEFAULTS.items():
        assert dycore.get_option(k) == v, k
    dycore.set_option("chunk_z", 7)
    assert dycore.get_option("chunk_z") == 7
    dycore.set_option("overlap", 1)
    assert dycore.get_option("overlap") == 1
    for bad in (("no_such_option", 1), ("overlap", 2), ("pipe", -1), ("chunk_z", -5), ("rccl_lanes", 3)):
        with pytest.raises(MWError):
            dycore.set_option(*bad)
    with pytest.raises(MWError):
        dycore.get_option("no_such_option")
    assert capi.lib().mw_build_flags() == 0                    # (the experiment builds of rounds 4-5 left the tree in round 6: one library, no optional parts)
    for gone in ("fused_state", "sched", "sched_mask"):
        with pytest.raises(MWError, match="unknown option"):
            dycore.set_option(gone, 0)
    dycore.set_option("debug_no_patch", 1)                     # (the negative control's switch is an ordinary test option now)
    dycore.set_option("debug_no_patch", 0)


def test_the_environment_no_longer_steers_a_handle(mw, monkeypatch):
    """Rounds 1-4 read ~30 MW_* variables per time step / per launch.  Setting them now changes nothing: same path, same bits."""
    from miniweatherml_amd import modules
    res, paths = [], []
    for env in ({}, {"MW_NO_SPEC": "1", "MW_NO_Y_ALL": "1", "MW_OVERLAP": "1", "MW_NO_WRAP": "1", "MW_CHUNK_Z": "3", "MW_FUSED_TRACERS": "0",
                     "MW_NO_FUSED_CONVERT": "1", "MW_DEBUG_NO_PATCH": "1", "MW_FUSED_STATE": "4", "MW_SCHED": "2"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        coupler, dycore, _ = modules.make_supercell(70, 24, 12, 1, 35000., 12000., 20000.)
        dt = dycore.compute_time_step(coupler)
        for _ in range(2):
            dycore.time_step(coupler, dt)
        res.append(gpu_fields(coupler))
        paths.append(dycore.path())
    assert paths[0] == paths[1] == "march ord5 K1 nens1 one_stream y_all conv_in_y tracers_fused 3d"
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k


def test_options_select_what_they_say(mw, monkeypatch):
    from miniweatherml_amd import modules
    want = {(): "march ord5 K1 nens1 one_stream y_all conv_in_y tracers_fused 3d",
            (("spec", 0),): "march ord5 K0 nens1 one_stream y_all conv_in_y tracers_fused 3d",
            (("y_all", 0),): "march ord5 K1 nens1 one_stream y_split conv_in_y tracers_fused 3d",
            (("overlap", 1),): "march ord5 K1 nens1 two_stream y_split conv_in_y tracers_fused 3d",
            (("fused_convert", 0),): "march ord5 K1 nens1 one_stream y_all conv_pass tracers_fused 3d",
            (("fused_tracers", 0),): "march ord5 K1 nens1 one_stream y_split conv_in_y tracers_unfused 3d",
            (("wrap", 0),): "march ord5 K1 nens1 one_stream y_all conv_pass tracers_fused 3d"}
    ref = None
    for opts, path in want.items():
        coupler, dycore, _ = modules.make_supercell(70, 24, 12, 1, 35000., 12000., 20000.)
        for k, v in opts:
            dycore.set_option(k, v)
        dt = dycore.compute_time_step(coupler)
        for _ in range(2):
            dycore.time_step(coupler, dt)
        assert dycore.path() == path, (opts, dycore.path())
        f = gpu_fields(coupler)
        if ref is None:
            ref = f
        elif opts[0][0] in ("y_all", "overlap", "fused_convert", "wrap"):      # these forms are the same arithmetic on the same data: bitwise
            for k in ref:
                assert np.array_equal(f[k], ref[k]), (opts, k)
        else:                                                  # (K = 0 and the unfused tracer stage: rounding-level differences at most)
            for k in ref:
                assert np.max(np.abs(f[k] - ref[k])) <= 1e-11 * max(np.max(np.abs(ref[k])), 1e-3), (opts, k)


@pytest.mark.parametrize("case", ["cloud_free", "one_storm", "city", "nens4", "ord3"])
def test_zero_tracer_shortcut_is_bit_neutral(mw, case):
    """Option zero_skip (default 1): the marching kernels skip the reconstructions of a tracer that is exactly zero over a wavefront's whole
    stencil (cloud and rain outside a storm, simple_city's vapour) -- the skipped reconstruction would have returned +0.  Same bits with the
    short-cut switched off, on a cloud-free state, on a state whose cloud / rain fill one box (wavefronts of all three kinds: empty, full,
    cut by the box's faces in x, y and z), on the city, with four members and at WENO-3; one step sub-cycled."""
    import torch
    from miniweatherml_amd import modules
    res = []
    for skip in (1, 0):
        if case == "city":
            coupler, dycore, _, _ = modules.make_simple_city(96, 48, 16, 1, 480., 240., 80., "building")
        else:
            nens, order = (4 if case == "nens4" else 1), (3 if case == "ord3" else 5)
            coupler, dycore, _ = modules.make_supercell(130, 44, 26, nens, 65000., 22000., 20000., ord=order)
            dm = coupler.get_data_manager_readwrite()
            if case != "cloud_free":
                box = torch.zeros_like(dm.get("density_dry"))
                box[3:15, 9:31, 37:101] = 1.0
                dm.get("cloud_liquid").copy_(3.0e-4 * box * dm.get("density_dry"))
                dm.get("precip_liquid").copy_(1.0e-4 * box * dm.get("density_dry") * (1.0 + 0.1 * torch.arange(nens, device=box.device)))
        dycore.set_option("zero_skip", skip)
        dycore.set_option("chunk_z", 7); dycore.set_option("chunk_f", 7); dycore.set_option("chunk_y", 9)
        dt = dycore.compute_time_step(coupler)
        for n in range(3):
            dycore.time_step(coupler, dt * (2.2 if n == 1 else 1.0))
        res.append(gpu_fields(coupler))
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k
    if case == "one_storm":
        assert float(res[0]["tracer1"].max()) > 0.0 and float((res[0]["tracer1"] == 0).mean()) > 0.5


@pytest.mark.parametrize("case", ["cloud_free", "one_storm", "specks", "city", "ord3", "ragged", "nens4", "nens3", "nens2_specks"])
def test_zero_row_maps_are_bit_neutral(mw, case):
    """Option zero_rows (default 1, round 5): per (level, row) and tracer a bit "may be non-zero", scanned from the sub-cycle's input and
    dilated by 3 rows / levels per RK stage (mw_march.h: k_zero_rows, k_zero_dilate); k_y_all and k_tracers_fused do not load rows whose bit
    is clear, k_y_all does not store y fluxes nobody loads.  The maps must be a superset of the non-zero rows: the same bits as with the
    maps switched off, and as with the whole zero short-cut switched off -- on a cloud-free state, on one box of cloud and rain, on single
    non-zero cells in the corners, on the faces and next to the periodic seams (a too small dilation shows here), on the city, at WENO-3;
    three steps, one of them sub-cycled (a sub-cycle rebuilds the maps from the slab instead of the coupler's arrays).  "ragged": 129 x 44 x 12
    cells -- the y kernel's last wavefront has 12 lanes (its iteration masks come from a ballot: the missing lanes' iterations must not read
    as "nothing to load / store"), cloud in exactly those cells, one chunk of 44 rows.  "nens4" / "nens3" / "nens2_specks": member-major handles
    (members in one workgroup for 2 and 4, the member-by-member pass for 3) -- one map set per member, cloud and rain differ from member to
    member (member 0 has none)."""
    import torch
    from miniweatherml_amd import modules
    from util import launched_kernels
    res, used = [], []
    for rows, skip in ((1, 1), (0, 1), (0, 0)):
        if case == "city":
            coupler, dycore, _, _ = modules.make_simple_city(96, 48, 16, 1, 480., 240., 80., "building")
        elif case == "ragged":
            coupler, dycore, _ = modules.make_supercell(129, 44, 12, 1, 64500., 22000., 20000.)
            dm = coupler.get_data_manager_readwrite()
            rho = dm.get("density_dry")
            cl = torch.zeros_like(rho)
            cl[11, 20:24, 119:129] = 2.0e-4; cl[11, 40, 128] = 1.0e-4; cl[0, 3, 0] = 1.0e-4
            dm.get("cloud_liquid").copy_(cl * rho); dm.get("precip_liquid").copy_(0.5 * cl * rho)
        elif case.startswith("nens"):
            nens = int(case[4])
            coupler, dycore, _ = modules.make_supercell(130, 44, 26, nens, 65000., 22000., 20000.)
            dm = coupler.get_data_manager_readwrite()
            rho = dm.get("density_dry")
            cl, pr = torch.zeros_like(rho), torch.zeros_like(rho)
            for e in range(1, nens):
                if case.endswith("specks"):
                    cl[0, 0, 0, e] = 2.0e-4; cl[25, 43, 129, e] = 2.0e-4; cl[12, 21, 64, e] = 1.0e-4; pr[9, 1, 127, e] = 1.0e-4
                else:
                    cl[3 + e:15, 9:31 - 3 * e, 37 + 5 * e:101, e] = 3.0e-4
                    pr[3 + e:12, 12:25, 40:90 - 7 * e, e] = 1.0e-4
            dm.get("cloud_liquid").copy_(cl * rho); dm.get("precip_liquid").copy_(pr * rho)
        else:
            coupler, dycore, _ = modules.make_supercell(130, 44, 26, 1, 65000., 22000., 20000., ord=(3 if case == "ord3" else 5))
            dm = coupler.get_data_manager_readwrite()
            rho = dm.get("density_dry")
            if case in ("one_storm", "ord3"):
                box = torch.zeros_like(rho)
                box[3:15, 9:31, 37:101] = 1.0
                dm.get("cloud_liquid").copy_(3.0e-4 * box * rho)
                dm.get("precip_liquid").copy_(1.0e-4 * box * rho)
            if case == "specks":
                cl, pr = torch.zeros_like(rho), torch.zeros_like(rho)
                for (k, j, i) in ((0, 0, 0), (25, 43, 129), (0, 43, 64), (25, 0, 5), (12, 21, 0), (12, 0, 70), (13, 43, 129), (7, 30, 100)):
                    cl[k, j, i] = 2.0e-4
                for (k, j, i) in ((25, 20, 64), (0, 22, 3), (9, 1, 127), (18, 42, 60)):
                    pr[k, j, i] = 1.0e-4
                dm.get("cloud_liquid").copy_(cl * rho); dm.get("precip_liquid").copy_(pr * rho)
        dycore.set_option("zero_skip", skip); dycore.set_option("zero_rows", rows)
        dycore.set_option("chunk_z", 7); dycore.set_option("chunk_f", 7); dycore.set_option("chunk_y", 44 if case == "ragged" else 9)
        dt = dycore.compute_time_step(coupler)
        launched_kernels(reset=True)
        for n in range(3):
            dycore.time_step(coupler, dt * (2.2 if n == 1 else 1.0))
        used.append(any("k_zero_rows" in k for k in launched_kernels()))
        res.append(gpu_fields(coupler))
    assert used == [True, False, False]
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k
        assert np.array_equal(res[0][k], res[2][k]), k
    if case == "specks":
        assert float(res[0]["tracer1"].max()) > 0.0 and float((res[0]["tracer1"] == 0).mean()) > 0.5


def test_zero_stores_survive_a_change_of_path(mw):
    """Option zero_stores (default 1): a lean iteration of the fused tracer kernel does not store zeros over a row that holds zeros already --
    the coupler's own arrays (known from the step's first scan), the stage slabs S1 / S2 (known from the PREVIOUS sub-cycle's maps).  The
    second only holds while nothing else writes the slabs: a step on the general kernels in between (set_strict(2): they use S1 as their
    stage slab, without maps) must invalidate it.  Sequence on one handle -- marching steps, one of them sub-cycled, a general step, marching
    steps again, the fluxes of the last stage read back (mw_dycore_get_fluxes reads slab S2 whole) -- against the same sequence with
    zero_stores = 0 and with the maps off: same bits, fluxes included."""
    import torch
    from miniweatherml_amd import modules
    res = []
    for opts in ({}, {"zero_stores": 0}, {"zero_rows": 0}):
        coupler, dycore, _ = modules.make_supercell(130, 44, 26, 1, 65000., 22000., 20000.)
        dm = coupler.get_data_manager_readwrite()
        rho = dm.get("density_dry")
        box = torch.zeros_like(rho)
        box[5:12, 15:27, 50:90] = 1.0
        dm.get("cloud_liquid").copy_(3.0e-4 * box * rho)
        dm.get("precip_liquid").copy_(1.0e-4 * box * rho)
        for k, v in opts.items():
            dycore.set_option(k, v)
        dt = dycore.compute_time_step(coupler)
        for n in range(3):
            dycore.time_step(coupler, dt * (2.2 if n == 1 else 1.0))
        dycore.set_strict(2)
        dycore.time_step(coupler, dt)
        dycore.set_strict(0)
        for n in range(3):
            dycore.time_step(coupler, dt)
        f = gpu_fields(coupler)
        for name, a in dycore.fluxes(coupler).items():
            f[name] = a.cpu().numpy().copy()
        res.append(f)
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k
        assert np.array_equal(res[0][k], res[2][k]), k
    assert float((res[0]["tracer1"] == 0).mean()) > 0.3


@pytest.mark.parametrize("seed", range(48))
def test_zero_row_maps_on_random_configurations(mw, seed):
    """Seeded sweep over what the zero-row maps depend on: grid extents (odd sizes, last wavefronts of any width, rows shorter than a wave,
    fewer levels than a chunk), chunk lengths of all three marching kernels, WENO order, the folded and the run-time configuration (with the
    switches at run time ALL tracers can vanish: water vapour is zeroed outside a box there), one to four members (member-major handles keep
    one map set per member), cloud / rain as boxes and single cells anywhere, sub-cycled steps.  Maps on (with zero_stores) against maps off on the same handle settings: same bits."""
    import torch
    from miniweatherml_amd import modules
    rng = np.random.default_rng(1000 + seed)
    nx, ny, nz = int(rng.integers(18, 150)), int(rng.integers(9, 60)), int(rng.integers(3, 40))
    order = 3 if seed % 5 == 4 else 5
    spec = 0 if seed % 3 == 2 else 1
    nens = (1, 1, 2, 1, 4, 3)[seed % 6]                          # (member-major handles: one map set per member; every other member cloud-free)
    chunks = {k: int(rng.integers(3, 40)) for k in ("chunk_y", "chunk_f", "chunk_z") if rng.random() < 0.6}
    nbox, nspeck = int(rng.integers(0, 3)), int(rng.integers(0, 6))
    boxes = [(rng.integers(0, nz), rng.integers(0, ny), rng.integers(0, nx), rng.integers(1, 8), rng.integers(1, 12), rng.integers(1, 30)) for _ in range(nbox)]
    specks = [(int(rng.integers(0, nz)), int(rng.integers(0, ny)), int(rng.integers(0, nx)), int(rng.integers(0, 2))) for _ in range(nspeck)]
    vbox = (int(rng.integers(0, max(1, nz - 2))), int(rng.integers(0, ny)), int(rng.integers(0, nx)))
    factors = [1.0, float(rng.choice([1.0, 2.2, 3.1])), 1.0]
    res = []
    for rows in (1, 0):
        coupler, dycore, _ = modules.make_supercell(nx, ny, nz, nens, 500.0 * nx, 500.0 * ny, 20000., ord=order)
        dm = coupler.get_data_manager_readwrite()
        rho = dm.get("density_dry")
        cl, pr = torch.zeros_like(rho), torch.zeros_like(rho)
        member = torch.tensor([1.0 if (e + seed) % 2 == 0 or nens == 1 else 0.0 for e in range(nens)], device=rho.device, dtype=rho.dtype)
        for (k, j, i, dk, dj, di) in boxes:
            cl[k:k + dk, j:j + dj, i:i + di] = 3.0e-4
            pr[k:k + dk, j:j + max(1, dj // 2), i:i + di] = 1.0e-4
        for (k, j, i, which) in specks:
            (pr if which else cl)[k, j, i] = 2.0e-4
        dm.get("cloud_liquid").copy_(cl * rho * member); dm.get("precip_liquid").copy_(pr * rho * member)
        if spec == 0:
            keep = torch.zeros_like(rho)
            keep[vbox[0]:vbox[0] + 6, vbox[1]:vbox[1] + 12, vbox[2]:vbox[2] + 40] = 1.0
            dm.get("water_vapor").mul_(keep)
        dycore.set_option("spec", spec); dycore.set_option("zero_rows", rows)
        dycore.set_option("zero_verify", rows)                       # (the maps' claims against the data in front of every launch that relies on them)
        for k, v in chunks.items():
            dycore.set_option(k, v)
        dt = dycore.compute_time_step(coupler)
        for f in factors:
            dycore.time_step(coupler, dt * f)
        res.append(gpu_fields(coupler))
        if rows:
            nviol, kinds = dycore.zero_violations()
            assert nviol == 0, ("zero_verify", kinds, seed, nx, ny, nz, nens, order, spec, chunks)   # (ran: >= 0; no claim contradicted)
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), (k, seed, nx, ny, nz, nens, order, spec, chunks)


def test_zero_row_maps_are_exactly_what_the_design_says(mw):
    """The ten maps of a sub-cycle (mw_debug_zero_maps) against a numpy restatement of their definition (mw_march.h): M0 = the rows
    in which a tracer that can vanish is non-zero (water vapour of the folded supercell configuration: always set), and per RK stage s the OR of
    M0 over 3s rows (periodic) and, for Qs / FNs / QYs, over levels k-3s-5 .. k+3s+1 / k-3s-6 .. k+3s+2 / k-3(s-1) .. k+3(s-1) (clamped).
    Equality both ways: a map that misses a row would be a wrong result (the bit-neutrality tests), a map that marks too much only a slower one
    -- nothing else would notice."""
    import torch
    from miniweatherml_amd import capi, modules
    nx, ny, nz = 130, 44, 26
    coupler, dycore, _ = modules.make_supercell(nx, ny, nz, 1, 65000., 22000., 20000.)
    dm = coupler.get_data_manager_readwrite()
    rho = dm.get("density_dry")
    cl, pr = torch.zeros_like(rho), torch.zeros_like(rho)
    cl[3:9, 9:15, 37:60] = 3.0e-4; cl[25, 43, 129] = 2.0e-4; cl[0, 0, 5] = 2.0e-4
    pr[12:14, 30:33, 100:120] = 1.0e-4; pr[20, 2, 64] = 1.0e-4
    dm.get("cloud_liquid").copy_(cl * rho); dm.get("precip_liquid").copy_(pr * rho)
    rows = [torch.ones(nz, ny, dtype=torch.bool), (cl[..., 0] != 0).any(dim=2).cpu(), (pr[..., 0] != 0).any(dim=2).cpu()]
    m0 = sum((r.numpy().astype(np.uint32) << v) for v, r in enumerate(rows))
    # ... and (round 6) the x segments: bits 4 .. 29 = "a tracer that can vanish may be non-zero in cells [s L, (s + 1) L) of the row", L = ceil(nx / 26),
    # set for every segment within 9 cells (periodic) of a non-zero cell
    cellnz = ((cl[..., 0] != 0) | (pr[..., 0] != 0)).cpu().numpy()
    grown = cellnz.copy()
    for d in range(1, 10):
        grown |= np.roll(cellnz, d, axis=2) | np.roll(cellnz, -d, axis=2)
    Lseg = (nx + 25) // 26
    for sgm in range((nx + Lseg - 1) // Lseg):
        m0 |= grown[:, :, sgm * Lseg:(sgm + 1) * Lseg].any(axis=2).astype(np.uint32) << np.uint32(4 + sgm)
    dycore.time_step(coupler, dycore.compute_time_step(coupler))
    L = capi.lib()
    dims = (C.c_int * 2)()
    n = L.mw_debug_zero_maps(dycore.h, None, 0, dims)
    assert n == 10 * nz * (ny + 18) and tuple(dims) == (nz, ny + 18)
    buf = np.zeros(n, dtype=np.uint32)
    assert L.mw_debug_zero_maps(dycore.h, buf.ctypes.data_as(C.c_void_p), n, dims) == n
    maps = buf.reshape(10, nz, ny + 18)[:, :, 9:9 + ny]

    def rows_or(a, r):
        out = a.copy()
        for d in range(1, r + 1):
            out |= np.roll(a, d, axis=1) | np.roll(a, -d, axis=1)
        return out

    def levels_or(a, lo, hi):
        out = np.zeros_like(a)
        for d in range(lo, hi + 1):
            out |= a[np.clip(np.arange(nz) + d, 0, nz - 1)]
        return out
    assert np.array_equal(maps[0], m0)
    for s in (1, 2, 3):
        b = rows_or(m0, 3 * s)
        assert np.array_equal(maps[s], levels_or(b, -3 * s - 5, 3 * s + 1)), ("Q", s)
        assert np.array_equal(maps[3 + s], levels_or(b, -3 * s - 6, 3 * s + 2)), ("FN", s)
        assert np.array_equal(maps[6 + s], levels_or(b, -3 * (s - 1), 3 * (s - 1))), ("QY", s)
    assert 0.02 < float((maps[7] & 6 != 0).mean()) < 0.6 and float((maps[1] & 6 != 0).mean()) < 1.0     # (the case has both kinds of row)
    seg = (maps[1] >> 4) & ((1 << 26) - 1)
    assert float((seg != 0).mean()) == float((maps[1] & 6 != 0).mean())                              # a row bit set <=> some segment set
    busy = seg[seg != 0]
    assert float(np.mean([bin(int(w)).count("1") for w in busy])) < 0.6 * ((nx + Lseg - 1) // Lseg)       # ... and busy rows are busy in part of their segments only

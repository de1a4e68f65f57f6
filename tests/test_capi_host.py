"""CPU-only checks of the C ABI: the shared library loads, exports every symbol include/mw_cdna4.h declares, the
host-only entry points agree with the oracle, and the device entry points fail LOUDLY without a GPU (no fallback)."""
import ctypes as C
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "mw_cdna4.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mw_[a-z0-9_]+)\s*\(", txt)) - {"mw_exchange_fn"})


def test_header_and_binding_tables_agree(mw):
    from miniweatherml_amd import capi
    assert header_functions() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol(mw):
    from miniweatherml_amd import capi
    L = C.CDLL(capi.LIB_PATH)
    for name in header_functions():
        assert hasattr(L, name), name


def test_no_oracle_or_cpu_fallback_in_product():
    """The product package must never import or link the oracle."""
    pkg = os.path.join(ROOT, "miniweatherml_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "mw_oracle" not in src and "libmw_oracle" not in src, os.path.join(dirpath, f)
    for f in ("mw_cdna4.h",):
        assert "oracle" not in open(os.path.join(ROOT, "include", f)).read().lower()


@pytest.mark.parametrize("nranks", [1, 2, 3, 4, 6, 8, 12, 16])
@pytest.mark.parametrize("grid", [(100, 1), (400, 400), (1024, 1024), (37, 53), (256, 512)])
def test_decompose_matches_oracle(mw, oracle, nranks, grid):
    from miniweatherml_amd import capi
    nxg, nyg = grid
    for rank in range(nranks):
        g = capi.Grid()
        capi.check(capi.lib().mw_decompose(nranks, rank, nxg, nyg, C.byref(g)))
        p, neigh = oracle.make_params(nxg, nyg, 8, nranks=nranks, rank=rank)
        assert (g.nproc_x, g.nproc_y, g.px, g.py) == (p.nproc_x, p.nproc_y, p.px, p.py)
        assert (g.nx, g.ny, g.i_beg, g.j_beg) == (p.nx, p.ny, p.i_beg, p.j_beg)
        assert list(g.neigh) == neigh
    # blocks tile the domain
    tot = 0
    for rank in range(nranks):
        g = capi.Grid()
        capi.check(capi.lib().mw_decompose(nranks, rank, nxg, nyg, C.byref(g)))
        tot += g.nx * g.ny
    assert tot == nxg * nyg


def test_reference_rank_grids():
    """1x1, 1x2, 2x2, 4x2 for 1/2/4/8 ranks in 3-D, nranks x 1 in 2-D."""
    from miniweatherml_amd import capi
    L = capi.lib()
    want = {1: (1, 1), 2: (1, 2), 4: (2, 2), 8: (4, 2)}
    for n, (npx, npy) in want.items():
        g = capi.Grid()
        capi.check(L.mw_decompose(n, 0, 1024, 1024, C.byref(g)))
        assert (g.nproc_x, g.nproc_y) == (npx, npy)
        capi.check(L.mw_decompose(n, 0, 1024, 1, C.byref(g)))
        assert (g.nproc_x, g.nproc_y) == (n, 1)


def test_constants_and_time_step_match_oracle(mw, oracle):
    from miniweatherml_amd import capi
    L = capi.lib()
    g = capi.Grid()
    capi.check(L.mw_default_constants(C.byref(g)))
    p, _ = oracle.make_params(200, 200, 50)
    for k in ("R_d", "R_v", "cp_d", "cp_v", "p0", "grav", "gamma_d", "kappa_d", "C0", "earthrot", "latitude"):
        assert getattr(g, k) == getattr(p, k), k
    capi.check(L.mw_decompose(1, 0, 200, 200, C.byref(g)))
    g.nz, g.xlen, g.ylen, g.zlen = 50, 1.0e5, 1.0e5, 2.0e4
    assert L.mw_dycore_compute_time_step(C.byref(g)) == oracle.lib().mwo_compute_time_step(p) == 0.5581395348837209


def test_exchange_plan_fifo_safe(mw):
    from miniweatherml_amd import capi
    L = capi.lib()
    for nranks in (2, 4, 8):
        for rank in range(nranks):
            g = capi.Grid()
            capi.check(L.mw_decompose(nranks, rank, 512, 512, C.byref(g)))
            peers, so, ro, act = (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)()
            capi.check(L.mw_exchange_plan(C.byref(g), peers, so, ro, act))
            assert list(so) == [0, 1, 2, 3] and list(ro) == [1, 0, 3, 2]
            assert list(act) == [int(g.nproc_x > 1)] * 2 + [int(g.nproc_y > 1)] * 2
            assert peers[0] == g.neigh[3] and peers[1] == g.neigh[5] and peers[2] == g.neigh[1] and peers[3] == g.neigh[7]


def test_device_entry_points_fail_loudly_without_gpu(mw):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from miniweatherml_amd import capi
    L = capi.lib()
    g = capi.Grid()
    capi.check(L.mw_decompose(1, 0, 16, 16, C.byref(g)))
    capi.check(L.mw_default_constants(C.byref(g)))
    g.nz, g.nens, g.num_tracers, g.idWV = 8, 1, 3, 0
    g.xlen, g.ylen, g.zlen = 8000., 8000., 20000.
    g.bc_z = capi.BC_WALL
    h = C.c_void_p()
    rc = L.mw_dycore_create(C.byref(h), C.byref(g), bytes([1, 1, 1]), bytes([1, 1, 1]), None)
    assert rc != 0
    assert b"no HIP device" in L.mw_last_error()
    with pytest.raises(capi.MWError):
        capi.check(rc)
    # round 4's device entry points: the validators and the ponni forward (argument checks first, then "no HIP device")
    out6 = (C.c_longlong * 6)()
    assert L.mw_validate_f64(C.c_void_p(0x1000), 10, out6, None) != 0 and b"no HIP device" in L.mw_last_error()
    assert L.mw_validate_f32(C.c_void_p(0x1000), 10, out6, None) != 0 and b"no HIP device" in L.mw_last_error()
    assert L.mw_validate_f64(None, 10, out6, None) != 0 and b"bad argument" in L.mw_last_error()
    from miniweatherml_amd import modules
    lay = (modules._PonniLayer * 1)(modules._PonniLayer(0, 3, 2, 0.0, 0))
    w = (C.c_float * 6)(*([0.5] * 6))
    assert L.mw_ponni_forward(C.cast(lay, C.c_void_p), 1, w, 6, 4, C.c_void_p(0x1000), C.c_void_p(0x2000), None) != 0
    assert b"no HIP device" in L.mw_last_error()
    assert L.mw_dycore_schedule(None) == -1
    assert L.mw_dycore_rccl_info(None, None, None, None) != 0 and b"null handle" in L.mw_last_error()


def test_create_rejects_bad_grids(mw):
    from miniweatherml_amd import capi
    L = capi.lib()
    g = capi.Grid()
    capi.check(L.mw_decompose(1, 0, 16, 16, C.byref(g)))
    capi.check(L.mw_default_constants(C.byref(g)))
    g.nz, g.nens, g.num_tracers, g.idWV = 8, 1, 0, 0        # no water_vapor tracer
    h = C.c_void_p()
    assert L.mw_dycore_create(C.byref(h), C.byref(g), None, None, None) != 0
    assert b"water_vapor" in L.mw_last_error()


def test_h5_weight_reader_matches_the_text_export(mw):
    """mw_h5_read_f32 (= ponni::load_h5_weights, microphysics_kessler_ponni.h:103-107) on the reference's shipped Keras weight file:
    shapes (in, out) = (5, 10), (10), (10, 4), (4) and values identical to the h5dump text export of the same file; errors are loud."""
    import numpy as np
    from miniweatherml_amd import modules
    from miniweatherml_amd.capi import MWError
    data = os.path.join(ROOT, "miniweatherml_amd", "data")
    h5 = os.path.join(data, "supercell_kessler_singlecell_model_weights.h5")
    a = modules.load_surrogate_weights(weights_h5=h5)
    b = modules.load_surrogate_weights(weights_txt=os.path.join(data, "kessler_surrogate_weights.txt"))
    assert [x.shape for x in a[:4]] == [(5, 10), (10,), (10, 4), (4,)]
    for x, y in zip(a, b):
        assert x.dtype == y.dtype and np.array_equal(x, y)
    assert all(np.array_equal(x, y) for x, y in zip(modules.load_surrogate_weights(), a))       # the default is the .h5
    with pytest.raises(MWError, match="no object named"):
        modules.load_h5_weights(h5, "/dense_6/dense_6", "kernel:1")
    with pytest.raises(MWError, match="not an old-style group|no object named"):
        modules.load_h5_weights(h5, "/dense_6/dense_6/kernel:0", "x")
    with pytest.raises(MWError, match="not an HDF5 file"):
        modules.load_h5_weights(os.path.join(ROOT, "README.md"), "/a", "b")
    with pytest.raises(MWError, match="cannot open"):
        modules.load_h5_weights(os.path.join(ROOT, "no_such_file.h5"), "/a", "b")


def test_h5_weight_reader_survives_malformed_files(mw, tmp_path):
    """The reader takes a path from a YAML key: truncated and corrupted copies of the shipped file must either read (unchanged
    bytes) or fail with a message -- never read outside the file image (every byte goes through a bounds-checked accessor)."""
    import numpy as np
    from miniweatherml_amd import modules
    from miniweatherml_amd.capi import MWError
    src = open(os.path.join(ROOT, "miniweatherml_amd", "data", "supercell_kessler_singlecell_model_weights.h5"), "rb").read()
    rng = np.random.default_rng(5)
    cases = [src[:n] for n in (96, 200, 1000, 2048, len(src) // 2, len(src) - 7)]
    for _ in range(200):                                       # byte flips in the metadata region, where the structure lives
        b = bytearray(src)
        for pos in rng.integers(8, min(len(b), 16384), size=int(rng.integers(1, 6))):
            b[pos] = int(rng.integers(0, 256))
        cases.append(bytes(b))
    for _ in range(60):                                        # 8-byte fields overwritten with huge values (addresses, sizes, dimensions)
        b = bytearray(src)
        pos = int(rng.integers(8, min(len(b), 16384) - 8))
        b[pos:pos + 8] = bytes([0xFF] * int(rng.integers(4, 9))).ljust(8, b"\x7f")
        cases.append(bytes(b))
    ok = bad = 0
    for n, blob in enumerate(cases):
        f = tmp_path / ("m%d.h5" % n)
        f.write_bytes(blob)
        try:
            w = modules.load_h5_weights(str(f), "/dense_6/dense_6", "kernel:0")
            assert w.size <= len(blob) // 4
            ok += 1
        except MWError as e:
            assert str(e)
            bad += 1
    assert bad >= 6 and ok + bad == len(cases)              # (at least the truncated copies; most random flips miss the structure)


def test_ponni_surface_host_side(tmp_path):
    """miniweatherml_amd/host/mw_ponni.h without a GPU: load_h5_weights<N> through the library's HDF5 reader, the layer classes,
    Inference::validate / print, a size mismatch between consecutive layers, a rank mismatch of load_h5_weights<N>, and the loud
    failure of forward_batch_parallel when there is no HIP device (no CPU fallback)."""
    import subprocess
    lib = os.path.join(ROOT, "miniweatherml_amd")
    src = tmp_path / "ponni_host.cpp"
    src.write_text(r'''
#include "%s/miniweatherml_amd/host/mw_ponni.h"
int main(int, char **argv) {
  try {
    auto W1 = ponni::load_h5_weights<2>(argv[1], "/dense_6/dense_6", "kernel:0");
    auto b1 = ponni::load_h5_weights<1>(argv[1], "/dense_6/dense_6", "bias:0");
    auto W2 = ponni::load_h5_weights<2>(argv[1], "/dense_7/dense_7", "kernel:0");
    auto b2 = ponni::load_h5_weights<1>(argv[1], "/dense_7/dense_7", "bias:0");
    printf("shapes %%d %%d %%d %%d %%d %%d w00 %%.9g\n", W1.extent(0), W1.extent(1), b1.extent(0), W2.extent(0), W2.extent(1), b2.extent(0), (double)W1.data[0]);
    ponni::Matvec<float> m1(W1), m2(W2); ponni::Bias<float> c1(b1), c2(b2); ponni::Relu<float> r1(c1.get_num_outputs(), 0.1);
    auto model = ponni::create_inference_model(m1, c1, r1, m2, c2);
    model.validate(); model.print();
    printf("io %%d %%d layers %%d params %%d\n", model.get_num_inputs(), model.get_num_outputs(), model.num_layers, (int)model.parameters().size());
    try { auto bad = ponni::create_inference_model(m1, c2); bad.validate(); printf("NOT REACHED\n"); }
    catch (std::exception &e) { printf("mismatch: %%s\n", e.what()); }
    try { auto w = ponni::load_h5_weights<1>(argv[1], "/dense_6/dense_6", "kernel:0"); printf("NOT REACHED\n"); }
    catch (std::exception &e) { printf("rank: %%s\n", e.what()); }
    if (argc_gpu_less()) {
      try { DeviceView<float> in{nullptr, {5, 16}}; in.ptr = (float *)0x1000; auto out = model.forward_batch_parallel(in); printf("NOT REACHED\n"); }
      catch (std::exception &e) { printf("nogpu: %%s\n", e.what()); }
    }
  } catch (std::exception &e) { fprintf(stderr, "endrun: %%s\n", e.what()); return 1; }
  return 0;
}
'''.replace("argc_gpu_less()", "mw_device_count() < 1") % ROOT)
    exe = tmp_path / "ponni_host"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-x", "c++", str(src), "-o", str(exe), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                           "-L" + lib, "-lmw_cdna4", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe), os.path.join(lib, "data", "supercell_kessler_singlecell_model_weights.h5")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    o = out.stdout
    assert "shapes 5 10 10 10 4 4" in o and "io 5 4 layers 5 params 104" in o
    assert "Matvec with 5 inputs and 10 outputs" in o and "Relu with 10 inputs and 10 outputs and negative_slope == 0.1" in o
    assert "mismatch: ERROR: layer 1 expects 4 inputs, but the layer before it has 10 outputs" in o
    assert "rank: ERROR: load_h5_weights<1>" in o and "has 2 dimensions" in o
    assert "NOT REACHED" not in o
    import torch
    if not torch.cuda.is_available():
        assert "nogpu:" in o and ("no HIP device available" in o or "device allocation failed" in o)


def test_round5_entry_points_host_side(mw):
    """The options / calibration / coverage entry points of round 5 as far as they go without a GPU: argument errors, the release build's
    flags, the thread count of the arithmetic-floor launch, an empty launch registry."""
    import torch
    from miniweatherml_amd import capi
    L = capi.lib()
    v = C.c_longlong(0)
    assert L.mw_dycore_set_option(None, b"pipe", 0) != 0 and b"null argument" in L.mw_last_error()
    assert L.mw_dycore_get_option(None, b"pipe", C.byref(v)) != 0
    assert L.mw_dycore_path(None) == b""
    assert L.mw_dycore_use_rccl_self(None) != 0
    if "MW_LIB_PATH" not in os.environ:
        assert L.mw_build_flags() == 0                           # the shipped library contains no experiment
    assert L.mw_calib_stage_arith_threads(16000000, 25) == 640000 and L.mw_calib_stage_arith_threads(1, 25) == 256
    assert L.mw_calib_stage_arith_threads(0, 25) == 0
    out5 = (C.c_double * 5)()
    assert L.mw_calib_fma64(0, 1.0, out5, None) != 0 and L.mw_calib_fma64(2, -1.0, out5, None) != 0
    assert L.mw_rccl_selftest_config(3, 0) != 0 and L.mw_rccl_selftest_config(0, 0) == 0
    if not torch.cuda.is_available():
        assert L.mw_calib_fma64(2, 0.1, out5, None) != 0 and b"no HIP device" in L.mw_last_error()
        assert L.mw_debug_launched_kernels(None, 0, 0) == 1      # nothing launched: an empty, terminated string


def test_dispatcher_path_space_is_well_formed():
    """tests/util.py: reachable_paths -- the Python statement of the dispatcher's rules that the GPU path matrix realises: every path
    spelled uniquely, the shipped configurations' paths among them."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import path_string, reachable_paths
    paths = [path_string(c) for c in reachable_paths()]
    assert len(paths) == len(set(paths)) == 293
    for must in ("march ord5 K1 nens1 one_stream y_all conv_in_y tracers_fused 3d",             # configs[1], [2]: one GPU
                 "march ord5 K1 mm_direct pipe y_all conv_pipe tracers_fused 3d transport",      # configs[3]: 8 GPUs, 4 members
                 "march ord5 K2 nens1 pipe y_all conv_pipe tracers_fused 3d transport",          # configs[4]: simple_city, 8 GPUs
                 "march ord3 K1 nens1 one_stream y_all conv_in_y tracers_fused 3d",              # the reference's GPU-benchmark order
                 "general-strict ord5 nens1", "general-fast ord9 fused_members transport"):
        assert must in paths, must


def test_every_option_is_documented_and_has_a_default_under_test():
    """The handle's option table (OPTS in csrc/mw_dycore.hip) against the header's list and the GPU test of the defaults: an option added to the
    library without its line in include/mw_cdna4.h or its entry in tests/test_gpu_options.py::DEFAULTS fails here, on the CPU."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "miniweatherml_amd", "csrc", "mw_dycore.hip")).read()
    keys = re.findall(r'\{"([a-z_0-9]+)", &DyOpts::[a-z_0-9]+, (-?\d+|0x[0-9a-f]+), [^,]+, (\d)\}', src)
    assert len(keys) >= 30
    header = open(os.path.join(root, "include", "mw_cdna4.h")).read()
    defaults = open(os.path.join(root, "tests", "test_gpu_options.py")).read()
    defaults = defaults[defaults.index("DEFAULTS = {"):defaults.index("}", defaults.index("DEFAULTS = {"))]
    for key, _, build in keys:
        assert '"%s"' % key in header, "option %s is not in the header's list" % key
        if build == "0":                                           # (experiment options have no default test on the release build)
            assert '"%s"' % key in defaults, "option %s has no entry in tests/test_gpu_options.py::DEFAULTS" % key
    assert '"fused_tracers"' in header and '"fused_tracers"' in defaults      # (the one key that is not a plain field)

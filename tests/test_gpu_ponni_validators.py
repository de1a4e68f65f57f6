"""Round 4: the ponni-shaped surface (mw_ponni_forward; C++ mirror miniweatherml_amd/host/mw_ponni.h) and the DataManager validators
(mw_validate_f64 / _f32; DataManager.h:385-483) on the GPU.

MLP tolerance: 1e-5 relative on the fp32 network output for the MFMA form (the matrix cores sum the same products in another order);
the strict form accumulates in index order and is compared BITWISE with a numpy float32 loop in that order."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ref_stack(layers, x):
    """index-order float32 restatement of a ponni layer stack (Matvec: acc = 0; acc += x[i] * W[i, o], i ascending)"""
    a = [x[i].astype(np.float32) for i in range(x.shape[0])]
    for l in layers:
        if l[0] == "matvec":
            W = np.asarray(l[1], np.float32)
            out = []
            for o in range(W.shape[1]):
                acc = np.zeros_like(a[0])
                for i in range(W.shape[0]):
                    acc = (acc + a[i] * W[i, o]).astype(np.float32)
                out.append(acc)
            a = out
        elif l[0] == "bias":
            b = np.asarray(l[1], np.float32)
            a = [(a[o] + b[o]).astype(np.float32) for o in range(len(a))]
        else:
            sl = np.float32(l[2] if len(l) > 2 else 0.0)
            a = [np.where(v > 0, v, (sl * v).astype(np.float32)).astype(np.float32) for v in a]
    return np.stack(a)


def surrogate_layers(mw):
    from miniweatherml_amd import modules
    W1, b1, W2, b2, si, so = modules.load_surrogate_weights()
    return [("matvec", W1), ("bias", b1), ("relu", 10, 0.1), ("matvec", W2), ("bias", b2)]


@pytest.mark.parametrize("batch", [1, 15, 16, 1000, 70001])
def test_ponni_forward_surrogate_stack(mw, batch):
    from miniweatherml_amd import modules
    layers = surrogate_layers(mw)
    rng = np.random.default_rng(batch)
    x = rng.uniform(-0.2, 1.2, (5, batch)).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    ref = ref_stack(layers, x)
    got = modules.ponni_forward(layers, xt).cpu().numpy()                       # MFMA tiles
    assert got.shape == (4, batch)
    assert np.max(np.abs(got - ref)) <= 1e-5 * max(1.0, float(np.max(np.abs(ref))))
    strict = modules.ponni_forward(layers, xt, strict=1).cpu().numpy()          # thread per element, index order
    assert np.array_equal(strict, ref)


def test_ponni_forward_generic_stack_and_validation(mw):
    from miniweatherml_amd import modules
    from miniweatherml_amd.capi import MWError
    rng = np.random.default_rng(7)
    layers = [("matvec", rng.normal(size=(3, 7))), ("relu", 7, 0.0), ("bias", rng.normal(size=7)), ("matvec", rng.normal(size=(7, 2))),
              ("relu", 2, 0.25)]
    x = rng.normal(size=(3, 513)).astype(np.float32)
    got = modules.ponni_forward(layers, torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(got, ref_stack(layers, x))                            # the generic kernel IS the index-order form
    with pytest.raises(MWError, match="expects 5 inputs"):                       # Inference::validate: sizes must chain
        modules.ponni_forward([("matvec", rng.normal(size=(3, 7))), ("bias", rng.normal(size=5))], torch.from_numpy(x).cuda())
    with pytest.raises(MWError, match="width"):
        modules.ponni_forward([("matvec", rng.normal(size=(3, 40)))], torch.from_numpy(x).cuda())


def test_ponni_forward_equals_fused_module_kernel(mw):
    """forward_batch_parallel on explicitly scaled inputs + the un-scaling of :196-201 reproduces the module's fused kernel (same MFMA
    tiles) up to the reciprocal-multiply of the fused scaling: 1e-6 of the output range."""
    from miniweatherml_amd import modules
    W1, b1, W2, b2, si, so = modules.load_surrogate_weights()
    rng = np.random.default_rng(3)
    n = 4096
    f = [rng.uniform(si[i, 0], si[i, 1], n) for i in range(5)]
    t = [torch.from_numpy(v).cuda() for v in f]
    fused = modules.mlp_forward(*t, W1, b1, W2, b2, si, so)
    x = np.stack([((f[i] - si[i, 0]) / (si[i, 1] - si[i, 0])).astype(np.float32) for i in range(5)])
    y = modules.ponni_forward(surrogate_layers(mw), torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64)
    for o in range(4):
        un = y[o] * (so[o, 1] - so[o, 0]) + so[o, 0]
        if o:
            un = np.maximum(0.0, un)
        assert np.max(np.abs(un - fused[o].cpu().numpy())) <= 1e-6 * (so[o, 1] - so[o, 0])


def test_validators_report_counts_and_first_index(mw, capsys):
    from miniweatherml_amd import modules
    from miniweatherml_amd.capi import MWError
    coupler, dycore, micro = modules.make_supercell(24, 20, 10, 1, 12000., 10000., 20000.)
    dm = coupler.get_data_manager_readwrite()
    assert dm.validate_all() == 0                                               # a fresh supercell state is clean
    temp, wv = dm.get("temp").view(-1), dm.get("water_vapor").view(-1)
    temp[1234] = float("nan"); temp[77] = float("nan"); temp[4000] = float("inf"); temp[3999] = -float("inf")
    wv[555] = -1.0e-9                                                           # positive-definite entry (add_tracer(..., positive = True))
    dm.get("uvel").view(-1)[5] = -3.0                                           # not positive-definite: negative values are fine
    assert dm.validate_nan("temp") == 2 and dm.validate_inf("temp") == 2 and dm.validate_pos("temp") == 0
    assert dm.validate_pos("water_vapor") == 1 and dm.validate("uvel") == 0
    err = capsys.readouterr().err
    assert "NaN discovered in: temp at global index: 77" in err and "inf discovered in: temp at global index: 3999" in err
    assert "water_vapor at global index: 555" in err
    assert dm.validate_all() == 5
    with pytest.raises(MWError):
        dm.validate("temp", die_on_failed_check=True)
    # fp32 entries take the f32 scan
    dm.register_and_allocate("f32_entry", "", (100,), dtype=torch.float32, positive=True)
    dm.get("f32_entry")[42] = -1.0
    assert dm.validate("f32_entry") == 1
    dm.unregister_and_deallocate("f32_entry")
    assert not dm.entry_exists("f32_entry")
    dm.clean_all_entries()
    assert dm.get_dirty_entries() == []
    dm.get("temp")
    assert dm.get_dirty_entries() == ["temp"] and dm.entry_is_dirty("temp")
    dm.clean_entry("temp")
    assert not dm.entry_is_dirty("temp")


@pytest.mark.parametrize("online", [0, 1])
def test_cpp_inference_ponni_driver_equals_python_mirror(mw, tmp_path, online):
    """examples/inference_ponni_driver.cpp = experiments/supercell_kessler_surrogate/inference_ponni.cpp:9-86 over the C++ mirrors
    (mw_ponni.h: load_h5_weights<N>, Matvec / Bias / Relu, create_inference_model, forward_batch_parallel; custom_modules::
    Microphysics_Kessler): every printed number equals the Python mirror's, bit for bit (same library underneath).
    online = 1: the NN result overwrites Kessler's (microphysics_kessler_ponni.h:273-276 un-commented; mw_ponni.h `online`) -- the run
    is then steered by the network, and the two mirrors must still agree in every bit."""
    from miniweatherml_amd import build, modules
    exe = os.path.join(ROOT, "examples", "inference_ponni_driver")
    if not os.path.exists(exe):
        build.build_examples(verbose=False)
    data = os.path.join(ROOT, "miniweatherml_amd", "data")
    nx, ny, nz, steps = 48, 1, 20, 3
    yml = tmp_path / "input.yaml"
    yml.write_text("---\n# the reference's keys (inputs/input_euler3d.yaml)\nsim_time: 86400\nnx_glob: %d\nny_glob: %d\nnz     : %d\nnens   : 1\n"
                   "xlen: 100000\nylen: 100000\nzlen: 20000\ninit_data: supercell\nout_prefix: test\ndt_gcm: 900\ndt_phys: 0.\nout_freq: -1.\n"
                   "keras_weights_h5: \"%s/supercell_kessler_singlecell_model_weights.h5\"\nnn_input_scaling: \"%s/kessler_surrogate_input_scaling.txt\"\n"
                   "nn_output_scaling: \"%s/kessler_surrogate_output_scaling.txt\"\n" % (nx, ny, nz, data, data, data))
    out = subprocess.run([exe, str(yml), str(steps), str(online)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Matvec with 5 inputs and 10 outputs" in out.stdout and "negative_slope == 0.1" in out.stdout      # model.print()
    m = re.search(r"steps (\d+) etime (\S+) maxw (\S+) sum_temp (\S+) sum_nn_temp (\S+) diffs (\S+) (\S+) (\S+) (\S+) ponni_out (\S+) (\S+) (\S+) (\S+) validate_all (\d+)", out.stdout)
    assert m, out.stdout
    v = [float(g) for g in m.groups()]
    assert int(v[0]) == steps and int(v[-1]) == 0
    surrogate = modules.Microphysics_Kessler_Surrogate()
    surrogate.online = bool(online)
    coupler, dycore, micro, nudger = modules.make_supercell(nx, ny, nz, 1, 1.0e5, 1.0e5, 2.0e4, micro=surrogate, with_nudger=True)
    etime = 0.0
    for _ in range(steps):                                                      # inference_ponni.cpp:69-82
        dt = dycore.compute_time_step(coupler)
        dycore.time_step(coupler, dt)
        micro.time_step(coupler, dt)
        d = micro._diffs if online else micro.mean_diffs(coupler)               # (:266-269: inside the module's time_step, BEFORE the overwrite)
        modules.sponge_layer(coupler, dt)
        nudger.nudge_to_column(coupler, dt)
        etime += dt
    dm = coupler.get_data_manager_readonly()

    def ssum(a):
        s = 0.0
        for x in a.ravel().tolist():
            s += x
        return s
    assert etime == v[1]
    assert float(dm.get("wvel", True).abs().max()) == v[2]
    assert ssum(dm.get("temp", True).cpu().numpy()) == v[3]
    assert ssum(micro._nn_out[0].cpu().numpy()) == v[4]
    assert [d["rho_v"], d["rho_c"], d["rho_r"], d["temp"]] == v[5:9]
    si = micro.scl_in
    x = np.stack([((dm.get(n, True).cpu().numpy().ravel() - si[i, 0]) / (si[i, 1] - si[i, 0])).astype(np.float32)
                  for i, n in enumerate(("temp", "density_dry", "water_vapor", "cloud_liquid", "precip_liquid"))])
    y = modules.ponni_forward(surrogate_layers(mw), torch.from_numpy(x).cuda()).cpu().numpy()
    assert [ssum(y[o].astype(np.float64)) for o in range(4)] == v[9:13]


def test_cpp_ponni_load_h5_weights_rank_mismatch_is_an_error(mw, tmp_path):
    """load_h5_weights<1> on a rank-2 dataset must end the run (the reference's template parameter is the array rank)."""
    src = tmp_path / "t.cpp"
    src.write_text('#include "%s/miniweatherml_amd/host/mw_ponni.h"\nint main(int, char **argv) { try { auto w = ponni::load_h5_weights<1>(argv[1], "/dense_6/dense_6", "kernel:0"); }\n'
                   ' catch (std::exception &e) { fprintf(stderr, "endrun: %%s\\n", e.what()); return 1; } return 0; }\n' % ROOT)
    exe = tmp_path / "t"
    lib = os.path.join(ROOT, "miniweatherml_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-x", "c++", str(src), "-o", str(exe), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                           "-L" + lib, "-lmw_cdna4", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe), os.path.join(lib, "data", "supercell_kessler_singlecell_model_weights.h5")], capture_output=True, text=True)
    assert out.returncode == 1 and "has 2 dimensions" in out.stderr

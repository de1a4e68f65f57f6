"""The C++ host side (miniweatherml_amd/host/mw_facade.h + examples/supercell_driver.cpp) mirrors the reference
driver's call sequence; it must give exactly what the Python mirror gives (same library underneath) and match the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_driver(*args):
    exe = os.path.join(ROOT, "examples", "supercell_driver")
    if not os.path.exists(exe):
        from miniweatherml_amd import build
        build.build_examples(verbose=False)
    out = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    m = re.search(r"etime (\S+) maxw (\S+) sum_density_dry (\S+) steps_per_s (\S+) validate_all (\d+)", out.stdout)
    assert m, out.stdout
    assert int(m.group(5)) == 0                               # DataManager::validate_all after the run: no NaN / inf / negative tracer
    return [float(v) for v in m.groups()[:4]]


def test_cpp_driver_matches_oracle_known_answers(mw, oracle):
    etime, maxw, sumr, _ = run_driver(32, 32, 16, 1, 16000., 16000., 20000., 3)
    assert abs(maxw - 9.35195661007467982e-01) <= 1e-10
    assert abs(sumr - 7.85159575943703931e+03) <= 1e-11 * 7.85e3
    assert abs(etime - 3 * 0.69767441860465118) < 1e-14


def test_cpp_driver_equals_python_mirror(mw):
    from miniweatherml_amd import modules
    etime, maxw, sumr, _ = run_driver(24, 20, 12, 2, 12000., 10000., 20000., 4, "supercell", 1)
    coupler, dycore, micro = modules.make_supercell(24, 20, 12, 2, 12000., 10000., 20000.)
    for _ in range(4):
        dt = dycore.compute_time_step(coupler)
        dycore.time_step(coupler, dt)
        micro.time_step(coupler, dt)
    w = coupler.get_data_manager_readonly().get("wvel", True).cpu().numpy()
    r = coupler.get_data_manager_readonly().get("density_dry", True).cpu().numpy()
    s = 0.0
    for x in r.ravel().tolist():
        s += x
    assert float(np.abs(w).max()) == maxw
    assert s == sumr


def test_cpp_driver_reports_endrun(mw):
    exe = os.path.join(ROOT, "examples", "supercell_driver")
    out = subprocess.run([exe, "16", "16", "8", "1", "8000", "8000", "20000", "1", "no_such_case"], capture_output=True, text=True)
    assert out.returncode == 1 and "Invalid init_data" in out.stderr


def test_cpp_driver_full_loop_equals_python(mw):
    from miniweatherml_amd import modules
    etime, maxw, sumr, _ = run_driver(20, 16, 12, 1, 10000., 8000., 20000., 3, "supercell", 2)
    coupler, dycore, micro, nudger = modules.make_supercell(20, 16, 12, 1, 10000., 8000., 20000., with_nudger=True)
    for _ in range(3):
        modules.supercell_step(coupler, dycore, micro, nudger)
    w = coupler.get_data_manager_readonly().get("wvel", True).cpu().numpy()
    r = coupler.get_data_manager_readonly().get("density_dry", True).cpu().numpy()
    s = 0.0
    for x in r.ravel().tolist():
        s += x
    assert float(np.abs(w).max()) == maxw and s == sumr


def test_cpp_simple_city_driver_equals_python_and_writes_the_files(mw, tmp_path):
    """examples/simple_city_driver.cpp = experiments/simple_city/driver.cpp:32-84 over the C++ facade: same numbers as the
    Python mirror, and both netCDF files (running output + time averages) are bit-identical to the Python mirror's."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cdf
    from miniweatherml_amd import modules
    exe = os.path.join(ROOT, "examples", "simple_city_driver")
    if not os.path.exists(exe):
        from miniweatherml_amd import build
        build.build_examples(verbose=False)
    args = [40, 40, 16, 1, 200., 200., 80., 4, "building", str(tmp_path / "cpp"), str(tmp_path / "cpp_avg.nc"), 2]
    out = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    m = re.search(r"etime (\S+) maxu (\S+) sum_density_dry (\S+)", out.stdout)
    etime, maxu, sumr = [float(v) for v in m.groups()]
    coupler, dycore, hs, ta = modules.make_simple_city(40, 40, 16, 1, 200., 200., 80., "building", out_prefix=str(tmp_path / "py"))
    dycore.output(coupler, 0.0)
    t = 0.0
    for s in range(4):
        t += modules.simple_city_step(coupler, dycore, hs, ta)
        if (s + 1) % 2 == 0:
            dycore.output(coupler, t)
    ta.finalize(coupler, str(tmp_path / "py_avg.nc"))
    u = coupler.get_data_manager_readonly().get("uvel", True).cpu().numpy()
    assert t == etime and float(np.abs(u).max()) == maxu
    assert open(str(tmp_path / "cpp.nc"), "rb").read() == open(str(tmp_path / "py.nc"), "rb").read()
    assert open(str(tmp_path / "cpp_avg.nc"), "rb").read() == open(str(tmp_path / "py_avg.nc"), "rb").read()
    r = cdf.Reader(str(tmp_path / "cpp.nc"))
    assert r.numrecs == 3 and list(r.get("t")) == [0.0, r.get("t")[1], etime]


def _multirank(world, args, tmp_path, extra_env=None):
    exe = os.path.join(ROOT, "examples", "supercell_multirank")
    if not os.path.exists(exe):
        from miniweatherml_amd import build
        build.build_examples(verbose=False)
    idf = str(tmp_path / "nccl_id")
    procs = []
    for r in range(world):
        env = dict(os.environ, MW_RANK=str(r), MW_WORLD=str(world), MW_ID_FILE=idf, MW_RUN_ID="pytest-%d" % os.getpid(), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([exe] + [str(a) for a in args], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-2000:]
        m = re.search(r"rank (\d+) of (\d+) block (\d+)x(\d+) at \((\d+),(\d+)\) etime (\S+) maxw (\S+) sum_density_dry (\S+) total_density_dry (\S+) validate_all (\d+)", o)
        assert m, o
        outs.append((m, e))
    return outs


def test_cpp_multirank_host_one_rank_over_rccl(mw, tmp_path):
    """examples/supercell_multirank.cpp with MW_FORCE_RCCL=1 on one GPU: the C++ host joins a 1-rank RCCL communicator
    (mw_dycore_use_rccl through the facade's dycore.use_rccl), and sponge_layer / ColumnNudger sum over it with
    mw_dycore_rccl_allreduce_sum.  A sum over one rank is the identity: same bits as the plain single-rank driver."""
    args = (24, 20, 12, 1, 12000., 10000., 20000., 3)
    (m, err), = _multirank(1, args, tmp_path, {"MW_FORCE_RCCL": "1"})
    assert "RCCL communicator: 1 ranks, this is rank 0" in err
    etime, maxw, sumr, _ = run_driver(*args, "supercell", 2)
    assert float(m.group(7)) == etime and float(m.group(8)) == maxw and float(m.group(9)) == sumr == float(m.group(10))
    assert int(m.group(11)) == 0


def test_cpp_multirank_host_between_gpus(mw, tmp_path):
    """Two C++ processes, one GPU each (skipped on the one-GPU development box): halo strips, the all-reduce of the column modules and
    the id hand-over without MPI or torch.  The dycore is decomposition-invariant; the column sums are added in another order
    (partial sums per rank): 1e-12."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    args = (48, 40, 12, 1, 24000., 20000., 20000., 3)
    outs = _multirank(2, args, tmp_path)
    etime, maxw, sumr, _ = run_driver(*args, "supercell", 2)
    tot = [float(m.group(10)) for m, _ in outs]
    assert tot[0] == tot[1] and abs(tot[0] - sumr) <= 1e-12 * sumr
    assert abs(max(float(m.group(8)) for m, _ in outs) - maxw) <= 1e-11 * maxw
    assert abs(sum(float(m.group(9)) for m, _ in outs) - sumr) <= 1e-12 * sumr


@pytest.mark.parametrize("shape", [(16, 12, 10, 2), (20, 1, 12, 1)])
def test_cpp_multifield_and_clone_into(mw, shape):
    """core::MultiField (model/core/MultipleFields.h:10-96) and core::Coupler::clone_into (coupler.h:85-106; DataManager.h:79-103) of the
    C++ facade, from a module with its OWN HIP kernels (examples/multifield_module.cpp): the aggregates are built the way
    column_nudging.h:28-33, :50-55 build them (`state.add_field(dm.get<real,4>(name))`), taken by value by a kernel that loops
    `state(l,k,j,i,iens)` over Bounds<5>(num_fields,nz,ny,nx,nens), and every cell of every field is checked on the host; a rank
    mismatch and more than max_fields fields end the run; the clone is an independent deep copy (grid, tracers, every entry; no options)
    on which a module runs while the original steps on."""
    from miniweatherml_amd import build
    exe = os.path.join(ROOT, "examples", "multifield_module")
    if not os.path.exists(exe):
        build.build_examples(verbose=False)
    out = subprocess.run([exe] + [str(v) for v in shape], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert re.search(r"multifield ok fields 5 cells %d clone_entries_checked 11" % (shape[0] * shape[1] * shape[2] * shape[3]), out.stdout), out.stdout


def test_cpp_driver_deferred_nudge_equals_the_eager_loop(mw):
    """examples/supercell_driver.cpp mode 3: ColumnNudger::nudge_to_column(coupler, dt, nullptr, nullptr, &dycore) -- the increments ride on the
    next dycore step's conversion (mw_nudge_to_column_deferred); the final fields are read through DataManager::get, whose access hook applies
    what is still parked.  Same numbers as mode 2 (the eager loop), bit for bit."""
    a = run_driver(40, 36, 16, 1, 20000., 18000., 20000., 6, "supercell", 2)
    b = run_driver(40, 36, 16, 1, 20000., 18000., 20000., 6, "supercell", 3)
    assert a[:3] == b[:3], (a, b)

"""bench.py's launch contract on a box without (enough) GPUs: it must refuse loudly instead of silently running one rank."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_more_gpus_than_visible_is_refused():
    import torch
    n = torch.cuda.device_count() + 1
    r = run(["--gpus", str(max(n, 2)), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "GPU(s) are visible" in r.stderr
    assert "cell-updates" not in r.stdout                      # no JSON line from a job that did not run as asked


def test_launcher_world_size_mismatch_is_refused():
    r = run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in r.stderr


def test_watchdog_ends_a_run_that_overstays_its_limit():
    """--timeout-s: a rank still running at the limit reports its last milestone and exits non-zero (a first RCCL contact between GPUs
    that hangs must not eat the caller's lease).  Here the limit expires while torch is still being imported."""
    r = run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--timeout-s", "0.05"])
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    assert "did not finish within" in r.stderr and "last milestone" in r.stderr
    assert "cell-updates" not in r.stdout


import pytest


@pytest.mark.gpu
def test_bench_config4_block_under_the_torchrun_launcher():
    """The launch line the driver uses for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- with N = 1 on the one GPU of the development box: RANK / LOCAL_RANK / WORLD_SIZE come
    from the launcher, the workload is config 4's per-GPU block (256 x 512 x 128, nens 4: the member-major path), and rank 0 prints
    ONE JSON line with the contract's keys."""
    import json
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "config4", "--steps", "2", "--warmup", "1",
           "--no-micro", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["scaling"] == "weak" and d["dtype"] == "f64"
    assert "256x512x128 nens=4" in d["config"]["workload"] and d["config"]["baseline_config"] == "configs[3] per-GPU block"
    assert d["value"] > 1e8                                     # a real run (the block takes ~28 ms per step)
    # the self-diagnosis of a launched job (round 6): what a first run on an 8-GPU node must say about itself
    m = d["multi_gpu"]
    assert m["world"] == 1 and m["rank_grid"] == "1x1" and len(m["per_rank"]) == 1
    r0 = m["per_rank"][0]
    for k in ("rank", "device", "ms_per_step", "wait_state_strips_ms_per_stage", "wait_tracer_strips_ms_per_stage", "one_rank_block_ms_per_step"):
        assert k in r0, k
    assert r0["ms_per_step"] > 1.0 and r0["one_rank_block_ms_per_step"] > 1.0
    assert 0.5 < m["efficiency_in_run"] < 1.5                   # one rank: the block IS the job
    assert "rccl_ranks_seen" in m and "ms_per_step_max_over_ranks" in m


@pytest.mark.gpu
def test_bench_micro_section_on_a_small_block():
    """The sections behind the timed region on a grid that takes seconds: the step at the sustained power limit, the storm / mature figures timed
    INSIDE the loop (hipEvents around the loop's own dycore steps) with the back-to-back figure of rounds 4-6 beside them, the state-qualified
    roofline, the RCCL self-loop blocks."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--nx", "96", "--ny", "96", "--nz", "40", "--steps", "3", "--warmup", "1", "--storm-steps", "160",
           "--mature-steps", "330", "--sustained-steps", "40", "--no-pmc", "--no-calib", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["value_sustained"] > 0 and d["sustained"]["ms_per_step"] > 0 and d["config"]["value_sustained"] == d["value_sustained"]
    for k, total in (("storm", 160), ("mature", 330)):
        s = d[k]
        assert s["ms_per_step"] > 0 and "inside the loop" in s["timing"] and ("iteration %d" % total) in s["state"]
        assert s["isolated_after"]["ms_per_step"] > 0
        assert 0.0 <= s["tiles_full_form"] <= s["isolated_after"]["tiles_full_form"] <= 1.0      # without Kessler in between the non-zero set only grows
        assert d["value_" + k] == s["cell_updates_per_s"]
    f = d["roofline"]["frac_by_state"]
    for k in ("cloud_free", "cloud_free_sustained", "storm", "mature", "developed"):
        assert f[k]["frac"] > 0 and f[k]["ms_per_step"] > 0, k
    assert d["simulation_loop"]["steps"] == 157 and d["simulation_loop"]["to_mature"]["steps"] == 327
    assert d["value_simulation_loop"] > 0 and d["value_developed"] > 0 and d["kessler"] and d["mlp"]

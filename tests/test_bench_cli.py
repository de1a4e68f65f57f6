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

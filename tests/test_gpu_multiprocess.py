"""One process per rank (as on a real node), all ranks on the one available GPU: process-level rendezvous, the
torch.distributed halo transport in its host-staged mode, decomposition, pack/unpack -- gathered result must equal
the single-rank GPU run bitwise."""
import os
import subprocess
import sys

import pytest

from test_multirank_cpu import free_port

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run_job(world, nxg, nyg, nz, nsteps, mode="dycore"):
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "gpu_dist_worker.py"), str(r), str(world), str(port), str(nxg),
                               str(nyg), str(nz), str(nsteps), mode], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])


def test_two_processes_3d():
    run_job(2, 24, 32, 12, 2)


def test_four_processes_2x2():
    run_job(4, 32, 32, 10, 2)


def test_two_processes_full_loop_with_allreduce():
    """dycore + Kessler + sponge_layer + ColumnNudger on 2 processes: halo strips AND the horizontal-mean all-reduce."""
    run_job(2, 24, 32, 12, 3, "full")

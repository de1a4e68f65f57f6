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
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL / device-memory sharing between processes needs it on this pool (as bench.py sets it)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "gpu_dist_worker.py"), str(r), str(world), str(port), str(nxg),
                               str(nyg), str(nz), str(nsteps), mode], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
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


def test_two_processes_full_loop_with_deferred_nudge():
    """The same loop with ColumnNudger.nudge_to_column(defer_to=dycore) on every rank: the sums go through the all-reduce as before, the
    increments are parked; a block of a decomposed domain applies them with a pass at the start of its next time step (its conversion is
    part of the pipelined exchange), and the DataManager applies the last ones when the fields are read."""
    run_job(2, 24, 32, 12, 3, "full_defer")


def _ngpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("world,nxg,nyg", [(2, 24, 32), (4, 32, 32), (8, 48, 32)])
def test_rccl_transport_between_gpus(world, nxg, nyg):
    """The built-in RCCL transport (mw_dycore_use_rccl: ncclSend/ncclRecv in one group on a side stream) between DISTINCT
    GPUs, one process per GPU: gathered blocks == the single-rank run, bitwise; the worker also asserts that exactly one
    librccl is mapped and that the library's entry points come from it.  Skipped on boxes with fewer GPUs (the development
    box has one); the driver's multi-GPU node runs it."""
    if _ngpus() < world:
        pytest.skip("needs %d GPUs, %d visible" % (world, _ngpus()))
    run_job(world, nxg, nyg, 12, 2, "rccl")


def test_rccl_full_loop_between_gpus():
    if _ngpus() < 2:
        pytest.skip("needs 2 GPUs, %d visible" % _ngpus())
    run_job(2, 24, 32, 12, 3, "rccl_full")

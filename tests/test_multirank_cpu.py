"""N > 1 path on CPU: the 2-D block decomposition (coupler.h:127-179) + 4-neighbour exchange pattern, run with
world_size-2 and -4 gloo jobs.  Every rank runs the CPU oracle on its block; strips travel over torch.distributed in
the order of the product's exchange plan.  The gathered result must equal a single-rank run BITWISE
(the dycore is decomposition-invariant)."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_job(world, nxg, nyg, nz, nsteps, bc=None):
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "gloo_worker.py"), str(r), str(world), str(port), str(nxg),
                               str(nyg), str(nz), str(nsteps)] + (["%d,%d,%d" % bc] if bc else []), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-2000:])


def test_two_ranks_3d_south_equals_north_peer():
    """2 ranks in 3-D -> 1x2 rank grid: south and north neighbour are the SAME rank (FIFO matching matters)."""
    run_job(2, 12, 16, 8, 2)


def test_two_ranks_2d_west_equals_east_peer():
    """2 ranks in 2-D -> 2x1 rank grid: west and east neighbour are the same rank."""
    run_job(2, 32, 1, 12, 2)


def test_four_ranks_2x2():
    run_job(4, 16, 16, 8, 2)


@pytest.mark.parametrize("world,bc", [(2, (2, 2, 2)), (2, (1, 0, 1)), (4, (2, 2, 2)), (4, (1, 1, 2))])
def test_wall_and_open_boundaries_across_ranks(world, bc):
    """Wall (2) / open (1) boundaries in decomposed directions (:782-825, :1040-1081 apply on the domain-edge ranks only): the
    gloo job must equal, bitwise, the same decomposition run with all ranks in one process -- and, where only undecomposed
    directions are non-periodic, the single-rank run."""
    run_job(world, 16, 16, 8, 2, bc)


def test_z_periodic_across_ranks():
    run_job(2, 12, 16, 8, 2, (0, 0, 0))

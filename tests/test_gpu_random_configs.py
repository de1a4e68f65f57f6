"""Seeded random configurations of the production path against the oracle: grid shape (odd sizes, 2-D and 3-D), ensemble size,
number of tracers and their positive / adds_mass flags, boundary conditions, z / y chunk sizes of the marching kernels, and a
rough state (random wind, sparse tracers) that keeps the FCT limiter busy.  Complements the hand-picked cases elsewhere."""
import os

import numpy as np
import pytest

from util import compare_fields, gpu_fields, push_fields, set_options

pytestmark = pytest.mark.gpu


def draw(seed):
    rng = np.random.default_rng(1000 + seed)
    two_d = rng.uniform() < 0.25
    nx = int(rng.integers(5, 75))
    ny = 1 if two_d else int(rng.integers(3, 18))
    nz = int(rng.integers(4, 26))
    nens = int(rng.choice([1, 1, 2, 3, 4]))
    nt = int(rng.integers(1, 5))
    pos = [1] + [int(rng.uniform() < 0.7) for _ in range(nt - 1)]
    adds = [1] + [int(rng.uniform() < 0.5) for _ in range(nt - 1)]
    bc = (int(rng.choice([0, 0, 1, 2])), 0 if two_d else int(rng.choice([0, 0, 1, 2])), int(rng.choice([1, 2, 2])))
    chunks = {k: int(rng.integers(3, 12)) for k in ("chunk_z", "chunk_f", "chunk_y")}
    return dict(nx=nx, ny=ny, nz=nz, nens=nens, nt=nt, pos=pos, adds=adds, bc=bc, chunks=chunks, rng=rng)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MW_RANDOM_SEEDS", "32"))))     # MW_RANDOM_SEEDS=300: a longer sweep
def test_random_configuration(mw, oracle, seed, monkeypatch):
    from miniweatherml_amd import modules
    c = draw(seed)
    set_options(monkeypatch, **c["chunks"])
    nx, ny, nz, nens, nt = c["nx"], c["ny"], c["nz"], c["nens"], c["nt"]
    xlen, ylen = 500.0 * nx, 500.0 * max(ny, 2)

    class Micro(modules.Microphysics_Kessler):
        def init(self, coupler):
            coupler.add_tracer("water_vapor", "Water Vapor", True, True)
            for t in range(1, nt):
                coupler.add_tracer("tr%d" % t, "", bool(c["pos"][t]), bool(c["adds"][t]))
    coupler, dycore, _ = modules.make_supercell(nx, ny, nz, nens, xlen, ylen, 20000., micro=Micro(), perturb=False)
    p, _ = oracle.make_params(nx, ny, nz, nens, xlen, ylen, 20000., num_tracers=nt)
    odyc = oracle.OracleDycore(p, tracer_positive=c["pos"], tracer_adds_mass=c["adds"])
    of = oracle.Fields(odyc.p)
    odyc.init("supercell", of)
    rng = c["rng"]
    of.temp *= 1.0 + 0.01 * rng.uniform(-1, 1, of.temp.shape)
    for a, amp in ((of.uvel, 15.0), (of.vvel, 15.0 if ny > 1 else 0.0), (of.wvel, 4.0)):
        a += amp * rng.uniform(-1, 1, a.shape)
    for t in range(1, nt):
        blob = rng.uniform(size=of.tracers[t].shape)
        of.tracers[t][...] = np.where(blob > 0.6, 1e-3 * rng.uniform(size=blob.shape), 0.0)
    if nens > 1:                                               # members differ
        of.uvel[..., 1:] += 0.5
    push_fields(coupler, of)
    dycore.set_bc(coupler, *c["bc"])
    odyc.p.bc_x, odyc.p.bc_y, odyc.p.bc_z = c["bc"]
    dt = dycore.compute_time_step(coupler)
    for step in range(2):
        dycore.time_step(coupler, dt)
        odyc.time_step(of, dt)
        compare_fields(gpu_fields(coupler), of.as_dict(), 1e-10, "seed %d %r step %d" % (seed, {k: v for k, v in c.items() if k != "rng"}, step + 1))


@pytest.mark.parametrize("seed", range(int(os.environ.get("MW_RANDOM_SEEDS_ORD", "12"))))
def test_random_configuration_other_weno_orders(mw, oracle, seed):
    """The same kind of draw at WENO orders 3, 7 and 9 (general kernels; orders 7 / 9 with their 4- / 5-cell halos), all boundary types,
    against the oracle build of that order."""
    from miniweatherml_amd import modules
    c = draw(500 + seed)
    order = (3, 7, 9)[seed % 3]
    Oo = oracle.with_order(order)
    nx, ny, nz, nens, nt = max(c["nx"], 10), c["ny"], max(c["nz"], 9), c["nens"], c["nt"]
    if ny > 1: ny = max(ny, 10)                                 # a 9-cell stencil plus its halo must fit the periodic wrap
    xlen, ylen = 500.0 * nx, 500.0 * max(ny, 2)

    class Micro(modules.Microphysics_Kessler):
        def init(self, coupler):
            coupler.add_tracer("water_vapor", "Water Vapor", True, True)
            for t in range(1, nt):
                coupler.add_tracer("tr%d" % t, "", bool(c["pos"][t]), bool(c["adds"][t]))
    coupler, dycore, _ = modules.make_supercell(nx, ny, nz, nens, xlen, ylen, 20000., micro=Micro(), perturb=False, ord=order)
    p, _ = Oo.make_params(nx, ny, nz, nens, xlen, ylen, 20000., num_tracers=nt)
    odyc = Oo.OracleDycore(p, tracer_positive=c["pos"], tracer_adds_mass=c["adds"])
    of = Oo.Fields(odyc.p)
    odyc.init("supercell", of)
    rng = c["rng"]
    of.temp *= 1.0 + 0.01 * rng.uniform(-1, 1, of.temp.shape)
    for a, amp in ((of.uvel, 15.0), (of.vvel, 15.0 if ny > 1 else 0.0), (of.wvel, 4.0)):
        a += amp * rng.uniform(-1, 1, a.shape)
    for t in range(1, nt):
        blob = rng.uniform(size=of.tracers[t].shape)
        of.tracers[t][...] = np.where(blob > 0.6, 1e-3 * rng.uniform(size=blob.shape), 0.0)
    push_fields(coupler, of)
    dycore.set_bc(coupler, *c["bc"])
    odyc.p.bc_x, odyc.p.bc_y, odyc.p.bc_z = c["bc"]
    dt = dycore.compute_time_step(coupler)
    for step in range(2):
        dycore.time_step(coupler, dt)
        odyc.time_step(of, dt)
        compare_fields(gpu_fields(coupler), of.as_dict(), 1e-10, "ord %d seed %d %r step %d" % (order, seed, {k: v for k, v in c.items() if k not in ("rng", "chunks")}, step + 1))

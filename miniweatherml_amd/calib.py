"""Host side of the calibration entry points (include/mw_cdna4.h: mw_calib_fma64, mw_calib_stage_arith): the measured fp64 ceiling of
this GPU and the arithmetic floor of one RK stage.  Used by bench.py (the bench line's fp64_valu / arith_floor fields) and
tools/calib.py.  No reference counterpart ("calibrate with an FMA microbenchmark")."""
import ctypes as C

import numpy as np
import torch

from . import capi
from .capi import check

SPEC_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 4.0        # 1024 SIMDs, one fp64 wave64 instruction per 4 cycles, 2.4 GHz peak engine clock


def fma64(waves_per_simd, seconds=0.5, device="cuda:0"):
    out = (C.c_double * 5)()
    with torch.cuda.device(device):
        st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        check(capi.lib().mw_calib_fma64(int(waves_per_simd), float(seconds), out, st))
    return {"waves_per_simd": waves_per_simd, "wave_instr_per_s": out[0], "ms": out[1], "clock_GHz_in_kernel": out[2] or None,
            "frac_of_spec": out[0] / SPEC_WAVE_INSTR_PER_S, "tflops_fp64": out[0] * 128 / 1e12, "cus": int(out[4])}


def stage_table(kind, nlev=32, seed=3):
    """(nlev, 8, 64) values of the eight reconstruction variables along a 64-cell row: "smooth" = the scales of the supercell's initial
    field (long waves), "cloud_free" = the same with cloud and rain exactly zero (the benchmark's headline state), "rough" = the scales of its developed storm with cell-to-cell noise (every stencil sees non-smooth data: the
    limiter's weights move away from the ideal ones, and the operands toggle many more bits per instruction -- power)."""
    amp = np.array([1.0e-2, 15.0, 8.0, 3.0, 1.5, 1.2e-2, 1.0e-3, 2.0e-5])
    k = np.arange(nlev)[:, None, None]
    i = np.arange(64)[None, None, :]
    v = np.arange(8)[None, :, None]
    if kind in ("smooth", "cloud_free"):
        t = np.sin(0.05 * i + 0.11 * k + 0.7 * v) * (0.5 + 0.5 * np.cos(0.03 * i - 0.07 * k))
    else:
        rng = np.random.default_rng(seed)
        t = 0.5 * np.sin(0.05 * i + 0.11 * k + 0.7 * v) + rng.uniform(-1.0, 1.0, (nlev, 8, 64))
    t = t * amp[None, :, None]
    t[:, 5:, :] = np.abs(t[:, 5:, :])                          # tracers are non-negative
    if kind == "cloud_free":
        t[:, 6:, :] = 0.0                                      # no cloud, no rain: the zero short-cut of the kernels applies (18 reconstructions per cell)
    return np.ascontiguousarray(t, dtype=np.float64)


def stage_arith(kind="smooth", cells=400 * 400 * 100, levels=25, device="cuda:0"):
    """-> dict(ms for `cells` cell-stages, ...).  levels = cells per thread (k_xz_state marches chunks of 25 levels on config 2)."""
    L = capi.lib()
    g = capi.Grid()
    check(L.mw_default_constants(C.byref(g)))
    hyt = 300.0
    bg = (C.c_double * 4)(1.0, hyt, g.C0 * hyt ** g.gamma_d, 1.0 / hyt)
    tab = torch.from_numpy(stage_table(kind)).to(device)
    nthr = L.mw_calib_stage_arith_threads(int(cells), int(levels))
    sink = torch.empty(nthr, dtype=torch.float64, device=device)
    out = (C.c_double * 3)()
    with torch.cuda.device(device):
        st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        check(L.mw_calib_stage_arith(C.c_void_p(tab.data_ptr()), tab.shape[0], int(cells), int(levels), 1 if kind == "cloud_free" else 3, bg,
                                     C.c_void_p(sink.data_ptr()), out, st))
        torch.cuda.synchronize()
    assert bool(torch.isfinite(sink).all()), "mw_calib_stage_arith produced non-finite values"
    done = out[1]
    return {"data": kind, "reconstructions_per_cell": 18 if kind == "cloud_free" else 24, "ms": out[0], "cells": done, "ms_per_stage_of_requested_cells": out[0] * cells / done, "levels_per_thread": levels,
            "workgroups": int(out[2]), "ns_per_cell": out[0] * 1e6 / done}

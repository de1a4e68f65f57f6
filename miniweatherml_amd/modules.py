"""Host-side mirror of the reference *modules* for the hot path (same class names, method names, argument
meaning, error behaviour), each a thin driver of the C ABI in include/mw_cdna4.h:

  Dynamics_Euler_Stratified_WenoFV   model/modules/dynamics_euler_stratified_wenofv.h:20-2196
  Microphysics_Kessler               model/modules/microphysics_kessler.h:8-346
  Microphysics_Kessler_Surrogate     experiments/supercell_kessler_surrogate/custom_modules/microphysics_kessler_ponni.h
  perturb_temperature                model/modules/perturb_temperature.h:8-67

No arithmetic happens in Python: everything runs in libmw_cdna4.so on the GPU (no fallback).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import capi
from .capi import MWError, check
from .coupler import Coupler, endrun

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def _ptr(t):
    if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
        raise MWError("field tensors must be contiguous fp64 CUDA tensors")
    return C.c_void_p(t.data_ptr())


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


# Options every NEW dycore handle of this process gets right after mw_dycore_create (mw_dycore_set_option; keys in include/mw_cdna4.h).
# Tests and A/B tools fill it (monkeypatch.setitem) where rounds 1-4 set MW_* environment variables; empty = the library's defaults.
DEFAULT_OPTIONS = {}
# (convenience of THIS Python host, read once at import -- the library itself never looks at it: MW_OPTIONS="pipe=0,chunk_z=7" pre-fills
#  the table for command-line A/B runs, e.g. tools/ab_bench.sh)
for _kv in filter(None, os.environ.get("MW_OPTIONS", "").split(",")):
    _k, _, _v = _kv.partition("=")
    DEFAULT_OPTIONS[_k.strip()] = int(_v)


# What the dispatcher chose in every time_step of this process since the list was last cleared (mw_dycore_path): the tests' comparisons
# drain it, so that each oracle comparison is logged with the kernel paths that produced the compared fields (tests/util.py).
PATH_LOG = []


class Dynamics_Euler_Stratified_WenoFV:
    ord = 5               # the reference's compile-time MW_ORD (:24-29); Dynamics_Euler_Stratified_WenoFV(ord=3) = its -DMW_ORD=3 build
    hs = 2
    num_state = 5
    idR, idU, idV, idW, idT = 0, 1, 2, 3, 4

    def __init__(self, ord=5):
        if ord not in (3, 5, 7, 9):
            endrun("ERROR: WENO order must be 3, 5, 7 or 9")
        self.ord, self.hs = ord, (ord - 1) // 2
        self.h = C.c_void_p(None)
        self.etime = 0.0
        self._tracer_ptrs = None
        self._fields = None

    def __del__(self):
        try:
            if self.h and self.h.value:
                capi.lib().mw_dycore_destroy(self.h)
                self.h = C.c_void_p(None)
        except Exception:
            pass

    # dynamics_euler_stratified_wenofv.h:70-77
    def compute_time_step(self, coupler):
        return capi.lib().mw_dycore_compute_time_step(C.byref(coupler.grid))

    def _bind(self, coupler):
        dm = coupler.get_data_manager_readwrite()
        names = coupler.get_tracer_names()
        self._fields = [dm.get(n) for n in ("density_dry", "uvel", "vvel", "wvel", "temp")]
        self._tracers = [dm.get(n) for n in names]
        arr = (C.c_void_p * max(1, len(names)))()
        for i, t in enumerate(self._tracers):
            arr[i] = _ptr(t).value
        self._tracer_ptrs = arr

    # dynamics_euler_stratified_wenofv.h:1197-1683
    def init(self, coupler):
        L = capi.lib()
        g = coupler.grid
        nz, ny, nx, nens = coupler.get_nz(), coupler.get_ny(), coupler.get_nx(), coupler.get_nens()
        # physical constants only if absent (:1227-1249)
        gc = capi.Grid()
        check(L.mw_default_constants(C.byref(gc)))
        for k in ("R_d", "cp_d", "R_v", "cp_v", "p0", "grav", "earthrot"):
            if not coupler.option_exists(k):
                coupler.set_option(k, getattr(gc, k))
        R_d, cp_d, p0 = coupler.get_option("R_d"), coupler.get_option("cp_d"), coupler.get_option("p0")
        if not coupler.option_exists("cv_d"): coupler.set_option("cv_d", cp_d - R_d)
        if not coupler.option_exists("gamma_d"): coupler.set_option("gamma_d", cp_d / coupler.get_option("cv_d"))
        if not coupler.option_exists("kappa_d"): coupler.set_option("kappa_d", R_d / cp_d)
        if not coupler.option_exists("cv_v"): coupler.set_option("cv_v", coupler.get_option("R_v") - coupler.get_option("cp_v"))
        if not coupler.option_exists("C0"):
            coupler.set_option("C0", gc.C0 if (R_d, cp_d, p0) == (gc.R_d, gc.cp_d, gc.p0) else
                               float(np.power(R_d * np.power(p0, -coupler.get_option("kappa_d")), coupler.get_option("gamma_d"))))
        coupler.set_option("latitude", 0.0)
        for k in ("R_d", "R_v", "cp_d", "cp_v", "p0", "grav", "gamma_d", "kappa_d", "C0", "earthrot", "latitude"):
            setattr(g, k, float(coupler.get_option(k)))
        dm = coupler.get_data_manager_readwrite()
        for n in ("density_dry", "uvel", "vvel", "wvel", "temp"):            # :1253-1257
            dm.register_and_allocate(n, "", (nz, ny, nx, nens))
        names = coupler.get_tracer_names()
        T = len(names)
        if "water_vapor" not in names:
            endrun("ERROR: a tracer named water_vapor must be registered before dycore.init (idWV, :1292)")
        pos = bytes(int(coupler.get_tracer_info(n)[2]) for n in names)
        adds = bytes(int(coupler.get_tracer_info(n)[3]) for n in names)
        g.num_tracers = T
        g.idWV = names.index("water_vapor")
        coupler.set_option("idWV", g.idWV)                                     # :1300
        dm.register_and_allocate("tracer_adds_mass", "", (T,), dtype=torch.bool)
        dm.get("tracer_adds_mass").copy_(torch.tensor([b != 0 for b in adds]))
        init_data = coupler.get_option("init_data")
        self.out_freq = coupler.get_option("out_freq", -1.0)
        if init_data not in capi.INIT_IDS:
            endrun("ERROR: Invalid init_data in yaml input file")              # :1310
        g.enable_gravity = int(bool(coupler.get_option("enable_gravity", True)))
        g.bc_x, g.bc_y, g.bc_z, g.use_immersed = capi.BC_PERIODIC, capi.BC_PERIODIC, capi.BC_WALL, 0
        with torch.cuda.device(coupler.device):
            check(L.mw_dycore_create(C.byref(self.h), C.byref(g), pos, adds, _stream_ptr(coupler.device)))
            for key, val in DEFAULT_OPTIONS.items():
                self.set_option(key, val)
            if self.ord != 5:
                check(L.mw_dycore_set_order(self.h, self.ord))              # before init: the supercell data uses `ord` GLL points
            self._bind(coupler)
            check(L.mw_dycore_init(self.h, capi.INIT_IDS[init_data], *[_ptr(t) for t in self._fields], self._tracer_ptrs))
        check(L.mw_dycore_get_grid(self.h, C.byref(g)))
        coupler.set_option("use_immersed_boundaries", bool(g.use_immersed))    # :1312,1426,1554
        coupler.add_option("bc_x", g.bc_x); coupler.add_option("bc_y", g.bc_y); coupler.add_option("bc_z", g.bc_z)
        # fields the reference registers for other modules (:1313, :1663-1682) -- zero-copy views of the handle's memory
        hy = [np.zeros((nz, nens)), np.zeros((nz, nens)), np.zeros((nz + 1, nens)), np.zeros((nz + 1, nens))]
        check(L.mw_dycore_get_background(self.h, *[a.ctypes.data_as(C.POINTER(C.c_double)) for a in hy]))
        self.hy_dens_cells, self.hy_dens_theta_cells, self.hy_dens_edges, self.hy_dens_theta_edges = hy
        dm.register_and_allocate("hy_dens_cells", "hydrostatic density cell averages", (nz, nens)).copy_(torch.from_numpy(hy[0]))
        dm.register_and_allocate("hy_dens_theta_cells", "hydrostatic density*theta cell averages", (nz, nens)).copy_(torch.from_numpy(hy[1]))
        self.etime = 0.0
        self.num_out = 0
        if self.out_freq >= 0.0:                                               # :1659: the initial state is record 0
            self.output(coupler, self.etime)

    def _wrap(self, ptr, shape):
        """Zero-copy CUDA tensor over library-owned device memory."""
        n = int(np.prod(shape))

        class _Arr:
            pass
        a = _Arr()
        a.__cuda_array_interface__ = dict(shape=(n,), typestr="<f8", data=(int(ptr), False), version=2)
        return torch.as_tensor(a, device="cuda").view(*shape)

    def fluxes(self, coupler):
        """state_flux_{x,y,z}, tracers_flux_{x,y,z} as the reference registers them (:1671-1676)."""
        out = (C.c_void_p * 6)()
        check(capi.lib().mw_dycore_get_fluxes(self.h, out))
        nz, ny, nx, nens, T = coupler.get_nz(), coupler.get_ny(), coupler.get_nx(), coupler.get_nens(), coupler.get_num_tracers()
        shp = [(5, nz, ny, nx + 1, nens), (5, nz, ny + 1, nx, nens), (5, nz + 1, ny, nx, nens),
               (T, nz, ny, nx + 1, nens), (T, nz, ny + 1, nx, nens), (T, nz + 1, ny, nx, nens)]
        names = ["state_flux_x", "state_flux_y", "state_flux_z", "tracers_flux_x", "tracers_flux_y", "tracers_flux_z"]
        with torch.cuda.device(coupler.device):
            return {n: self._wrap(out[i], shp[i]) for i, n in enumerate(names)}

    def immersed_proportion(self, coupler):
        p = capi.lib().mw_dycore_immersed_proportion(self.h)
        with torch.cuda.device(coupler.device):
            return self._wrap(p, (coupler.get_nz(), coupler.get_ny(), coupler.get_nx(), coupler.get_nens()))

    # dynamics_euler_stratified_wenofv.h:81-198 (file output excluded)
    def time_step(self, coupler, dt_phys):
        with torch.cuda.device(coupler.device):
            check(capi.lib().mw_dycore_time_step(self.h, *[_ptr(t) for t in self._fields], self._tracer_ptrs, float(dt_phys)))
        self._parked = False                                    # (parked column increments were consumed by the conversion, or applied at entry)
        self.etime += dt_phys
        PATH_LOG.append(self.path())
        if len(PATH_LOG) > 4096:                                # (a long run that nobody drains: keep the distinct entries)
            PATH_LOG[:] = sorted(set(PATH_LOG))
        # :183-186.  out_freq == 0: the reference's etime/0. is +inf >= num_out+1, i.e. a record after every step
        if self.out_freq >= 0.0 and (self.out_freq == 0.0 or self.etime / self.out_freq >= self.num_out + 1):
            self.output(coupler, self.etime)
            self.num_out += 1

    # one compute_tendencies(state(coupler), dt) (:204-552): returns (state_tend, tracers_tend); fluxes via .fluxes()
    def compute_tendencies(self, coupler, dt):
        nz, ny, nx, nens, T = coupler.get_nz(), coupler.get_ny(), coupler.get_nx(), coupler.get_nens(), coupler.get_num_tracers()
        st = torch.zeros((5, nz, ny, nx, nens), dtype=torch.float64, device=coupler.device)
        tt = torch.zeros((T, nz, ny, nx, nens), dtype=torch.float64, device=coupler.device)
        with torch.cuda.device(coupler.device):
            check(capi.lib().mw_dycore_compute_tendencies(self.h, *[_ptr(t) for t in self._fields], self._tracer_ptrs, float(dt),
                                                          _ptr(st), _ptr(tt)))
        return st, tt

    def set_strict(self, strict):
        check(capi.lib().mw_dycore_set_strict(self.h, int(strict)))

    def set_option(self, key, value):
        """A run-time option of this handle (mw_dycore_set_option: schedule, kernel forms, chunk sizes, transport; include/mw_cdna4.h)."""
        check(capi.lib().mw_dycore_set_option(self.h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_longlong(0)
        check(capi.lib().mw_dycore_get_option(self.h, key.encode(), C.byref(v)))
        return v.value

    _parked = False       # column increments may be parked in the handle (set by ColumnNudger.nudge_to_column(defer_to=self), cleared by time_step / flush)

    def flush_pending(self):
        """Applies column increments that ColumnNudger.nudge_to_column(..., defer_to=self) parked in this handle (mw_dycore_flush_pending); a
        no-op when there are none.  The coupler's DataManager calls it in front of every field access."""
        if self._parked:
            self._parked = False
            check(capi.lib().mw_dycore_flush_pending(self.h))

    def pending(self):
        """(parked now, [rode on a conversion, applied by a pass]) -- mw_dycore_pending."""
        out = (C.c_ulonglong * 2)()
        return bool(capi.lib().mw_dycore_pending(self.h, out)), [int(out[0]), int(out[1])]

    def zero_violations(self):
        """(total, [4]) of option zero_verify's counters (mw_debug_zero_violations; a test aid): claims of the zero-row maps that the data
        contradicted; total = -1 when the option never ran on this handle."""
        out = (C.c_ulonglong * 4)()
        n = capi.lib().mw_debug_zero_violations(self.h, out)
        return n, [int(v) for v in out]

    def set_bc(self, coupler, bc_x, bc_y, bc_z):
        check(capi.lib().mw_dycore_set_bc(self.h, bc_x, bc_y, bc_z))
        coupler.set_option("bc_x", bc_x); coupler.set_option("bc_y", bc_y); coupler.set_option("bc_z", bc_z)

    def schedule(self):
        """What the last time_step ran (mw_dycore_schedule): dict(streams = "one" | "two" | "pipelined", y_all, general_kernels)."""
        v = capi.lib().mw_dycore_schedule(self.h)
        return dict(code=v, streams=("one stream", "two streams (state | tracers, tracer stream at high priority)",
                                     "one compute stream, strip exchange on a side stream beside the inner y rows / the tracer stage")[v & 3],
                    y_all=bool(v & 4), general_kernels=bool(v & 8))

    def path(self):
        """The dispatcher's decisions of the last time_step, spelled out (mw_dycore_path)."""
        return (capi.lib().mw_dycore_path(self.h) or b"").decode()

    def rccl_info(self):
        """(ranks, rank, lanes) as the installed RCCL transport's communicator reports them (ncclCommCount / ncclCommUserRank)."""
        n, r, l = C.c_int(-1), C.c_int(-1), C.c_int(0)
        check(capi.lib().mw_dycore_rccl_info(self.h, C.byref(n), C.byref(r), C.byref(l)))
        return n.value, r.value, l.value

    def profile(self, enable):
        check(capi.lib().mw_dycore_profile(self.h, int(enable)))

    def profile_get(self, which):
        ms, n = C.c_double(), C.c_longlong()
        check(capi.lib().mw_dycore_profile_get(self.h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value


class Microphysics_Kessler:
    num_tracers = 3
    ID_V, ID_C, ID_R = 0, 1, 2

    def __init__(self):                                                        # microphysics_kessler.h:29-41
        self.R_d, self.cp_d = 287., 1003.
        self.cv_d = self.cp_d - self.R_d
        self.gamma_d = self.cp_d / self.cv_d
        self.kappa_d = self.R_d / self.cp_d
        self.R_v, self.cp_v = 461., 1859.
        self.cv_v = self.R_v - self.cp_v
        self.p0, self.grav = 1.e5, 9.81
        self._ws = None
        self.strict = 0          # 1: the strict path (reference operation order, glibc's pow / exp): bit-identical to the CPU oracle

    def set_strict(self, strict):
        self.strict = int(bool(strict))

    @staticmethod
    def get_num_tracers():
        return 3

    def micro_name(self):
        return "kessler"

    def init(self, coupler):                                                   # :51-96
        coupler.add_tracer("water_vapor", "Water Vapor", True, True)
        coupler.add_tracer("cloud_liquid", "Cloud liquid", True, True)
        coupler.add_tracer("precip_liquid", "precip_liquid", True, True)
        dm = coupler.get_data_manager_readwrite()
        dm.register_and_allocate("precl", "precipitation rate", (coupler.get_ny(), coupler.get_nx(), coupler.get_nens()),
                                 ["y", "x", "nens"])
        coupler.set_option("micro", "kessler")
        for k in ("R_d", "cp_d", "cv_d", "gamma_d", "kappa_d", "R_v", "cp_v", "cv_v", "p0", "grav"):
            coupler.set_option(k, getattr(self, k))

    def time_step(self, coupler, dt, return_rainsplit=False):                  # :99-162
        dm = coupler.get_data_manager_readwrite()
        rho_v, rho_c, rho_r = dm.get("water_vapor"), dm.get("cloud_liquid"), dm.get("precip_liquid")
        rho_d, temp, precl = dm.get("density_dry", readonly=True), dm.get("temp"), dm.get("precl")
        nz = coupler.get_nz()
        ncol = coupler.get_ny() * coupler.get_nx() * coupler.get_nens()
        L = capi.lib()
        nbytes = L.mw_kessler_workspace_bytes(nz, ncol)
        if self._ws is None or self._ws.numel() * 8 < nbytes:
            self._ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=coupler.device)
        rs = C.c_int(0)
        check(L.mw_kessler_set_strict(self.strict))
        with torch.cuda.device(coupler.device):
            check(L.mw_kessler_time_step(nz, ncol, coupler.get_dz(), float(dt), _ptr(rho_v), _ptr(rho_c), _ptr(rho_r), _ptr(rho_d),
                                         _ptr(temp), _ptr(precl), _ptr(self._ws), C.byref(rs) if return_rainsplit else None,
                                         _stream_ptr(coupler.device)))
        return rs.value if return_rainsplit else None


def load_h5_weights(fname, group, dataset):
    """ponni::load_h5_weights<N> (microphysics_kessler_ponni.h:103-107): one float32 dataset of a Keras HDF5 weight file, through the
    library's dependency-free reader (mw_h5.cpp)."""
    L = capi.lib()
    dims, nd = (C.c_longlong * 8)(), C.c_int(0)
    fb, gb, db = str(fname).encode(), group.encode(), dataset.encode()
    check(L.mw_h5_read_f32(fb, gb, db, None, 0, dims, C.byref(nd)))
    shape = tuple(int(dims[i]) for i in range(nd.value))
    out = np.empty(shape, dtype=np.float32)
    check(L.mw_h5_read_f32(fb, gb, db, out.ctypes.data_as(C.POINTER(C.c_float)), out.size, dims, C.byref(nd)))
    return out


def load_surrogate_weights(weights_txt=None, in_scaling_txt=None, out_scaling_txt=None, weights_h5=None):
    """The Keras weights + min/max scaling tables (microphysics_kessler_ponni.h:97-135).  weights_h5: the reference's
    `keras_weights_h5` file, read like ponni::load_h5_weights does (:103-107); weights_txt: a text export of the 104 values
    (tools/export_mlp_weights.sh).  Default: the reference's shipped weight file (miniweatherml_amd/data/)."""
    if weights_txt is None and weights_h5 is None:
        weights_h5 = os.path.join(_DATA, "supercell_kessler_singlecell_model_weights.h5")
    if weights_h5 is not None:
        W1 = load_h5_weights(weights_h5, "/dense_6/dense_6", "kernel:0")       # Matvec 1   (in, out) = (5, 10)
        b1 = load_h5_weights(weights_h5, "/dense_6/dense_6", "bias:0")
        W2 = load_h5_weights(weights_h5, "/dense_7/dense_7", "kernel:0")       # Matvec 2   (10, 4)
        b2 = load_h5_weights(weights_h5, "/dense_7/dense_7", "bias:0")
        if W1.shape != (5, 10) or b1.shape != (10,) or W2.shape != (10, 4) or b2.shape != (4,):
            endrun("surrogate weight file: expected Dense(5->10) and Dense(10->4)")
        w = None
    else:
        w = np.loadtxt(weights_txt, dtype=np.float64, comments="#").astype(np.float32)
        if w.size != 104:
            endrun("surrogate weight file must hold 104 values")
        W1, b1, W2, b2 = w[:50].reshape(5, 10).copy(), w[50:60].copy(), w[60:100].reshape(10, 4).copy(), w[100:104].copy()
    scl_in = np.loadtxt(in_scaling_txt or os.path.join(_DATA, "kessler_surrogate_input_scaling.txt")).reshape(5, 2)
    scl_out = np.loadtxt(out_scaling_txt or os.path.join(_DATA, "kessler_surrogate_output_scaling.txt")).reshape(4, 2)
    return W1, b1, W2, b2, np.ascontiguousarray(scl_in), np.ascontiguousarray(scl_out)


def mlp_forward(temp, rho_d, rho_v, rho_c, rho_r, W1, b1, W2, b2, scl_in, scl_out, outs=None, strict=0):
    """model.forward_batch_parallel with the fused scaling (microphysics_kessler_ponni.h:176-202).  strict = 1: the thread-per-cell
    form that accumulates in index order (bit-identical to the CPU restatement) instead of the MFMA kernels."""
    fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
    if outs is None:
        outs = [torch.empty_like(temp) for _ in range(4)]
    check(capi.lib().mw_mlp_set_strict(int(bool(strict))))
    with torch.cuda.device(temp.device):
        check(capi.lib().mw_mlp_forward(temp.numel(), _ptr(temp), _ptr(rho_d), _ptr(rho_v), _ptr(rho_c), _ptr(rho_r),
                                        W1.ctypes.data_as(fp), b1.ctypes.data_as(fp), W2.ctypes.data_as(fp), b2.ctypes.data_as(fp),
                                        scl_in.ctypes.data_as(dp), scl_out.ctypes.data_as(dp), *[_ptr(o) for o in outs],
                                        _stream_ptr(temp.device)))
    return outs


class _PonniLayer(C.Structure):
    """mw_ponni_layer_t"""
    _fields_ = [("kind", C.c_int), ("n_in", C.c_int), ("n_out", C.c_int), ("negative_slope", C.c_float), ("offset", C.c_int)]


def ponni_forward(layers, x, strict=0):
    """ponni::Inference::forward_batch_parallel (microphysics_kessler_ponni.h:189) for a stack of ponni layers on a float32 CUDA tensor
    x of shape (num_in, batch), batch fastest -> (num_out, batch).  layers: ("matvec", W (in, out)) | ("bias", b) | ("relu", n, slope)
    in the order of ponni::create_inference_model's arguments (:109).  The C++ mirror is miniweatherml_amd/host/mw_ponni.h."""
    recs, params = [], []
    for l in layers:
        kind = {"matvec": 0, "bias": 1, "relu": 2}[l[0]]
        off = sum(p.size for p in params)
        if kind == 0:
            W = np.ascontiguousarray(l[1], dtype=np.float32); params.append(W.ravel()); recs.append((0, W.shape[0], W.shape[1], 0.0, off))
        elif kind == 1:
            b = np.ascontiguousarray(l[1], dtype=np.float32); params.append(b.ravel()); recs.append((1, b.size, b.size, 0.0, off))
        else:
            recs.append((2, int(l[1]), int(l[1]), float(l[2]) if len(l) > 2 else 0.0, off))
    arr = (_PonniLayer * len(recs))(*[_PonniLayer(*r) for r in recs])
    flat = np.concatenate(params).astype(np.float32) if params else np.zeros(1, np.float32)
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or not x.is_contiguous():
        endrun("ponni_forward: x must be a contiguous float32 (num_in, batch) tensor")
    out = torch.empty((recs[-1][2], x.shape[1]), dtype=torch.float32, device=x.device)
    check(capi.lib().mw_mlp_set_strict(int(bool(strict))))
    with torch.cuda.device(x.device):
        rc = capi.lib().mw_ponni_forward(C.cast(arr, C.c_void_p), len(recs), flat.ctypes.data_as(C.POINTER(C.c_float)),
                                         int(sum(p.size for p in params)), x.shape[1], C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()),
                                         _stream_ptr(x.device))
    check(capi.lib().mw_mlp_set_strict(0))
    check(rc)
    return out


class Microphysics_Kessler_Surrogate(Microphysics_Kessler):
    """custom_modules::Microphysics_Kessler of the surrogate experiment: NN inference beside the true Kessler
    (microphysics_kessler_ponni.h:149-278).  The NN result is returned (and diffed) but not written back,
    exactly like the reference with lines :273-276 commented out."""

    online = False        # True = the four deep_copy_to lines :273-276 un-commented: the NN result replaces Kessler's
    mlp_strict = 0        # 1: the thread-per-cell MLP that accumulates in index order (bit-identical to the CPU restatement; tests)

    def init(self, coupler, weights_txt=None, in_scaling_txt=None, out_scaling_txt=None, weights_h5=None):
        super().init(coupler)
        self.W1, self.b1, self.W2, self.b2, self.scl_in, self.scl_out = load_surrogate_weights(weights_txt, in_scaling_txt,
                                                                                              out_scaling_txt, weights_h5)
        self._nn_out = None

    def time_step(self, coupler, dt):
        dm = coupler.get_data_manager_readwrite()
        temp, rho_d = dm.get("temp"), dm.get("density_dry", readonly=True)
        rho_v, rho_c, rho_r = dm.get("water_vapor"), dm.get("cloud_liquid"), dm.get("precip_liquid")
        self._nn_out = mlp_forward(temp, rho_d, rho_v, rho_c, rho_r, self.W1, self.b1, self.W2, self.b2, self.scl_in, self.scl_out,
                                   self._nn_out, strict=self.mlp_strict)
        super().time_step(coupler, dt)
        if self.online:                                                        # :273-276
            self._diffs = self.mean_diffs(coupler)                             # (the prints of :266-269 come first)
            for dst, src in zip((temp, rho_v, rho_c, rho_r), self._nn_out):
                dst.copy_(src)
        return self._nn_out            # (temp_tmp, rho_v_tmp, rho_c_tmp, rho_r_tmp)

    def mean_diffs(self, coupler):
        """The four 'Relative diff' prints (:266-269): mean(NN - Kessler)."""
        dm = coupler.get_data_manager_readonly()
        t, v, c, r = self._nn_out
        if getattr(self, "_ws_mean", None) is None:
            self._ws_mean = torch.empty(1024, dtype=torch.float64, device=coupler.device)
        out = {}
        with torch.cuda.device(coupler.device):
            for key, nn, name in (("rho_v", v, "water_vapor"), ("rho_c", c, "cloud_liquid"), ("rho_r", r, "precip_liquid"), ("temp", t, "temp")):
                m = C.c_double(0.0)
                check(capi.lib().mw_mean_diff(nn.numel(), _ptr(nn), _ptr(dm.get(name, True)), _ptr(self._ws_mean), C.byref(m),
                                              _stream_ptr(coupler.device)))
                out[key] = m.value
        return out


def perturb_temperature(coupler, thermal=True, random=False):                   # perturb_temperature.h:8-67
    temp = coupler.get_data_manager_readwrite().get("temp")
    with torch.cuda.device(coupler.device):
        if random:      # :25-39 (splitmix64 in place of the unavailable yakl::Random, see include/mw_cdna4.h)
            check(capi.lib().mw_perturb_temperature_random(C.byref(coupler.grid), _ptr(temp), _stream_ptr(coupler.device)))
        if thermal:     # :41-66
            check(capi.lib().mw_perturb_temperature(C.byref(coupler.grid), _ptr(temp), _stream_ptr(coupler.device)))


def use_rccl_allreduce(coupler, dycore):
    """The column modules' sums over ranks (sponge_layer, ColumnNudger) on the dycore handle's own RCCL communicator
    (mw_dycore_rccl_allreduce_sum, ctx = the handle) instead of torch.distributed -- what a C++ host does (host/mw_facade.h)."""
    fn = C.cast(capi.lib().mw_dycore_rccl_allreduce_sum, capi.ALLREDUCE_FN)
    coupler._allreduce = (fn, dycore.h, dycore)


def _torch_allreduce(coupler, group=None):
    """-> (mw_allreduce_fn, ctx) for the sponge / nudger horizontal means: the override installed by use_rccl_allreduce, else
    torch.distributed (RCCL on GPUs); (NULL, None) on one rank."""
    ov = getattr(coupler, "_allreduce", None)
    if ov is not None:
        return ov[0], ov[1]
    import torch.distributed as dist
    if coupler.get_nranks() <= 1 or not dist.is_initialized():
        return C.cast(None, capi.ALLREDUCE_FN), None
    dev = coupler.device

    def cb(ctx, buf, n, stream):
        try:
            class _A:
                pass
            a = _A()
            a.__cuda_array_interface__ = dict(shape=(int(n),), typestr="<f8", data=(int(buf), False), version=2)
            st = torch.cuda.ExternalStream(stream, device=dev) if stream else torch.cuda.default_stream(dev)
            with torch.cuda.device(dev), torch.cuda.stream(st):
                t = torch.as_tensor(a, device=dev)
                if dist.get_backend(group) == "gloo":
                    h = t.cpu(); dist.all_reduce(h, group=group); t.copy_(h)
                else:
                    dist.all_reduce(t, group=group)
            return 0
        except Exception as e:                                   # pragma: no cover
            import sys
            print("allreduce callback failed: %r" % (e,), file=sys.stderr)
            return 1
    fn = capi.ALLREDUCE_FN(cb)
    coupler._allreduce_keep = fn                                 # (the callback object must outlive the call)
    return fn, None


def _field_ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = _ptr(t).value
    return arr


def _column_ws(coupler, nf, holder):
    nbytes = capi.lib().mw_column_workspace_bytes(C.byref(coupler.grid), nf)
    ws = getattr(holder, "_ws_col", None)
    if ws is None or ws.numel() * 8 < nbytes:
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=coupler.device)
        holder._ws_col = ws
    return ws


def set_column_strict(strict):
    """1: sponge_layer / ColumnNudger add their horizontal sums in the reference's serial order (bit-identical to the CPU restatement);
    0: the deterministic tree sums (default).  Process-wide (mw_column_set_strict)."""
    check(capi.lib().mw_column_set_strict(int(bool(strict))))


def sponge_layer(coupler, dt, time_scale=60.0):
    """modules::sponge_layer(coupler, dt, time_scale), model/modules/sponge_layer.h:8-77."""
    dm = coupler.get_data_manager_readwrite()
    fields = [dm.get(n) for n in ("density_dry", "uvel", "vvel", "wvel", "temp")] + [dm.get(n) for n in coupler.get_tracer_names()]
    ws = _column_ws(coupler, len(fields), coupler)
    fn, ctx = _torch_allreduce(coupler)
    with torch.cuda.device(coupler.device):
        check(capi.lib().mw_sponge_layer(C.byref(coupler.grid), _field_ptr_array(fields), len(fields), float(dt), float(time_scale),
                                         _ptr(ws), fn, ctx, _stream_ptr(coupler.device)))


class ColumnNudger:
    """modules::ColumnNudger, model/modules/column_nudging.h:9-108."""
    num_fields = 5

    def __init__(self):
        self.column = None

    @staticmethod
    def _state(coupler):
        dm = coupler.get_data_manager_readwrite()
        return [dm.get(n) for n in ("density_dry", "uvel", "vvel", "temp", "water_vapor")]

    def set_column(self, coupler):                                             # :15-36
        self.column = torch.zeros((5, coupler.get_nz(), coupler.get_nens()), dtype=torch.float64, device=coupler.device)
        ws = _column_ws(coupler, 5, self)
        fn, ctx = _torch_allreduce(coupler)
        with torch.cuda.device(coupler.device):
            check(capi.lib().mw_column_average(C.byref(coupler.grid), _field_ptr_array(self._state(coupler)), _ptr(self.column), _ptr(ws),
                                               fn, ctx, _stream_ptr(coupler.device)))

    def nudge_to_column(self, coupler, dt, defer_to=None):                     # :39-66
        """defer_to = the dycore module (round 6, no reference counterpart): the second pass -- state += dt (column - average) / 900 -- is not
        run; the increments are parked in the dycore handle and its next time_step adds them while it converts the coupler's fields (bit for
        bit the same result, one pass over five fields less per loop iteration).  Whoever reads a field through the DataManager in between
        triggers the pass after all (DataManager.before_access), so nobody ever SEES un-nudged values."""
        if self.column is None:
            endrun("ColumnNudger.nudge_to_column before set_column")
        ws = _column_ws(coupler, 5, self)
        fn, ctx = _torch_allreduce(coupler)
        state = self._state(coupler)
        with torch.cuda.device(coupler.device):
            if defer_to is not None:
                dm = coupler.get_data_manager_readwrite()
                if getattr(defer_to, "_flush_hook_dm", None) is not dm:
                    import weakref
                    ref = weakref.ref(defer_to)
                    dm.before_access.append(lambda: ref() is not None and ref().h and ref().flush_pending())
                    defer_to._flush_hook_dm = dm
                check(capi.lib().mw_nudge_to_column_deferred(defer_to.h, _field_ptr_array(state), _ptr(self.column), float(dt), _ptr(ws), fn, ctx))
                defer_to._parked = True
            else:
                check(capi.lib().mw_nudge_to_column(C.byref(coupler.grid), _field_ptr_array(state), _ptr(self.column), float(dt),
                                                    _ptr(ws), fn, ctx, _stream_ptr(coupler.device)))


# ---------------------------------------------------------------------------------------------------------------------
# File output and the simple_city custom modules (rank 3)
# ---------------------------------------------------------------------------------------------------------------------
def _barrier(coupler):
    import torch.distributed as dist
    if coupler.get_nranks() > 1 and dist.is_initialized():
        dist.barrier()


class _NcFile:
    """RAII wrapper over the mw_nc_* writer (CDF-5 by default, like the reference's NC_64BIT_DATA)."""

    def __init__(self, path, create, fmt=5, header_align=0, var_align=0):
        self.h = C.c_void_p(None)
        L = capi.lib()
        if create:
            check(L.mw_nc_create(C.byref(self.h), path.encode(), fmt, header_align, var_align))
        else:
            check(L.mw_nc_open(C.byref(self.h), path.encode()))

    def def_dim(self, name, n):
        d = C.c_int(-1)
        check(capi.lib().mw_nc_def_dim(self.h, name.encode(), int(n), C.byref(d)))
        return d.value

    def def_var(self, name, dims):
        v = C.c_int(-1)
        arr = (C.c_int * max(1, len(dims)))(*dims)
        check(capi.lib().mw_nc_def_var(self.h, name.encode(), len(dims), arr, C.byref(v)))
        return v.value

    def enddef(self):
        check(capi.lib().mw_nc_enddef(self.h))

    def varid(self, name):
        v = C.c_int(-1)
        check(capi.lib().mw_nc_inq_varid(self.h, name.encode(), C.byref(v)))
        return v.value

    def dimlen(self, name):
        n = C.c_longlong(-1)
        check(capi.lib().mw_nc_inq_dimlen(self.h, name.encode(), C.byref(n)))
        return n.value

    def def_var_typed(self, name, nc_type, dims):
        v = C.c_int(-1)
        arr = (C.c_int * max(1, len(dims)))(*dims)
        check(capi.lib().mw_nc_def_var_typed(self.h, name.encode(), nc_type, len(dims), arr, C.byref(v)))
        return v.value

    def put_typed(self, varid, start, count, data):
        """data: a contiguous numpy array of the variable's own type (int32 / float32 / float64)."""
        data = np.ascontiguousarray(data)
        st = (C.c_longlong * max(1, len(start)))(*start)
        ct = (C.c_longlong * max(1, len(count)))(*count)
        check(capi.lib().mw_nc_put_vara(self.h, varid, st, ct, data.ctypes.data_as(C.c_void_p)))

    def put(self, varid, start, count, data):
        data = np.ascontiguousarray(data, dtype=np.float64)
        st = (C.c_longlong * max(1, len(start)))(*start)
        ct = (C.c_longlong * max(1, len(count)))(*count)
        check(capi.lib().mw_nc_put_vara_double(self.h, varid, st, ct, data.ctypes.data_as(C.c_void_p)))

    def put_field(self, varid, record, coupler, tensor):
        with torch.cuda.device(coupler.device):
            check(capi.lib().mw_output_put_field(self.h, varid, record, C.byref(coupler.grid), _ptr(tensor), _stream_ptr(coupler.device)))

    def set_numrecs(self, n):
        check(capi.lib().mw_nc_set_numrecs(self.h, int(n)))

    def close(self):
        if self.h and self.h.value:
            h, self.h = self.h, C.c_void_p(None)
            check(capi.lib().mw_nc_close(h))


def _coords(coupler):
    """x/y/z cell-centre coordinates of this rank's block (:2133-2147)."""
    g = coupler.grid
    dx, dy, dz = coupler.get_dx(), coupler.get_dy(), coupler.get_dz()
    return ((np.arange(g.nx) + g.i_beg + 0.5) * dx, (np.arange(g.ny) + g.j_beg + 0.5) * dy, (np.arange(g.nz) + 0.5) * dz)


def dycore_output(coupler, etime, fmt=5, barrier=None):
    """Dynamics_Euler_Stratified_WenoFV::output(coupler, etime), dynamics_euler_stratified_wenofv.h:2019-2191, shared-file
    branch (:2092-2188): one CDF-5 file `<out_prefix>.nc`, dims x,y,z (global sizes) and unlimited t, variables x,y,z,t and
    one (t,z,y,x) double variable per coupler field (ensemble member 0).  etime == 0 creates the file (main rank), later calls
    append a record.  `file_per_process` (:2038-2090): see _dycore_output_per_process."""
    if coupler.get_option("file_per_process", False):
        return _dycore_output_per_process(coupler, etime, fmt)
    barrier = barrier or (lambda: _barrier(coupler))
    path = str(coupler.get_option("out_prefix")) + ".nc"
    names = ["density_dry", "uvel", "vvel", "wvel", "temp"] + list(coupler.get_tracer_names())
    g = coupler.grid
    xs, ys, zs = _coords(coupler)
    main = coupler.is_mainproc()
    if etime == 0:
        if main:
            nc = _NcFile(path, True, fmt, 1048576, 1048576)                  # nc_header_align_size / nc_var_align_size, :2103-2104
            dx_, dy_, dz_ = nc.def_dim("x", coupler.get_nx_glob()), nc.def_dim("y", coupler.get_ny_glob()), nc.def_dim("z", g.nz)
            dt_ = nc.def_dim("t", 0)
            for n_, d_ in (("x", [dx_]), ("y", [dy_]), ("z", [dz_]), ("t", [dt_])):
                nc.def_var(n_, d_)
            for n_ in names:
                nc.def_var(n_, [dt_, dz_, dy_, dx_])
            nc.enddef()
            nc.put(nc.varid("z"), [0], [g.nz], zs)
            nc.put(nc.varid("t"), [0], [1], [0.0])
        barrier()
        if not main:
            nc = _NcFile(path, False)
        nc.put(nc.varid("x"), [g.i_beg], [g.nx], xs)
        nc.put(nc.varid("y"), [g.j_beg], [g.ny], ys)
        rec = 0
    else:
        nc = _NcFile(path, False)
        rec = nc.dimlen("t")
        if main:
            nc.put(nc.varid("t"), [rec], [1], [float(etime)])
    dm = coupler.get_data_manager_readonly()
    for n_ in names:
        nc.put_field(nc.varid(n_), rec, coupler, dm.get(n_))
    barrier()                                                                  # every rank's block is in the file ...
    if main:
        nc.set_numrecs(rec + 1)                                                # ... before the record becomes visible
    nc.close()
    barrier()


def _dycore_output_per_process(coupler, etime, fmt=5):
    """The `file_per_process` branch (:2038-2090): every rank writes `<out_prefix>_<rank, 8 digits>.nc` with its LOCAL x/y sizes
    and its own coordinate values; no communication.  (The reference's SimpleNetCDF produces a NetCDF-4 container here; this
    writer produces the classic CDF-5 layout with the same dimensions, variables and values.)"""
    path = "%s_%08d.nc" % (coupler.get_option("out_prefix"), coupler.get_myrank())
    names = ["density_dry", "uvel", "vvel", "wvel", "temp"] + list(coupler.get_tracer_names())
    g = coupler.grid
    xs, ys, zs = _coords(coupler)
    local = capi.Grid.from_buffer_copy(g)                                       # hyperslab offsets are local in a per-rank file
    local.i_beg, local.j_beg = 0, 0
    if etime == 0:
        nc = _NcFile(path, True, fmt)
        dx_, dy_, dz_, dt_ = nc.def_dim("x", g.nx), nc.def_dim("y", g.ny), nc.def_dim("z", g.nz), nc.def_dim("t", 0)
        for n_, d_ in (("x", [dx_]), ("y", [dy_]), ("z", [dz_]), ("t", [dt_])):
            nc.def_var(n_, d_)
        for n_ in names:
            nc.def_var(n_, [dt_, dz_, dy_, dx_])
        nc.enddef()
        nc.put(nc.varid("x"), [0], [g.nx], xs); nc.put(nc.varid("y"), [0], [g.ny], ys); nc.put(nc.varid("z"), [0], [g.nz], zs)
        rec = 0
    else:
        nc = _NcFile(path, False)
        rec = nc.dimlen("t")
    nc.put(nc.varid("t"), [rec], [1], [float(etime)])
    dm = coupler.get_data_manager_readonly()
    for n_ in names:
        with torch.cuda.device(coupler.device):
            check(capi.lib().mw_output_put_field(nc.h, nc.varid(n_), rec, C.byref(local), _ptr(dm.get(n_)), _stream_ptr(coupler.device)))
    nc.set_numrecs(rec + 1)
    nc.close()


Dynamics_Euler_Stratified_WenoFV.output = lambda self, coupler, etime, **kw: dycore_output(coupler, etime, **kw)

_SIX = ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor")


class Horizontal_Sponge:
    """custom_modules::Horizontal_Sponge, experiments/simple_city/custom_modules/horizontal_sponge.h:7-194."""

    def __init__(self):
        self.column = None                                                     # (6, nz, nens): col_rho_d .. col_rho_v
        self.sponge_cells, self.time_scale = 10, 1.0

    def init(self, coupler, sponge_cells=10, time_scale=1.0):                  # :18-91
        dm = coupler.get_data_manager_readonly()
        self.column = torch.zeros((6, coupler.get_nz(), coupler.get_nens()), dtype=torch.float64, device=coupler.device)
        with torch.cuda.device(coupler.device):
            check(capi.lib().mw_horizontal_sponge_column(C.byref(coupler.grid), _field_ptr_array([dm.get(n) for n in _SIX]),
                                                         _ptr(self.column), _stream_ptr(coupler.device)))
        import torch.distributed as dist
        if coupler.get_nranks() > 1 and dist.is_initialized():                 # MPI_Bcast from the main rank, :73-78
            if dist.get_backend() == "gloo":
                h = self.column.cpu(); dist.broadcast(h, 0); self.column.copy_(h)
            else:
                dist.broadcast(self.column, 0)
        self.sponge_cells, self.time_scale = int(sponge_cells), float(time_scale)

    def _override(self, l, val):
        self.column[l].fill_(float(val))

    def override_rho_d(self, val): self._override(0, val)                      # noqa: E704   :94-99
    def override_uvel(self, val): self._override(1, val)                       # noqa: E704
    def override_vvel(self, val): self._override(2, val)                       # noqa: E704
    def override_wvel(self, val): self._override(3, val)                       # noqa: E704
    def override_temp(self, val): self._override(4, val)                       # noqa: E704
    def override_rho_v(self, val): self._override(5, val)                      # noqa: E704

    def apply(self, coupler, dt, x1=True, x2=True, y1=True, y2=True):          # :101-192
        if self.column is None:
            endrun("Horizontal_Sponge.apply before init")
        dm = coupler.get_data_manager_readwrite()
        with torch.cuda.device(coupler.device):
            check(capi.lib().mw_horizontal_sponge_apply(C.byref(coupler.grid), _field_ptr_array([dm.get(n) for n in _SIX]), _ptr(self.column),
                                                        self.sponge_cells, self.time_scale, float(dt), int(x1), int(x2), int(y1), int(y2),
                                                        _stream_ptr(coupler.device)))


class Time_Averager:
    """custom_modules::Time_Averager, experiments/simple_city/custom_modules/time_averager.h:7-143."""

    def __init__(self):
        self.etime = 0.0

    def init(self, coupler):                                                   # :10-35
        dm = coupler.get_data_manager_readwrite()
        shape = (coupler.get_nz(), coupler.get_ny(), coupler.get_nx(), coupler.get_nens())
        for n in _SIX:
            dm.register_and_allocate("time_avg_" + n, "", shape)
            dm.get("time_avg_" + n).zero_()
        self.etime = 0.0

    def accumulate(self, coupler, dt):                                         # :37-78
        dm = coupler.get_data_manager_readwrite()
        with torch.cuda.device(coupler.device):
            check(capi.lib().mw_time_average_accumulate(C.byref(coupler.grid), _field_ptr_array([dm.get(n) for n in _SIX]),
                                                        _field_ptr_array([dm.get("time_avg_" + n) for n in _SIX]), float(self.etime),
                                                        float(dt), _stream_ptr(coupler.device)))
        self.etime += dt

    def finalize(self, coupler, path="time_averaged_fields.nc", fmt=5, barrier=None):   # :80-141
        barrier = barrier or (lambda: _barrier(coupler))
        g = coupler.grid
        xs, ys, zs = _coords(coupler)
        if coupler.is_mainproc():
            nc = _NcFile(path, True, fmt)
            dx_, dy_, dz_ = nc.def_dim("x", coupler.get_nx_glob()), nc.def_dim("y", coupler.get_ny_glob()), nc.def_dim("z", g.nz)
            for n_, d_ in (("x", [dx_]), ("y", [dy_]), ("z", [dz_])):
                nc.def_var(n_, d_)
            for n_ in _SIX:
                nc.def_var(n_, [dz_, dy_, dx_])
            nc.enddef()
            nc.put(nc.varid("z"), [0], [g.nz], zs)
        barrier()
        if not coupler.is_mainproc():
            nc = _NcFile(path, False)
        nc.put(nc.varid("x"), [g.i_beg], [g.nx], xs)
        nc.put(nc.varid("y"), [g.j_beg], [g.ny], ys)
        dm = coupler.get_data_manager_readonly()
        for n_ in _SIX:
            nc.put_field(nc.varid(n_), -1, coupler, dm.get("time_avg_" + n_))
        nc.close()
        barrier()


class StatisticsGatherer:
    """custom_modules::StatisticsGatherer, experiments/supercell_kessler_surrogate/custom_modules/gather_micro_statistics.h:9-90:
    which share of the cells has active microphysics (input = the coupler cloned before micro.time_step, output = after)."""

    def __init__(self):
        self.numer, self.denom, self.num_out = 0.0, 0.0, 0
        self.last_mask = None

    def gather_micro_statistics(self, inp, out, dt, etime, keep_mask=False):    # :19-58
        names = ("temp", "water_vapor", "cloud_liquid", "precip_liquid")
        a = [inp.get_data_manager_readonly().get(n, True) for n in names]
        b = [out.get_data_manager_readonly().get(n, True) for n in names]
        mask = torch.empty((inp.get_nz(), inp.get_ny(), inp.get_nx()), dtype=torch.uint8, device=inp.device) if keep_mask else None
        cnt = C.c_longlong(0)
        with torch.cuda.device(inp.device):
            check(capi.lib().mw_micro_active_count(C.byref(inp.grid), _field_ptr_array(a), _field_ptr_array(b),
                                                   C.c_void_p(mask.data_ptr()) if keep_mask else None, C.byref(cnt), _stream_ptr(inp.device)))
        self.last_mask = mask
        if etime > (self.num_out + 1) * 200:                                   # :54
            self.print(inp)
            self.num_out += 1
        self.numer += float(cnt.value)
        self.denom += float(inp.get_nz() * inp.get_ny() * inp.get_nx())
        return cnt.value

    def ratio(self, coupler):                                                  # MPI_Reduce(SUM) of numer and denom, :77-86
        import torch.distributed as dist
        v = torch.tensor([self.numer, self.denom], dtype=torch.float64)
        if coupler.get_nranks() > 1 and dist.is_initialized():
            v = v.to(coupler.device) if dist.get_backend() != "gloo" else v
            dist.all_reduce(v)
            v = v.cpu()
        return float(v[0] / v[1]) if float(v[1]) > 0 else float("nan")

    def print(self, coupler):
        r = self.ratio(coupler)
        if coupler.is_mainproc():
            print("*** Ratio Active ***:  %10.6e" % r, flush=True)

    def finalize(self, coupler):                                               # :89
        self.print(coupler)


class DataGenerator:
    """custom_modules::DataGenerator, experiments/supercell_kessler_surrogate/custom_modules/generate_micro_surrogate_data.h:11-156:
    samples (5 inputs x 2-cell vertical stencil, 4 outputs; fp32) of the microphysics' effect, about half of them from active
    cells, appended to one file per rank.  The file is classic netCDF (CDF-5) instead of the reference's NetCDF-4 container, all
    variables are defined when it is created, and the random numbers come from splitmix64 (include/mw_cdna4.h)."""

    ratio_active = 0.4                       # :47-49 (from gather_statistics)
    desired_samples_per_time_step = 50.0     # :53
    desired_ratio_active = 0.5               # :55

    def init(self, coupler, directory="."):                                    # :17-33
        self.fname = os.path.join(directory, "supercell_kessler_data_task_%d.nc" % coupler.get_myrank())
        nc = _NcFile(self.fname, True, 5)
        ds, dvi, dst, dvo = nc.def_dim("nsamples", 0), nc.def_dim("num_vars_in", 5), nc.def_dim("sten_size", 2), nc.def_dim("num_vars_out", 4)
        for n in ("time_step_size", "dx", "dy", "dz", "xlen", "ylen", "zlen"):
            nc.def_var_typed(n, 6, [])
        nc.def_var_typed("only_two_dimensions", 4, [])
        nc.def_var_typed("inputs", 5, [ds, dvi, dst])
        nc.def_var_typed("outputs", 5, [ds, dvo])
        nc.enddef()
        nc.close()
        self._meta_written = False
        if coupler.is_mainproc():
            with open(os.path.join(directory, "supercell_kessler_metadata.txt"), "w") as f:
                f.write("This dataset contains data for training a surrogate model to emulate Kessler microphysics.\n\n"
                        "vars_in : temperature, dry air density, water vapor density, cloud liquid density, precipitation density\n"
                        "vars_out: temperature, water vapor density, cloud liquid density, precipitation density\n")

    def generate_samples_stencil(self, inp, out, dt, etime, seed=None):        # :35-153
        import time as _time
        nx, ny, nz, nranks, myrank = inp.get_nx(), inp.get_ny(), inp.get_nz(), inp.get_nranks(), inp.get_myrank()
        ncell = nx * ny * nz
        want_act = self.desired_ratio_active * self.desired_samples_per_time_step / nranks          # :58-59
        want_inact = (1 - self.desired_ratio_active) * self.desired_samples_per_time_step / nranks
        thr_act = want_act / (self.ratio_active * ncell)                                             # :61-62
        thr_inact = want_inact / ((1 - self.ratio_active) * ncell)
        max_mag = (2 ** 64 - 1) // (nranks + ncell)                                                  # :84-85
        seed = (int(_time.time()) if seed is None else int(seed)) % max_mag
        key0 = ((seed + myrank) * ncell) % (2 ** 64)
        names = ("temp", "water_vapor", "cloud_liquid", "precip_liquid")
        a = [inp.get_data_manager_readonly().get(n, True) for n in names]
        b = [out.get_data_manager_readonly().get(n, True) for n in names]
        rho_d = inp.get_data_manager_readonly().get("density_dry", True)
        dev = inp.device
        mask = torch.empty(ncell, dtype=torch.uint8, device=dev)
        L = capi.lib()
        with torch.cuda.device(dev):
            check(L.mw_micro_sample_mask(C.byref(inp.grid), _field_ptr_array(a), _field_ptr_array(b), key0, thr_act, thr_inact,
                                         C.c_void_p(mask.data_ptr()), _stream_ptr(dev)))
            cells = torch.nonzero(mask).flatten().contiguous()                 # ascending = the reference's (k,j,i) loop order
            n = int(cells.numel())
            ins = torch.empty((n, 5, 2), dtype=torch.float32, device=dev)
            outs = torch.empty((n, 4), dtype=torch.float32, device=dev)
            check(L.mw_micro_gather_samples(C.byref(inp.grid), _ptr(rho_d), _field_ptr_array(a), _field_ptr_array(b),
                                            C.c_void_p(cells.data_ptr()), n, C.c_void_p(ins.data_ptr()), C.c_void_p(outs.data_ptr()),
                                            _stream_ptr(dev)))
        self.last_cells = cells.cpu().numpy()                                 # k*ny*nx + j*nx + i of the samples (diagnostic)
        nc = _NcFile(self.fname, False)
        ul = nc.dimlen("nsamples")                                             # :116
        if not self._meta_written:                                             # :118-125 (`if (!nc.varExists(..)) nc.write(..)`)
            for name, val in (("time_step_size", dt), ("dx", inp.get_dx()), ("dy", inp.get_dy()), ("dz", inp.get_dz()),
                              ("xlen", inp.get_xlen()), ("ylen", inp.get_ylen()), ("zlen", inp.get_zlen())):
                nc.put_typed(nc.varid(name), [], [], np.array([val], dtype=np.float64))
            nc.put_typed(nc.varid("only_two_dimensions"), [], [], np.array([0 if inp.get_ny_glob() == 1 else 1], dtype=np.int32))
            self._meta_written = True
        if n:
            nc.put_typed(nc.varid("inputs"), [ul, 0, 0], [n, 5, 2], ins.cpu().numpy())
            nc.put_typed(nc.varid("outputs"), [ul, 0], [n, 4], outs.cpu().numpy())
            nc.set_numrecs(ul + n)
        nc.close()
        return n


def install_exchange(dycore, coupler, transport="rccl", group=None):
    """Picks the halo-exchange transport for a multi-rank run TOGETHER on all ranks: the built-in RCCL transport unless any
    rank cannot set it up, in which case every rank switches to the torch.distributed point-to-point transport (ranks on
    different transports would deadlock).  Returns the transport in use: "none" (one rank), "rccl" or "torch"."""
    import torch.distributed as dist
    if coupler.get_nranks() <= 1:
        return "none"
    dev = coupler.device if dist.get_backend(group) != "gloo" else "cpu"

    def any_rank(failed):
        flag = torch.tensor([1 if failed else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        return int(flag.item()) == 1

    if transport == "rccl":
        failed, why = False, ""
        if not capi.lib().mw_rccl_library_path(None):                        # checked BEFORE the collective communicator set-up
            failed, why = True, "no librccl available to libmw_cdna4"
        if any_rank(failed):
            transport = "torch"
        else:
            try:
                use_rccl_exchange(dycore, coupler, group)
            except MWError as e:
                failed, why = True, str(e)
            if any_rank(failed):
                transport = "torch"
        if failed:
            import sys
            print("rank %d: built-in RCCL transport unavailable (%s)" % (coupler.get_myrank(), why), file=sys.stderr)
    if transport == "torch":
        use_torch_distributed_exchange(dycore, coupler, group)                # replaces (and frees) a half-installed RCCL transport
    return transport


def use_rccl_exchange(dycore, coupler, group=None):
    """Slab halo exchange over RCCL point-to-point inside the library (mw_rccl.cpp): rank 0 creates the ncclUniqueId,
    torch.distributed broadcasts it, every rank joins.  Replaces the MPI_Isend/Irecv of halo_exchange (:641-723)."""
    import torch.distributed as dist
    L = capi.lib()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    # 128 id bytes + 1 status byte.  Rank 0 ALWAYS reaches the broadcast -- with status 1 and a zero id when it could not create the
    # id -- and every rank raises only after it: a rank that left before the collective would leave the others waiting in it.
    ident = torch.zeros(129, dtype=torch.uint8, device=coupler.device)
    why = ""
    if rank == 0:
        buf = C.create_string_buffer(128)
        if L.mw_rccl_unique_id(buf) != 0:
            why = L.mw_last_error().decode(errors="replace")
            ident[128] = 1
        else:
            ident[:128].copy_(torch.tensor(list(buf.raw), dtype=torch.uint8))
    dist.broadcast(ident, 0, group=group)
    host = ident.cpu().tolist()
    if host[128]:
        raise MWError("rank 0 could not create the ncclUniqueId" + (": " + why if why else ""))
    with torch.cuda.device(coupler.device):
        check(L.mw_dycore_use_rccl(dycore.h, bytes(host[:128]), world, rank))


def use_rccl_self_exchange(dycore, coupler):
    """The self-loop test transport (mw_dycore_use_rccl_self): this one rank plays every rank of the coupler's rank grid over a 1-rank
    RCCL communicator -- the real send / receive groups on the side stream(s), on one GPU.  The column modules' sums go through the
    handle's communicator too (times the number of blocks)."""
    with torch.cuda.device(coupler.device):
        check(capi.lib().mw_dycore_use_rccl_self(dycore.h))
    use_rccl_allreduce(coupler, dycore)


def use_torch_distributed_exchange(dycore, coupler, group=None, host_staged=False):
    """Alternative transport for the same exchange: torch.distributed point-to-point ops (backend "nccl" = RCCL) issued
    from the library's exchange callback in the order of mw_exchange_plan.  Same wire pattern as the built-in transport.
    host_staged=True copies the strips through host buffers first -- the reference's non-GPU-aware-MPI mode
    (dynamics_euler_stratified_wenofv.h:687-722); works with the gloo backend."""
    import torch.distributed as dist
    L = capi.lib()
    peers, so, ro, act = (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)()
    check(L.mw_exchange_plan(C.byref(coupler.grid), peers, so, ro, act))
    dev = coupler.device

    def wrap(ptr, n):
        class _A:
            pass
        a = _A()
        a.__cuda_array_interface__ = dict(shape=(int(n),), typestr="<f8", data=(int(ptr), False), version=2)
        return torch.as_tensor(a, device=dev)

    def cb(ctx, sW, sE, sS, sN, rW, rE, rS, rN, nWE, nSN, stream):
        try:
            st = torch.cuda.ExternalStream(stream, device=dev) if stream else torch.cuda.default_stream(dev)
            with torch.cuda.device(dev), torch.cuda.stream(st):
                sb, rb, cnt = [sW, sE, sS, sN], [rW, rE, rS, rN], [nWE, nWE, nSN, nSN]
                if host_staged:
                    hs = {d: wrap(sb[d], cnt[d]).cpu() for d in range(4) if act[d] and cnt[d] and sb[d]}     # syncs the stream
                    hr = {d: torch.empty(cnt[d], dtype=torch.float64) for d in range(4) if act[d] and cnt[d] and rb[d]}
                    reqs = [dist.isend(hs[so[o]], peers[so[o]], group=group) for o in range(4) if so[o] in hs]
                    reqs += [dist.irecv(hr[ro[o]], peers[ro[o]], group=group) for o in range(4) if ro[o] in hr]
                    for r in reqs:
                        r.wait()
                    for d, t in hr.items():
                        wrap(rb[d], cnt[d]).copy_(t, non_blocking=False)
                    return 0
                ops = []
                for o in range(4):
                    d = so[o]
                    if act[d] and cnt[d] and sb[d]:
                        ops.append(dist.P2POp(dist.isend, wrap(sb[d], cnt[d]), peers[d], group))
                for o in range(4):
                    d = ro[o]
                    if act[d] and cnt[d] and rb[d]:
                        ops.append(dist.P2POp(dist.irecv, wrap(rb[d], cnt[d]), peers[d], group))
                if ops:
                    for r in dist.batch_isend_irecv(ops):
                        r.wait()
            return 0
        except Exception as e:                                   # pragma: no cover
            import sys
            print("exchange callback failed: %r" % (e,), file=sys.stderr)
            return 1

    dycore._xchg_cb = capi.EXCHANGE_FN(cb)                       # keep alive
    check(L.mw_dycore_set_exchange(dycore.h, dycore._xchg_cb, None))


def make_supercell(nx_glob, ny_glob, nz, nens=1, xlen=1.0e5, ylen=1.0e5, zlen=2.0e4, init_data="supercell", device="cuda:0",
                   nranks=1, myrank=0, micro=None, enable_gravity=None, perturb=True, with_nudger=False, ord=5):
    """The set-up sequence of experiments/supercell_example/driver.cpp:41-61 (column nudger excluded)."""
    coupler = Coupler(device)
    coupler.set_option("out_prefix", "test")
    coupler.set_option("init_data", init_data)
    coupler.set_option("out_freq", -1.0)
    if enable_gravity is not None:
        coupler.set_option("enable_gravity", bool(enable_gravity))
    coupler.distribute_mpi_and_allocate_coupled_state(nz, ny_glob, nx_glob, nens, nranks, myrank)
    coupler.set_grid(xlen, ylen, zlen)
    micro = micro or Microphysics_Kessler()
    dycore = Dynamics_Euler_Stratified_WenoFV(ord)
    micro.init(coupler)
    dycore.init(coupler)
    if with_nudger:
        nudger = ColumnNudger()
        nudger.set_column(coupler)                 # driver.cpp:60: set the column BEFORE perturbing
    if perturb:
        perturb_temperature(coupler)
    if with_nudger:
        return coupler, dycore, micro, nudger
    return coupler, dycore, micro


def make_simple_city(nx_glob, ny_glob, nz, nens=1, xlen=2400.0, ylen=2400.0, zlen=120.0, init_data="city", device="cuda:0",
                     nranks=1, myrank=0, out_prefix="test", ord=5):
    """The set-up sequence of experiments/simple_city/driver.cpp:32-62: only water_vapor is registered (zero), gravity off,
    dycore.init -> horiz_sponge.init(coupler, 10, 1.) -> time_averager.init."""
    coupler = Coupler(device)
    coupler.set_option("out_prefix", out_prefix)
    coupler.set_option("init_data", init_data)
    coupler.set_option("out_freq", -1.0)
    coupler.set_option("enable_gravity", False)
    coupler.distribute_mpi_and_allocate_coupled_state(nz, ny_glob, nx_glob, nens, nranks, myrank)
    coupler.set_grid(xlen, ylen, zlen)
    coupler.add_tracer("water_vapor", "water_vapor", True, True)               # driver.cpp:55-56
    coupler.get_data_manager_readwrite().get("water_vapor").zero_()
    dycore, horiz_sponge, time_averager = Dynamics_Euler_Stratified_WenoFV(ord), Horizontal_Sponge(), Time_Averager()
    dycore.init(coupler)
    horiz_sponge.init(coupler, 10, 1.0)
    time_averager.init(coupler)
    return coupler, dycore, horiz_sponge, time_averager


def simple_city_step(coupler, dycore, horiz_sponge, time_averager, dtphys=None):
    """One iteration of experiments/simple_city/driver.cpp:66-79."""
    if dtphys is None or dtphys <= 0:
        dtphys = dycore.compute_time_step(coupler)
    horiz_sponge.apply(coupler, dtphys, True, True, False, False)
    dycore.time_step(coupler, dtphys)
    sponge_layer(coupler, dtphys, 1)
    time_averager.accumulate(coupler, dtphys)
    return dtphys


def supercell_step(coupler, dycore, micro, nudger, dtphys=None, defer_nudge=False):
    """One iteration of the reference's time loop, experiments/supercell_example/driver.cpp:66-79.  defer_nudge: the nudger's increments ride
    on the next dycore step's conversion instead of a pass of their own (ColumnNudger.nudge_to_column(defer_to=...)): same bits."""
    if dtphys is None:
        dtphys = dycore.compute_time_step(coupler)
    dycore.time_step(coupler, dtphys)
    micro.time_step(coupler, dtphys)
    sponge_layer(coupler, dtphys)
    nudger.nudge_to_column(coupler, dtphys, defer_to=dycore if defer_nudge else None)
    return dtphys

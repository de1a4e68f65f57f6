"""ctypes binding of the CPU oracle (oracle/mw_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (miniweatherml_amd) never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# The reconstruction order is a COMPILE-time choice in the reference (-DMW_ORD) and in this restatement: libmw_oracle.so is the
# ord-5 build; `with_order(3)` returns a second instance of this module bound to libmw_oracle_ord3.so.
_LIB_PATH = os.path.join(_HERE, globals().get("_MW_ORACLE_LIB_NAME", "libmw_oracle.so"))


def with_order(order):
    """This module bound to the oracle build of the given WENO order (3, 5, 7 or 9)."""
    import importlib.util
    import sys
    if order == 5:
        return sys.modules[__name__]
    if order not in (3, 7, 9):
        raise ValueError("oracle builds exist for WENO orders 3, 5, 7 and 9")
    name = __name__ + "_ord%d" % order
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.abspath(__file__))
    mod = importlib.util.module_from_spec(spec)
    mod._MW_ORACLE_LIB_NAME = "libmw_oracle_ord%d.so" % order
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    assert mod.lib().mwo_order() == order
    return mod

DATA_THERMAL, DATA_SUPERCELL, DATA_CITY, DATA_BUILDING = 0, 1, 2, 3
BC_PERIODIC, BC_OPEN, BC_WALL = 0, 1, 2
INIT_IDS = {"thermal": DATA_THERMAL, "supercell": DATA_SUPERCELL, "city": DATA_CITY, "building": DATA_BUILDING}


class Params(C.Structure):
    _fields_ = [
        ("nz", C.c_int), ("ny", C.c_int), ("nx", C.c_int), ("nens", C.c_int), ("num_tracers", C.c_int),
        ("nx_glob", C.c_longlong), ("ny_glob", C.c_longlong), ("i_beg", C.c_longlong), ("j_beg", C.c_longlong),
        ("xlen", C.c_double), ("ylen", C.c_double), ("zlen", C.c_double),
        ("px", C.c_int), ("py", C.c_int), ("nproc_x", C.c_int), ("nproc_y", C.c_int),
        ("bc_x", C.c_int), ("bc_y", C.c_int), ("bc_z", C.c_int),
        ("use_immersed", C.c_int), ("enable_gravity", C.c_int),
        ("idWV", C.c_int),
        ("R_d", C.c_double), ("R_v", C.c_double), ("cp_d", C.c_double), ("cp_v", C.c_double), ("p0", C.c_double),
        ("grav", C.c_double), ("gamma_d", C.c_double), ("kappa_d", C.c_double), ("C0", C.c_double),
        ("earthrot", C.c_double), ("latitude", C.c_double),
    ]


XCHG_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int,
                      C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                      C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                      C.c_longlong, C.c_longlong)

ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_longlong)

_lib = None


def build(force=False):
    """Compile oracle/libmw_oracle.so with the committed Makefile (g++, seconds)."""
    src = os.path.join(_HERE, "mw_oracle.cpp")
    libs = [os.path.join(_HERE, n) for n in ("libmw_oracle.so", "libmw_oracle_ord3.so", "libmw_oracle_ord7.so", "libmw_oracle_ord9.so",
                                              "libmw_powcheck.so")]
    csrc = os.path.join(os.path.dirname(_HERE), "miniweatherml_amd", "csrc")
    newest = max(os.path.getmtime(f) for f in (src, os.path.join(_HERE, "weno79.inc"), os.path.join(_HERE, "pow_check.cpp"),
                                               os.path.join(csrc, "mw_glibc_pow.h"), os.path.join(csrc, "mw_glibc_pow_tables.h")))
    if force or any(not os.path.exists(l) or os.path.getmtime(l) < newest for l in libs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def powcheck():
    """(libm_pow, glibc_pow_restated) on numpy arrays: std::pow of the host's C library and the product's restatement of it
    (miniweatherml_amd/csrc/mw_glibc_pow.h compiled for the host) -- see oracle/pow_check.cpp."""
    build()
    L = C.CDLL(os.path.join(_HERE, "libmw_powcheck.so"))
    dp = C.POINTER(C.c_double)
    L.mwo_libm_pow.argtypes = [C.c_longlong, dp, dp, dp]
    L.mwo_glibc_pow_restated.argtypes = [C.c_longlong, dp, dp, dp, C.POINTER(C.c_ubyte)]
    L.mwo_glibc_pow_restated.restype = C.c_longlong

    def libm_pow(x, y):
        x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
        out = np.empty_like(x)
        L.mwo_libm_pow(x.size, x.ctypes.data_as(dp), y.ctypes.data_as(dp), out.ctypes.data_as(dp))
        return out

    def restated(x, y):
        x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
        out, ok = np.empty_like(x), np.empty(x.size, dtype=np.uint8)
        L.mwo_glibc_pow_restated(x.size, x.ctypes.data_as(dp), y.ctypes.data_as(dp), out.ctypes.data_as(dp), ok.ctypes.data_as(C.POINTER(C.c_ubyte)))
        return out, ok.astype(bool)
    return libm_pow, restated


def expcheck(fn="exp"):
    """(libm_fn, restated_fn) for fn = "exp" or "cos": the host C library's function and the product's restatement of it (mw_glibc_pow.h)."""
    build()
    L = C.CDLL(os.path.join(_HERE, "libmw_powcheck.so"))
    dp = C.POINTER(C.c_double)
    f_ref, f_re = getattr(L, "mwo_libm_" + fn), getattr(L, "mwo_glibc_%s_restated" % fn)
    f_ref.argtypes = [C.c_longlong, dp, dp]
    f_re.argtypes = [C.c_longlong, dp, dp, C.POINTER(C.c_ubyte)]
    f_re.restype = C.c_longlong

    def libm_fn(x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        f_ref(x.size, x.ctypes.data_as(dp), out.ctypes.data_as(dp))
        return out

    def restated(x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out, ok = np.empty_like(x), np.empty(x.size, dtype=np.uint8)
        f_re(x.size, x.ctypes.data_as(dp), out.ctypes.data_as(dp), ok.ctypes.data_as(C.POINTER(C.c_ubyte)))
        return out, ok.astype(bool)
    return libm_fn, restated


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    dp = C.POINTER(C.c_double)
    L.mwo_weno5.argtypes = [dp, dp, dp]
    L.mwo_order.restype = C.c_int
    L.mwo_weno5_ideal_weights.argtypes = [dp]
    L.mwo_compute_C0.restype = C.c_double
    L.mwo_compute_C0.argtypes = [C.c_double] * 4
    L.mwo_compute_time_step.restype = C.c_double
    L.mwo_compute_time_step.argtypes = [C.POINTER(Params)]
    L.mwo_decompose.argtypes = [C.c_int, C.c_int, C.c_longlong, C.c_longlong] + [C.POINTER(C.c_int)] * 4 + \
        [C.POINTER(C.c_longlong)] * 4 + [C.POINTER(C.c_int)]
    L.mwo_create.restype = C.c_void_p
    L.mwo_create.argtypes = [C.POINTER(Params), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mwo_destroy.argtypes = [C.c_void_p]
    L.mwo_set_exchange.argtypes = [C.c_void_p, XCHG_FN, C.c_void_p]
    L.mwo_params_ptr.restype = C.POINTER(Params)
    L.mwo_params_ptr.argtypes = [C.c_void_p]
    for name in ("mwo_hy_dens_cells", "mwo_hy_dens_theta_cells", "mwo_hy_dens_edges", "mwo_hy_dens_theta_edges",
                 "mwo_immersed_proportion"):
        getattr(L, name).restype = dp
        getattr(L, name).argtypes = [C.c_void_p]
    L.mwo_flux.restype = dp
    L.mwo_flux.argtypes = [C.c_void_p, C.c_int]
    L.mwo_city_dims.argtypes = [C.POINTER(Params), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mwo_city_building_heights.argtypes = [C.c_int, C.c_int, dp]
    pdp = C.POINTER(dp)
    L.mwo_init.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, pdp]
    L.mwo_perturb_temperature.argtypes = [C.POINTER(Params), dp]
    L.mwo_perturb_temperature_random.argtypes = [C.POINTER(Params), dp, C.c_int]
    L.mwo_stage_tendencies.argtypes = [C.c_void_p, dp, dp, dp, dp, dp, pdp, C.c_double, dp, dp]
    L.mwo_time_step.argtypes = [C.c_void_p, dp, dp, dp, dp, dp, pdp, C.c_double]
    L.mwo_kessler_time_step.restype = C.c_int
    L.mwo_kessler_time_step.argtypes = [C.c_int, C.c_longlong, C.c_double, C.c_double, dp, dp, dp, dp, dp, dp]
    L.mwo_sponge_layer.argtypes = [C.POINTER(Params), pdp, C.c_int, C.c_double, C.c_double, ALLREDUCE_FN, C.c_void_p]
    L.mwo_column_average.argtypes = [C.POINTER(Params), pdp, dp, ALLREDUCE_FN, C.c_void_p]
    L.mwo_nudge_to_column.argtypes = [C.POINTER(Params), pdp, dp, C.c_double, ALLREDUCE_FN, C.c_void_p]
    L.mwo_horizontal_sponge_apply.argtypes = [C.POINTER(Params), pdp, dp, C.c_int, C.c_double, C.c_double] + [C.c_int] * 4
    L.mwo_time_average_accumulate.argtypes = [C.POINTER(Params), pdp, pdp, C.c_double, C.c_double]
    ip = C.POINTER(C.c_int)
    L.mwo_micro_active.restype = C.c_longlong
    L.mwo_micro_active.argtypes = [C.c_int] * 4 + [pdp, pdp, ip]
    L.mwo_micro_sample_thresholds.argtypes = [C.c_int] * 4 + [C.c_double, dp, dp]
    L.mwo_micro_sample_mask.restype = C.c_longlong
    L.mwo_micro_sample_mask.argtypes = [C.c_int] * 4 + [pdp, pdp, C.c_ulonglong, C.c_double, C.c_double, ip]
    L.mwo_micro_gather_samples.restype = C.c_longlong
    L.mwo_micro_gather_samples.argtypes = [C.c_int] * 4 + [dp, pdp, pdp, ip, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    fp = C.POINTER(C.c_float)
    L.mwo_mlp_forward.argtypes = [C.c_longlong, dp, dp, dp, dp, dp, fp, fp, fp, fp, dp, dp, dp, dp, dp, dp]
    _lib = L
    return L


def _dp(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def make_params(nx_glob, ny_glob, nz, nens=1, xlen=1.0e5, ylen=1.0e5, zlen=2.0e4, num_tracers=3, idWV=0,
                enable_gravity=True, nranks=1, rank=0):
    """Grid + constants as the drivers set them up (micro.init first, then dycore.init):
    microphysics_kessler.h:29-41,86-95 and dynamics_euler_stratified_wenofv.h:1227-1249; decomposition
    coupler.h:127-179."""
    L = lib()
    p = Params()
    npx, npy, px, py = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    ib, ie, jb, je = C.c_longlong(), C.c_longlong(), C.c_longlong(), C.c_longlong()
    neigh = (C.c_int * 9)()
    L.mwo_decompose(nranks, rank, nx_glob, ny_glob, npx, npy, px, py, ib, ie, jb, je, neigh)
    p.nz, p.ny, p.nx, p.nens, p.num_tracers = nz, je.value - jb.value + 1, ie.value - ib.value + 1, nens, num_tracers
    p.nx_glob, p.ny_glob, p.i_beg, p.j_beg = nx_glob, ny_glob, ib.value, jb.value
    p.xlen, p.ylen, p.zlen = xlen, ylen, zlen
    p.px, p.py, p.nproc_x, p.nproc_y = px.value, py.value, npx.value, npy.value
    p.bc_x, p.bc_y, p.bc_z = BC_PERIODIC, BC_PERIODIC, BC_WALL
    p.use_immersed, p.enable_gravity, p.idWV = 0, int(enable_gravity), idWV
    p.R_d, p.cp_d, p.R_v, p.cp_v, p.p0, p.grav = 287., 1003., 461., 1859., 1.e5, 9.81
    cv_d = p.cp_d - p.R_d
    p.gamma_d = p.cp_d / cv_d
    p.kappa_d = p.R_d / p.cp_d
    p.C0 = L.mwo_compute_C0(p.R_d, p.p0, p.kappa_d, p.gamma_d)
    p.earthrot, p.latitude = 7.292115e-5, 0.0
    return p, list(neigh)


class Fields:
    """Coupler-side fields (nz,ny,nx,nens) fp64: density_dry, uvel, vvel, wvel, temp + tracers."""

    def __init__(self, p):
        shp = (p.nz, p.ny, p.nx, p.nens)
        self.rho_d = np.zeros(shp)
        self.uvel = np.zeros(shp)
        self.vvel = np.zeros(shp)
        self.wvel = np.zeros(shp)
        self.temp = np.zeros(shp)
        self.tracers = [np.zeros(shp) for _ in range(p.num_tracers)]

    def tracer_ptrs(self):
        arr = (C.POINTER(C.c_double) * max(1, len(self.tracers)))()
        for t, a in enumerate(self.tracers):
            arr[t] = _dp(a)
        return arr

    def copy(self):
        import copy
        return copy.deepcopy(self)

    def as_dict(self):
        d = {"density_dry": self.rho_d, "uvel": self.uvel, "vvel": self.vvel, "wvel": self.wvel, "temp": self.temp}
        for t, a in enumerate(self.tracers):
            d["tracer%d" % t] = a
        return d


class OracleDycore:
    def __init__(self, p, tracer_positive=None, tracer_adds_mass=None):
        L = lib()
        nt = p.num_tracers
        pos = (C.c_int * max(1, nt))(*([1] * nt if tracer_positive is None else [int(x) for x in tracer_positive]))
        adds = (C.c_int * max(1, nt))(*([1] * nt if tracer_adds_mass is None else [int(x) for x in tracer_adds_mass]))
        self.h = L.mwo_create(C.byref(p), pos, adds)
        self.L = L
        self._cb = None

    @property
    def p(self):
        return self.L.mwo_params_ptr(self.h).contents

    def close(self):
        if self.h:
            self.L.mwo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_exchange(self, pyfunc):
        self._cb = XCHG_FN(pyfunc)
        self.L.mwo_set_exchange(self.h, self._cb, None)

    def _arr(self, fn, shape):
        ptr = fn(self.h)
        return np.ctypeslib.as_array(ptr, shape=shape)

    def hy(self):
        p = self.p
        return dict(hy_dens_cells=self._arr(self.L.mwo_hy_dens_cells, (p.nz, p.nens)).copy(),
                    hy_dens_theta_cells=self._arr(self.L.mwo_hy_dens_theta_cells, (p.nz, p.nens)).copy(),
                    hy_dens_edges=self._arr(self.L.mwo_hy_dens_edges, (p.nz + 1, p.nens)).copy(),
                    hy_dens_theta_edges=self._arr(self.L.mwo_hy_dens_theta_edges, (p.nz + 1, p.nens)).copy())

    def immersed_proportion(self):
        p = self.p
        return self._arr(self.L.mwo_immersed_proportion, (p.nz, p.ny, p.nx, p.nens))

    def fluxes(self):
        p = self.p
        nt = max(1, p.num_tracers)
        shp = {0: (5, p.nz, p.ny, p.nx + 1, p.nens), 1: (5, p.nz, p.ny + 1, p.nx, p.nens), 2: (5, p.nz + 1, p.ny, p.nx, p.nens),
               3: (nt, p.nz, p.ny, p.nx + 1, p.nens), 4: (nt, p.nz, p.ny + 1, p.nx, p.nens), 5: (nt, p.nz + 1, p.ny, p.nx, p.nens)}
        names = ["state_flux_x", "state_flux_y", "state_flux_z", "tracers_flux_x", "tracers_flux_y", "tracers_flux_z"]
        return {names[w]: np.ctypeslib.as_array(self.L.mwo_flux(self.h, w), shape=shp[w]).copy() for w in range(6)}

    def init(self, init_data, f):
        self.L.mwo_init(self.h, INIT_IDS[init_data], _dp(f.rho_d), _dp(f.uvel), _dp(f.vvel), _dp(f.wvel), _dp(f.temp),
                        f.tracer_ptrs())

    def time_step(self, f, dt_phys):
        self.L.mwo_time_step(self.h, _dp(f.rho_d), _dp(f.uvel), _dp(f.vvel), _dp(f.wvel), _dp(f.temp), f.tracer_ptrs(),
                             float(dt_phys))

    def stage_tendencies(self, f, dt):
        p = self.p
        st = np.zeros((5, p.nz, p.ny, p.nx, p.nens))
        tt = np.zeros((max(1, p.num_tracers), p.nz, p.ny, p.nx, p.nens))
        self.L.mwo_stage_tendencies(self.h, _dp(f.rho_d), _dp(f.uvel), _dp(f.vvel), _dp(f.wvel), _dp(f.temp),
                                    f.tracer_ptrs(), float(dt), _dp(st), _dp(tt))
        return st, tt

    def compute_time_step(self):
        return self.L.mwo_compute_time_step(self.L.mwo_params_ptr(self.h))


def perturb_temperature(p, temp, thermal=True, random=False, myrank=0):
    if random:
        lib().mwo_perturb_temperature_random(C.byref(p), _dp(temp), int(myrank))
    if thermal:
        lib().mwo_perturb_temperature(C.byref(p), _dp(temp))


def weno5(stencil):
    s = np.ascontiguousarray(stencil, dtype=np.float64)          # `ord` values (the bound build's order)
    coefs = np.zeros(lib().mwo_order())
    gll = np.zeros(2)
    lib().mwo_weno5(_dp(s), _dp(coefs), _dp(gll))
    return coefs, gll


def weno5_ideal_weights():
    w = np.zeros(4)
    lib().mwo_weno5_ideal_weights(_dp(w))
    return w


def kessler_time_step(dz, dt, rho_v, rho_c, rho_r, rho_d, temp, precl):
    """Arrays (nz, ...) modified in place; returns rainsplit."""
    nz = rho_v.shape[0]
    ncol = rho_v.size // nz
    return lib().mwo_kessler_time_step(nz, ncol, float(dz), float(dt), _dp(rho_v), _dp(rho_c), _dp(rho_r), _dp(rho_d),
                                       _dp(temp), _dp(precl))


def mlp_forward(temp, rho_d, rho_v, rho_c, rho_r, W1, b1, W2, b2, scl_in, scl_out):
    n = temp.size
    outs = [np.zeros(temp.shape) for _ in range(4)]
    lib().mwo_mlp_forward(n, _dp(temp), _dp(rho_d), _dp(rho_v), _dp(rho_c), _dp(rho_r), _fp(W1), _fp(b1), _fp(W2), _fp(b2),
                          _dp(scl_in), _dp(scl_out), *[_dp(o) for o in outs])
    return outs


def micro_active(in4, out4):
    """StatisticsGatherer: (count, active mask (nz,ny,nx)) from the four fields before / after the microphysics (nz,ny,nx,nens)."""
    nz, ny, nx, nens = in4[0].shape
    act = np.zeros((nz, ny, nx), dtype=np.int32)
    n = lib().mwo_micro_active(nz, ny, nx, nens, _ptr_array(in4), _ptr_array(out4), act.ctypes.data_as(C.POINTER(C.c_int)))
    return int(n), act


def micro_sample_thresholds(nz, ny, nx, nranks, desired_samples_per_time_step=50.0):
    a, b = C.c_double(), C.c_double()
    lib().mwo_micro_sample_thresholds(nz, ny, nx, nranks, float(desired_samples_per_time_step), C.byref(a), C.byref(b))
    return a.value, b.value


def micro_samples(rho_d, in4, out4, key0, thr_active, thr_inactive):
    """DataGenerator: (do_sample mask (nz,ny,nx), inputs (n,5,2) float32, outputs (n,4) float32) in the reference's loop order."""
    nz, ny, nx, nens = in4[0].shape
    mask = np.zeros((nz, ny, nx), dtype=np.int32)
    ip = C.POINTER(C.c_int)
    n = lib().mwo_micro_sample_mask(nz, ny, nx, nens, _ptr_array(in4), _ptr_array(out4), C.c_ulonglong(key0 % 2 ** 64), thr_active,
                                    thr_inactive, mask.ctypes.data_as(ip))
    ins, outs = np.zeros((max(n, 1), 5, 2), dtype=np.float32), np.zeros((max(n, 1), 4), dtype=np.float32)
    m = lib().mwo_micro_gather_samples(nz, ny, nx, nens, _dp(rho_d), _ptr_array(in4), _ptr_array(out4), mask.ctypes.data_as(ip),
                                       ins.ctypes.data_as(C.POINTER(C.c_float)), outs.ctypes.data_as(C.POINTER(C.c_float)))
    assert m == n
    return mask, ins[:n], outs[:n]


def _ptr_array(arrs):
    a = (C.POINTER(C.c_double) * len(arrs))()
    for i, x in enumerate(arrs):
        a[i] = _dp(x)
    return a


def _ar(allreduce):
    return ALLREDUCE_FN(allreduce) if allreduce else C.cast(None, ALLREDUCE_FN)


def sponge_layer(p, f, dt, time_scale=60.0, allreduce=None):
    """modules::sponge_layer on Fields f (density_dry, uvel, vvel, wvel, temp, tracers...)."""
    arrs = [f.rho_d, f.uvel, f.vvel, f.wvel, f.temp] + list(f.tracers)
    cb = _ar(allreduce)
    lib().mwo_sponge_layer(C.byref(p), _ptr_array(arrs), len(arrs), float(dt), float(time_scale), cb, None)


class ColumnNudger:
    """modules::ColumnNudger (column_nudging.h): set_column / nudge_to_column on (density_dry, uvel, vvel, temp, water_vapor)."""

    def _state(self, f, idWV=0):
        return [f.rho_d, f.uvel, f.vvel, f.temp, f.tracers[idWV]]

    def set_column(self, p, f, allreduce=None):
        self.column = np.zeros((5, p.nz, p.nens))
        cb = _ar(allreduce)
        lib().mwo_column_average(C.byref(p), _ptr_array(self._state(f, p.idWV)), _dp(self.column), cb, None)

    def nudge_to_column(self, p, f, dt, allreduce=None):
        cb = _ar(allreduce)
        lib().mwo_nudge_to_column(C.byref(p), _ptr_array(self._state(f, p.idWV)), _dp(self.column), float(dt), cb, None)


class HorizontalSponge:
    """custom_modules::Horizontal_Sponge (simple_city): column = cell (k,0,0,iens) of the main rank; cosine relaxation strips."""

    @staticmethod
    def _six(f, idWV=0):
        return [f.rho_d, f.uvel, f.vvel, f.wvel, f.temp, f.tracers[idWV]]

    def init(self, p, f, sponge_cells=10, time_scale=1.0):
        self.column = np.stack([a[:, 0, 0, :].copy() for a in self._six(f, p.idWV)])      # (6, nz, nens)
        self.sponge_cells, self.time_scale = int(sponge_cells), float(time_scale)

    def apply(self, p, f, dt, x1=True, x2=True, y1=True, y2=True):
        lib().mwo_horizontal_sponge_apply(C.byref(p), _ptr_array(self._six(f, p.idWV)), _dp(self.column), self.sponge_cells,
                                          self.time_scale, float(dt), int(x1), int(x2), int(y1), int(y2))


class TimeAverager:
    """custom_modules::Time_Averager (simple_city): running time mean of the six fields."""

    def init(self, p, f):
        self.avg = [np.zeros_like(a) for a in HorizontalSponge._six(f, p.idWV)]
        self.etime = 0.0

    def accumulate(self, p, f, dt):
        lib().mwo_time_average_accumulate(C.byref(p), _ptr_array(HorizontalSponge._six(f, p.idWV)), _ptr_array(self.avg), self.etime, float(dt))
        self.etime += dt


def city_building_heights(p):
    nby, nbx = C.c_int(), C.c_int()
    lib().mwo_city_dims(C.byref(p), nby, nbx)
    h = np.zeros((max(1, nby.value), max(1, nbx.value)))
    lib().mwo_city_building_heights(nby.value, nbx.value, _dp(h))
    return h


def supercell_setup(nx_glob, ny_glob, nz, nens=1, xlen=1.0e5, ylen=1.0e5, zlen=2.0e4, init_data="supercell",
                    num_tracers=3, perturb=True, enable_gravity=True, nranks=1, rank=0):
    """micro.init -> dycore.init -> perturb_temperature, as experiments/supercell_example/driver.cpp:58-61
    (column nudger excluded: out of scope)."""
    p, neigh = make_params(nx_glob, ny_glob, nz, nens, xlen, ylen, zlen, num_tracers=num_tracers,
                           enable_gravity=enable_gravity, nranks=nranks, rank=rank)
    dyc = OracleDycore(p)
    f = Fields(dyc.p)
    dyc.init(init_data, f)
    if perturb:
        perturb_temperature(dyc.p, f.temp)
    return dyc, f

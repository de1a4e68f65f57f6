"""ctypes binding of libmw_cdna4.so -- the C ABI declared in include/mw_cdna4.h.

This is the only way Python reaches the hot path.  If the shared library is missing or cannot be loaded
the import of the symbols FAILS LOUDLY (no CPU fallback, no oracle).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MW_LIB_PATH") or os.path.join(_HERE, "libmw_cdna4.so")      # (MW_LIB_PATH: compiler-flag experiments)

MW_MAX_TRACERS = 16
DATA_THERMAL, DATA_SUPERCELL, DATA_CITY, DATA_BUILDING = 0, 1, 2, 3
BC_PERIODIC, BC_OPEN, BC_WALL = 0, 1, 2
INIT_IDS = {"thermal": DATA_THERMAL, "supercell": DATA_SUPERCELL, "city": DATA_CITY, "building": DATA_BUILDING}


class Grid(C.Structure):
    """mw_grid_t"""
    _fields_ = [
        ("nz", C.c_int), ("ny", C.c_int), ("nx", C.c_int), ("nens", C.c_int), ("num_tracers", C.c_int),
        ("nx_glob", C.c_longlong), ("ny_glob", C.c_longlong), ("i_beg", C.c_longlong), ("j_beg", C.c_longlong),
        ("xlen", C.c_double), ("ylen", C.c_double), ("zlen", C.c_double),
        ("px", C.c_int), ("py", C.c_int), ("nproc_x", C.c_int), ("nproc_y", C.c_int),
        ("neigh", C.c_int * 9),
        ("bc_x", C.c_int), ("bc_y", C.c_int), ("bc_z", C.c_int),
        ("use_immersed", C.c_int), ("enable_gravity", C.c_int),
        ("idWV", C.c_int),
        ("R_d", C.c_double), ("R_v", C.c_double), ("cp_d", C.c_double), ("cp_v", C.c_double), ("p0", C.c_double),
        ("grav", C.c_double), ("gamma_d", C.c_double), ("kappa_d", C.c_double), ("C0", C.c_double),
        ("earthrot", C.c_double), ("latitude", C.c_double),
    ]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                          C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_void_p)

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p)

# every symbol include/mw_cdna4.h declares (checked by tests/test_capi_symbols.py against the header text)
SYMBOLS = {
    "mw_last_error": (C.c_char_p, []),
    "mw_device_count": (C.c_int, []),
    "mw_decompose": (C.c_int, [C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.POINTER(Grid)]),
    "mw_default_constants": (C.c_int, [C.POINTER(Grid)]),
    "mw_dycore_compute_time_step": (C.c_double, [C.POINTER(Grid)]),
    "mw_dycore_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(Grid), C.c_char_p, C.c_char_p, C.c_void_p]),
    "mw_dycore_destroy": (None, [C.c_void_p]),
    "mw_dycore_init": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.POINTER(C.c_void_p)]),
    "mw_dycore_set_background": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_double)] * 4 + [C.c_void_p]),
    "mw_dycore_get_background": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_double)] * 4),
    "mw_dycore_immersed_proportion": (C.c_void_p, [C.c_void_p]),
    "mw_dycore_get_grid": (C.c_int, [C.c_void_p, C.POINTER(Grid)]),
    "mw_dycore_set_bc": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mw_dycore_set_strict": (C.c_int, [C.c_void_p, C.c_int]),
    "mw_dycore_set_order": (C.c_int, [C.c_void_p, C.c_int]),
    "mw_dycore_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_longlong]),
    "mw_dycore_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong)]),
    "mw_build_flags": (C.c_int, []),
    "mw_calib_fma64": (C.c_int, [C.c_int, C.c_double, C.POINTER(C.c_double), C.c_void_p]),
    "mw_calib_stage_arith_threads": (C.c_longlong, [C.c_longlong, C.c_int]),
    "mw_calib_stage_arith": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_double), C.c_void_p]),
    "mw_debug_spin": (C.c_int, [C.c_longlong, C.c_void_p]),
    "mw_debug_launched_kernels": (C.c_longlong, [C.c_char_p, C.c_longlong, C.c_int]),
    "mw_debug_zero_maps": (C.c_longlong, [C.c_void_p, C.c_void_p, C.c_longlong, C.POINTER(C.c_int)]),
    "mw_debug_zero_violations": (C.c_longlong, [C.c_void_p, C.POINTER(C.c_ulonglong)]),
    "mw_dycore_path": (C.c_char_p, [C.c_void_p]),
    "mw_dycore_use_rccl_self": (C.c_int, [C.c_void_p]),
    "mw_rccl_selftest_config": (C.c_int, [C.c_int, C.c_int]),
    "mw_dycore_time_step": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.POINTER(C.c_void_p), C.c_double]),
    "mw_dycore_compute_tendencies": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.POINTER(C.c_void_p), C.c_double,
                                                                               C.c_void_p, C.c_void_p]),
    "mw_dycore_get_fluxes": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mw_dycore_get_etime": (C.c_double, [C.c_void_p]),
    "mw_dycore_schedule": (C.c_int, [C.c_void_p]),
    "mw_dycore_rccl_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mw_dycore_rccl_allreduce_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "mw_dycore_rccl_bcast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]),
    "mw_dycore_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "mw_dycore_profile_get": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "mw_calib_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "mw_perturb_temperature": (C.c_int, [C.POINTER(Grid), C.c_void_p, C.c_void_p]),
    "mw_perturb_temperature_random": (C.c_int, [C.POINTER(Grid), C.c_void_p, C.c_void_p]),
    "mw_exchange_plan": (C.c_int, [C.POINTER(Grid)] + [C.POINTER(C.c_int)] * 4),
    "mw_dycore_set_exchange": (C.c_int, [C.c_void_p, EXCHANGE_FN, C.c_void_p]),
    "mw_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "mw_dycore_use_rccl": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    "mw_rccl_selftest": (C.c_int, [C.c_longlong, C.c_void_p]),
    "mw_rccl_selftest_lanes": (C.c_int, []),
    "mw_strict_pow": (C.c_int, [C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mw_rccl_library_path": (C.c_char_p, [C.POINTER(C.c_int)]),
    "mw_weno5_edges": (C.c_int, [C.c_longlong, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mw_kessler_workspace_bytes": (C.c_longlong, [C.c_int, C.c_longlong]),
    "mw_kessler_set_strict": (C.c_int, [C.c_int]),
    "mw_mlp_set_strict": (C.c_int, [C.c_int]),
    "mw_column_set_strict": (C.c_int, [C.c_int]),
    "mw_kessler_time_step": (C.c_int, [C.c_int, C.c_longlong, C.c_double, C.c_double] + [C.c_void_p] * 7 +
                             [C.POINTER(C.c_int), C.c_void_p]),
    "mw_h5_read_f32": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_float), C.c_longlong, C.POINTER(C.c_longlong),
                                 C.POINTER(C.c_int)]),
    "mw_mean_diff": (C.c_int, [C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_void_p]),
    "mw_kessler_math_probe": (C.c_int, [C.c_longlong, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mw_column_workspace_bytes": (C.c_longlong, [C.POINTER(Grid), C.c_int]),
    "mw_sponge_layer": (C.c_int, [C.POINTER(Grid), C.POINTER(C.c_void_p), C.c_int, C.c_double, C.c_double, C.c_void_p, ALLREDUCE_FN,
                                  C.c_void_p, C.c_void_p]),
    "mw_column_average": (C.c_int, [C.POINTER(Grid), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, ALLREDUCE_FN, C.c_void_p, C.c_void_p]),
    "mw_nudge_to_column": (C.c_int, [C.POINTER(Grid), C.POINTER(C.c_void_p), C.c_void_p, C.c_double, C.c_void_p, ALLREDUCE_FN,
                                     C.c_void_p, C.c_void_p]),
    "mw_nudge_to_column_deferred": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_double, C.c_void_p, ALLREDUCE_FN, C.c_void_p]),
    "mw_dycore_flush_pending": (C.c_int, [C.c_void_p]),
    "mw_dycore_pending": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong)]),
    "mw_nc_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.c_longlong, C.c_longlong]),
    "mw_nc_def_dim": (C.c_int, [C.c_void_p, C.c_char_p, C.c_longlong, C.POINTER(C.c_int)]),
    "mw_nc_def_var": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mw_nc_def_var_typed": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mw_nc_put_vara": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_void_p]),
    "mw_nc_enddef": (C.c_int, [C.c_void_p]),
    "mw_nc_open": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p]),
    "mw_nc_inq_varid": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "mw_nc_inq_dimlen": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong)]),
    "mw_nc_put_vara_double": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_void_p]),
    "mw_nc_set_numrecs": (C.c_int, [C.c_void_p, C.c_longlong]),
    "mw_nc_close": (C.c_int, [C.c_void_p]),
    "mw_output_put_field": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.POINTER(Grid), C.c_void_p, C.c_void_p]),
    "mw_horizontal_sponge_column": (C.c_int, [C.POINTER(Grid), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    "mw_horizontal_sponge_apply": (C.c_int, [C.POINTER(Grid), C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_double, C.c_double] +
                                   [C.c_int] * 4 + [C.c_void_p]),
    "mw_time_average_accumulate": (C.c_int, [C.POINTER(Grid), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_double, C.c_double,
                                             C.c_void_p]),
    "mw_micro_active_count": (C.c_int, [C.POINTER(Grid), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_longlong),
                                        C.c_void_p]),
    "mw_micro_sample_mask": (C.c_int, [C.POINTER(Grid), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_ulonglong, C.c_double, C.c_double,
                                       C.c_void_p, C.c_void_p]),
    "mw_micro_gather_samples": (C.c_int, [C.POINTER(Grid), C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_longlong,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "mw_ponni_forward": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mw_validate_f64": (C.c_int, [C.c_void_p, C.c_longlong, C.POINTER(C.c_longlong), C.c_void_p]),
    "mw_validate_f32": (C.c_int, [C.c_void_p, C.c_longlong, C.POINTER(C.c_longlong), C.c_void_p]),
    "mw_mlp_forward": (C.c_int, [C.c_longlong] + [C.c_void_p] * 5 + [C.POINTER(C.c_float)] * 4 +
                       [C.POINTER(C.c_double)] * 2 + [C.c_void_p] * 4 + [C.c_void_p]),
}

_lib = None


class MWError(RuntimeError):
    pass


def lib():
    """Load libmw_cdna4.so (built by miniweatherml_amd.build).  Raises if it is absent: no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own ROCm runtime (libamdhip64.so.7 / libhsa-runtime64 / librccl).  It must be loaded
    # BEFORE libmw_cdna4.so so that both share ONE HIP/HSA runtime in the process (loading /opt/rocm's copy first
    # leaves torch with "No HIP GPUs are available").  Pure C/C++ hosts simply use /opt/rocm's runtime.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise MWError("libmw_cdna4.so not found at %s -- run `python -m miniweatherml_amd.build` "
                      "(there is no CPU fallback for the hot path)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise MWError(lib().mw_last_error().decode("utf-8", "replace"))

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import mw_oracle
    mw_oracle.lib()
    return mw_oracle


@pytest.fixture(scope="session")
def mw():
    """The product package with the HIP library loaded.  Fails loudly if the .so is missing."""
    import miniweatherml_amd
    from miniweatherml_amd import capi
    capi.lib()
    return miniweatherml_amd


@pytest.fixture(autouse=True)
def _fresh_path_log():
    """Every test starts with an empty dispatcher-path log and launch registry, so that a comparison is only credited with the time steps
    of its own test (tests/util.py: compare_fields drains both)."""
    capi = sys.modules.get("miniweatherml_amd.capi")
    if capi is not None and getattr(capi, "_lib", None) is not None:
        from util import drain_paths, launched_kernels
        drain_paths()
        launched_kernels(reset=True)
    yield


def pytest_sessionstart(session):
    from util import PARITY_LOG
    try:
        os.remove(PARITY_LOG)
    except OSError:
        pass


def pytest_sessionfinish(session, exitstatus):
    """Condenses the observed parity errors of this session (tests/util.py: compare_fields) into one JSON next to the log:
    per comparison the worst relative error over the fields and whether the sensitivity fallback was needed."""
    import json
    from util import PARITY_LOG
    if not os.path.exists(PARITY_LOG):
        return
    rows = [json.loads(ln) for ln in open(PARITY_LOG)]
    failed = [r for r in rows if not r["passed"]]          # comparisons that raised (negative controls expect to)
    all_passed = [r for r in rows if r["passed"]]
    rows = [r for r in all_passed if r["fields"]]          # (record_comparison rows carry no per-field numbers: coverage only)
    out = {"comparisons": len(rows), "failed_comparisons (negative controls included)": sorted({r["what"] for r in failed}),
           "needed_fallback": sorted({r["what"] for r in rows if any(f["needed_fallback"] for f in r["fields"].values())}),
           "worst_rel_by_tolerance": {}, "cases": []}
    out["worst_fraction_of_limit_by_tolerance"] = {}
    # comparisons held to bit-equality (the strict kernel path, "mode 1": tolerance 0) that passed
    out["bitwise_equal_to_the_oracle"] = sorted({r["what"] for r in rows if r.get("bitwise")})
    out["note"] = ("rel = max|diff| / max|field|.  The limit of a field is tol * max|field| plus an absolute floor for fields that are "
                   "identically ~0 in the oracle (tests/util.py: ABS_FLOOR, e.g. vvel in a y-symmetric run), so a large `rel` of such a "
                   "field is not a violation; worst_fraction_of_limit = max|diff| / limit is the number to read (<= 1 passes).  Cases in "
                   "needed_fallback exceeded the plain limit and passed on the allow-listed 10 x oracle-sensitivity fallback.")
    for r in rows:
        worst_field = max(r["fields"], key=lambda k: r["fields"][k]["rel"])
        wr = r["fields"][worst_field]["rel"]
        key = "%g" % r["tol"]
        if not any(f["needed_fallback"] for f in r["fields"].values()):
            # only fields whose limit is the relative one (no absolute floor in play) enter the relative statistic
            rels = [f["rel"] for f in r["fields"].values() if f["scale"] > 0 and f["limit_abs"] <= 1.0000001 * r["tol"] * f["scale"]]
            if rels:
                out["worst_rel_by_tolerance"][key] = max(out["worst_rel_by_tolerance"].get(key, 0.0), max(rels))
            frac = max(((f["max_abs_diff"] / f["limit_abs"]) for f in r["fields"].values() if f["limit_abs"] > 0), default=0.0)
            out["worst_fraction_of_limit_by_tolerance"][key] = max(out["worst_fraction_of_limit_by_tolerance"].get(key, 0.0), frac)
        out["cases"].append({"what": r["what"], "tol": r["tol"], "passed": r["passed"], "worst_field": worst_field, "worst_rel": wr,
                             "fallback_allowed": r["fallback_allowed"],
                             "needed_fallback": sorted(k for k, f in r["fields"].items() if f["needed_fallback"]),
                             "rel": {k: f["rel"] for k, f in r["fields"].items()}})
    cov = _coverage(session, all_passed)
    if cov is not None:
        out["coverage"] = cov
    json.dump(out, open(os.path.join(os.path.dirname(PARITY_LOG), "parity_summary.json"), "w"), indent=1)
    if cov is not None and cov["enforced"] and (cov["missing_paths"] or cov["unknown_paths"] or cov["kernels_never_compared"]):
        print("\nCOVERAGE FAILURE (tests/conftest.py): dispatcher paths without a passed oracle comparison: %r; paths the library reported that "
              "tests/util.py:reachable_paths does not know: %r; compiled kernel instantiations no oracle comparison exercised: %r"
              % (cov["missing_paths"], cov["unknown_paths"], cov["kernels_never_compared"]))
        session.exitstatus = 1


# kernel families of the dispatcher: everything mw_dycore_init / time_step / compute_tendencies / get_fluxes / perturb_temperature can
# launch (the calibration kernels and test aids of mw_calib.h, and the other modules' kernels, are not part of this path)
_DISPATCHED = {"k_y_all", "k_y_state", "k_y_tracers", "k_xz_state", "k_tracers_fused", "k_tracer_patch", "k_xz_tracers", "k_tracer_update",
               "k_flux", "k_fct", "k_update", "k_coupler_to_state", "k_coupler_to_state_fast", "k_coupler_to_member", "k_member_to_coupler",
               "k_member_to_fused", "k_halo_xyz", "k_pack_xy", "k_unpack_xy", "k_init_cells", "k_perturb_temperature",
               "k_perturb_temperature_random", "k_zero_rows", "k_zero_merge", "k_zero_halo", "k_zero_dilate"}


def _compiled_dycore_kernels(lib_path):
    """Mangled names of every dispatcher kernel compiled into the library: the `.kd` kernel descriptors of its embedded gfx950 code
    object whose function name (the length-prefixed identifier behind _ZN2mw) is one of the dispatcher's families."""
    import re
    out = set()
    for m in re.finditer(rb"(_ZN2mw(\d+)([0-9A-Za-z_]+))\.kd\x00", open(lib_path, "rb").read()):
        n = int(m.group(2))
        if m.group(3)[:n].decode() in _DISPATCHED:
            out.add(m.group(1).decode())
    return sorted(out)


def _demangle(names):
    import shutil
    import subprocess
    if not names or not shutil.which("c++filt"):
        return list(names)
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return [ln.replace("void ", "").split("(")[0] for ln in r.stdout.split("\n") if ln] if r.returncode == 0 else list(names)


def _coverage(session, rows):
    """Path-coverage matrix of this session (round 5): which dispatcher paths (mw_dycore_path) and which compiled kernel instantiations
    (mw_debug_launched_kernels against the kernel descriptors found in the loaded library itself) produced fields that a PASSED oracle
    comparison checked.
    Enforced -- the session fails -- when the whole path matrix ran (tests/test_gpu_path_matrix.py, i.e. a full `-m gpu` run)."""
    import collections
    from util import path_string, reachable_paths
    rows = [r for r in rows if r.get("paths") or r.get("kernels")]
    if not rows:
        return None
    seen = collections.Counter(p for r in rows for p in r.get("paths", []))
    want = [path_string(c) for c in reachable_paths()]
    items = getattr(session, "items", [])
    ran = sum(1 for it in items if "test_gpu_path_matrix.py::test_path[" in it.nodeid)
    # enforced on a FULL `-m gpu` session only: the whole matrix ran and every tests/test_gpu_*.py file contributed items (a run of
    # selected files still reports its coverage, without failing on what the files left out cover)
    import glob
    all_files = {os.path.basename(f) for f in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_*.py"))}
    ran_files = {os.path.basename(it.nodeid.split("::")[0]) for it in items}
    enforced = ran == len(want) and all_files <= ran_files and session.exitstatus == 0
    base = lambda p: p                                            # noqa: E731
    cov = {"enforced": enforced, "paths_compared": dict(sorted(seen.items())), "reachable_paths": len(want),
           "missing_paths": sorted(set(want) - set(seen)), "unknown_paths": sorted(p for p in seen if base(p) not in set(want)),
           "kernels_never_compared": [], "kernels_compiled": None}
    try:
        from miniweatherml_amd import capi
        compiled = _compiled_dycore_kernels(capi.LIB_PATH)
        hit = set(k for r in rows for k in r.get("kernels", []))
        cov["kernels_compiled"] = len(compiled)
        cov["kernels_compared"] = len([n for n in compiled if n in hit])
        cov["kernels_never_compared"] = sorted(_demangle([n for n in compiled if n not in hit]))
    except Exception as e:                                        # pragma: no cover
        cov["kernel_list_error"] = repr(e)
    return cov

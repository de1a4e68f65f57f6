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
    rows = [r for r in rows if r["passed"]]
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
    json.dump(out, open(os.path.join(os.path.dirname(PARITY_LOG), "parity_summary.json"), "w"), indent=1)

"""Turns rocprofv3 CSV output (kernel stats + separate PMC passes) into the committed per-round summary under profiles/.
usage: python tools/summarize_profiles.py <tag> <stats_dir> <fetch_dir> <write_dir> <sq_dir> [cells_per_launch]"""
import collections
import csv
import glob
import json
import os
import sys

tag, stats_dir, fdir, wdir, sqdir = sys.argv[1:6]
cells = float(sys.argv[6]) if len(sys.argv) > 6 else 400 * 400 * 100.0
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(n):
    n = n.replace("void ", "").replace("mw::", "")
    return n.split("(")[0]


def pmc(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


stats = {}
for f in glob.glob(os.path.join(stats_dir, "*", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "mw::" in r["Name"]:
            stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])}
F, W, SQ = pmc(fdir), pmc(wdir), pmc(sqdir)
import hashlib
_h = hashlib.sha256()
for _f in ("mw_march.h", "mw_weno.h", "mw_weno79.h", "mw_common.h", "mw_calib.h", "mw_dycore.hip"):      # = bench.py KERNEL_SOURCES
    _h.update(open(os.path.join(root, "miniweatherml_amd", "csrc", _f), "rb").read())
out = {"tag": tag, "cells_per_launch": cells, "kernel_sources_sha16": _h.hexdigest()[:16],
       "note": "FETCH_SIZE x2 (calibrated with mw_calib_copy: 8 B/lane streaming reads report exactly 1/2 on gfx950), WRITE_SIZE x1; KiB -> bytes",
       "kernels": {}}
for k, s in sorted(stats.items(), key=lambda kv: -kv[1]["pct"]):
    e = dict(s)
    if k in F and k in W:
        e["hbm_read_bytes"] = 2.0 * F[k]["FETCH_SIZE"] * 1024
        e["hbm_write_bytes"] = W[k]["WRITE_SIZE"] * 1024
        e["hbm_GBps"] = (e["hbm_read_bytes"] + e["hbm_write_bytes"]) / (s["avg_us"] * 1e-6) / 1e9
    if k in SQ and "SQ_INSTS_VALU" in SQ[k]:
        m = SQ[k]
        e["valu_instr_per_cell"] = m["SQ_INSTS_VALU"] * 64 / cells
        if "GRBM_GUI_ACTIVE" in m:
            cyc = m["GRBM_GUI_ACTIVE"] / 8
            e["valu_busy_frac"] = m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc
            e["clock_GHz_under_profile"] = cyc / (s["avg_us"] * 1e-6) / 1e9
        e["waves"] = m.get("SQ_WAVES")
    out["kernels"][k] = e
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", "%s_summary.json" % tag), "w"), indent=1)
print("%-28s %6s %9s %6s %9s %9s %8s %9s %6s" % ("kernel", "calls", "avg_us", "pct", "rd_GB", "wr_GB", "GB/s", "instr/cell", "VALU%"))
for k, e in out["kernels"].items():
    print("%-28s %6d %9.1f %6.2f %9.3f %9.3f %8.0f %9.0f %6.0f" % (k[:28], e["calls"], e["avg_us"], e["pct"], e.get("hbm_read_bytes", 0) / 1e9,
          e.get("hbm_write_bytes", 0) / 1e9, e.get("hbm_GBps", 0), e.get("valu_instr_per_cell", 0), 100 * e.get("valu_busy_frac", 0)))

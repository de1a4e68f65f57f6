#!/bin/bash
# rocprofv3 evidence for the Kessler and MLP kernels (through gpurun): kernel stats, then separate PMC passes (never combined with
# tracing).  usage: bash tools/profile_micro.sh <tag>   -> gpurun_out/micro_<tag>_summary.json (copy to profiles/)
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for w in kessler_scattered kessler_storm mlp; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pm_${tag}_${w}_stats -- python3 $R/tools/micro_driver.py $w --iters 20 > /dev/null 2> $R/gpurun_out/pm_${tag}_${w}.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pm_${tag}_${w}_f -- python3 $R/tools/micro_driver.py $w --iters 3 > /dev/null 2>> $R/gpurun_out/pm_${tag}_${w}.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pm_${tag}_${w}_w -- python3 $R/tools/micro_driver.py $w --iters 3 > /dev/null 2>> $R/gpurun_out/pm_${tag}_${w}.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pm_${tag}_${w}_sq -- python3 $R/tools/micro_driver.py $w --iters 3 > /dev/null 2>> $R/gpurun_out/pm_${tag}_${w}.err
done
cd $R
python3 - "$tag" <<'PY'
import collections, csv, glob, json, sys
tag = sys.argv[1]
cells = 400 * 400 * 100.0
def short(n): return n.replace("void ", "").replace("mw::", "").split("(")[0]
def pmc(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
out = {"tag": tag, "cells": cells, "note": "hbm bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes; FETCH_SIZE reports 1/2 for 8 B/lane streams on gfx950, "
       "calibrated with mw_calib_copy); algorithmic 72 B per cell; per workload: kernels of libmw_cdna4 only", "workloads": {}}
for w in ("kessler_scattered", "kessler_storm", "mlp"):
    base = "gpurun_out/pm_%s_%s" % (tag, w)
    stats = {}
    for f in glob.glob(base + "_stats/*/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if "mw::" in r["Name"] and ("kessler" in r["Name"] or "mlp" in r["Name"]):
                stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
    F, W, SQ = pmc(base + "_f"), pmc(base + "_w"), pmc(base + "_sq")
    ks, tot_us, tot_b = {}, 0.0, 0.0
    for k, s in stats.items():
        e = dict(s)
        if k in F and k in W:
            e["hbm_read_bytes"] = 2.0 * F[k]["FETCH_SIZE"] * 1024; e["hbm_write_bytes"] = W[k]["WRITE_SIZE"] * 1024
            tot_b += e["hbm_read_bytes"] + e["hbm_write_bytes"]
        if k in SQ:
            m = SQ[k]
            e["valu_instr_per_cell"] = m.get("SQ_INSTS_VALU", 0) * 64 / cells
            if m.get("GRBM_GUI_ACTIVE"):
                e["valu_busy_frac"] = m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (m["GRBM_GUI_ACTIVE"] / 8)
                if "SQ_VALU_MFMA_BUSY_CYCLES" in m: e["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (m["GRBM_GUI_ACTIVE"] / 8)
            if "SQ_INSTS_VALU_MFMA_MOPS_F32" in m: e["mfma_mops_f32"] = m["SQ_INSTS_VALU_MFMA_MOPS_F32"]
        ks[k] = e
        if s["avg_us"] > 2: tot_us += s["avg_us"]
    out["workloads"][w] = {"kernels": ks, "us_per_call_all_kernels": tot_us, "alg_GBps": cells * 72 / (tot_us * 1e-6) / 1e9 if tot_us else None,
                           "frac_of_8TBps": cells * 72 / (tot_us * 1e-6) / 8e12 if tot_us else None, "hbm_bytes_per_call": tot_b}
json.dump(out, open("gpurun_out/micro_%s_summary.json" % tag, "w"), indent=1)
for w, v in out["workloads"].items():
    print(w, "us/call %.1f  frac %.3f  traffic %.2f GB" % (v["us_per_call_all_kernels"], v["frac_of_8TBps"] or 0, v["hbm_bytes_per_call"] / 1e9))
    for k, e in v["kernels"].items():
        print("   %-28s calls %4d avg_us %8.1f instr/cell %7.1f busy %.2f rd %.2f wr %.2f GB" % (k, e["calls"], e["avg_us"], e.get("valu_instr_per_cell", 0), e.get("valu_busy_frac", 0), e.get("hbm_read_bytes", 0) / 1e9, e.get("hbm_write_bytes", 0) / 1e9))
PY
find gpurun_out/pm_${tag}_* -name "*counter_collection.csv" -size +4M -delete

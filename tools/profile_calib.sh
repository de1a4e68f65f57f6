#!/bin/bash
# rocprofv3 evidence for the calibration kernels (round 5): kernel stats of tools/calib.py, then one PMC pass (GRBM_GUI_ACTIVE + VALU
# counters: the clock the FMA kernel really ran at, its VALU busy fraction, the floor kernel's instruction count).  Through gpurun:
#   bash tools/profile_calib.sh <tag>   ->  gpurun_out/calib_<tag>_{stats,pmc,*.json}
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/calib_${tag}_stats -- python3 $R/tools/calib.py 1.0 > $R/gpurun_out/calib_${tag}.json 2> $R/gpurun_out/calib_${tag}.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $R/gpurun_out/calib_${tag}_pmc -- python3 $R/tools/calib.py 0.3 > /dev/null 2>> $R/gpurun_out/calib_${tag}.err
cd $R
python3 - "$tag" <<'PY'
import collections, csv, glob, json, sys
tag = sys.argv[1]
def short(n): return n.replace("void ", "").replace("mw::", "").split("(")[0]
stats = {}
for f in glob.glob("gpurun_out/calib_%s_stats/*/*kernel_stats.csv" % tag):
    for r in csv.DictReader(open(f)):
        if "k_calib" in r["Name"]:
            stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/calib_%s_pmc/*/*counter_collection.csv" % tag):
    for r in csv.DictReader(open(f)):
        if "k_calib" in r["Kernel_Name"]:
            key = (short(r["Kernel_Name"]), r["Dispatch_Id"])
            rows[key][r["Counter_Name"]] = float(r["Counter_Value"])
            rows[key]["us"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3
            rows[key]["grid"] = float(r["Grid_Size"])
pm = []
for (k, d), m in sorted(rows.items(), key=lambda kv: int(kv[0][1])):
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    e = {"kernel": k, "dispatch": int(d), "us": m["us"], "threads": m["grid"], "waves": m.get("SQ_WAVES"), "valu_wave_instr": m.get("SQ_INSTS_VALU")}
    if cyc and m["us"] > 0:
        e["clock_GHz"] = cyc / (m["us"] * 1e-6) / 1e9
        e["valu_busy_frac"] = m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / cyc
        e["valu_wave_instr_per_s"] = m.get("SQ_INSTS_VALU", 0) / (m["us"] * 1e-6)
    if k == "k_calib_stage_arith" and m.get("SQ_INSTS_VALU"):
        e["valu_instr_per_cell"] = m["SQ_INSTS_VALU"] * 64 / 1.6e7
    pm.append(e)
out = {"tag": tag, "kernel_stats": stats, "pmc_per_dispatch": pm, "tool_output": json.load(open("gpurun_out/calib_%s.json" % tag)),
       "note": "clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; valu_busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / cycles; k_calib_fma64 dispatches: warm-up x2 then the "
               "long run, at 1 / 2 / 4 / 8 wavefronts per SIMD; k_calib_stage_arith: smooth and rough tables at 25 and 100 levels per thread, two launches each"}
json.dump(out, open("gpurun_out/calib_%s_summary.json" % tag, "w"), indent=1)
for e in pm:
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items()})
PY
find gpurun_out/calib_${tag}_pmc gpurun_out/calib_${tag}_stats -name "*.csv" -size +8M -delete

"""tools/profile_storm_vs_initial.sh's CSVs -> one JSON: per kernel and state the average duration, shader clock, cycles, L2 hit rate, fabric
requests and HBM bytes, the storm / initial ratios, and the un-profiled power / clock samples.  python tools/summarize_storm_vs_initial.py <dir>"""
import collections, csv, glob, json, os, sys
D = sys.argv[1]


def short(n):
    return n.replace("void ", "").replace("mw::", "").split("(")[0]


def load(mode, name):
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(D, "%s_%s" % (mode, name), "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "mw::" not in r["Kernel_Name"]:
                continue
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            try:
                dur[k][r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3
            except (KeyError, ValueError):
                pass
    return ({k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}, {k: sum(d.values()) / len(d) for k, d in dur.items() if d})


def stats(mode):
    out = {}
    for f in glob.glob(os.path.join(D, "%s_stats" % mode, "*", "*kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            if "mw::" in r["Name"]:
                out[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) * 1e-3}
    return out


res = {"what": "six dycore steps (400 x 400 x 100) on the cloud-free initial state ('initial') and on the developed storm after 2600 steps of the complete loop "
               "('storm'), one process each; rocprofv3 --kernel-trace --stats for the un-countered durations, separate --pmc passes for the counters "
               "(clock_GHz = GRBM_GUI_ACTIVE / 8 / the pass's own duration)", "kernels": {}}
st = {"initial": stats("initial"), "storm": stats("run")}
per = {}
for label, mode in (("initial", "initial"), ("storm", "run")):
    clk, clk_us = load(mode, "clk"); l2, _ = load(mode, "l2"); ea, _ = load(mode, "ea"); es, _ = load(mode, "eastall"); rd, _ = load(mode, "rd"); wr, _ = load(mode, "wr")
    for k in clk:
        e = per.setdefault(k, {}).setdefault(label, {})
        c = clk[k]
        e["avg_us_stats_pass"] = st[label].get(k, {}).get("avg_us"); e["calls"] = st[label].get(k, {}).get("calls")
        e["avg_us_counter_pass"] = clk_us.get(k)
        if c.get("GRBM_GUI_ACTIVE") and clk_us.get(k):
            e["clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / (clk_us[k] * 1e-6) / 1e9
        for n in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVES", "GRBM_GUI_ACTIVE"):
            if n in c:
                e[n] = c[n]
        if c.get("GRBM_GUI_ACTIVE") and "SQ_ACTIVE_INST_VALU" in c:
            e["valu_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (c["GRBM_GUI_ACTIVE"] / 8)
        if k in l2 and "TCC_HIT_sum" in l2[k]:
            h, m = l2[k]["TCC_HIT_sum"], l2[k].get("TCC_MISS_sum", 0.0)
            e["TCC_HIT_sum"], e["TCC_MISS_sum"], e["l2_hit_rate"] = h, m, (h / (h + m) if h + m else None)
        for src in (ea, es):
            for n, v in src.get(k, {}).items():
                e[n] = v
        if k in rd and "FETCH_SIZE" in rd[k]:
            e["hbm_read_GB"] = 2.0 * rd[k]["FETCH_SIZE"] * 1024 / 1e9
        if k in wr and "WRITE_SIZE" in wr[k]:
            e["hbm_write_GB"] = wr[k]["WRITE_SIZE"] * 1024 / 1e9
for k, e in per.items():
    if "initial" in e and "storm" in e:
        r = {}
        for n in ("avg_us_stats_pass", "avg_us_counter_pass", "clock_GHz", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "l2_hit_rate",
                  "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "hbm_read_GB", "hbm_write_GB"):
            a, b = e["initial"].get(n), e["storm"].get(n)
            if a and b:
                r[n] = b / a
        e["storm_over_initial"] = r
    if (e.get("initial", {}).get("avg_us_stats_pass") or 0) > 40:
        res["kernels"][k] = e
try:
    res["power_clock_unprofiled"] = json.loads(open(os.path.join(D, "power_clock.json")).read().strip().splitlines()[-1])
except Exception as ex:
    res["power_clock_unprofiled"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
if os.path.exists(os.path.join(D, "failed.txt")):
    res["failed_passes"] = open(os.path.join(D, "failed.txt")).read().split("\n")
print(json.dumps(res, indent=1))

#!/bin/bash
# One extra rocprofv3 counter pass over 2 bench steps: bash tools/pmc_pass.sh <tag> COUNTER [COUNTER ...]   (through gpurun)
# Prints the per-kernel average of each counter per launch; raw CSVs stay under gpurun_out/pmc_<tag>.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${tag} -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-micro --no-pmc --no-calib > /dev/null 2> $R/gpurun_out/pmc_${tag}.err
cd $R
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv" % tag):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void mw::", "").replace("mw::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
names = sorted({c for k in acc for c in acc[k]})
print("%-34s %6s " % ("kernel", "calls") + " ".join("%22s" % n for n in names))
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    n = len(cnt[k])
    print("%-34s %6d " % (k[:34], n) + " ".join("%22.4g" % (acc[k][c] / n) for c in names))
PY
find gpurun_out/pmc_${tag} -name "*counter_collection.csv" -size +8M -delete

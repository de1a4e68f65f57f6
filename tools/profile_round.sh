#!/bin/bash
# Collects the per-round rocprofv3 evidence on the GPU box: kernel stats of the default bench command, then three separate
# PMC passes (never combined with tracing).  usage (through gpurun): bash tools/profile_round.sh <tag> [bench args]
# Output: gpurun_out/prof_<tag>_{stats,f,w,sq}; summarise with tools/summarize_profiles.py.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-micro --no-pmc --no-calib $*"      # (no nested rocprofv3 children under the profiler)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${tag}_bench.json 2> $R/gpurun_out/prof_${tag}_stats.err
PARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-micro --no-pmc --no-calib $*"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_f -- python3 $R/bench.py $PARGS > /dev/null 2> $R/gpurun_out/prof_${tag}_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_w -- python3 $R/bench.py $PARGS > /dev/null 2> $R/gpurun_out/prof_${tag}_w.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $R/gpurun_out/prof_${tag}_sq -- python3 $R/bench.py $PARGS > /dev/null 2> $R/gpurun_out/prof_${tag}_sq.err
cd $R
# keep only the small CSVs (the counter_collection files are large: aggregate them here)
python3 tools/summarize_profiles.py ${MW_ROUND:-r06}_${tag} gpurun_out/prof_${tag}_stats gpurun_out/prof_${tag}_f gpurun_out/prof_${tag}_w gpurun_out/prof_${tag}_sq > gpurun_out/prof_${tag}_table.txt 2>&1
cp profiles/${MW_ROUND:-r06}_${tag}_summary.json gpurun_out/
find gpurun_out/prof_${tag}_f gpurun_out/prof_${tag}_w gpurun_out/prof_${tag}_sq -name "*counter_collection.csv" -size +8M -delete
tail -25 gpurun_out/prof_${tag}_table.txt

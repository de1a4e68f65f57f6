#!/bin/bash
# Round 6, VERDICT item 3: WHY does every kernel run the developed storm 12-27 % slower than the initial state at the same instructions and bytes?
# Counters, not prose: per kernel and state the shader clock (GRBM_GUI_ACTIVE / 8 / duration), wave / busy / wait cycles, L2 hit rate, fabric
# requests; plus un-profiled power / clock samples of steady loops on both states (tools/power_clock_sample.py).
#   bash tools/profile_storm_vs_initial.sh <tag>      (through gpurun; ~6 minutes)
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/svi_${tag}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python3 $R/tools/storm_state.py save --steps 2600 --file /tmp/storm.pt > $OUT/save.log 2>&1
rocprofv3 -L > $OUT/counters_available.txt 2>&1
pass() {   # pass <state-mode> <name> COUNTER...
  mode=$1; name=$2; shift 2
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/${mode}_${name} -- python3 $R/tools/storm_state.py $mode --n 6 --file /tmp/storm.pt > /dev/null 2> $OUT/${mode}_${name}.err || echo "pass ${mode}_${name} failed" >> $OUT/failed.txt
}
for mode in initial run; do
  pass $mode clk GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES
  pass $mode l2 TCC_HIT_sum TCC_MISS_sum
  pass $mode ea TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
  pass $mode eastall TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum
  pass $mode rd FETCH_SIZE
  pass $mode wr WRITE_SIZE
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${mode}_stats -- python3 $R/tools/storm_state.py $mode --n 6 --file /tmp/storm.pt > /dev/null 2> $OUT/${mode}_stats.err
done
python3 $R/tools/power_clock_sample.py --file /tmp/storm.pt --steps 300 > $OUT/power_clock.json 2> $OUT/power_clock.err
cd $R
python3 tools/summarize_storm_vs_initial.py $OUT > $OUT/summary.json
find $OUT -name "*.csv" -size +6M -delete
find $OUT -name "*.db" -delete

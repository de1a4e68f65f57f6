#!/bin/bash
# Per-kernel profile of the dycore step ON A GIVEN STATE (the headline profile, tools/profile_round.sh, is the cloud-free initial state):
# rocprofv3 kernel stats, then separate --pmc passes (never combined with tracing), summarised into profiles/r06_<state>_summary.json + table.
#   bash tools/profile_state.sh developed            bench.py's seeded stress state (cloud / rain rims everywhere)
#   bash tools/profile_state.sh storm [steps]        the complete supercell loop spun up for 2600 iterations (725 s) first
#   bash tools/profile_state.sh mature               ... for 12900 iterations (3600 s)
state=${1:-developed}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/state_${state}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
case $state in
  developed) MODE="developed" ;;
  storm)     python3 $R/tools/storm_state.py save --steps ${2:-2600} --file /tmp/state.pt > $OUT/save.log 2>&1; MODE="run --file /tmp/state.pt" ;;
  mature)    python3 $R/tools/storm_state.py save --steps ${2:-12900} --file /tmp/state.pt > $OUT/save.log 2>&1; MODE="run --file /tmp/state.pt" ;;
  *) echo "unknown state $state"; exit 2 ;;
esac
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/storm_state.py $MODE --n 8 > /dev/null 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 $R/tools/storm_state.py $MODE --n 3 > /dev/null 2> $OUT/f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 $R/tools/storm_state.py $MODE --n 3 > /dev/null 2> $OUT/w.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT/sq -- python3 $R/tools/storm_state.py $MODE --n 3 > /dev/null 2> $OUT/sq.err
cd $R
python3 tools/summarize_profiles.py r06_${state} $OUT/stats $OUT/f $OUT/w $OUT/sq > $OUT/table.txt 2>&1
cp profiles/r06_${state}_summary.json $OUT/ 2>/dev/null
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv
find $OUT -name "*counter_collection.csv" -size +8M -delete; find $OUT -name "*.db" -delete
tail -30 $OUT/table.txt

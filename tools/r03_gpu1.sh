#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dev_stats -- python3 $R/tools/developed_only.py 6 > /dev/null 2> $R/gpurun_out/prof_dev_stats.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $R/gpurun_out/prof_dev_sq -- python3 $R/tools/developed_only.py 2 > /dev/null 2> $R/gpurun_out/prof_dev_sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_dev_f -- python3 $R/tools/developed_only.py 2 > /dev/null 2> $R/gpurun_out/prof_dev_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_dev_w -- python3 $R/tools/developed_only.py 2 > /dev/null 2> $R/gpurun_out/prof_dev_w.err
cd $R
python3 tools/summarize_profiles.py r03_developed gpurun_out/prof_dev_stats gpurun_out/prof_dev_f gpurun_out/prof_dev_w gpurun_out/prof_dev_sq 2>&1 | tail -16
cp profiles/r03_developed_summary.json gpurun_out/
find gpurun_out/prof_dev_f gpurun_out/prof_dev_w gpurun_out/prof_dev_sq -name "*counter_collection.csv" -size +8M -delete

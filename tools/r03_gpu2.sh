#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r03_full_gpu_tests.txt 2>&1
tail -4 gpurun_out/r03_full_gpu_tests.txt
bash tools/profile_round.sh ord3 --ord 3 > gpurun_out/r03_prof_ord3.txt 2>&1
tail -16 gpurun_out/r03_prof_ord3.txt
python bench.py --ord 3 > gpurun_out/bench_r03_ord3.json 2> gpurun_out/bench_r03_ord3.err
cut -c1-300 gpurun_out/bench_r03_ord3.json

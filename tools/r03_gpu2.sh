#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r03_full_gpu_tests.txt 2>&1
tail -5 gpurun_out/r03_full_gpu_tests.txt
bash tools/profile_round.sh v1 > gpurun_out/r03_prof_v1.txt 2>&1
tail -30 gpurun_out/r03_prof_v1.txt
python bench.py > gpurun_out/bench_r03_v1.json 2> gpurun_out/bench_r03_v1.err
cat gpurun_out/bench_r03_v1.json | cut -c1-600

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r03_t5.txt 2>&1
grep -E "passed|failed" gpurun_out/r03_t5.txt | tail -2

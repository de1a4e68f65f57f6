#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r03_t4.txt 2>&1
grep -E "^FAILED|passed|failed" gpurun_out/r03_t4.txt | head -60

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03_t2.txt 2>&1
tail -15 gpurun_out/r03_t2.txt

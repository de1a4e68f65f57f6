#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/ab_bench.sh "MW_X=1" "MW_OVERLAP=1" "MW_OVERLAP=1 MW_EARLY_YT=1" "MW_OVERLAP=1 MW_EARLY_YT=1 MW_TSTREAM_PRIO=0" > gpurun_out/r03_ab4.txt 2>&1
cat gpurun_out/r03_ab4.txt

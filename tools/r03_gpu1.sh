#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh "MW_X=1" "MW_LIB_PATH=$GRAFT_REPO_ROOT/miniweatherml_amd/variants/libmw_nt.so" > gpurun_out/r03_ab5.txt 2>&1
cat gpurun_out/r03_ab5.txt

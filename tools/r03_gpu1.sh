#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh "MW_X=1" "MW_CHUNK_Z=20" "MW_CHUNK_F=20" "MW_CHUNK_Z=17" "MW_CHUNK_F=17" "MW_CHUNK_Y=40" "MW_CHUNK_Y=45" "MW_CHUNK_YT=25" "MW_CHUNK_YT=34" 2>&1 | grep -v rep3 > gpurun_out/r03_ab8.txt
cat gpurun_out/r03_ab8.txt

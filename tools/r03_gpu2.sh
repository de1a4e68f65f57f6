#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
bash tools/profile_round.sh v2 > gpurun_out/r03_prof_v2.txt 2>&1
tail -16 gpurun_out/r03_prof_v2.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1 2 3; do
  MW_MEMBER_DIRECT=$v python bench.py --workload config4 --no-micro --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('direct=$v', '%.4g'%d['value'], '%.2f ms'%d['ms_per_step'], {k:round(x,2) for k,x in d['kernel_ms_per_step'].items() if x>0.01})"
done
MW_MEMBER_DIRECT=3 timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_dycore_parity.py tests/test_gpu_random_configs.py -x -q -m gpu 2>&1 | tail -2

#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/city_timing.py 2>&1 | grep -v amdgpu.ids | tail -2
MW_NO_SPEC=1 python tools/city_timing.py 2>&1 | grep -v amdgpu.ids | tail -2 | sed 's/^/nospec /'
for g in "200 200 50" "1024 1024 100"; do set -- $g; python bench.py --nx $1 --ny $2 --nz $3 --no-micro --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$g', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'])"; done
python bench.py --full-loop --no-micro --no-cpu-baseline 2>/dev/null > gpurun_out/bench_r03_full_loop.json; python -c "
import json; d=json.loads(open('gpurun_out/bench_r03_full_loop.json').read()); print('full loop', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'])"
python bench.py --workload config4 --no-micro --no-cpu-baseline --steps 10 2>/dev/null > gpurun_out/bench_r03_config4_block.json; python -c "
import json; d=json.loads(open('gpurun_out/bench_r03_config4_block.json').read()); print('config4 block', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'])"

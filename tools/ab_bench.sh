#!/bin/bash
# A/B timing of kernel variants in ONE gpurun call: bash tools/ab_bench.sh "MW_OPTIONS=chunk_z=20" "MW_LIB_PATH=.../libmw_x.so" ...
# (MW_OPTIONS = handle options for the Python host, modules.DEFAULT_OPTIONS; MW_LIB_PATH = another build of the library)
# Each variant: bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-micro, three repetitions, interleaved.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
  for v in "$@"; do
    out=$(env $v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-micro 2>/dev/null)
    echo "$v rep$rep $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); k=d["kernel_ms_per_step"]; print("ms/step %.3f  stage %.3f | "%(d["ms_per_step"], d["roofline"]["avg_launch_ms"]) + " ".join("%s %.3f"%(n,v) for n,v in k.items() if v>0.001))')"
  done
done

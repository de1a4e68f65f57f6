bash tools/profile_round.sh v1 > gpurun_out/r06_profile_round.log 2>&1
cp profiles/r06_v1_summary.json gpurun_out/ 2>/dev/null
python bench.py > gpurun_out/bench_r06_final.json 2> gpurun_out/bench_r06_final.err
python bench.py --full-loop --no-micro --no-cpu-baseline --no-pmc --no-calib > gpurun_out/bench_r06_full_loop.json 2>/dev/null
python bench.py --workload config3 --no-micro --no-cpu-baseline --no-pmc --no-calib > gpurun_out/bench_r06_config3.json 2>/dev/null
python bench.py --workload config4 --no-micro --no-cpu-baseline --no-calib > gpurun_out/bench_r06_config4.json 2>/dev/null
python bench.py --workload config5 --no-micro --no-cpu-baseline --no-pmc --no-calib > gpurun_out/bench_r06_config5.json 2>/dev/null
python bench.py --ord 3 --no-micro --no-cpu-baseline --no-pmc --no-calib > gpurun_out/bench_r06_ord3.json 2>/dev/null

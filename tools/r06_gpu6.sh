python -m pytest tests/test_gpu_options.py tests/test_gpu_multirank.py tests/test_gpu_full_loop.py tests/test_gpu_rccl_selfloop.py tests/test_gpu_path_matrix.py tests/test_gpu_dycore_parity.py -m gpu -q -x > gpurun_out/r06_t6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06_t6.log
python bench.py --no-selfloop --no-cpu-baseline > gpurun_out/r06_bench2.json 2> gpurun_out/r06_bench2.err

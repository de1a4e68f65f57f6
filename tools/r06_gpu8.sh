python -m pytest tests/test_gpu_dycore_parity.py tests/test_gpu_multirank.py tests/test_gpu_options.py tests/test_gpu_random_configs.py tests/test_gpu_configs.py -m gpu -q -x > gpurun_out/r06_t8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06_t8.log
python tools/oneoff/developed_classes.py > gpurun_out/r06_developed_classes2.txt 2>&1

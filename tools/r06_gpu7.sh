python -m pytest tests -m gpu -q > gpurun_out/r06_t7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06_t7.log

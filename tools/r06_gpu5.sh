python -m pytest tests/test_gpu_options.py tests/test_gpu_full_loop.py -m gpu -q -x -k "slab or deferred or defaults" > gpurun_out/r06_t5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06_t5.log
python tools/slab_ab.py > gpurun_out/r06_slab_ab.json 2> gpurun_out/r06_slab_ab.err
python tools/nudge_ab.py > gpurun_out/r06_nudge_ab.json 2> gpurun_out/r06_nudge_ab.err

#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kessler_mlp.py tests/test_glibc_pow.py tests/test_gpu_cpp_facade.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -12

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r06_gpu_tests.txt
cat gpurun_out/r06_gpu_tests.txt

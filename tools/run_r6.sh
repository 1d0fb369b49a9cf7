#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
INFL="v_prev main" KINDS="0" tools/ab_r6.sh
for v in v_prev main; do LIBDEFLATE_AMD_LIB=$PWD/libdeflate_amd/libdeflate_amd$([ $v = main ] || echo _$v).so timeout 200 python tools/microbench.py inflate --chunks 262144 --size 4096 --fmt zlib 2>&1 | grep "flate\["; done
timeout 300 python tools/bench_stream.py 1 16 64 2>&1 | grep -v amdgpu | cut -c1-60
timeout 300 python tools/bench_stream.py --mix 16 2>&1 | grep -v amdgpu | cut -c1-60

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
INFL="v_prev main" tools/ab_r6.sh
for v in v_prev main; do for k in 5 0; do LIBDEFLATE_AMD_LIB=$PWD/libdeflate_amd/libdeflate_amd$([ $v = main ] || echo _$v).so timeout 120 python tools/microbench.py inflate --chunks 256 --kind $k 2>&1 | grep "flate\["; done; done
timeout 600 python -m pytest tests/test_inflate_gpu.py tests/test_fuzz_gpu.py tests/test_stream_gpu.py -x -q -s 2>&1 | grep -E "passed|failed|slow|ms" | tail -8

#!/bin/bash
# scratch driver of one gpurun call (rewritten per call)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
INFL="main v_cb512 v_cb576 v_cb768" DEFL="main" LEVELS_OF="main" SMALL="main" tools/ab_r6.sh
for v in v_cb576 v_cb768; do
  LIBDEFLATE_AMD_LIB=$PWD/libdeflate_amd/libdeflate_amd_$v.so timeout 600 python -m pytest tests/test_inflate_gpu.py -x -q 2>&1 | tail -3
done
timeout 300 python -m pytest tests/test_deflate_gpu.py -x -q -k "recorded or ratio or roundtrip" 2>&1 | tail -3

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in main; do echo "== $v"; timeout 300 python -m pytest tests/test_inflate_gpu.py -x -q -s -k slow_decompression 2>&1 | grep -E "slow_decomp|passed|failed"; done
INFL="main" KINDS=" " tools/ab_r6.sh
timeout 600 python -m pytest tests/test_inflate_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -2

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export LIBDEFLATE_AMD_LIB=$PWD/libdeflate_amd/libdeflate_amd_prof.so
for k in "" "--kind 0" "--kind 5"; do
timeout 120 python tools/microbench.py inflate --chunks 4096 $k 2>&1 | grep -v amdgpu.ids | grep -E "sync|iterations|copy groups|slot batches|par rounds|inflate\["
done

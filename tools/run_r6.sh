#!/bin/bash
# scratch driver of one gpurun call (rewritten per call)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r6_gputests_2.txt 2>&1; tail -3 gpurun_out/r6_gputests_2.txt
DEFL="main v_ilpd" LEVELS_OF="v_ilpd" SMALL="v_ilpd" tools/ab_r6.sh
{ for k in 5 0; do for n in 256 1024; do echo "== lone streams: kind $k x $n"; LIBDEFLATE_AMD_LIB=$PWD/libdeflate_amd/libdeflate_amd_prof.so timeout 120 python tools/microbench.py inflate --chunks $n --kind $k 2>&1 | grep -v amdgpu.ids; done; done; } > gpurun_out/r6_inflate_lone_profile.txt
cat gpurun_out/r6_inflate_lone_profile.txt
for k in 5 0 6; do for n in 256 1024; do timeout 120 python tools/microbench.py inflate --chunks $n --kind $k 2>&1 | grep "flate\["; done; done

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/prof_all.sh r06 > gpurun_out/r06_prof_all.log 2>&1
mkdir -p /tmp/p && cp gpurun_out/pmc_*.json /tmp/p/ && for f in /tmp/p/pmc_*.json; do b=$(basename $f); cp $f profiles/r06_$b; done   # so that this run's bench line finds profiles of this build
timeout 900 python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_stderr.log; tail -c 300 gpurun_out/r06_bench_line.json
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_tests.txt 2>&1; tail -3 gpurun_out/r06_gpu_tests.txt
{ for k in -1 0 5 6; do LIBDEFLATE_AMD_LIB=$PWD/libdeflate_amd/libdeflate_amd_prof.so timeout 120 python tools/microbench.py inflate --chunks 4096 --kind $k 2>&1 | grep -v amdgpu.ids; done; for k in 5 0; do echo "== lone streams (one per CU): kind $k"; LIBDEFLATE_AMD_LIB=$PWD/libdeflate_amd/libdeflate_amd_prof.so timeout 120 python tools/microbench.py inflate --chunks 256 --kind $k 2>&1 | grep -v amdgpu.ids; done; } > gpurun_out/r06_inflate_phase_profile.txt
LIBDEFLATE_AMD_LIB=$PWD/libdeflate_amd/libdeflate_amd_prof.so timeout 120 python tools/microbench.py deflate --chunks 4096 --level 6 --iters 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_deflate_phase_profile.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r06_gpu_tests.txt
cat gpurun_out/r06_gpu_tests.txt
bash tools/prof_all.sh r06 > gpurun_out/prof_all.log 2>&1
for w in bench l1 l9 opt small inflate64k stream; do [ -f gpurun_out/pmc_$w.json ] && cp gpurun_out/pmc_$w.json profiles/r06_pmc_$w.json; done
timeout 1200 python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_err.txt
tail -c 600 gpurun_out/r06_bench_line.json
PC=$PWD/libdeflate_amd/libdeflate_amd_profc.so
{
for k in "" "--kind 0" "--kind 5" "--kind 6"; do LIBDEFLATE_AMD_LIB=$PC timeout 120 python tools/microbench.py inflate --chunks 4096 $k 2>&1 | grep -v amdgpu.ids; done
for k in 5 0; do LIBDEFLATE_AMD_LIB=$PC timeout 120 python tools/microbench.py inflate --chunks 256 --kind $k 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r06_inflate_phase_profile.txt
LIBDEFLATE_AMD_LIB=$PWD/libdeflate_amd/libdeflate_amd_prof.so timeout 120 python tools/microbench.py deflate --chunks 4096 --level 6 --iters 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_deflate_phase_profile.txt
tail -3 gpurun_out/r06_deflate_phase_profile.txt
{
for l in 10 11 12; do timeout 200 python tools/microbench.py deflate --chunks 4096 --level $l --iters 3 2>&1 | grep "flate\["; timeout 200 python tools/ratio_levels.py $l 2>&1 | grep "size 65536"; done
} > gpurun_out/r06_levels_10_12.txt 2>&1
cat gpurun_out/r06_levels_10_12.txt

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for a in "1 32" "4 16" "16 8" "64 6" "256 4" "1024 3"; do python tools/bench_single_api.py $a 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04_cl2.log; done
LDA_HOST_THREADS=8 python tools/bench_single_api.py 256 4 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04_cl2.log
LDA_HOST_THREADS=1 python tools/bench_single_api.py 256 4 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04_cl2.log
cat gpurun_out/r04_cl2.log

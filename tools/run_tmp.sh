cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r04_t4.log
python tools/bench_stream.py 1 16 64 2>&1 | grep -v amdgpu >> gpurun_out/r04_t4.log
for a in "1 32" "16 8" "64 6"; do python tools/bench_single_api.py $a 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04_t4.log; done
python tools/bench_host_batch.py 2>&1 | grep -v amdgpu.ids | tail -4 >> gpurun_out/r04_t4.log
cat gpurun_out/r04_t4.log

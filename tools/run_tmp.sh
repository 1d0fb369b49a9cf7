R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_threads_gpu.py tests/test_stream_gpu.py -x -q 2>&1 | tail -2 > gpurun_out/r04_w4.log
python -m pytest tests/test_deflate_gpu.py -x -q -k "large or 4gib" 2>&1 | tail -1 >> gpurun_out/r04_w4.log
python tools/bench_stream.py 1 16 64 2>&1 | grep -v amdgpu >> gpurun_out/r04_w4.log
python tools/bench_single_api.py 16 8 2>&1 | grep -v amdgpu >> gpurun_out/r04_w4.log
python tools/bench_single_api.py 256 4 2>&1 | grep -v amdgpu >> gpurun_out/r04_w4.log
python tools/bench_host_batch.py 2>&1 | grep -v amdgpu | tail -2 >> gpurun_out/r04_w4.log
LDA_HOST_THREADS=8 python tools/bench_stream.py 16 64 2>&1 | grep -v amdgpu >> gpurun_out/r04_w4.log
cat gpurun_out/r04_w4.log

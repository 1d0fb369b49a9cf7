R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_stream_gpu.py tests/test_fuzz_gpu.py tests/test_threads_gpu.py -x -q 2>&1 | tail -2 > gpurun_out/r04_w3.log
python -m pytest tests/test_deflate_gpu.py -x -q -k over_4gib 2>&1 | tail -1 >> gpurun_out/r04_w3.log
python tools/bench_stream.py 0.0625 1 16 64 2>&1 | grep -v amdgpu >> gpurun_out/r04_w3.log
timeout 200 python tools/fuzz_stream.py $(seq 800 812) 2>&1 | tail -1 >> gpurun_out/r04_w3.log
cat gpurun_out/r04_w3.log

R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_deflate_gpu.py -x -q -k over_4gib -s 2>&1 | grep -E "passed|failed|Error" | tail -3 > gpurun_out/r04_w2.log
python -m pytest tests/test_stream_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -2 >> gpurun_out/r04_w2.log
python tools/bench_stream.py 1 16 64 2>&1 | grep -v amdgpu >> gpurun_out/r04_w2.log
python tools/bench_stream.py --mix 16 64 2>&1 | grep -v amdgpu >> gpurun_out/r04_w2.log
timeout 300 python tools/fuzz_stream.py $(seq 700 716) 2>&1 | tail -1 >> gpurun_out/r04_w2.log
cat gpurun_out/r04_w2.log

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_stream_gpu.py -x -q -s 2>&1 | tail -12 > gpurun_out/r04_w1.log
python tools/bench_stream.py 1 16 64 2>&1 | grep -v amdgpu >> gpurun_out/r04_w1.log
timeout 300 python tools/fuzz_stream.py $(seq 500 512) 2>&1 | tail -2 >> gpurun_out/r04_w1.log
LDA_STREAM_WINDOW=65536 timeout 300 python tools/fuzz_stream.py $(seq 600 612) 2>&1 | tail -2 >> gpurun_out/r04_w1.log
cat gpurun_out/r04_w1.log

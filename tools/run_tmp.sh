cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python tools/bench_stream.py 1 16 64 2>&1 | grep -v amdgpu > gpurun_out/r04_bs8.log
python tools/bench_stream.py --mix 16 2>&1 | grep -v amdgpu >> gpurun_out/r04_bs8.log
python -m pytest tests/test_stream_gpu.py -x -q 2>&1 | tail -2 >> gpurun_out/r04_bs8.log
python -m pytest tests/test_deflate_gpu.py -x -q -k over_4gib 2>&1 | tail -2 >> gpurun_out/r04_bs8.log
timeout 300 python tools/fuzz_stream.py $(seq 400 412) 2>&1 | tail -2 >> gpurun_out/r04_bs8.log
cat $R/gpurun_out/r04_bs8.log

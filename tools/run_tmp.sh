cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python tools/bench_stream.py 0.125 1 16 2>&1 | grep -v amdgpu > gpurun_out/r04_bs10.log
python -m pytest tests/test_stream_gpu.py -x -q 2>&1 | tail -2 >> gpurun_out/r04_bs10.log
cd /tmp
rm -rf $R/gpurun_out/trace_s2; mkdir -p $R/gpurun_out/trace_s2
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace_s2 -o s --output-format csv -- python $R/tools/bench_stream.py 0.125 > /dev/null 2>&1
for f in $(find $R/gpurun_out/trace_s2 -name "*kernel_stats.csv"); do head -8 $f >> $R/gpurun_out/r04_bs10.log; done
find $R/gpurun_out/trace_s2 -name "*_trace.csv" -delete
cat $R/gpurun_out/r04_bs10.log

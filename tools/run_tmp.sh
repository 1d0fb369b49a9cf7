cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest $R/tests/test_stream_gpu.py -x -q 2>&1 | tail -5 > $R/gpurun_out/r04_bs3.log
python $R/tools/bench_stream.py 1 4 16 64 256 >> $R/gpurun_out/r04_bs3.log 2>&1
python $R/tools/bench_stream.py --mix 1 16 64 >> $R/gpurun_out/r04_bs3.log 2>&1
rm -rf $R/gpurun_out/trace_s1; mkdir -p $R/gpurun_out/trace_s1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace_s1 -o s --output-format csv -- python $R/tools/bench_stream.py 16 > /dev/null 2>&1
find $R/gpurun_out/trace_s1 -name "*kernel_stats.csv" | head -1 | xargs cat | head -14 >> $R/gpurun_out/r04_bs3.log
find $R/gpurun_out/trace_s1 -name "*kernel_trace.csv" -delete
cat $R/gpurun_out/r04_bs3.log

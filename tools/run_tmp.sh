#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
prof() {
  D=$GRAFT_REPO_ROOT/gpurun_out/trace_tmp_$1
  rm -rf $D; mkdir -p $D
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $D -o x --output-format csv -- "${@:2}" > $D/stdout.log 2>&1)
  echo "== $1"; grep "MiB" $D/stdout.log
  find $D -name "*kernel_stats.csv" | head -1 | xargs cat | head -14
  find $D -name "*kernel_trace.csv" -delete
}
{
prof s1 python $GRAFT_REPO_ROOT/tools/bench_stream.py 1
LDA_STREAM_DEBUG=1 python tools/bench_stream.py 1 2>&1 | tail -12
timeout 300 python -m pytest tests/test_deflate_gpu.py -m gpu -x -q 2>&1 | tail -3
} 2>&1 | grep -v amdgpu.ids > gpurun_out/tmp_stream.txt
cat gpurun_out/tmp_stream.txt

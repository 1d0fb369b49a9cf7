#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench.py run (run via gpurun).
# Writes gpurun_out/trace_rNN/ ; copy the *_kernel_stats.csv into profiles/.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
tag=${1:-r01}
rm -rf $R/gpurun_out/trace_$tag; mkdir -p $R/gpurun_out/trace_$tag
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace_$tag -o bench --output-format csv -- \
    python $R/bench.py --steps 3 --warmup 1 --no-cpu --configs headline > $R/gpurun_out/trace_$tag/bench_stdout.log 2>&1 || true
tail -2 $R/gpurun_out/trace_$tag/bench_stdout.log
find $R/gpurun_out/trace_$tag -name "*kernel_stats.csv" | head -1 | xargs cat | head -12
# keep only the small summaries
find $R/gpurun_out/trace_$tag -name "*kernel_trace.csv" -delete

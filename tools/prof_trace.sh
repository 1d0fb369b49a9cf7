#!/bin/bash
# rocprofv3 kernel trace + stats of one workload (run via gpurun).
# usage: tools/prof_trace.sh <tag> [bench|small|opt|stream|l1|l9|inflate64k]
# Writes gpurun_out/trace_<tag>_<what>/ ; copy the *_kernel_stats.csv into profiles/.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
tag=${1:-r01}
what=${2:-bench}
case $what in
  bench)  cmd="python $R/bench.py --steps 3 --warmup 1 --no-cpu --configs headline" ;;
  small)  cmd="python $R/tools/microbench.py deflate --size 4096 --chunks 1048576 --level 9 --fmt zlib --iters 2" ;;
  opt)    cmd="python $R/tools/microbench.py deflate --chunks 4096 --level 12 --iters 2" ;;
  stream) cmd="python $R/tools/bench_stream.py 16" ;;
  l1)     cmd="python $R/tools/microbench.py deflate --chunks 4096 --level 1 --fmt deflate --iters 3" ;;
  l9)     cmd="python $R/tools/microbench.py deflate --chunks 4096 --level 9 --iters 3" ;;
  inflate64k) cmd="python $R/tools/microbench.py inflate --chunks 65536 --fmt gzip" ;;
esac
D=$R/gpurun_out/trace_${tag}_$what
rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D -o $what --output-format csv -- $cmd > $D/stdout.log 2>&1 || true
tail -2 $D/stdout.log
find $D -name "*kernel_stats.csv" | head -1 | xargs cat | head -12
# keep only the small summaries
find $D -name "*kernel_trace.csv" -delete

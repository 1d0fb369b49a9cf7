cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LDA_STREAM_DEBUG=1 python $R/tools/bench_stream.py --mix 16 2>&1 | grep -v planned | head -30 > $R/gpurun_out/r04_bs4.log
cat $R/gpurun_out/r04_bs4.log

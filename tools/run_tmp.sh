cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python tools/fuzz_stream.py $(seq 300 330) > gpurun_out/r04_fz.log 2>&1
tail -12 gpurun_out/r04_fz.log

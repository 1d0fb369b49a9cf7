#!/bin/bash
# prof_all.sh TAG - every profile the round's documents quote, in one gpurun
# call: rocprofv3 kernel stats of each workload (tools/prof_trace.sh), the PMC
# passes (tools/prof_pmc.sh), the host-to-host figures of the single-buffer
# calls (tools/bench_stream.py, tools/bench_single_api.py).  Everything lands
# in gpurun_out/; tools/collect_profiles.sh TAG copies the summaries into
# profiles/TAG_*.
R=$GRAFT_REPO_ROOT
tag=${1:-r06}
cd $R
for w in bench l1 l9 small opt inflate64k stream; do
  timeout 600 tools/prof_trace.sh $tag $w > gpurun_out/trace_${tag}_$w.log 2>&1
done
for w in bench l1 l9 opt small inflate64k stream; do
  timeout 900 tools/prof_pmc.sh $w > gpurun_out/pmc_$w.log 2>&1
done
{
  timeout 300 python tools/bench_stream.py 1 4 16 64 256
  timeout 300 python tools/bench_stream.py --mix 16 64
  timeout 300 python tools/bench_stream.py --stored 16 256
  timeout 300 python tools/bench_stream.py --fixed 16 256
} > gpurun_out/${tag}_bench_stream.txt 2>&1
{ for m in 1 16 256 1024; do timeout 600 python tools/bench_single_api.py $m $((m < 64 ? 16 : 3)); done; } > gpurun_out/${tag}_bench_single_api.txt 2>&1
ls gpurun_out | head -80

#!/bin/bash
# collect_profiles.sh TAG - copy what tools/prof_all.sh left in gpurun_out/ into
# profiles/TAG_* (the committed evidence the documents cite)
cd "$(dirname "$0")/.."
tag=${1:-r06}
for w in bench l1 l9 small opt inflate64k stream; do
  f=$(find gpurun_out/trace_${tag}_$w -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" profiles/${tag}_${w}_kernel_stats.csv
done
for w in bench l1 l9 opt small inflate64k stream; do
  [ -f gpurun_out/pmc_$w.json ] && cp gpurun_out/pmc_$w.json profiles/${tag}_pmc_$w.json
done
for f in bench_stream bench_single_api; do
  [ -f gpurun_out/${tag}_$f.txt ] && grep -v "amdgpu.ids" gpurun_out/${tag}_$f.txt > profiles/${tag}_$f.txt
done
[ -f gpurun_out/${tag}_bench_line.json ] && grep "^{" gpurun_out/${tag}_bench_line.json | tail -1 > profiles/${tag}_bench_line.json
bash tools/prof_resources.sh > profiles/${tag}_kernel_resources.txt 2>/dev/null
ls -la profiles | grep ${tag}_

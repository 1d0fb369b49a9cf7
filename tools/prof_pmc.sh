#!/bin/bash
# PMC passes of one workload (separate from any tracing run; counters only:
# no trace domains in the same rocprofv3 command).
# usage: tools/prof_pmc.sh <bench|small|opt|l1|l9|inflate64k|stream>   (run on the GPU box via gpurun)
#   bench   bench.py's headline batch (4096 distinct 64 KiB chunks, gzip level 6:
#           lda_deflate_batch_kernel, lda_inflate_wave_kernel, CRC-32)
#   small   1 Mi/4 x 4 KiB zlib level 9 (lda_deflate_small_kernel), tools/microbench.py
#   opt     4096 x 64 KiB level 12 (lda_deflate_opt_kernel), tools/microbench.py
#   l1, l9  4096 x 64 KiB level 1 / 9 (lda_deflate_batch_kernel), tools/microbench.py
#   inflate64k  65 536 gzip streams of 64 KiB (lda_inflate_wave_kernel), tools/microbench.py
#   stream  one 16 MiB gzip stream through libdeflate_gzip_decompress (lda_stream_*)
# writes gpurun_out/pmc_<what>.json: per kernel, counter -> value per launch
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
what=${1:-bench}
case $what in
  bench)  cmd="python $R/bench.py --steps 2 --warmup 1 --no-cpu --configs headline" ;;
  small)  cmd="python $R/tools/microbench.py deflate --size 4096 --chunks 262144 --level 9 --fmt zlib --iters 2" ;;
  opt)    cmd="python $R/tools/microbench.py deflate --chunks 4096 --level 12 --iters 2" ;;
  l1)     cmd="python $R/tools/microbench.py deflate --chunks 4096 --level 1 --fmt deflate --iters 2" ;;
  l9)     cmd="python $R/tools/microbench.py deflate --chunks 4096 --level 9 --iters 2" ;;
  inflate64k) cmd="python $R/tools/microbench.py inflate --chunks 65536" ;;
  stream) cmd="python $R/tools/bench_stream.py 16" ;;
  *) echo "unknown workload $what"; exit 2 ;;
esac
rm -rf $R/gpurun_out/pmc_$what; mkdir -p $R/gpurun_out/pmc_$what
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp -d $R/gpurun_out/pmc_$what/$tag -o out --output-format csv -- \
     $cmd > /dev/null 2>&1 || echo "pass failed: $grp"
done
python - <<PY
import csv, glob, collections, json
out = collections.defaultdict(dict)
for f in sorted(glob.glob("$R/gpurun_out/pmc_$what/*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("lda_"):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k, d in agg.items():
        for c, v in d.items():
            out[k][c + "_per_launch"] = v / max(1, len(disp[k]))
        out[k]["launches_seen"] = max(out[k].get("launches_seen", 0), len(disp[k]))
import sys
sys.path.insert(0, "$R")
import bench
out["_workload"] = {"what": "$what", "command": "$cmd".replace("$R/", ""),
                    "kernel_src_sha16": bench.kernel_src_digest()}
json.dump(out, open("$R/gpurun_out/pmc_$what.json", "w"), indent=1, sort_keys=True)
for k, d in out.items():
    if k != "_workload":
        print(k, {c: f"{v:.4g}" for c, v in d.items()})
PY

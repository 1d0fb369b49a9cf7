#!/bin/bash
# A second set of PMC passes over tools/microbench.py deflate (level 6, 4096 x 64 KiB):
# instruction fetch / instruction cache, branches, LDS latency.  Run via gpurun.
# writes gpurun_out/pmc_extra.json
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cmd="python $R/tools/microbench.py deflate --chunks 4096 --level ${LEVEL:-6} --iters 2"
rm -rf $R/gpurun_out/pmc_extra; mkdir -p $R/gpurun_out/pmc_extra
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_IFETCH_LEVEL" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp -d $R/gpurun_out/pmc_extra/$tag -o out --output-format csv -- \
     $cmd > /dev/null 2>&1 || echo "pass failed: $grp"
done
python - <<PY
import csv, glob, collections, json
out = collections.defaultdict(dict)
for f in sorted(glob.glob("$R/gpurun_out/pmc_extra/*/**/*counter_collection.csv", recursive=True)):
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
json.dump(out, open("$R/gpurun_out/pmc_extra.json", "w"), indent=1, sort_keys=True)
for k, d in out.items():
    print(k, {c: f"{v:.4g}" for c, v in d.items()})
PY

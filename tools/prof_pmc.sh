#!/bin/bash
# PMC passes for the compress / inflate kernels (separate from any tracing run).
# usage: tools/prof_pmc.sh <deflate|inflate>   (run on the GPU box via gpurun)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
what=${1:-deflate}
mkdir -p $R/gpurun_out/pmc_$what
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp -d $R/gpurun_out/pmc_$what/$tag -o out --output-format csv -- \
     python $R/tools/microbench.py $what --chunks 4096 --iters 2 > /dev/null 2>&1 || echo "pass failed: $grp"
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
json.dump(out, open("$R/gpurun_out/pmc_$what.json", "w"), indent=1, sort_keys=True)
for k, d in out.items():
    print(k, {c: f"{v:.4g}" for c, v in d.items()})
PY

#!/bin/bash
# ab_small.sh NAME... - the small-buffer compress kernel of variant builds side
# by side: 262 144 x 4 KiB zlib level 9 and 6 (tools/microbench.py), then the
# small batches of tools/fuzz_deflate.py (seeds 0 4 8 ... = small) under a timeout.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "$@"; do
  lib=libdeflate_amd/libdeflate_amd_$v.so
  [ "$v" = main ] && lib=libdeflate_amd/libdeflate_amd.so
  out=gpurun_out/abs_$v.txt
  : > $out
  for l in 9 6; do
    LIBDEFLATE_AMD_LIB=$PWD/$lib timeout -k 5 200 python tools/microbench.py deflate \
        --chunks 262144 --size 4096 --level $l --fmt zlib --iters 5 >> $out 2>&1 || echo "FAILED level $l rc=$?" >> $out
  done
  if [ -z "$NOFUZZ" ]; then
    LIBDEFLATE_AMD_LIB=$PWD/$lib timeout -k 5 300 python tools/fuzz_deflate.py $(seq 0 4 160) 2>&1 | tail -2 >> $out || echo "FUZZ FAILED rc=$?" >> $out
  fi
  echo "== $v"; grep -E "deflate\[|FAIL|seed" $out
done

#!/bin/bash
# ab.sh NAME... - time and check builds of the library side by side on one GPU
# box: for every NAME (a `make VARIANT=NAME` build, libdeflate_amd_NAME.so;
# "main" = the product build) the level-6 compress kernel on 4096 x 64 KiB
# (tools/microbench.py) and, unless NODIGEST=1, the digests of the fixed input
# set (tools/digest_deflate.py --quick: every stream decoded by zlib).
# LEVELS="6 1 9" times more levels.  Every step runs under `timeout`: a
# schedule that deadlocks must not take the box with it.
# Output: gpurun_out/ab_NAME.txt.  A tuning aid, not part of the product.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
LEVELS=${LEVELS:-6}
CHUNKS=${CHUNKS:-4096}
for v in "$@"; do
  lib=libdeflate_amd/libdeflate_amd_$v.so
  [ "$v" = main ] && lib=libdeflate_amd/libdeflate_amd.so
  out=gpurun_out/ab_$v.txt
  : > $out
  for l in $LEVELS; do
    LIBDEFLATE_AMD_LIB=$PWD/$lib timeout -k 5 ${TMO:-150} python tools/microbench.py deflate \
        --chunks $CHUNKS --level $l --iters 5 >> $out 2>&1 || echo "FAILED/TIMEOUT level $l rc=$?" >> $out
  done
  if [ -z "$NODIGEST" ]; then
    LIBDEFLATE_AMD_LIB=$PWD/$lib timeout -k 5 ${DTMO:-150} python tools/digest_deflate.py ${DIGEST_ARGS:---quick} \
        > gpurun_out/dig_$v.txt 2>&1 || echo "DIGEST FAILED rc=$?" >> $out
    md5sum gpurun_out/dig_$v.txt >> $out
    grep -c . gpurun_out/dig_$v.txt >> $out
  fi
  echo "== $v"; grep -E "deflate\[|FAILED|DIGEST|dig_" $out
done

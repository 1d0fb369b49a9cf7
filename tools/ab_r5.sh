#!/bin/bash
# ab_r5.sh - round 5's side-by-side run of variant builds on one GPU box (a tuning
# aid like ab.sh): level-6 compress kernel + level-6 digests of every variant named,
# then the other kernels of the builds in $WIDE.  Output: gpurun_out/r5_*.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${TMO:-70}
lib() { [ "$1" = main ] && echo $PWD/libdeflate_amd/libdeflate_amd.so || echo $PWD/libdeflate_amd/libdeflate_amd_$1.so; }
for v in "$@"; do
  out=gpurun_out/r5_$v.txt; : > $out
  LIBDEFLATE_AMD_LIB=$(lib $v) timeout -k 5 $T python tools/microbench.py deflate --chunks 4096 --level 6 --iters 5 >> $out 2>&1 || echo "FAILED/TIMEOUT L6 rc=$?" >> $out
  LIBDEFLATE_AMD_LIB=$(lib $v) timeout -k 5 $T python tools/digest_deflate.py --levels=6 > gpurun_out/r5dig_$v.txt 2>&1 || echo "DIGEST FAILED rc=$?" >> $out
  md5sum < gpurun_out/r5dig_$v.txt >> $out
  echo "== $v"; grep -E "deflate\[|FAILED|DIGEST" $out; tail -n 1 $out
done
for v in $LEVELS_OF; do	# more levels of the compress kernels only
  out=gpurun_out/r5l_$v.txt; : > $out
  for l in ${WLEVELS:-9 1 12}; do
    LIBDEFLATE_AMD_LIB=$(lib $v) timeout -k 5 $T python tools/microbench.py deflate --chunks 4096 --level $l --iters 5 >> $out 2>&1 || echo "FAILED L$l rc=$?" >> $out
  done
  echo "== levels $v"; grep -E "flate\[|FAILED" $out
done
for v in $WIDE; do
  out=gpurun_out/r5w_$v.txt; : > $out
  for l in 9 1; do
    LIBDEFLATE_AMD_LIB=$(lib $v) timeout -k 5 $T python tools/microbench.py deflate --chunks 4096 --level $l --iters 5 >> $out 2>&1 || echo "FAILED L$l rc=$?" >> $out
  done
  LIBDEFLATE_AMD_LIB=$(lib $v) timeout -k 5 $T python tools/microbench.py inflate --chunks 4096 >> $out 2>&1 || echo "FAILED inflate rc=$?" >> $out
  LIBDEFLATE_AMD_LIB=$(lib $v) timeout -k 5 $T python tools/microbench.py inflate --chunks 65536 >> $out 2>&1 || echo "FAILED inflate64k rc=$?" >> $out
  LIBDEFLATE_AMD_LIB=$(lib $v) timeout -k 5 90 python tools/microbench.py deflate --chunks 262144 --size 4096 --level 9 --fmt zlib --iters 5 >> $out 2>&1 || echo "FAILED small rc=$?" >> $out
  echo "== wide $v"; grep -E "flate\[|FAILED" $out
done
for v in $SMALL; do
  out=gpurun_out/r5s_$v.txt; : > $out
  LIBDEFLATE_AMD_LIB=$(lib $v) timeout -k 5 90 python tools/microbench.py deflate --chunks 262144 --size 4096 --level 9 --fmt zlib --iters 5 >> $out 2>&1 || echo "FAILED small rc=$?" >> $out
  echo "== small $v"; grep -E "flate\[|FAILED" $out
done

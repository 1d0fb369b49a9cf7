#!/bin/bash
# ab_r6.sh - round 6's side-by-side runs of variant builds on one GPU box (a tuning
# aid like ab_r5.sh).  INFL="v1 v2 .." inflate kernel of those builds (4096 and 65 536
# streams, and the binary / 16-symbol kinds alone); DEFL="v1 .." level-6 compress +
# digests; LEVELS_OF / SMALL as in ab_r5.sh.  Output: gpurun_out/r6_*.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${TMO:-90}
lib() { [ "$1" = main ] && echo $PWD/libdeflate_amd/libdeflate_amd.so || echo $PWD/libdeflate_amd/libdeflate_amd_$1.so; }
mb() { LIBDEFLATE_AMD_LIB=$(lib $1) timeout -k 5 $T python tools/microbench.py "${@:2}" 2>&1 | grep -v amdgpu.ids; }
for v in $INFL; do
  out=gpurun_out/r6i_$v.txt; : > $out
  mb $v inflate --chunks 4096 >> $out || echo "FAILED inflate rc=$?" >> $out
  mb $v inflate --chunks 4096 >> $out
  mb $v inflate --chunks 65536 >> $out || echo "FAILED inflate64k rc=$?" >> $out
  for k in ${KINDS:-0 5 6}; do mb $v inflate --chunks 4096 --kind $k >> $out; done
  echo "== inflate $v"; grep -E "flate\[|FAILED" $out
done
for v in $DEFL; do
  out=gpurun_out/r6d_$v.txt; : > $out
  mb $v deflate --chunks 4096 --level 6 --iters 5 >> $out || echo "FAILED L6 rc=$?" >> $out
  LIBDEFLATE_AMD_LIB=$(lib $v) timeout -k 5 $T python tools/digest_deflate.py --levels=6 > gpurun_out/r6dig_$v.txt 2>&1 || echo "DIGEST FAILED rc=$?" >> $out
  md5sum < gpurun_out/r6dig_$v.txt >> $out
  echo "== deflate $v"; grep -E "flate\[|FAILED|DIGEST" $out; tail -n 1 $out
done
for v in $LEVELS_OF; do
  out=gpurun_out/r6l_$v.txt; : > $out
  for l in ${WLEVELS:-9 1 12}; do
    mb $v deflate --chunks 4096 --level $l --iters 5 >> $out || echo "FAILED L$l rc=$?" >> $out
  done
  echo "== levels $v"; grep -E "flate\[|FAILED" $out
done
for v in $SMALL; do
  out=gpurun_out/r6s_$v.txt; : > $out
  T=120 mb $v deflate --chunks 262144 --size 4096 --level 9 --fmt zlib --iters 5 >> $out || echo "FAILED small rc=$?" >> $out
  echo "== small $v"; grep -E "flate\[|FAILED" $out
done

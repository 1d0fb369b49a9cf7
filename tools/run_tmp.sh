cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python tools/digest_deflate.py --quick > gpurun_out/dig_new2.txt 2>&1
diff <(grep -v big gpurun_out/dig_new.txt) <(grep -v big gpurun_out/dig_new2.txt) > gpurun_out/r04_huff2.log && echo "digests identical to the first parallel build (except big)" >> gpurun_out/r04_huff2.log
python tools/microbench.py deflate --chunks 4096 --level 6 2>&1 | grep "deflate\[" >> gpurun_out/r04_huff2.log
python tools/microbench.py deflate --size 4096 --chunks 262144 --level 9 --fmt zlib 2>&1 | grep "deflate\[" >> gpurun_out/r04_huff2.log
LIBDEFLATE_AMD_LIB=$R/libdeflate_amd/libdeflate_amd_prof.so python tools/microbench.py deflate --chunks 4096 --level 6 2>&1 | grep -E "two trees|mc:|deflate\[" | tail -6 >> gpurun_out/r04_huff2.log
cat gpurun_out/r04_huff2.log

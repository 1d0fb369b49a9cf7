cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_deflate_gpu.py -q -k "over_4gib" 2>&1 | tail -5 > gpurun_out/r04_t3.log
python -m pytest tests -m gpu -q --deselect tests/test_deflate_gpu.py::test_input_over_4gib --deselect tests/test_stream_gpu.py 2>&1 | tail -8 >> gpurun_out/r04_t3.log
LIBDEFLATE_AMD_LIB=$R/libdeflate_amd/libdeflate_amd_prof.so python tools/microbench.py deflate --chunks 4096 --level 6 > gpurun_out/r04_prof6.log 2>&1
cat gpurun_out/r04_t3.log; cat gpurun_out/r04_prof6.log

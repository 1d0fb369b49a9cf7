R=$GRAFT_REPO_ROOT
cd $R
bash tools/prof_trace.sh r04 bench > gpurun_out/r04_prof.log 2>&1
bash tools/prof_trace.sh r04 small >> gpurun_out/r04_prof.log 2>&1
bash tools/prof_trace.sh r04 opt >> gpurun_out/r04_prof.log 2>&1
bash tools/prof_trace.sh r04 stream >> gpurun_out/r04_prof.log 2>&1
bash tools/prof_pmc.sh bench >> gpurun_out/r04_prof.log 2>&1
bash tools/prof_pmc.sh small >> gpurun_out/r04_prof.log 2>&1
cat gpurun_out/r04_prof.log | cut -c1-400

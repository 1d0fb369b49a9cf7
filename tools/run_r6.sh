#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
INFL="main v_pf1 v_pf4 main" tools/ab_r6.sh

#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
INFL="main v_prio0 v_prio2" KINDS=" " tools/ab_r6.sh

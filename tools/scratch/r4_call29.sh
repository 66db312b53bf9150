#!/bin/bash
# round-4 call 29: pass 1's gathers two trips ahead (-DSRT_SCAT_GATHER2=1) against one: A/B at 200 k rays
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c29
RAYS=200000 PMC=0 TIMES=2 bash tools/scat_exp.sh "one|-" "two|gather2" "oneb|-" "twob|gather2" "onec|-" "twoc|gather2"

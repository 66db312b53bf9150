#!/bin/bash
# round-4 call 23: config[4] at FULL size, the round's final library against the library of profiles/r04 v31 (round 3's scattered kernel), twice
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c23
RAYS=1000000 PMC=0 TIMES=1 bash tools/scat_exp.sh "final|-" "v31|v31" "finalb|-" "v31b|v31" 2>&1 | tee gpurun_out/c23/ab.txt

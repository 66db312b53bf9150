#!/bin/bash
# round-4 call 26: pass 1's first gather issued before the points are handed round: A/B at 200 k rays against the build before it
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c26
RAYS=200000 PMC=0 TIMES=2 bash tools/scat_exp.sh "ahead1|ahead1" "hoist|-" "ahead1b|ahead1" "hoistb|-" "ahead1c|ahead1" "hoistc|-"

#!/bin/bash
# usage: tools/pmc_pass.sh <outdir> <name> "<counters>" [bench args...]   (run on the GPU box via gpurun)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
NAME=$2; CNT="$3"; shift 3
timeout -k 10 400 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/$NAME -- python3 bench.py --traffic off --other-configs 0 --steps 1 --warmup 0 --cpu-seconds 0 "$@" > $O/$NAME.log 2>&1
grep -h trace_kernel $O/$NAME/*/*counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}' | sed 's/"//g'

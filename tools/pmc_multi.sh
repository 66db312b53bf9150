#!/bin/bash
# usage: tools/pmc_multi.sh <outdir> [bench args...] -- several PMC passes over the default bench (run on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R; shift
i=0
while read -r CNT; do
  [ -z "$CNT" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/p$i -- python3 bench.py --traffic off --other-configs 0 --steps 1 --warmup 0 --cpu-seconds 0 "$@" > $O/p$i.log 2>&1
  grep -h trace_kernel $O/p$i/*/*counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}' | sed 's/"//g'
done < $R/tools/pmc_sets.txt

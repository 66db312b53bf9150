#!/bin/bash
# tools/measure_scat.sh <tag> -- config[4] at full size: bench line (with the CPU baseline) and rocprofv3 kernel stats of the same
# workload (one launch), then the config[1] bench line.  Outputs under gpurun_out/<tag>/ (run on the GPU box via gpurun).
set -e
TAG=${1:-scat}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -k 10 500 python bench.py --traffic off --other-configs 0 --workload scattered825k --steps 1 --warmup 1 > $O/bench_scat.log 2>$O/bench_scat.err && tail -1 $O/bench_scat.log | cut -c1-300
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --traffic off --other-configs 0 --workload scattered825k --steps 1 --warmup 0 --cpu-seconds 0 --damping-rays 0 > $O/stats.log 2>&1
cat $O/stats/*/*kernel_stats.csv | head -4 | cut -c1-220
timeout -k 10 300 python bench.py --traffic off --other-configs 0 --workload ngo100k > $O/bench_ngo.log 2>$O/bench_ngo.err && tail -1 $O/bench_ngo.log | cut -c1-300

#!/bin/bash
# round-4 call 3: full GPU suite (no -x), config[4] A/B at 200k rays (v28 sources, the reverted tree, rsq pivots), the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c3
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/c3/pytest_gpu.log 2>&1; tail -5 gpurun_out/c3/pytest_gpu.log
RAYS=200000 PMC=0 TIMES=2 bash tools/scat_exp.sh "v28|v28" "head|-" "rsq|rsq" "v28b|v28" "headb|-" "rsqb|rsq" 2>&1 | tee gpurun_out/c3/ab.txt
cd /tmp && export TMPDIR=/tmp && cd $R
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/c3/bench.log 2> gpurun_out/c3/bench.err; tail -1 gpurun_out/c3/bench.log | cut -c1-300

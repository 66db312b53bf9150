#!/bin/bash
# round-4 call 2: full GPU suite (no -x), then config[4] A/B at 200k rays: v28 sources, HEAD, HEAD with the 8-way split, HEAD without the sub-sort
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c2
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/c2/pytest_gpu.log 2>&1; tail -5 gpurun_out/c2/pytest_gpu.log
RAYS=200000 PMC=0 TIMES=2 bash tools/scat_exp.sh "v28|v28" "head|-" "nine0|nine0" "nosub|-|SRT_SCATTERED_SUBSORT=0" "v28b|v28" "headb|-" 2>&1 | tee gpurun_out/c2/ab.txt

#!/bin/bash
# round-4 call 28: the pair loop's first buffer started by the weights pass (SRT_SCAT_DMA_EARLY): scattered parity suites, then A/B
# at 200 k rays against the build before it
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c28
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scattered_paths.py tests/test_gcpm_golden.py -x -q -m gpu -k "scattered or gcpm" > gpurun_out/c28/tests.log 2>&1; rc=$?; tail -3 gpurun_out/c28/tests.log
[ $rc -eq 0 ] || exit $rc
RAYS=200000 PMC=0 TIMES=2 bash tools/scat_exp.sh "batch|batch" "early|-" "batchb|batch" "earlyb|-" "batchc|batch" "earlyc|-"

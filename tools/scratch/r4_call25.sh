#!/bin/bash
# round-4 call 25: the weights pass's prologue constants from the model (u11, 1/radius) and the staging records read one trip ahead:
# scattered parity suites, then A/B at 200 k rays in one call -- head (4594654) | ahead0 (constants only) | new (both), twice
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c25
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scattered_paths.py tests/test_gcpm_golden.py -x -q -m gpu -k "scattered or gcpm" > gpurun_out/c25/tests.log 2>&1; rc=$?; tail -3 gpurun_out/c25/tests.log
[ $rc -eq 0 ] || exit $rc
RAYS=200000 PMC=0 TIMES=2 bash tools/scat_exp.sh "head|head" "ahead0|ahead0" "new|-" "head2|head" "ahead0b|ahead0" "new2|-" &&
RAYS=200000 PMC=1 TIMES=1 bash tools/scat_exp.sh "headp|head" "newp|-"

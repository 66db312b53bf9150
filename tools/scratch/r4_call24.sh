#!/bin/bash
# round-4 call 24: the generate_monomials orders (4, 5) on the device: parity tests, the scattered suites around them, what a fit
# costs, and the round's kernel against HEAD's (libsrt_hip_head.so) at 200 k rays in the same call
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c24
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scattered_paths.py tests/test_gcpm_golden.py -x -q -m gpu -k "scattered or gcpm" > gpurun_out/c24/tests.log 2>&1; rc=$?; tail -5 gpurun_out/c24/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/scratch/order45_cost.py > gpurun_out/c24/cost.log 2>&1 && cat gpurun_out/c24/cost.log &&
RAYS=200000 PMC=0 TIMES=2 bash tools/scat_exp.sh "head|head" "new|-" "head2|head" "new2|-"

#!/bin/bash
# round-4 call 36: the hand-off's exponential out of line (its constants no longer spilled and reloaded inside the chain) against v34:
# every kept row bit-equal, scattered parity suites, A/B at 200 k rays
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c36
timeout -k 10 300 python tools/scratch/scat_rows_hash.py > gpurun_out/c36/hash_new.log 2>&1 &&
SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_v34.so timeout -k 10 300 python tools/scratch/scat_rows_hash.py > gpurun_out/c36/hash_old.log 2>&1 &&
cat gpurun_out/c36/hash_new.log gpurun_out/c36/hash_old.log && cmp gpurun_out/c36/hash_new.log gpurun_out/c36/hash_old.log && echo "HASHES EQUAL" &&
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scattered_paths.py tests/test_gcpm_golden.py -x -q -m gpu -k "scattered or gcpm" > gpurun_out/c36/tests.log 2>&1; rc=$?; tail -3 gpurun_out/c36/tests.log
[ $rc -eq 0 ] || exit $rc
RAYS=200000 PMC=0 TIMES=2 bash tools/scat_exp.sh "v34|v34" "expo|-" "v34b|v34" "expob|-" "v34c|v34" "expoc|-"

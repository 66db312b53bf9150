#!/bin/bash
# round-4 call 37: the staging pointer wave-uniform in the three passes (no spill of it, no reload per trip) against v35:
# every kept row bit-equal, scattered parity suites, A/B at 200 k rays
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c37
timeout -k 10 300 python tools/scratch/scat_rows_hash.py > gpurun_out/c37/hash_new.log 2>&1 &&
SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_v35.so timeout -k 10 300 python tools/scratch/scat_rows_hash.py > gpurun_out/c37/hash_old.log 2>&1 &&
cat gpurun_out/c37/hash_new.log gpurun_out/c37/hash_old.log && cmp gpurun_out/c37/hash_new.log gpurun_out/c37/hash_old.log && echo "HASHES EQUAL" &&
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scattered_paths.py tests/test_gcpm_golden.py -x -q -m gpu -k "scattered or gcpm" > gpurun_out/c37/tests.log 2>&1; rc=$?; tail -3 gpurun_out/c37/tests.log
[ $rc -eq 0 ] || exit $rc
RAYS=200000 PMC=0 TIMES=2 bash tools/scat_exp.sh "v35|v35" "uni|-" "v35b|v35" "unib|-" "v35c|v35" "unic|-"

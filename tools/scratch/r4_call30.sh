#!/bin/bash
# round-4 call 30: no store drain between the passes of a stencil (SRT_SCAT_WAVE_ORDER=1: the wave's own issue order) against
# __syncthreads() there (libsrt_hip_sync.so): every kept row bit-equal (hashes, short and long lists), the scattered parity suites,
# then A/B at 200 k rays
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c30
timeout -k 10 300 python tools/scratch/scat_rows_hash.py > gpurun_out/c30/hash_new.log 2>&1 &&
SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_sync.so timeout -k 10 300 python tools/scratch/scat_rows_hash.py > gpurun_out/c30/hash_sync.log 2>&1 &&
cat gpurun_out/c30/hash_new.log gpurun_out/c30/hash_sync.log && cmp gpurun_out/c30/hash_new.log gpurun_out/c30/hash_sync.log && echo "HASHES EQUAL" &&
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scattered_paths.py tests/test_gcpm_golden.py -x -q -m gpu -k "scattered or gcpm" > gpurun_out/c30/tests.log 2>&1; rc=$?; tail -3 gpurun_out/c30/tests.log
[ $rc -eq 0 ] || exit $rc
RAYS=200000 PMC=0 TIMES=2 bash tools/scat_exp.sh "sync|sync" "order|-" "syncb|sync" "orderb|-" "syncc|sync" "orderc|-"

#!/bin/bash
# round-4 call 33: the caller's private arrays addressed as scratch in coop_stencil (no flat stores in front of the LDS-only syncs)
# against v33's library: every kept row bit-equal, scattered parity suites, A/B at 200 k rays
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c33
timeout -k 10 300 python tools/scratch/scat_rows_hash.py > gpurun_out/c33/hash_new.log 2>&1 &&
SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_v33.so timeout -k 10 300 python tools/scratch/scat_rows_hash.py > gpurun_out/c33/hash_old.log 2>&1 &&
cat gpurun_out/c33/hash_new.log gpurun_out/c33/hash_old.log && cmp gpurun_out/c33/hash_new.log gpurun_out/c33/hash_old.log && echo "HASHES EQUAL" &&
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scattered_paths.py tests/test_gcpm_golden.py -x -q -m gpu -k "scattered or gcpm" > gpurun_out/c33/tests.log 2>&1; rc=$?; tail -3 gpurun_out/c33/tests.log
[ $rc -eq 0 ] || exit $rc
RAYS=200000 PMC=0 TIMES=2 bash tools/scat_exp.sh "v33|v33" "priv|-" "v33b|v33" "privb|-" "v33c|v33" "privc|-"

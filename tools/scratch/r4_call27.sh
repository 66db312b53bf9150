#!/bin/bash
# round-4 call 27: the owner's nine stencil numbers fetched together (one round trip instead of six per stencil): scattered parity
# suites, then A/B at 200 k rays against the build before it
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c27
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scattered_paths.py -x -q -m gpu -k "scattered" > gpurun_out/c27/tests.log 2>&1; rc=$?; tail -3 gpurun_out/c27/tests.log
[ $rc -eq 0 ] || exit $rc
RAYS=200000 PMC=0 TIMES=2 bash tools/scat_exp.sh "ahead1|ahead1" "batch|-" "ahead1b|ahead1" "batchb|-" "ahead1c|ahead1" "batchc|-"

#!/bin/bash
# round-4 call 34: offset_F's constants through the scalar cache (libsrt_hip_scal.so) against v34's library: Ngo parity tests, config[1] A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c34
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trace.py tests/test_root1_golden.py -x -q -m gpu -k "not scattered and not 256 and not full" > gpurun_out/c34/tests.log 2>&1; rc=$?; tail -3 gpurun_out/c34/tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/scratch/ab_ngo.sh v34 scal v34 scal

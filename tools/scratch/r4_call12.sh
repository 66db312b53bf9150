#!/bin/bash
# round-4 call 12: the scattered model's second-order tier (SRT_SCAT_TAYLOR): parity tests, then config[4] A/B with counters at 200k rays
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c12
timeout -k 10 600 python -m pytest tests/test_gpu_scattered_paths.py tests/test_gpu_parity.py tests/test_gcpm_golden.py tests/test_gpu_trajectory_stats.py tests/test_gpu_trace.py -m gpu -q -k "scattered or gcpm or config5" > gpurun_out/c12/pytest.log 2>&1; tail -3 gpurun_out/c12/pytest.log
RAYS=200000 PMC=1 TIMES=2 bash tools/scat_exp.sh "taylor|-" "notaylor|notaylor" "taylorb|-" "notaylorb|notaylor" 2>&1 | tee gpurun_out/c12/ab.txt

#!/bin/bash
# round-4 call 18: two samples per lane and trip in the second-order tier (exp / log chains interleaved): parity tests, A/B at 200k rays with counters
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c19
timeout -k 10 600 python -m pytest tests/test_gpu_scattered_paths.py tests/test_gpu_parity.py tests/test_gcpm_golden.py tests/test_gpu_trajectory_stats.py tests/test_gpu_trace.py -m gpu -q -k "scattered or gcpm or config5" > gpurun_out/c19/pytest.log 2>&1; tail -3 gpurun_out/c19/pytest.log
RAYS=200000 PMC=1 TIMES=2 bash tools/scat_exp.sh "ns2o|-" "ns1|ns1" "ns2ob|-" "ns1b|ns1" 2>&1 | tee gpurun_out/c19/ab.txt

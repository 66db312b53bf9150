#!/bin/bash
# round-4 call 20: LDS-only wave syncs where only LDS data is handed over (no wait for scratch stores in flight): scattered parity tests,
# config[4] A/B at 200k rays with counters, twice
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c21
timeout -k 10 600 python -m pytest tests/test_gpu_scattered_paths.py tests/test_gpu_parity.py tests/test_gcpm_golden.py tests/test_gpu_trajectory_stats.py tests/test_gpu_trace.py -m gpu -q -k "scattered or gcpm or config5" > gpurun_out/c21/pytest.log 2>&1; tail -3 gpurun_out/c21/pytest.log
RAYS=200000 PMC=1 TIMES=2 bash tools/scat_exp.sh "sync2|-" "base|base" "sync2b|-" "baseb|base" 2>&1 | tee gpurun_out/c21/ab.txt

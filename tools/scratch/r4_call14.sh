#!/bin/bash
# round-4 call 14: kept counts made in the weights pass + zero-padded records (no index test, no count in the pair loop): scattered
# parity tests, then config[4] A/B at 200k rays with counters, twice
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c14
timeout -k 10 600 python -m pytest tests/test_gpu_scattered_paths.py tests/test_gpu_parity.py tests/test_gcpm_golden.py tests/test_gpu_trajectory_stats.py tests/test_gpu_trace.py -m gpu -q -k "scattered or gcpm or config5" > gpurun_out/c14/pytest.log 2>&1; tail -3 gpurun_out/c14/pytest.log
RAYS=200000 PMC=1 TIMES=2 bash tools/scat_exp.sh "kept|-" "tay|tay" "keptb|-" "tayb|tay" 2>&1 | tee gpurun_out/c14/ab.txt

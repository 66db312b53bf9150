#!/bin/bash
# round-4 call 5: the Ngo model's right-hand sides spread over idle lanes in tail mode: GPU tests that touch the Ngo model and the new
# tests of this round, then config[1] A/B (spread / not spread), two rounds, plus the 1 M-ray launch
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c5
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_trace.py::test_full_size_config5_scattered_1m > gpurun_out/c5/pytest_gpu.log 2>&1; tail -4 gpurun_out/c5/pytest_gpu.log
bash tools/scratch/ab_ngo.sh nospread spread 2>&1 | tee gpurun_out/c5/ab_ngo.txt
RAYS=1000000 bash tools/scratch/ab_ngo.sh nospread spread 2>&1 | tee gpurun_out/c5/ab_ngo_1m.txt

#!/bin/bash
# round-4 call 10: Ngo stages in tail mode fused (density + offset evaluations in one exchange): Ngo-touching GPU tests, then config[1] A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c10
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trace.py tests/test_gpu_trajectory_stats.py tests/test_root1_golden.py tests/test_gpu_cli.py tests/test_igrf.py tests/test_t04.py -m gpu -q -x -k "not full_size" > gpurun_out/c10/pytest_gpu.log 2>&1; tail -4 gpurun_out/c10/pytest_gpu.log
bash tools/scratch/ab_ngo.sh nofuse fusetail 2>&1 | tee gpurun_out/c10/ab_ngo.txt
RAYS=1000000 bash tools/scratch/ab_ngo.sh nofuse fusetail 2>&1 | tee gpurun_out/c10/ab_ngo_1m.txt

#!/bin/bash
# round-4 call 4: the fused scattered path -- parity tests under the fused builds, then config[4] A/B with traffic counters at 200k rays
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c4
for v in fused1 fused2; do
  SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_$v.so timeout -k 10 600 python -m pytest tests/test_gpu_scattered_paths.py tests/test_gpu_parity.py tests/test_gcpm_golden.py -m gpu -q -k "scattered or gcpm" > gpurun_out/c4/pytest_$v.log 2>&1; echo "$v: $(tail -1 gpurun_out/c4/pytest_$v.log)"
done
timeout -k 10 300 python -m pytest tests/test_gcpm_golden.py -m gpu -q > gpurun_out/c4/pytest_gcpm_head.log 2>&1; tail -1 gpurun_out/c4/pytest_gcpm_head.log
RAYS=200000 PMC=1 TIMES=2 bash tools/scat_exp.sh "head|-" "fused1|fused1" "w1|w1" "fused2|fused2" "headb|-" "fused1b|fused1" 2>&1 | tee gpurun_out/c4/ab.txt

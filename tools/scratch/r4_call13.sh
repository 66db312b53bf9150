#!/bin/bash
# round-4 call 13: second-order tier, final form: scattered parity tests, then config[4] at FULL size, A/B twice, VALU counter once each
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c13
timeout -k 10 600 python -m pytest tests/test_gpu_scattered_paths.py tests/test_gpu_parity.py tests/test_gcpm_golden.py tests/test_gpu_trajectory_stats.py tests/test_gpu_trace.py -m gpu -q -k "scattered or gcpm or config5" > gpurun_out/c13/pytest.log 2>&1; tail -3 gpurun_out/c13/pytest.log
RAYS=1000000 PMC=0 TIMES=1 bash tools/scat_exp.sh "taylor|-" "notaylor|notaylor" "taylorb|-" "notaylorb|notaylor" 2>&1 | tee gpurun_out/c13/ab.txt
cd /tmp && export TMPDIR=/tmp && cd $R
bash tools/pmc_valu.sh notaylor 2>&1 | tee gpurun_out/c13/valu.txt
cp stanford_raytracer_amd/lib/libsrt_hip.so stanford_raytracer_amd/lib/libsrt_hip_taylor.so; bash tools/pmc_valu.sh taylor 2>&1 | tee -a gpurun_out/c13/valu.txt

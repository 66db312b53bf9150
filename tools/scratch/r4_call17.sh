#!/bin/bash
# round-4 call 17: which stencils take the second-order tier (timing build, 50k rays)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c17
SRT_PHASE_TIMING=1 SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_timing.so timeout -k 10 600 python bench.py --traffic off --other-configs 0 --workload scattered825k --rays 50000 --steps 1 --warmup 0 --cpu-seconds 0 --damping-rays 0 > gpurun_out/c17/out.log 2> gpurun_out/c17/err.log; grep "srt tier\|srt phase" gpurun_out/c17/err.log

#!/bin/bash
# A/B two builds of the library in ONE gpurun call (boxes differ by several %): tools/ab.sh "<flagsA>" "<flagsB>" [bench args]
# Build the variants locally first:  tools/ab_build.sh A "<flags>"
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for round in 1 2; do
  for v in "$@"; do
    SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_$v.so timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --damping-rays 0 --traffic off --other-configs 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'steps/s %.4g' % d['value'], int(d['roofline']['accepted_steps_per_launch']))"
  done
done

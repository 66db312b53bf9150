#!/bin/bash
# tools/scratch/ab_wl.sh <workload> "<bench args>" variantA variantB ... -- A/B builds of the library on one workload in ONE gpurun call
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
WL=$1; ARGS=$2; shift 2
for round in 1 2; do
  for v in "$@"; do
    SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_$v.so timeout -k 10 300 python bench.py --workload $WL $ARGS --cpu-seconds 0 --damping-rays 0 --traffic off --other-configs 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', '$WL', 'kernel_ms', round(d['roofline']['kernel_ms'],2), 'steps/s %.4g' % d['value'], int(d['roofline']['accepted_steps_per_launch']), 'occ %.3f' % d['detail']['lane_occupancy'])"
  done
done

#!/bin/bash
# tools/scratch/ab_ngo.sh <variant...> -- BASELINE config[1] (100k rays, Ngo model) for library variants, two rounds in one call
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for round in 1 2; do
  for v in "$@"; do
    SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_$v.so timeout -k 10 300 python bench.py --workload ngo100k ${RAYS:+--rays $RAYS} --steps 5 --warmup 2 --traffic off --other-configs 0 --cpu-seconds 0 --damping-rays 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', 'kernel_ms %.2f steps/s %.4g' % (d['roofline']['kernel_ms'], d['value']))"
  done
done

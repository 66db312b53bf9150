#!/bin/bash
# tools/scratch/ab_field.sh <variant...> -- the field-option workloads (T04_s, IGRF on config[2]'s set and grid) for library variants, one call
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "$@"; do
  for w in ${WL:-interp_t04_64k interp_igrf200k}; do
    SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_$v.so timeout -k 10 400 python bench.py --workload $w ${RAYS:+--rays $RAYS} --steps 2 --warmup 1 --traffic off --other-configs 0 --cpu-seconds 0 --damping-rays 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v $w', 'kernel_ms %.1f steps/s %.4g' % (d['roofline']['kernel_ms'], d['value']))"
  done
done

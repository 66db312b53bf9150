#!/bin/bash
# tools/scratch/ab_scat_full.sh <variant...> -- config[4] at FULL size (1M rays) for library variants; env MARGINS="0.125 0.08 .." sweeps the
# candidate blocks' margin (SRT_SCATTERED_MARGIN) for the first variant at 200k rays
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for m in $MARGINS; do
  SRT_SCATTERED_MARGIN=$m SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_$1.so timeout -k 10 400 python bench.py --traffic off --other-configs 0 --workload scattered825k --rays 200000 --steps 1 --warmup 1 --cpu-seconds 0 --damping-rays 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1 margin=$m 200k rays', 'kernel_ms %.1f steps/s %.4g' % (d['roofline']['kernel_ms'], d['value']))"
done
for v in "$@"; do
  SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_$v.so timeout -k 10 400 python bench.py --traffic off --other-configs 0 --workload scattered825k --steps 1 --warmup 1 --cpu-seconds 0 --damping-rays 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v 1M rays', 'kernel_ms %.1f steps/s %.4g' % (d['roofline']['kernel_ms'], d['value']))"
done

#!/bin/bash
# tools/scratch/ab_scat3.sh <variant...> -- config[4] A/B in ONE gpurun call (boxes differ by several %): for every library variant
# (stanford_raytracer_amd/lib/libsrt_hip_<v>.so, built by tools/ab_build.sh) the scattered825k workload at RAYS rays
# (default 100000) with the candidate blocks on / off and 8 / 4 waves per CU.  Prints kernel ms and steps/s.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
RAYS=${RAYS:-100000}
for v in "$@"; do
  for cfg in ${CFGS:-1:8 0:8 1:4 0:4}; do
    B=${cfg%%:*}; W=${cfg##*:}
    SRT_SCATTERED_BLOCKS=$B SRT_WAVES_PER_CU=$W SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_$v.so timeout -k 10 400 python bench.py --traffic off --other-configs 0 --workload scattered825k --rays $RAYS --steps 1 --warmup 1 --cpu-seconds 0 --damping-rays 0 2>gpurun_out/ab3_err.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v blocks=$B waves/CU=$W', 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'steps/s %.4g' % d['value'], int(d['roofline']['accepted_steps_per_launch']))" || { tail -5 gpurun_out/ab3_err.log; exit 1; }
  done
done

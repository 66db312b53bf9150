#!/bin/bash
# tools/pmc_valu.sh <variant...> -- VALU instruction count and busy cycles of the scattered trace kernel per library variant (one
# PMC pass each, RAYS rays; launch times on this pool differ by +-2.5 % run to run, instruction counts do not)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
RAYS=${RAYS:-100000}
for v in "$@"; do
  O=$R/gpurun_out/pmc_valu_$v; rm -rf $O; mkdir -p $O
  SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_$v.so timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/p -- python3 bench.py --traffic off --other-configs 0 --workload scattered825k --rays $RAYS --steps 1 --warmup 0 --cpu-seconds 0 --damping-rays 0 > $O/p.log 2>&1 || exit 1
  grep -h trace_kernel $O/p/*/*counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}' | sed 's/"//g' | tr '\n' ' ' | sed "s/^/$v: /"; echo
  python -c "import json; d=json.loads(open('$O/p.log').readlines()[-1]); print('$v accepted', int(d['roofline']['accepted_steps_per_launch']), 'kernel_ms %.1f' % d['roofline']['kernel_ms'])"
done

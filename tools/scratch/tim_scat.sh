#!/bin/bash
# tools/scratch/tim_scat.sh <variant...> -- cycle breakdown of coop_stencil (variants built with -DSRT_PHASE_TIMING), 100 k rays
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "$@"; do
  SRT_PHASE_TIMING=1 SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_$v.so timeout -k 10 300 python bench.py --workload scattered825k --rays 100000 --steps 1 --warmup 0 --cpu-seconds 0 --damping-rays 0 --traffic off --other-configs 0 > gpurun_out/tim_$v.log 2>&1 || exit 1
  echo "== $v"; grep -a "phase cycles" gpurun_out/tim_$v.log | python -c "
import sys
h=[int(x) for x in sys.stdin.read().split(':')[1].split()]
n=h[8]; names=['scan','pass1','reduce','base','pass2','solve','handoff','own','count','shared_total','n_list','cands_or_rebuilds','block_len','direct_calls','d7_metres','rebuild_cycles']
print(' '.join(('%s=%.3f' if names[i] in ('cands_or_rebuilds','direct_calls') else '%s=%.0f')%(names[i],h[i]/n) for i in range(len(h)) if i!=8), 'stencils=%d'%n)"
  python -c "import sys,json; d=json.loads(open('gpurun_out/tim_$v.log').readlines()[-1]); print('kernel_ms', round(d['roofline']['kernel_ms'],1), 'steps/s %.4g' % d['value'])"
done

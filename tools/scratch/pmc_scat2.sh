#!/bin/bash
# tools/scratch/pmc_scat2.sh <variant...> -- stall-side counters of the scattered trace kernel (A/B of library variants)
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
RAYS=${RAYS:-100000}
for v in "$@"; do
  O=$R/gpurun_out/pmc2_scat_$v; mkdir -p $O
  for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_FLAT SQ_IFETCH"; do
    N=$(echo $C | cut -d' ' -f1)
    SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_$v.so timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$N -- python3 bench.py --traffic off --other-configs 0 --workload scattered825k --rays $RAYS --steps 1 --warmup 0 --cpu-seconds 0 --damping-rays 0 > $O/$N.log 2>&1 || { tail -3 $O/$N.log; continue; }
    echo "== $v"; grep -h trace_kernel $O/$N/*/*counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}' | sed 's/"//g'
  done
done

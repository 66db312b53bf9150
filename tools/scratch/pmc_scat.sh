#!/bin/bash
# tools/scratch/pmc_scat.sh <variant...> -- instruction-mix counters of the scattered trace kernel for library variants (A/B)
# env: RAYS (default 100000), WAVES (SRT_WAVES_PER_CU, default unset)
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
RAYS=${RAYS:-100000}
for v in "$@"; do
  O=$R/gpurun_out/pmc_scat_$v; mkdir -p $O
  for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    N=$(echo $C | cut -d' ' -f1)
    SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_$v.so timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$N -- python3 bench.py --traffic off --other-configs 0 --workload scattered825k --rays $RAYS --steps 1 --warmup 0 --cpu-seconds 0 --damping-rays 0 > $O/$N.log 2>&1
    echo "== $v"; grep -h trace_kernel $O/$N/*/*counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}' | sed 's/"//g'
  done
done

#!/bin/bash
# tools/scratch/pmc_scat3.sh <variant...> -- memory-side counters of the scattered trace kernel
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
RAYS=${RAYS:-100000}
for v in "$@"; do
  O=$R/gpurun_out/pmc3_scat_$v; mkdir -p $O
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_BUSY_avr GRBM_GUI_ACTIVE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
    N=$(echo $C | cut -d' ' -f1)
    SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_$v.so timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$N -- python3 bench.py --traffic off --other-configs 0 --workload scattered825k --rays $RAYS --steps 1 --warmup 0 --cpu-seconds 0 --damping-rays 0 > $O/$N.log 2>&1 || { echo "FAILED $N"; tail -2 $O/$N.log; continue; }
    echo "== $v"; grep -h trace_kernel $O/$N/*/*counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}' | sed 's/"//g'
  done
done

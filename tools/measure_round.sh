#!/bin/bash
# tools/measure_round.sh <tag> -- the round's evidence in ONE gpurun call (run on the GPU box):
#   GPU parity tests, the default bench line (live PMC traffic, other_configs, CPU baselines), the rocprofv3 kernel
#   stats of the same workload, and instruction-mix PMC passes.  Outputs under gpurun_out/<tag>/ ; copy what is to
#   be judged into profiles/<round>/.
set -e
TAG=${1:-meas}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 && tail -2 $O/pytest_gpu.log
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>$O/bench.err && tail -1 $O/bench.log | cut -c1-600
cp gpurun_out/traffic_interp256.json $O/ 2>/dev/null || true
# kernel statistics of the same workload (the bench's own child passes would nest profilers: --traffic off)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 1 --traffic off --other-configs 0 --cpu-seconds 0 --damping-rays 0 > $O/stats.log 2>&1
cat $O/stats/*/*kernel_stats.csv | head -5 | cut -c1-220
for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" \
         "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$N -- python3 bench.py --pmc-child --steps 1 --warmup 0 > $O/pmc_$N.log 2>&1
  grep -h trace_kernel $O/pmc_$N/*/*counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}' | sed 's/"//g' | tee -a $O/pmc_summary.txt
done

#!/bin/bash
# tools/measure_round.sh <tag> -- the round's evidence in ONE gpurun call (run on the GPU box):
#   GPU parity tests, the default bench line (live PMC traffic, other_configs, CPU baselines), and for EVERY workload that
#   carries a roofline in that line (interp256, scattered825k, ngo100k, interp_igrf200k, interp_t04_64k) the rocprofv3
#   kernel stats of one launch and an instruction-mix PMC pass.  Outputs under gpurun_out/<tag>/ ; copy what is to be
#   judged into profiles/<round>/.
TAG=${1:-meas}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
fi
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>$O/bench.err && tail -1 $O/bench.log | cut -c1-400
cp gpurun_out/traffic_interp256.json $O/ 2>/dev/null || true
for W in ${WLS:-interp256 scattered825k ngo100k interp_igrf200k interp_t04_64k}; do
  # kernel statistics of one launch of the workload (the bench's own child passes would nest profilers: --traffic off)
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$W -- python3 bench.py --workload $W --steps 2 --warmup 1 --traffic off --other-configs 0 --cpu-seconds 0 --damping-rays 0 > $O/stats_$W.log 2>&1
  echo "== $W"; cat $O/stats_$W/*/*kernel_stats.csv | head -3 | cut -c1-260
  for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    N=$(echo $C | cut -d' ' -f1)
    timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${W}_$N -- python3 bench.py --pmc-child --workload $W --steps 1 --warmup 0 > $O/pmc_${W}_$N.log 2>&1
    grep -h trace_kernel $O/pmc_${W}_$N/*/*counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}' | sed 's/"//g' | sed "s/^/$W /" | tee -a $O/pmc_summary.txt
  done
done

#!/bin/bash
# tools/measure_round.sh <tag> [workload] -- the round's evidence in ONE gpurun call (run on the GPU box):
#   GPU parity tests, the default bench line (with the CPU baseline), rocprofv3 kernel stats of the same command,
#   and the PMC passes (FETCH_SIZE / WRITE_SIZE separately, then instruction-mix counters).
# Outputs under gpurun_out/<tag>/ ; copy what is to be judged into profiles/.
set -e
TAG=${1:-meas}; WL=${2:-interp256}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 && tail -2 $O/pytest_gpu.log
timeout -k 10 600 python bench.py --workload $WL > $O/bench.log 2>$O/bench.err && tail -1 $O/bench.log | cut -c1-400
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload $WL --steps 2 --warmup 1 --cpu-seconds 0 > $O/stats.log 2>&1
cat $O/stats/*/*kernel_stats.csv | head -4 | cut -c1-200
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" \
         "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_SALU"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$N -- python3 bench.py --workload $WL --steps 1 --warmup 0 --cpu-seconds 0 > $O/pmc_$N.log 2>&1
  grep -h trace_kernel $O/pmc_$N/*/*counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}' | sed 's/"//g'
done
cd $R && python3 tools/traffic_json.py $O $WL && cp profiles/traffic_$WL.json $O/

#!/bin/bash
# tools/collect_profiles.sh <gpurun_out tag> <round dir, e.g. r04> <version, e.g. v30> -- copy what tools/measure_round.sh left under
# gpurun_out/<tag>/ into the tracked profiles/<round>/ (file names as in the earlier rounds)
set -e
T=gpurun_out/$1; P=profiles/$2; V=$3
mkdir -p $P
tail -1 $T/bench.log > $P/bench_default_$V.json
tail -3 $T/pytest_gpu.log > $P/pytest_gpu_$V.txt
cp $T/pmc_summary.txt $P/pmc_all_workloads_$V.txt
cp $T/traffic_interp256.json $P/traffic_interp256_$V.json
cp $T/traffic_interp256.json profiles/traffic_interp256.json
for W in interp256 scattered825k ngo100k interp_igrf200k interp_t04_64k; do
  cat $T/stats_$W/*/*kernel_stats.csv > $P/kernel_stats_${W}_$V.csv
done
ls -la $P | tail -12

#!/bin/bash
# tools/scratch/ab_tail.sh -- the launch's tail at a strong-scaling shard size (500k rays of config[2]): ray_order 1 (Morton) against 2
# (likely-short rays last), two rounds in one gpurun call
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for round in 1 2; do
  for o in 1 2; do
    timeout -k 10 300 python bench.py --rays ${RAYS:-500000} --ray-order $o --steps 3 --warmup 1 --cpu-seconds 0 --damping-rays 0 --traffic off --other-configs 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ray_order=$o kernel_ms %.2f steps/s %.4g lane_occupancy %.4f' % (d['roofline']['kernel_ms'], d['value'], d['detail']['lane_occupancy']))"
  done
done

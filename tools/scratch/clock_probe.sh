#!/bin/bash
# tools/scratch/clock_probe.sh -- the shader clock and power the GPU reports (rocm-smi) while a workload's launch is running:
# evidence for which kernels are power-bound (HISTORY.md section 2.5)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for spec in "scattered825k 300000" "interp256 1000000" "ngo100k 1000000"; do
  set -- $spec
  python bench.py --workload $1 --rays $2 --steps ${STEPS:-6} --warmup 1 --traffic off --other-configs 0 --cpu-seconds 0 --damping-rays 0 > /tmp/clk_$1.log 2>/dev/null &
  PID=$!
  sleep ${WAIT:-25}
  echo "== $1 ($2 rays) while its launches run"
  for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | sed 's/  */ /g' | tr '\n' ';'; echo; sleep 1.5; done
  wait $PID
  tail -1 /tmp/clk_$1.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('   kernel_ms %.1f' % d['roofline']['kernel_ms'])"
done

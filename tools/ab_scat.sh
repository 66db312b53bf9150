set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "$@"; do
  SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_$v.so timeout -k 10 300 python bench.py --traffic off --other-configs 0 --workload scattered825k --rays 100000 --steps 1 --warmup 1 --cpu-seconds 0 --damping-rays 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'steps/s %.4g' % d['value'], int(d['roofline']['accepted_steps_per_launch']))"
done

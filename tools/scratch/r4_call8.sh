#!/bin/bash
# round-4 call 8: right-hand sides spread over idle lanes in the INTERP kernel's tail (-DSRT_INTERP_SPREAD=1): interp parity tests
# under that build, then the headline workload at 500 k rays (an 8-GPU strong-scaling shard) and 1 M rays, A/B twice
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c8
SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_ispread.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trace.py tests/test_gcpm_golden.py tests/test_root1_golden.py -m gpu -q -k "interp and not full_size" > gpurun_out/c8/pytest_ispread.log 2>&1; tail -2 gpurun_out/c8/pytest_ispread.log
for round in 1 2; do
  for rays in 500000 1000000; do
    for v in base ispread; do
      LIBENV=""; [ "$v" != "base" ] && LIBENV="SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_$v.so"
      env $LIBENV timeout -k 10 300 python bench.py --rays $rays --steps 5 --warmup 2 --traffic off --other-configs 0 --cpu-seconds 0 --damping-rays 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v rays $rays', 'kernel_ms %.2f steps/s %.4g occupancy %.4f' % (d['roofline']['kernel_ms'], d['value'], d['detail']['lane_occupancy']))"
    done
  done
done 2>&1 | tee gpurun_out/c8/ab.txt

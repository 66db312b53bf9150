#!/bin/bash
# tools/scat_exp.sh "name|libvariant|ENV=val ENV2=val" ... -- config[4] A/B in ONE gpurun call (boxes differ by several %):
# for every configuration the scattered825k workload at RAYS rays (default 100000): kernel ms (TIMES launches), then one PMC
# pass each for FETCH_SIZE, WRITE_SIZE and SQ_INSTS_VALU (skipped with PMC=0).  libvariant "-" = the in-tree library, else
# stanford_raytracer_amd/lib/libsrt_hip_<variant>.so (tools/ab_build.sh).  Prints one line per configuration.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
RAYS=${RAYS:-100000}; PMC=${PMC:-1}; TIMES=${TIMES:-2}
ARGS="--traffic off --other-configs 0 --workload scattered825k --rays $RAYS --cpu-seconds 0 --damping-rays 0"
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; v=${rest%%|*}; envs=${rest#*|}
  [ "$envs" = "$rest" ] && envs=""
  LIBENV=""; [ "$v" != "-" ] && LIBENV="SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_$v.so"
  O=$R/gpurun_out/exp_$name; rm -rf $O; mkdir -p $O
  env $envs $LIBENV timeout -k 10 600 python bench.py $ARGS --steps $TIMES --warmup 1 > $O/time.log 2>$O/time.err || { echo "$name: timing run FAILED"; tail -3 $O/time.err; continue; }
  line=$(python -c "import json; d=json.loads(open('$O/time.log').readlines()[-1]); print('kernel_ms %.1f steps/s %.4g accepted %d' % (d['roofline']['kernel_ms'], d['value'], d['roofline']['accepted_steps_per_launch']))")
  pm=""
  if [ "$PMC" != "0" ]; then
    for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
      N=$(echo $C | cut -d' ' -f1)
      env $envs $LIBENV timeout -k 10 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$N -- python3 bench.py $ARGS --steps 1 --warmup 0 > $O/$N.log 2>&1 || { pm="$pm $N=FAILED"; continue; }
      pm="$pm $(grep -h trace_kernel $O/$N/*/*counter_collection.csv | awk -F, '{print $(NF-3)"="$(NF-2)}' | sed 's/"//g' | tr '\n' ' ')"
    done
  fi
  echo "$name: $line $pm"
done

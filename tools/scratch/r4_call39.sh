#!/bin/bash
# round-4 call 39 (information for the next round, nothing committed from it): the stencil service's wave-uniform values in scalar
# registers (libsrt_hip_unic.so) against the round's final library, rows hashed, A/B at 200 k rays
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c39
timeout -k 10 300 python tools/scratch/scat_rows_hash.py > gpurun_out/c39/hash_final.log 2>&1 &&
SRT_LIB_OVERRIDE=$R/stanford_raytracer_amd/lib/libsrt_hip_unic.so timeout -k 10 300 python tools/scratch/scat_rows_hash.py > gpurun_out/c39/hash_unic.log 2>&1 &&
cmp gpurun_out/c39/hash_final.log gpurun_out/c39/hash_unic.log && echo "HASHES EQUAL" &&
RAYS=200000 PMC=0 TIMES=2 bash tools/scat_exp.sh "final|-" "unic|unic" "finalb|-" "unicb|unic" "finalc|-" "unicc|unic"

#!/bin/bash
# tools/probes/build.sh [name...] -- compile the hardware probes for gfx950 next to their sources (binaries are git-ignored)
set -e
cd "$(dirname "$0")"
for src in ${@:-*.hip}; do
  src=${src%.hip}.hip
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -I../../stanford_raytracer_amd/csrc -I../../include -o ${src%.hip} $src
done

#!/bin/bash
# tools/ab_build.sh NAME "<extra hipcc flags>"  -> stanford_raytracer_amd/lib/libsrt_hip_NAME.so
set -e
cd /root/repo
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared $2 -o stanford_raytracer_amd/lib/libsrt_hip_$1.so stanford_raytracer_amd/csrc/srt_api.hip stanford_raytracer_amd/csrc/srt_host.cpp stanford_raytracer_amd/csrc/srt_scattered_host.cpp

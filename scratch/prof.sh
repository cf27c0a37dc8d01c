#!/bin/bash
# in-kernel phase timing: builds an instrumented copy of the library (-DCAGPU_ABLATE) next to the product one
set -e
cp gym_collision_avoidance_amd/libcagpu.so /tmp/libcagpu_product.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared -DCAGPU_ABLATE -Iinclude gym_collision_avoidance_amd/csrc/cagpu.hip -o gym_collision_avoidance_amd/libcagpu.so
python scratch/prof_phases.py; python scratch/wgprof.py
cp /tmp/libcagpu_product.so gym_collision_avoidance_amd/libcagpu.so

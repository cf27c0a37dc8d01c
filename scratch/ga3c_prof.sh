#!/bin/bash
cp gym_collision_avoidance_amd/libcagpu.so /tmp/libcagpu_product.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-pass-failed -fPIC -shared -DCAGPU_ABLATE -Iinclude gym_collision_avoidance_amd/csrc/cagpu.hip -o gym_collision_avoidance_amd/libcagpu.so 2>/dev/null
python scratch/ga3c_prof.py
cp /tmp/libcagpu_product.so gym_collision_avoidance_amd/libcagpu.so

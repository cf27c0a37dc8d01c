#!/bin/bash
# GA3C kernel: 32- vs 64-row tiles (instrumented builds next to the product library)
cp gym_collision_avoidance_amd/libcagpu.so /tmp/libcagpu_product.so
for tm in 32 64; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-pass-failed -fPIC -shared -DGA3C_TM=$tm -Iinclude gym_collision_avoidance_amd/csrc/cagpu.hip -o gym_collision_avoidance_amd/libcagpu.so 2>/dev/null
  echo "== TM=$tm"; python scratch/ga3c_split.py | tail -7
done
cp /tmp/libcagpu_product.so gym_collision_avoidance_amd/libcagpu.so

#!/bin/bash
# compiler-flag experiments on the whole library (instrumented copies next to the product build)
cp gym_collision_avoidance_amd/libcagpu.so /tmp/libcagpu_product.so
BASE="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-pass-failed -mllvm -disable-machine-licm -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared -Iinclude gym_collision_avoidance_amd/csrc/cagpu.hip -o gym_collision_avoidance_amd/libcagpu.so"
python bench.py --steps 200 --warmup 50 --no-cpu-baseline > /dev/null 2>&1
for extra in "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-schedule-relaxed-occupancy=true" "-mllvm -enable-post-misched=false" "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy"; do
  if /opt/rocm/bin/hipcc $BASE $extra 2>/tmp/flags.err; then
    r=$(timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), round(d['rollout']['ms_per_step']*1e3,2), d['episode_stats']['episodes'])")
    echo "[$extra] step / rollout us, episodes: $r"
  else
    echo "[$extra] does not compile: $(tail -1 /tmp/flags.err | cut -c1-120)"
  fi
done
cp /tmp/libcagpu_product.so gym_collision_avoidance_amd/libcagpu.so

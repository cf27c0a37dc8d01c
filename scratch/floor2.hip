// scratch/floor2.hip -- launch floor by workgroup geometry, and what the store policy of the big output costs at the
// END of a launch: each launch writes 11.5 MB (the observation block of 4096 x 10 agents) with (a) plain stores,
// (b) non-temporal stores, (c) system-scope (write-through) stores; back-to-back launches on one stream, HIP events.
// build: hipcc --offload-arch=gfx950 -O2 scratch/floor2.hip -o scratch/floor2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NT>
__global__ __launch_bounds__(NT) void k_empty(float*, int) {
  extern __shared__ unsigned char smem[];
  if (threadIdx.x == 10000) smem[0] = 1;
}
// every workgroup writes `per_wg` floats, mode 0 plain / 1 nontemporal / 2 system-scope relaxed atomic store
template <int NT>
__global__ __launch_bounds__(NT) void k_store(float* out, int mode_per) {
  extern __shared__ unsigned char smem[];
  const int mode = mode_per >> 24, per_wg = mode_per & 0xFFFFFF;
  float* dst = out + (size_t)blockIdx.x * per_wg;
  for (int q = threadIdx.x; q < per_wg; q += NT) {
    const float v = (float)q;
    if (mode == 0) dst[q] = v;
    else if (mode == 1) __builtin_nontemporal_store(v, dst + q);
    else __hip_atomic_store(dst + q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (threadIdx.x == 10000) smem[0] = 1;
}

template <typename K>
float run(K kern, float* buf, int arg, int grid, int nt, int lds, int n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), lds, 0, buf, arg);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), lds, 0, buf, arg);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / n;
}

int main() {
  const int grid = 1024;
  const int per_wg = 40 * 69;  // floats of one 4-env tile's observation rows
  float* buf;
  hipMalloc(&buf, (size_t)grid * per_wg * 4 * 2);
  printf("empty 1024 x 256, 24 KB LDS   %.2f us\n", run(k_empty<256>, buf, 0, grid, 256, 24 * 1024, 3000));
  printf("empty 1024 x 512, 34 KB LDS   %.2f us\n", run(k_empty<512>, buf, 0, grid, 512, 34 * 1024, 3000));
  printf("empty 1024 x 512,  1 KB LDS   %.2f us\n", run(k_empty<512>, buf, 0, grid, 512, 1024, 3000));
  printf("empty  512 x 512, 34 KB LDS   %.2f us\n", run(k_empty<512>, buf, 0, 512, 512, 34 * 1024, 3000));
  printf("empty 2048 x 256, 17 KB LDS   %.2f us\n", run(k_empty<256>, buf, 0, 2048, 256, 17 * 1024, 3000));
  for (int kb : {1, 8, 16, 20, 24, 28, 31, 32, 33, 36, 40})
    printf("empty 1024 x 512, %2d KB LDS   %.2f us\n", kb, run(k_empty<512>, buf, 0, grid, 512, kb * 1024, 3000));
  for (int g : {256, 512, 768, 1024, 1280, 2048})
    printf("empty %4d x 512, 33 KB LDS   %.2f us\n", g, run(k_empty<512>, buf, 0, g, 512, 33 * 1024, 3000));
  for (int mode = 0; mode < 3; ++mode)
    printf("1024 x 512 storing 11.3 MB, mode %d (0 plain, 1 nontemporal, 2 system scope)   %.2f us\n", mode,
           run(k_store<512>, buf, (mode << 24) | per_wg, grid, 512, 34 * 1024, 3000));
  for (int mode = 0; mode < 3; ++mode)
    printf("1024 x 256 storing 11.3 MB, mode %d   %.2f us\n", mode, run(k_store<256>, buf, (mode << 24) | per_wg, grid, 256, 24 * 1024, 3000));
  return 0;
}

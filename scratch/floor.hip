// scratch/floor.hip -- launch floor of the step kernel's geometry on MI355X: back-to-back launches (one stream) of
//   (a) an empty kernel, (b) a kernel that only loads the 16 state arrays of its 40 agents on wave 0 and stores 9 back,
//   (c) the same + a dependent chain of `spin` f64 FMAs on wave 0 between load and store (a stand-in for the step),
// each with 1024 workgroups x 256 threads and 24 KB of LDS (4 workgroups per CU).  HIP events over 2000 launches.
// build: hipcc --offload-arch=gfx950 -O2 scratch/floor.hip -o scratch/floor ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Arr { double* a[16]; };

__global__ __launch_bounds__(256) void k_empty(Arr, int) {
  extern __shared__ unsigned char smem[];
  if (threadIdx.x == 1000) smem[0] = 1;
}
__global__ __launch_bounds__(256) void k_ldst(Arr s, int spin) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x;
  if (lane < 40) {
    const long i = (long)blockIdx.x * 40 + lane;
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = s.a[q][i];
    double acc = v[0];
    if (spin > 0) {
#pragma unroll 8
      for (int q = 0; q < spin; ++q) acc = __builtin_fma(acc, 1.0000001, v[1]);
    } else if (spin < -100000) {  // four independent float chains, interleaved (does a wave issue them back to back?)
      float f0 = (float)acc, f1 = f0 + 1.f, f2 = f0 + 2.f, f3 = f0 + 3.f, fb = (float)v[1];
#pragma unroll 8
      for (int q = 0; q < -spin - 100000; ++q) {
        f0 = __builtin_fmaf(f0, 1.0000001f, fb); f1 = __builtin_fmaf(f1, 1.0000001f, fb);
        f2 = __builtin_fmaf(f2, 1.0000001f, fb); f3 = __builtin_fmaf(f3, 1.0000001f, fb);
      }
      acc = (f0 + f1) + (f2 + f3);
    } else if (spin < 0) {  // float chain
      float fa = (float)acc, fb = (float)v[1];
#pragma unroll 8
      for (int q = 0; q < -spin; ++q) fa = __builtin_fmaf(fa, 1.0000001f, fb);
      acc = fa;
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) s.a[q][i] = v[q] + (spin ? acc * 1e-300 : 0.0);
  }
  __syncthreads();
  if (threadIdx.x == 1000) smem[0] = 1;
}

template <typename K>
float run(K kern, Arr s, int spin, int grid, int lds, int n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, s, spin);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, s, spin);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / n;
}

// the same empty kernel as a captured graph of n kernel nodes, replayed
float run_graph(Arr s, int grid, int lds, int n, int reps) {
  hipStream_t st;
  hipStreamCreate(&st);
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), lds, st, s, 0);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
  hipEventRecord(e1, st);
  hipStreamSynchronize(st);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / (n * reps);
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 1024;
  const int lds = 24 * 1024;
  Arr s;
  for (int q = 0; q < 16; ++q) { hipMalloc(&s.a[q], (size_t)grid * 40 * 8); hipMemset(s.a[q], 0, (size_t)grid * 40 * 8); }
  printf("grid %d x 256 threads, %d B LDS\n", grid, lds);
  printf("empty kernel              %.2f us / launch\n", run(k_empty, s, 0, grid, lds, 2000));
  printf("empty kernel, graph of 200   %.2f us / kernel node\n", run_graph(s, grid, lds, 200, 10));
  printf("load 16 + store 9 arrays  %.2f us / launch\n", run(k_ldst, s, 0, grid, lds, 2000));
  for (int spin : {1000, 4000, -1000, -4000, -101000, -104000}) printf("  + %5d dependent FMAs (negative: f32) %.2f us / launch\n", spin, run(k_ldst, s, spin, grid, lds, 2000));
  return 0;
}

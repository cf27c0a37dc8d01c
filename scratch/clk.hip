#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float* out, unsigned long long* t, int n) {
  float x = out[0];
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < n; ++i) x = __builtin_fmaf(x, 1.0000001f, 0.5f);
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  out[threadIdx.x + blockIdx.x * blockDim.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
int main() {
  float* d; unsigned long long* t; hipMalloc(&d, 1 << 24); hipMalloc(&t, 64);
  hipMemset(d, 0, 1 << 24);
  for (int grid : {1, 256, 683, 2048, 8192}) for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int n = 2000000;
    hipEventRecord(a); hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, 0, d, t, n); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("grid %5d: %.3f ms, clock64 ticks %llu (%.3f GHz), wall_clock64 ticks %llu (%.1f MHz), cycles/fma %.2f\n", grid, ms, h[0], h[0] / (ms * 1e6), h[1], h[1] / (ms * 1e3), (double)h[0] / n);
  }
  int clk; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0); printf("attr clock %d kHz\n", clk);
  return 0;
}

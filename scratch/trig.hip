#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s\n", hipGetErrorString(e_)); return 1; } } while (0)
template <int OP>
__global__ void bench(double* out, unsigned long long* t, int n) {
  double x = out[threadIdx.x] + 0.3 + 0.01 * threadIdx.x, y = 0.7;
  unsigned long long c0 = clock64();
  for (int i = 0; i < n; ++i) {
    if (OP == 0) x = atan2(y, x) + 0.5;
    if (OP == 1) { double s, c; sincos(x, &s, &c); x = s + c; }
    if (OP == 2) x = sqrt(x * x + y) ;
    if (OP == 3) x = y / (x + 1.5);
    if (OP == 4) x = (double)atan2f((float)y, (float)x) + 0.5;
    if (OP == 5) { float s, c; sincosf((float)x, &s, &c); x = s + c; }
    if (OP == 6) x = x * 1.0000001 + 0.5;
    if (OP == 7) { float f = (float)x; f = sqrtf(f * f + 0.7f); x = f; }
    if (OP == 8) { float f = (float)x; f = 0.7f / (f + 1.5f); x = f; }
  }
  unsigned long long c1 = clock64();
  out[threadIdx.x + blockIdx.x * blockDim.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = c1 - c0;
}
int main() {
  double* d; unsigned long long* t; CHK(hipMalloc(&d, 1 << 22)); CHK(hipMalloc(&t, 64)); CHK(hipMemset(d, 0, 1 << 22));
  const char* names[] = {"atan2 f64", "sincos f64", "sqrt f64 (+mul,add)", "div f64 (+add)", "atan2f", "sincosf", "fma f64", "sqrtf (+cvt)", "divf (+cvt)"};
  int n = 20000;
  for (int waves = 1; waves <= 4; waves *= 2) for (int op = 0; op < 9; ++op) {
    dim3 g(256), b(64 * waves);   // waves per CU (one block per CU)
    switch (op) {
      case 0: hipLaunchKernelGGL(bench<0>, g, b, 0, 0, d, t, n); break; case 1: hipLaunchKernelGGL(bench<1>, g, b, 0, 0, d, t, n); break;
      case 2: hipLaunchKernelGGL(bench<2>, g, b, 0, 0, d, t, n); break; case 3: hipLaunchKernelGGL(bench<3>, g, b, 0, 0, d, t, n); break;
      case 4: hipLaunchKernelGGL(bench<4>, g, b, 0, 0, d, t, n); break; case 5: hipLaunchKernelGGL(bench<5>, g, b, 0, 0, d, t, n); break;
      case 6: hipLaunchKernelGGL(bench<6>, g, b, 0, 0, d, t, n); break; case 7: hipLaunchKernelGGL(bench<7>, g, b, 0, 0, d, t, n); break;
      case 8: hipLaunchKernelGGL(bench<8>, g, b, 0, 0, d, t, n); break;
    }
    CHK(hipDeviceSynchronize());
    unsigned long long h; CHK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
    printf("waves/CU %d  %-22s %8.1f cycles per call\n", waves, names[op], (double)h / n);
  }
  return 0;
}

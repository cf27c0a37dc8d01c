// scratch/f16_denorm.hip -- does v_mfma_f32_16x16x32_f16 keep fp16 denormal inputs, and does v_cvt_f16_f32 produce them?
// (the question behind a two-plane fp16 split of float32 operands: the low plane of a value below 0.125 is an fp16 denormal)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out, float tiny) {
  const int lane = threadIdx.x;
  f16x8 a, b;
  const _Float16 t = static_cast<_Float16>(tiny);   // v_cvt_f16_f32 of 2^-20: an fp16 denormal (min normal 2^-14)
  for (int e = 0; e < 8; ++e) { a[e] = (e == 0 && lane < 16) ? t : static_cast<_Float16>(0.f); b[e] = (e == 0 && lane < 16) ? static_cast<_Float16>(1.0f) : static_cast<_Float16>(0.f); }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (lane == 0) { out[0] = c[0]; out[1] = static_cast<float>(t); }
}
int main() {
  float* d; hipMalloc(&d, 16);
  k<<<1, 64>>>(d, 9.5367431640625e-07f);
  float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  printf("fp16(2^-20) back as float: %.9g (expected 9.53674316e-07); MFMA(2^-20 * 1.0) = %.9g\n", h[1], h[0]);
  k<<<1, 64>>>(d, 3.0e-06f);
  hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  printf("fp16(3e-6) back as float: %.9g; MFMA = %.9g\n", h[1], h[0]);
  return 0;
}

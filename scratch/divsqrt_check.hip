// scratch/divsqrt_check.hip -- the lean correctly rounded float divide / square root of cagpu.hip (the compiler's own
// sequences minus their range handling) against the compiler's `/` and sqrtf (-fhip-fp32-correctly-rounded-divide-sqrt)
// on 2^28 random operands per range.  build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off
//   -fhip-fp32-correctly-rounded-divide-sqrt scratch/divsqrt_check.hip -o scratch/divsqrt_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float div_lean(float a, float b) {
  float y = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, y, 1.0f);
  y = __builtin_fmaf(e, y, y);
  float q = a * y;
  float r = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(r, y, q);
  r = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(r, y, q);
}
__device__ __forceinline__ float sqrt_lean(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float s_dn = __int_as_float(__float_as_int(s) - 1), s_up = __int_as_float(__float_as_int(s) + 1);
  const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
  float o = (r_dn <= 0.0f) ? s_dn : s;
  o = (r_up > 0.0f) ? s_up : o;
  return (x == 0.0f) ? x : o;
}
__device__ __forceinline__ double divd(double a, double b) {
  double y = __builtin_amdgcn_rcp(b);
  y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
  const double q = a * y;
  return __builtin_fma(__builtin_fma(-b, q, a), y, q);
}
__device__ __forceinline__ double sqrtd(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  return (x == 0.0) ? x : g;
}
__device__ __forceinline__ double mkd(uint32_t h, uint32_t h2, int elo, int ehi) {
  const unsigned long long man = (static_cast<unsigned long long>(h & 0xFFFFFu) << 32) | h2, sgn = static_cast<unsigned long long>(h >> 31) << 63;
  const unsigned long long ex = static_cast<unsigned long long>(elo + static_cast<int>((h >> 20) & 0x7FF) % (ehi - elo + 1) + 1023);
  return __longlong_as_double(static_cast<long long>(sgn | (ex << 52) | man));
}
__device__ unsigned long long g_bad[4];
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// operands: sign random, exponent uniform in [elo, ehi], mantissa random
__device__ __forceinline__ float mk(uint32_t h, int elo, int ehi) {
  const uint32_t man = h & 0x7FFFFFu, sgn = (h >> 31) << 31;
  const uint32_t ex = static_cast<uint32_t>(elo + static_cast<int>((h >> 23) & 0xFF) % (ehi - elo + 1) + 127);
  return __uint_as_float(sgn | (ex << 23) | man);
}
__global__ void k(int elo, int ehi, unsigned seed) {
  const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  const uint32_t h0 = mix(static_cast<uint32_t>(i) * 2654435761u + seed), h1 = mix(h0 + 0x9E3779B9u), h2 = mix(h1 ^ 0x85ebca6bu);
  const float a = mk(h0, elo, ehi), b = mk(h1, elo, ehi), x = fabsf(mk(h2, 2 * elo, 2 * ehi > 120 ? 120 : 2 * ehi));
  const float q0 = a / b, q1 = div_lean(a, b);
  const float s0 = sqrtf(x), s1 = sqrt_lean(x);
  const float r0 = 1.0f / b, r1 = div_lean(1.0f, b);
  if (__float_as_uint(q0) != __float_as_uint(q1)) atomicAdd(&g_bad[0], 1ull);
  if (__float_as_uint(s0) != __float_as_uint(s1)) atomicAdd(&g_bad[1], 1ull);
  if (__float_as_uint(r0) != __float_as_uint(r1)) atomicAdd(&g_bad[2], 1ull);
  // float64: exponents 5 x as wide
  const double da = mkd(mix(h2 + 1u), mix(h2 + 2u), 5 * elo, 5 * ehi), db = mkd(mix(h2 + 3u), mix(h2 + 4u), 5 * elo, 5 * ehi);
  const double dx = fabs(mkd(mix(h2 + 5u), mix(h2 + 6u), 8 * elo, 8 * ehi > 1000 ? 1000 : 8 * ehi));
  if (__double_as_longlong(da / db) != __double_as_longlong(divd(da, db))) atomicAdd(&g_bad[3], 1ull);
  if (__double_as_longlong(sqrt(dx)) != __double_as_longlong(sqrtd(dx))) atomicAdd(&g_bad[3], 1ull << 32);
}
int main() {
  const int ranges[][2] = {{-20, 10}, {-40, 40}, {-60, 60}, {-126, 127}};
  for (auto& rg : ranges) {
    unsigned long long z[4] = {0, 0, 0, 0}, h[4];
    hipMemcpyToSymbol(HIP_SYMBOL(g_bad), z, sizeof(z));
    for (unsigned rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(k, dim3(1 << 18), dim3(256), 0, 0, rg[0], rg[1], rep * 7919u + 1u);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bad), sizeof(h));
    printf("exponents of a, b in [%4d, %3d] (2^28 cases): a / b differs %llu, sqrt differs %llu, 1 / b differs %llu | float64 (exponents x 5, sqrt x 8): divide differs %llu, sqrt differs %llu\n",
           rg[0], rg[1], h[0], h[1], h[2], h[3] & 0xFFFFFFFFull, h[3] >> 32);
  }
  return 0;
}

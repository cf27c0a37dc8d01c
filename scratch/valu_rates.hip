// scratch/valu_rates.hip -- issue cost of the VALU instruction classes the step kernels are made of, on a FULL device
// (1024 workgroups x 512 threads = 8 waves per SIMD, every wave issuing the same class from 8 independent chains):
// cycles per wave-instruction per SIMD = launch time x 2.4 GHz / (instructions per wave x 8 waves per SIMD).  Answers what
// a float64 instruction, a transcendental, a 32-bit integer multiply, a DPP move or a ds_bpermute costs RELATIVE to v_fma_f32
// when the kernel is issue-bound (the fused n-step kernel: SQ_ACTIVE_INST_VALU 87 % of the step) -- the number the
// "filtered float64" work is priced with (profiles/r05_kernel_geometry.md).
// build: hipcc --offload-arch=gfx950 -O2 scratch/valu_rates.hip -o scratch/valu_rates ; run on the GPU box
// (under rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES for the counter view of the same launches).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL_F32(name, ASM)                                                                  \
  __global__ __launch_bounds__(512, 8) void name(float* out, int iters, float seed) {          \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
    const float b = seed * 0.5f + 1.0f;                                                        \
    for (int i = 0; i < iters; ++i) {                                                          \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                          \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                   \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); \
      }                                                                                        \
    }                                                                                          \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456f) out[threadIdx.x] = a0;              \
  }

#define KERNEL_F64(name, ASM)                                                                  \
  __global__ __launch_bounds__(512, 8) void name(float* out, int iters, float seed) {          \
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
    const double b = seed * 0.5 + 1.0;                                                         \
    for (int i = 0; i < iters; ++i) {                                                          \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                          \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                   \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); \
      }                                                                                        \
    }                                                                                          \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456) out[threadIdx.x] = (float)a0;        \
  }

#define A_FMA32(n) "v_fma_f32 %" #n ", %" #n ", %8, %8\n"
#define A_MUL32(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define A_RCP32(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define A_SQRT32(n) "v_sqrt_f32 %" #n ", %" #n "\n"
#define A_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define A_MAX32(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define A_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define A_MUL24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define A_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define A_CMP32(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define A_DPP(n) "v_mov_b32_dpp %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_BPERM(n) "ds_bpermute_b32 %" #n ", %8, %" #n "\ns_waitcnt lgkmcnt(0)\n"
#define A_CVT3264(n) "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define A_FMA64(n) "v_fma_f64 %" #n ", %" #n ", %8, %8\n"
#define A_MUL64(n) "v_mul_f64 %" #n ", %" #n ", %8\n"
#define A_ADD64(n) "v_add_f64 %" #n ", %" #n ", %8\n"
#define A_RCP64(n) "v_rcp_f64 %" #n ", %" #n "\n"
#define A_RSQ64(n) "v_rsq_f64 %" #n ", %" #n "\n"
#define A_CMP64(n) "v_cmp_lt_f64 vcc, %" #n ", %8\n"
#define A_MAX64(n) "v_max_f64 %" #n ", %" #n ", %8\n"
#define A_RNDNE64(n) "v_rndne_f64 %" #n ", %" #n "\n"
// (round 6) packed float32: one instruction, two float32 operations per lane on a 64-bit register pair
#define A_PKFMA32(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %8\n"
#define A_PKMUL32(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define A_PKADD32(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define A_CVT64(n) "v_cvt_f32_f64 %" #n ", %" #n "\n"

KERNEL_F32(k_fma32, A_FMA32)
KERNEL_F32(k_mul32, A_MUL32)
KERNEL_F32(k_rcp32, A_RCP32)
KERNEL_F32(k_sqrt32, A_SQRT32)
KERNEL_F32(k_cndmask, A_CNDMASK)
KERNEL_F32(k_max32, A_MAX32)
KERNEL_F32(k_mullo, A_MULLO)
KERNEL_F32(k_mul24, A_MUL24)
KERNEL_F32(k_and, A_AND)
KERNEL_F32(k_cmp32, A_CMP32)
KERNEL_F32(k_dpp, A_DPP)
KERNEL_F32(k_bperm, A_BPERM)
KERNEL_F32(k_cvti, A_CVT3264)
KERNEL_F64(k_fma64, A_FMA64)
KERNEL_F64(k_mul64, A_MUL64)
KERNEL_F64(k_add64, A_ADD64)
KERNEL_F64(k_rcp64, A_RCP64)
KERNEL_F64(k_rsq64, A_RSQ64)
KERNEL_F64(k_cmp64, A_CMP64)
KERNEL_F64(k_max64, A_MAX64)
KERNEL_F64(k_rndne64, A_RNDNE64)
KERNEL_F64(k_pk_fma32, A_PKFMA32)
KERNEL_F64(k_pk_mul32, A_PKMUL32)
KERNEL_F64(k_pk_add32, A_PKADD32)

template <typename K>
void run(const char* name, K kern, float* out, int grid, int threads) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, iters, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_wave = (double)iters * 32.0;                           // instructions of the class per wave
  const double waves_per_simd = (double)grid * (threads / 64) / 1024.0;   // 256 CUs x 4 SIMDs
  const double cyc = ms * 1e-3 / 5.0 * 2.4e9 / (per_wave * waves_per_simd);
  printf("%-12s grid %5d x %3d  %8.1f us / launch  %6.2f cycles per wave-instruction per SIMD\n", name, grid, threads, ms * 1e3 / 5.0, cyc);
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  for (int pass = 0; pass < 2; ++pass) {
    const int grid = pass == 0 ? 1024 : 256, threads = pass == 0 ? 512 : 256;   // 8 waves / SIMD, then 1 wave / SIMD
    printf("---- %d waves per SIMD\n", pass == 0 ? 8 : 1);
#define RUN(k) run(#k, k, out, grid, threads)
    RUN(k_fma32); RUN(k_mul32); RUN(k_max32); RUN(k_and); RUN(k_cndmask); RUN(k_cmp32); RUN(k_mul24); RUN(k_mullo); RUN(k_cvti);
    RUN(k_rcp32); RUN(k_sqrt32); RUN(k_dpp); RUN(k_bperm);
    RUN(k_fma64); RUN(k_mul64); RUN(k_add64); RUN(k_max64); RUN(k_cmp64); RUN(k_rndne64); RUN(k_rcp64); RUN(k_rsq64);
    RUN(k_pk_fma32); RUN(k_pk_mul32); RUN(k_pk_add32);
  }
  return 0;
}

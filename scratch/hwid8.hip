// scratch/hwid8.hip -- where do the EIGHT waves of a pipelined-kernel workgroup land?  The step kernel's geometry (1024
// workgroups x 512 threads, 37.4 KB LDS -> 4 workgroups per CU = 8 waves per SIMD) with HW_ID / XCC_ID recorded per wave:
// which SIMD holds wave 0 (the S-agent chain) and wave 4 (the O-agent chain) of each of a CU's four workgroups?
// build: hipcc --offload-arch=gfx950 -O2 scratch/hwid8.hip -o scratch/hwid8 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(512) void probe(unsigned* out, int spin) {
  extern __shared__ unsigned char smem[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  volatile unsigned char* sm = smem;
  float acc = threadIdx.x;
  for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;  // stay resident for a while
  sm[threadIdx.x] = (unsigned char)acc;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    unsigned* o = out + (blockIdx.x * 8 + (threadIdx.x >> 6)) * 2;
    o[0] = hw; o[1] = xcc;
  }
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 1024;
  const int lds = argc > 2 ? atoi(argv[2]) : 37408;
  const int spin = argc > 3 ? atoi(argv[3]) : 4000;
  unsigned* d;
  (void)hipMalloc(&d, grid * 16 * 4);
  (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(grid), dim3(512), lds, 0, d, spin);
  (void)hipDeviceSynchronize();
  std::vector<unsigned> h(grid * 16);
  (void)hipMemcpy(h.data(), d, grid * 16 * 4, hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
  int simd_of_wave[8][4] = {{0}};
  int rel[8][4] = {{0}};   // SIMD of wave w relative to the SIMD of wave 0 of the same workgroup
  std::map<unsigned, std::vector<int>> per_cu;  // CU -> blocks
  for (int b = 0; b < grid; ++b) {
    const int s0 = (h[(b * 8) * 2] >> 4) & 3;
    for (int w = 0; w < 8; ++w) {
      const unsigned hw = h[(b * 8 + w) * 2];
      const int simd = (hw >> 4) & 3;
      simd_of_wave[w][simd]++;
      rel[w][(simd - s0) & 3]++;
    }
    const unsigned hw0 = h[(b * 8) * 2], xcc = h[(b * 8) * 2 + 1] & 0xF;
    per_cu[(xcc << 16) | (hw0 & 0xFF00)].push_back(b);
  }
  printf("grid %d x 512, lds %d: CUs seen %zu\n", grid, lds, per_cu.size());
  for (int w = 0; w < 8; ++w) printf("  wave %d: absolute simd0..3 %4d %4d %4d %4d | relative to wave 0: %4d %4d %4d %4d\n", w, simd_of_wave[w][0], simd_of_wave[w][1],
                                     simd_of_wave[w][2], simd_of_wave[w][3], rel[w][0], rel[w][1], rel[w][2], rel[w][3]);
  int wgs_hist[8] = {0}, chain_hist[12] = {0}, shown = 0;
  for (auto& kv : per_cu) {
    wgs_hist[kv.second.size() < 7 ? kv.second.size() : 7]++;
    int chain[4] = {0};
    for (int b : kv.second) { chain[(h[(b * 8) * 2] >> 4) & 3]++; chain[(h[(b * 8 + 4) * 2] >> 4) & 3]++; }
    int mx = 0;
    for (int q = 0; q < 4; ++q) mx = chain[q] > mx ? chain[q] : mx;
    chain_hist[mx < 11 ? mx : 11]++;
    if (shown++ < 12) {
      printf("  CU %05x:", kv.first);
      for (int b : kv.second) printf("  wg %4d w0->s%d w4->s%d", b, (h[(b * 8) * 2] >> 4) & 3, (h[(b * 8 + 4) * 2] >> 4) & 3);
      printf("\n");
    }
  }
  printf("workgroups per CU -> CUs:");
  for (int q = 0; q < 8; ++q) printf(" %d:%d", q, wgs_hist[q]);
  printf("\nmax chain waves (wave 0 + wave 4 of every workgroup) on one SIMD of a CU -> CUs:");
  for (int q = 0; q < 12; ++q) printf(" %d:%d", q, chain_hist[q]);
  printf("\n");
  return 0;
}

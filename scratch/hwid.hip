// scratch/hwid.hip -- where does the dispatcher put the waves of a workgroup?  Launches the step kernel's geometry
// (1024 workgroups x 256 threads, 23 KB LDS -> 4 workgroups per CU) and records HW_ID / XCC_ID of every wave.
// build: hipcc --offload-arch=gfx950 -O2 scratch/hwid.hip -o scratch/hwid ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
  extern __shared__ unsigned char smem[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = wall_clock64();
  volatile unsigned char* sm = smem;
  float acc = threadIdx.x;
  for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;  // stay resident for a while
  sm[threadIdx.x] = (unsigned char)acc;
  __syncthreads();
  const unsigned long long t1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) {
    unsigned* o = out + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    o[0] = hw; o[1] = xcc; o[2] = (unsigned)t0; o[3] = (unsigned)t1;
  }
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 1024;
  const int lds = argc > 2 ? atoi(argv[2]) : 23 * 1024;
  const int spin = argc > 3 ? atoi(argv[3]) : 4000;
  unsigned* d;
  hipMalloc(&d, grid * 16 * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds, 0, d, spin);
  hipDeviceSynchronize();
  std::vector<unsigned> h(grid * 16);
  hipMemcpy(h.data(), d, grid * 16 * 4, hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
  int simd_of_wave[4][4] = {{0}};
  std::map<unsigned, std::vector<int>> per_cu;  // (xcc, se, sh, cu) -> list of (block, simd of wave 0)
  for (int b = 0; b < grid; ++b) {
    for (int w = 0; w < 4; ++w) {
      const unsigned hw = h[(b * 4 + w) * 4], xcc = h[(b * 4 + w) * 4 + 1] & 0xF;
      const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      simd_of_wave[w][simd]++;
      if (w == 0) per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b * 4 + simd);
    }
  }
  printf("wave index -> SIMD histogram (grid %d)\n", grid);
  for (int w = 0; w < 4; ++w) printf("  wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, simd_of_wave[w][0], simd_of_wave[w][1], simd_of_wave[w][2], simd_of_wave[w][3]);
  printf("CUs seen: %zu\n", per_cu.size());
  int shown = 0;
  int hist[8] = {0};
  for (auto& kv : per_cu) {
    int cnt[4] = {0};
    for (int v : kv.second) cnt[v & 3]++;
    int mx = 0;
    for (int q = 0; q < 4; ++q) mx = cnt[q] > mx ? cnt[q] : mx;
    hist[mx < 7 ? mx : 7]++;
    if (shown < 24) {
      printf("  xcc %u se %u sh %u cu %2u: blocks", kv.first >> 12, (kv.first >> 8) & 7, (kv.first >> 4) & 1, kv.first & 15);
      for (int v : kv.second) printf(" %d(s%d)", v >> 2, v & 3);
      printf("\n");
      ++shown;
    }
  }
  printf("max wave-0 count on one SIMD of a CU -> number of CUs:");
  for (int q = 0; q < 8; ++q) printf(" %d:%d", q, hist[q]);
  printf("\n");
  // for candidate rotations: how balanced would the agent wave be?
  for (int shift = -1; shift <= 10; ++shift) {
    int worst_hist[8] = {0};
    std::map<unsigned, std::vector<int>> cnt;
    for (int b = 0; b < grid; ++b) {
      const int aw = shift < 0 ? 0 : ((b >> shift) & 3);
      const unsigned hw = h[(b * 4 + aw) * 4], xcc = h[(b * 4 + aw) * 4 + 1] & 0xF;
      const unsigned key = (xcc << 12) | (hw & 0xFF00);
      auto& v = cnt[key];
      if (v.empty()) v.assign(4, 0);
      v[(hw >> 4) & 3]++;
    }
    for (auto& kv : cnt) { int mx = 0; for (int q : kv.second) mx = q > mx ? q : mx; worst_hist[mx < 7 ? mx : 7]++; }
    printf("aw_shift %2d: max agent waves per SIMD -> CUs:", shift);
    for (int q = 1; q < 8; ++q) printf(" %d:%d", q, worst_hist[q]);
    printf("\n");
  }
  return 0;
}

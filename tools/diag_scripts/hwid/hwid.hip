// Where do the four waves of a 256-thread workgroup with ~78 KB of LDS land (two workgroups per CU)? Prints, for the first CUs, the
// workgroups in order of arrival with the SIMD and wave slot of each of their waves.   hipcc --offload-arch=gfx950 -O2 hwid.hip -o hwid
// Measured (round 6, MI355X): every workgroup has one wave per SIMD, and the two workgroups that share a CU NEVER have their wave 0 on the
// same SIMD (256 of 256 CUs: e.g. w0..w3 on SIMDs 1 3 0 2 and 3 0 2 1) — the serial wave-0 roles of k_solve_chain do not collide; what two
// co-resident workgroups cost each other there (60 -> 87 us) is not that.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void k(unsigned *out, long long *tm, int spin) {
  extern __shared__ double sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID, 32 bits
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;      // HW_REG_XCC_ID
  id = (id & 0xffff) | (xcc << 16);
  long long t0 = wall_clock64();
  double x = sm[threadIdx.x] = threadIdx.x;
  for (int i = 0; i < spin; i++) x = __builtin_fma(x, 1.0000001, 1e-9);
  sm[threadIdx.x] = x;
  if (lane == 0) { out[blockIdx.x * 4 + wave] = id; if (wave == 0) tm[blockIdx.x] = t0; }
}
int main() {
  const int B = 2048;
  unsigned *out; long long *tm;
  hipMalloc(&out, B * 4 * 4); hipMalloc(&tm, B * 8);
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 78 * 1024);
  hipLaunchKernelGGL(k, dim3(B), dim3(256), 78 * 1024, 0, out, tm, 20000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(B * 4); std::vector<long long> ht(B);
  hipMemcpy(h.data(), out, B * 16, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), tm, B * 8, hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh [12], se [15:13] (gfx950: se may be wider), ... xcc in XCC_ID register
  std::vector<int> order(B); for (int i = 0; i < B; i++) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return ht[a] < ht[b]; });
  int shown = 0;
  for (int key = 0; key < 4 && shown < 4; key++) {
    unsigned cu_key = 0xffffffff;
    for (int oi = 0; oi < B; oi++) {
      int b = order[oi];
      unsigned cu = h[b * 4] >> 8;
      if (cu_key == 0xffffffff) { bool used = false; for (int p = 0; p < oi; p++) if ((h[order[p] * 4] >> 8) == cu) used = true; if (used && key) continue; if (key) { /* take the key-th distinct */ } cu_key = cu; }
      if (cu != cu_key) continue;
      printf("cu %06x wg %4d t %lld :", cu, b, (ht[b] - ht[order[0]]));
      for (int w = 0; w < 4; w++) printf("  w%d simd %u slot %u", w, (h[b * 4 + w] >> 4) & 3, h[b * 4 + w] & 15);
      printf("\n");
    }
    shown++;
    break;
  }
  // statistics: how often do all four waves sit on four different SIMDs; how often is wave 0 on SIMD 0
  int perm = 0, w0s0 = 0; int hist[4] = {0, 0, 0, 0};
  for (int b = 0; b < B; b++) {
    int m = 0; for (int w = 0; w < 4; w++) m |= 1 << ((h[b * 4 + w] >> 4) & 3);
    perm += m == 15; hist[(h[b * 4] >> 4) & 3]++;
  }
  printf("workgroups with one wave per SIMD: %d of %d; SIMD of wave 0: %d %d %d %d\n", perm, B, hist[0], hist[1], hist[2], hist[3]);
  // pairs of workgroups on the same CU at the same time: first two arrivals per CU
  int same = 0, diff = 0;
  std::vector<unsigned> seen; std::vector<int> first;
  for (int oi = 0; oi < B; oi++) {
    int b = order[oi]; unsigned cu = h[b * 4] >> 8;
    size_t p = std::find(seen.begin(), seen.end(), cu) - seen.begin();
    if (p == seen.size()) { seen.push_back(cu); first.push_back(b); }
    else if (first[p] >= 0) { int a = first[p]; (((h[a * 4] >> 4) & 3) == ((h[b * 4] >> 4) & 3) ? same : diff)++; first[p] = -1; }
  }
  printf("first two workgroups of a CU: wave 0 on the same SIMD %d times, on different SIMDs %d times (%zu CUs seen)\n", same, diff, seen.size());
  return 0;
}

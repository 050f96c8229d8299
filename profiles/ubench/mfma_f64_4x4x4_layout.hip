// Probe of the lane layout of v_mfma_f64_4x4x4_4b_f64 (4 independent 4x4x4 blocks per instruction).
// Every lane feeds distinct primes so that each product a*b identifies its (lane_a, lane_b) pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double *out, double *timing, int iters) {
  const int l = threadIdx.x;
  double a = 1.0 + l, b = 1000.0 * (1.0 + l);   // a*b = 1000 (1+la)(1+lb)
  double c = 0.0;
  c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
  out[l] = c;
  double acc[8];
  for (int q = 0; q < 8; q++) acc[q] = 0;
  long long t0 = clock64();
  for (int i = 0; i < iters; i++)
#pragma unroll
    for (int q = 0; q < 8; q++) acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[q], 0, 0, 0);
  long long t1 = clock64();
  double s = 0; for (int q = 0; q < 8; q++) s += acc[q];
  out[64 + l] = s;
  if (l == 0) timing[0] = (double)(t1 - t0) / (iters * 8.0);
}
int main() {
  double *d, *t; hipMalloc(&d, 1024 * 8); hipMalloc(&t, 64);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t, 1000); hipDeviceSynchronize();
  double h[64], ht; hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost);
  printf("cycles per 4x4x4 MFMA: %.1f\n", ht);
  // decode: D[lane] = sum over 4 (la, lb) pairs of 1000 (1+la)(1+lb); brute force which 4 pairs
  for (int l = 0; l < 64; l++) {
    printf("lane %2d D=%.0f :", l, h[l]);
    // try hypothesis: block = l>>4 ; lanes in the same block; D[i][j] with k = 0..3
    int found = 0;
    for (int ia = 0; ia < 4 && !found; ia++) for (int ib = 0; ib < 4 && !found; ib++) for (int sa = 1; sa <= 4 && !found; sa *= 4) for (int sb = 1; sb <= 4 && !found; sb *= 4) {
      // a-lane = base + ia*sa' + k*ka ... enumerate simple forms: a lanes {base + ia*(sa==1?1:4) + k*(sa==1?4:1)}
      const int base = (l >> 4) << 4;
      double sum = 0;
      for (int kk = 0; kk < 4; kk++) {
        const int la = base + (sa == 1 ? ia + 4 * kk : 4 * ia + kk);
        const int lb = base + (sb == 1 ? ib + 4 * kk : 4 * ib + kk);
        sum += 1000.0 * (1 + la) * (1 + lb);
      }
      if (fabs(sum - h[l]) < 0.5) { printf(" a: i=%d %s  b: j=%d %s", ia, sa == 1 ? "lane=i+4k" : "lane=4i+k", ib, sb == 1 ? "lane=j+4k" : "lane=4j+k"); found = 1; }
    }
    if (!found) printf(" no simple same-block match");
    printf("\n");
  }
  return 0;
}

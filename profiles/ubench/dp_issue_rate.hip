// Issue rate of FP64 VALU instructions for ONE wave on an otherwise idle CU (the situation of the critical chain of k_solve):
// independent v_fma_f64, independent v_fmac_f64_dpp row_newbcast, a dependent v_mul_f64 chain, v_rsq_f64, v_readlane pairs.
//   hipcc --offload-arch=gfx950 -O3 profiles/ubench/dp_issue_rate.hip -o profiles/ubench/dp_issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 200
__global__ __launch_bounds__(64) void k(double *p, long long *out) {
  double a[16], m = p[threadIdx.x];
  for (int i = 0; i < 16; i++) a[i] = p[64 + i * 64 + threadIdx.x];
  long long t0, t1;
  // 1. independent plain FMA
  t0 = clock64();
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
  }
  t1 = clock64(); if (threadIdx.x == 0) out[0] = t1 - t0;
  // 2. independent DPP FMA
  t0 = clock64();
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(m));
  }
  t1 = clock64(); if (threadIdx.x == 0) out[1] = t1 - t0;
  // 3. dependent multiply chain
  double c = m;
  t0 = clock64();
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(c) : "v"(m));
  }
  t1 = clock64(); if (threadIdx.x == 0) out[2] = t1 - t0;
  // 4. dependent rsq chain
  t0 = clock64();
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_rsq_f64 %0, %0\n\ts_nop 1" : "+v"(c));
  }
  t1 = clock64(); if (threadIdx.x == 0) out[3] = t1 - t0;
  // 5. dependent DPP fmac chain (same accumulator)
  t0 = clock64();
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(m));
  }
  t1 = clock64(); if (threadIdx.x == 0) out[4] = t1 - t0;
  // 6. independent 32-bit mov (issue baseline)
  int b[16];
  for (int i = 0; i < 16; i++) b[i] = threadIdx.x + i;
  t0 = clock64();
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(b[i]) : "v"(b[(i + 1) & 15]));
  }
  t1 = clock64(); if (threadIdx.x == 0) out[5] = t1 - t0;
  double s = c; for (int i = 0; i < 16; i++) s += a[i] + b[i];
  p[threadIdx.x] = s;
}
int main() {
  double *p; long long *o, h[6];
  hipMalloc(&p, 8 * 64 * 20); hipMalloc(&o, 64); hipMemset(p, 0, 8 * 64 * 20);
  for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, o);
  hipMemcpy(h, o, 48, hipMemcpyDeviceToHost);
  const char *nm[6] = {"independent v_fma_f64", "independent v_fmac_f64_dpp row_newbcast", "dependent v_mul_f64 chain", "dependent v_rsq_f64 chain (+s_nop 1)", "dependent v_fmac_f64_dpp chain (+s_nop 1)", "independent v_add_u32"};
  for (int i = 0; i < 6; i++) printf("%-48s %6.2f shader cycles per instruction\n", nm[i], (double)h[i] / (REP * 16));
  return 0;
}

// Micro-benchmark (gfx950): do FP64 matrix-core instructions and FP64 vector instructions overlap on one SIMD?
// 512-thread workgroups = 2 waves per SIMD. Mode 0: every wave runs a dependent v_mfma_f64_16x16x4 chain; mode 1: every wave an
// FMA stream (16 independent chains); mode 2: on each SIMD one wave runs the MFMA chain and the other the FMA stream; mode 3: one
// wave interleaves both in a single instruction stream. If the two pipes were independent, mode 2 / 3 would take max(t0, t1)
// per unit of work, not the sum.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dbl4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(double *out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;                 // waves 0..3 -> SIMD 0..3, waves 4..7 -> SIMD 0..3 again
  const bool do_mfma = mode == 0 || (mode == 2 && wave < 4) || mode == 3;
  const bool do_fma = mode == 1 || (mode == 2 && wave >= 4) || mode == 3;
  dbl4 acc = {0, 0, 0, 0};
  double f[16];
  for (int q = 0; q < 16; q++) f[q] = q;
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
  for (int i = 0; i < iters; i++) {
    if (do_mfma && do_fma) {           // one stream: 1 MFMA (64 clk of the matrix pipe) + 16 FMAs (16 x 4 clk of the vector pipe)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 16; q++) f[q] = __builtin_fma(f[q], a, b);
    } else if (do_mfma) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    } else if (do_fma) {
#pragma unroll
      for (int q = 0; q < 16; q++) f[q] = __builtin_fma(f[q], a, b);
    }
  }
  double s = acc[0] + acc[1] + acc[2] + acc[3];
  for (int q = 0; q < 16; q++) s += f[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double *d; hipMalloc(&d, 1 << 24);
  const int iters = 200000, blocks = 256;
  const char *names[4] = {"all waves MFMA chain (1 MFMA / iter)", "all waves FMA stream (16 FMA / iter)", "per SIMD: one wave MFMA, one wave FMA",
                          "every wave: 1 MFMA + 16 FMA interleaved"};
  for (int mode = 0; mode < 4; mode++) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, 1000, mode); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, iters, mode); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d  %-44s %8.3f ms  = %.1f ns per iteration\n", mode, names[mode], ms, ms * 1e6 / iters);
  }
  return 0;
}

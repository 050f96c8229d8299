// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 vs v_fma_f64 on gfx950 (one wave per SIMD
// and two waves per SIMD). Build: hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip -o mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dbl4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k_mfma(double *out, int iters) {
  dbl4 acc[NACC];
  for (int q = 0; q < NACC; q++) acc[q] = (dbl4){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int q = 0; q < NACC; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0;
  for (int q = 0; q < NACC; q++) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / ((double)iters * NACC);
}
template <int NACC>
__global__ void k_fma(double *out, int iters) {
  double acc[NACC];
  for (int q = 0; q < NACC; q++) acc[q] = q;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
  long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int q = 0; q < NACC; q++) acc[q] = __builtin_fma(acc[q], a, b);
  }
  long long t1 = clock64();
  double s = 0;
  for (int q = 0; q < NACC; q++) s += acc[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / ((double)iters * NACC);
}
int main() {
  double *d; hipMalloc(&d, 1 << 24);
  double h;
  for (int threads : {64, 256, 512}) {
    for (int blocks : {1, 256}) {
      hipLaunchKernelGGL(k_mfma<8>, dim3(blocks), dim3(threads), 0, 0, d, 2000); hipDeviceSynchronize();
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0); hipLaunchKernelGGL(k_mfma<8>, dim3(blocks), dim3(threads), 0, 0, d, 20000); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
      double flops = (double)blocks * (threads / 64) * 20000.0 * 8 * 2048;
      printf("mfma_f64_16x16x4  blocks %3d threads %3d : %.1f shader-clk per MFMA per wave, %.2f TFLOP/s\n", blocks, threads, h, flops / ms * 1e-9);
      hipEventRecord(e0); hipLaunchKernelGGL(k_fma<16>, dim3(blocks), dim3(threads), 0, 0, d, 20000); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
      flops = (double)blocks * threads * 20000.0 * 16 * 2;
      printf("v_fma_f64         blocks %3d threads %3d : %.1f shader-clk per FMA  per wave, %.2f TFLOP/s\n", blocks, threads, h, flops / ms * 1e-9);
    }
  }
  // dependent-accumulator latency: 1, 2, 4 independent chains per wave (one wave per SIMD)
  {
    hipLaunchKernelGGL(k_mfma<1>, dim3(256), dim3(256), 0, 0, d, 20000); hipDeviceSynchronize(); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("mfma_f64_16x16x4 1 chain : %.1f clk per MFMA\n", h);
    hipLaunchKernelGGL(k_mfma<2>, dim3(256), dim3(256), 0, 0, d, 20000); hipDeviceSynchronize(); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("mfma_f64_16x16x4 2 chains: %.1f clk per MFMA\n", h);
    hipLaunchKernelGGL(k_mfma<4>, dim3(256), dim3(256), 0, 0, d, 20000); hipDeviceSynchronize(); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("mfma_f64_16x16x4 4 chains: %.1f clk per MFMA\n", h);
  }
  return 0;
}

// Round 5: how fast does a LONE wave run on a mostly idle MI355X? One wave, a chain of dependent FP64 fmas / a loop of
// LDS reads, timed with the 100 MHz wall clock — right after an idle gap, between short kernels, and after the device has been kept
// busy for a few milliseconds. (Build: hipcc --offload-arch=gfx950 -O3 lone_wave_clock.hip -o lone_wave_clock)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
__device__ __forceinline__ unsigned long long wall() { return __builtin_readcyclecounter(); }
__global__ void chain(double *out, unsigned long long *t, int n) {
  double a = out[0], b = 1.0000001, c = 1e-9;
  unsigned long long t0 = wall_clock64();
  unsigned long long c0 = clock64();
#pragma unroll 32
  for (int i = 0; i < n; i++) a = __builtin_fma(a, b, c);
  unsigned long long c1 = clock64();
  unsigned long long t1 = wall_clock64();
  out[1] = a;
  if (threadIdx.x == 0) { t[0] = t1 - t0; t[1] = c1 - c0; }
}
__global__ void ldsloop(double *out, unsigned long long *t, int n) {
  __shared__ double buf[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = i * 1e-3;
  __syncthreads();
  double s = 0.0;
  int idx = threadIdx.x;
  unsigned long long t0 = wall_clock64();
#pragma unroll 32
  for (int i = 0; i < n; i++) { s += buf[idx & 1023]; idx += 17; }
  unsigned long long t1 = wall_clock64();
  out[2 + threadIdx.x] = s;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}
// four independent chains: the issue rate of a lone wave
__global__ void chain4(double *out, unsigned long long *t, int n) {
  double a0 = out[0], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001, c = 1e-9;
  unsigned long long t0 = wall_clock64();
#pragma unroll 8
  for (int i = 0; i < n; i++) { a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c); }
  unsigned long long t1 = wall_clock64();
  out[1] = a0 + a1 + a2 + a3;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}
// a loop the compiler may not unroll: the cost of a taken branch
__global__ void loop1(double *out, unsigned long long *t, int n) {
  double a = out[0], b = 1.0000001, c = 1e-9;
  unsigned long long t0 = wall_clock64();
#pragma unroll 1
  for (int i = 0; i < n; i++) a = __builtin_fma(a, b, c);
  unsigned long long t1 = wall_clock64();
  out[1] = a;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void busy(double *out, int n) {
  double a = out[0] + threadIdx.x;
  for (int i = 0; i < n; i++) a = __builtin_fma(a, 1.0000001, 1e-9);
  if (a == 12345.678) out[5] = a;
}
int main() {
  double *out; unsigned long long *t, h[2];
  hipMalloc(&out, 4096); hipMalloc(&t, 64); hipMemset(out, 0, 4096);
  const int n = 20000;
  auto run = [&](const char *what) {
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, out, t, n); hipDeviceSynchronize();
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("%-44s dependent fma: %6.2f ns each (%5.2f shader-clock ticks of clock64)", what, h[0] * 10.0 / n, (double)h[1] / n);
    hipLaunchKernelGGL(ldsloop, dim3(1), dim3(64), 0, 0, out, t, n); hipDeviceSynchronize();
    hipMemcpy(h, t, 8, hipMemcpyDeviceToHost);
    printf("   LDS read + add: %6.2f ns", h[0] * 10.0 / n);
    hipLaunchKernelGGL(chain4, dim3(1), dim3(64), 0, 0, out, t, n); hipDeviceSynchronize();
    hipMemcpy(h, t, 8, hipMemcpyDeviceToHost);
    printf("   4 independent fmas: %6.2f ns per fma", h[0] * 10.0 / n / 4);
    hipLaunchKernelGGL(loop1, dim3(1), dim3(64), 0, 0, out, t, n); hipDeviceSynchronize();
    hipMemcpy(h, t, 8, hipMemcpyDeviceToHost);
    printf("   loop of one fma (not unrolled): %6.2f ns per iteration\n", h[0] * 10.0 / n);
  };
  run("first kernels of the process");
  run("again at once");
  usleep(200000);
  run("after 200 ms idle");
  for (int k = 0; k < 3; k++) { hipLaunchKernelGGL(busy, dim3(2048), dim3(256), 0, 0, out, 400000); }
  hipDeviceSynchronize();
  run("right after ~ms of a full-device kernel");
  run("again");
  usleep(2000);
  run("2 ms later");
  usleep(20000);
  run("20 ms later");
  return 0;
}

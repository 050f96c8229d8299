// gfbe_devutil.h — device helpers shared by the kernel translation units (gfbe_kernels.hip, gfbe_solve.hip):
// fixed-order wave / block reductions, the Ceres constants of the dogleg strategy, triangular index decoding.
#pragma once
#include "gfbe_device.h"

namespace gfd {

#define GF_MIN_DIAG 1e-6
#define GF_MAX_DIAG 1e32
#define GF_MIN_MU 1e-8
#define GF_MAX_MU 1.0
#define GF_MU_INC 10.0

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}
// Deterministic block reduction (sum) for blockDim.x <= 1024; result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double *scratch /*>=16*/) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) for (int i = 0; i < (int)((blockDim.x + 63) >> 6); i++) r += scratch[i];
  return r;
}
__device__ __forceinline__ double block_max(double v, double *scratch) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) for (int i = 0; i < (int)((blockDim.x + 63) >> 6); i++) r = fmax(r, scratch[i]);
  return r;
}

// NQ quantities per thread reduced over the block with ONE pair of barriers (bit q of maxmask: maximum instead of sum): wave
// sums by the shuffle tree, then thread q adds the <= 16 wave values in wave order — the same order as block_sum / block_max,
// so the results are bit-identical to NQ separate calls. scratch: (16 + 1) * NQ doubles; results in scratch[16 * NQ + q] for
// every thread after the call.
template <int NQ>
__device__ __forceinline__ void block_reduce_multi(const double (&v)[NQ], unsigned maxmask, double *scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const double r = ((maxmask >> q) & 1) ? wave_max(v[q]) : wave_sum(v[q]);
    if (lane == 0) scratch[q * 16 + wid] = r;
  }
  __syncthreads();
  if (threadIdx.x < NQ) {
    const int q = threadIdx.x;
    double r = 0.0;
    for (int i = 0; i < nw; i++) r = ((maxmask >> q) & 1) ? fmax(r, scratch[q * 16 + i]) : r + scratch[q * 16 + i];
    scratch[16 * NQ + q] = r;
  }
  __syncthreads();
}

// landmark sharding: tile t of a window is evaluated by rank t % world (gfbe_set_allreduce)
#define TILE_OWNED(d, tile) ((d).world == 1 || (tile) % (d).world == (d).rank)

__device__ __forceinline__ double clamp_diag(double x) { return fmin(fmax(x, GF_MIN_DIAG), GF_MAX_DIAG); }

__device__ __forceinline__ void tri_decode(int e, int &a, int &b) {
  a = (int)((sqrtf(8.0f * e + 1.0f) - 1.0f) * 0.5f);
  while ((a + 1) * (a + 2) / 2 <= e) a++;
  while (a * (a + 1) / 2 > e) a--;
  b = e - a * (a + 1) / 2;   // b <= a
}

}  // namespace gfd

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

// (round 6) s_setprio at the head of the short, latency-bound kernels of a throughput batch: their few waves share SIMDs with the streaming
// kernels of the batch's other parts (GFBE_PRIO_SMALL = 0: the priorities are left alone)
#ifndef GFBE_PRIO_SMALL
#define GFBE_PRIO_SMALL 0
#endif
#if GFBE_PRIO_SMALL
#define GFBE_SMALL_KERNEL_PRIO() __builtin_amdgcn_s_setprio(3)
#else
#define GFBE_SMALL_KERNEL_PRIO() do { } while (0)
#endif

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
// tid: the thread's index in the order the sums are taken in — threadIdx.x, or the logical index of a kernel that renumbers its waves
// (k_solve_chain: by the SIMD they sit on)
template <int NQ>
__device__ __forceinline__ void block_reduce_multi(const double (&v)[NQ], unsigned maxmask, double *scratch, int tid = -1) {
  if (tid < 0) tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const double r = ((maxmask >> q) & 1) ? wave_max(v[q]) : wave_sum(v[q]);
    if (lane == 0) scratch[q * 16 + wid] = r;
  }
  __syncthreads();
  if (tid < NQ) {
    const int q = tid;
    double r = 0.0;
    for (int i = 0; i < nw; i++) r = ((maxmask >> q) & 1) ? fmax(r, scratch[q * 16 + i]) : r + scratch[q * 16 + i];
    scratch[16 * NQ + q] = r;
  }
  __syncthreads();
}

// Speculative linearisation (BatchDev::spec): the batch seen with set `lb` of the linearisation's outputs in the places of the first
// set — the kernels between two linearisations index d.lm_hP, d.imu_part, ... as before. lb = 0 (and every batch without the
// second set): the batch itself.
__device__ __forceinline__ BatchDev lin_view(const BatchDev &d, const int lb) {
  BatchDev v = d;
  if (d.spec && lb) {
    v.lm_Hll = d.lm_Hll2; v.lm_gl = d.lm_gl2; v.lm_hC = d.lm_hC2; v.lm_hP = d.lm_hP2; v.lm_sw = d.lm_sw2; v.vis_part = d.vis_part2;
    v.imu_part = d.imu_part2; v.wheel_part = d.wheel_part2; v.plane_part = d.plane_part2; v.anchor_part = d.anchor_part2; v.prior_g = d.prior_g2;
    v.lio_part = d.lio_part2; v.gnss_J = d.gnss_J2; v.gnss_r = d.gnss_r2; v.gnss_cost = d.gnss_cost2;
    if (d.schur_part2) v.schur_part = d.schur_part2;      // (k_linschur: the landmark elimination belongs to the set)
  }
  return v;
}

// landmark sharding: tile t of a window is evaluated by rank t % world (gfbe_set_allreduce)
#define TILE_OWNED(d, tile) ((d).world == 1 || (tile) % (d).world == (d).rank)

// DoglegStrategy::StepAccepted's mu (k_accept; the pass that linearises a candidate forms the landmark weights with it)
__device__ __forceinline__ double mu_after_accept(const double mu) { return fmax(GF_MIN_MU, 2.0 * mu / GF_MU_INC); }
__device__ __forceinline__ double clamp_diag(double x) { return fmin(fmax(x, GF_MIN_DIAG), GF_MAX_DIAG); }

// ---- Landmark rows of the normal equations as k_vis<0, false> leaves them (round 4; a batch with constant extrinsic and td):
//   per factor   d = G^T w (3 doubles: lm_hP[k][0..2])        per landmark   D = sum of its d, x = P_w - P_0 (lm_hC[0..2], [3..5])
// instead of the 6 + 6 tangent entries. With the frame constants Rt = [ R_f (row-major, 9) | t_f = P_f - P_0 (3) ] (FrameConst, slot (f, f)
// of the pair table) the H_pl blocks are
//   start pose i:       [ D ; R_i^T ((x - t_i) x D) ]         observing pose j:   [ -d ; R_j^T (d x (x - t_j)) ]
// and a landmark row times a step of the poses needs no block at all: with u_f = R_f dtheta_f (the frame's rotation step in the world)
//   h_l . delta = D . v_i(x) - sum_k d_k . v_jk(x),    v_f(x) = dp_f + u_f x (x - t_f)   (the displacement of the point with frame f).
#define LM_RT_OFF 12          // doubles before R inside a FrameConst record (W 9, wt 3)
__device__ __forceinline__ void lm_row_block(const double *Rt, const double *dv, const double *x, const bool start, double *out) {
  const double e0 = x[0] - Rt[9], e1 = x[1] - Rt[10], e2 = x[2] - Rt[11];
  double c0, c1, c2;
  if (start) { c0 = __builtin_fma(e1, dv[2], -(e2 * dv[1])); c1 = __builtin_fma(e2, dv[0], -(e0 * dv[2])); c2 = __builtin_fma(e0, dv[1], -(e1 * dv[0])); }
  else { c0 = __builtin_fma(dv[1], e2, -(dv[2] * e1)); c1 = __builtin_fma(dv[2], e0, -(dv[0] * e2)); c2 = __builtin_fma(dv[0], e1, -(dv[1] * e0)); }
#pragma unroll
  for (int q = 0; q < 3; q++) {
    out[q] = start ? dv[q] : -dv[q];
    out[3 + q] = __builtin_fma(Rt[q], c0, __builtin_fma(Rt[3 + q], c1, Rt[6 + q] * c2));
  }
}
// v_f(x) . dvec for the two steps a back-substitution needs (Gauss-Newton y and Cauchy v), fs = [ dp_y (3) | u_y (3) | dp_v (3) | u_v (3) | t (3) ]
__device__ __forceinline__ void lm_row_dot2(const double *fs, const double *dv, const double *x, double &oy, double &ov) {
  const double e0 = x[0] - fs[12], e1 = x[1] - fs[13], e2 = x[2] - fs[14];
  const double y0 = fs[0] + __builtin_fma(fs[4], e2, -(fs[5] * e1)), y1 = fs[1] + __builtin_fma(fs[5], e0, -(fs[3] * e2)), y2 = fs[2] + __builtin_fma(fs[3], e1, -(fs[4] * e0));
  const double v0 = fs[6] + __builtin_fma(fs[10], e2, -(fs[11] * e1)), v1 = fs[7] + __builtin_fma(fs[11], e0, -(fs[9] * e2)), v2 = fs[8] + __builtin_fma(fs[9], e1, -(fs[10] * e0));
  oy = __builtin_fma(dv[0], y0, __builtin_fma(dv[1], y1, dv[2] * y2));
  ov = __builtin_fma(dv[0], v0, __builtin_fma(dv[1], v1, dv[2] * v2));
}
enum { LM_FS = 15 };      // doubles per frame of the staged steps above

__device__ __forceinline__ void tri_decode(int e, int &a, int &b) {
  a = (int)((sqrtf(8.0f * e + 1.0f) - 1.0f) * 0.5f);
  while ((a + 1) * (a + 2) / 2 <= e) a++;
  while (a * (a + 1) / 2 > e) a--;
  b = e - a * (a + 1) / 2;   // b <= a
}

}  // namespace gfd

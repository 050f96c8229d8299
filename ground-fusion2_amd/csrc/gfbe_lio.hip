// gfbe_lio.hip — LiDAR point-to-plane factors on the device (BASELINE configs[4]).
//
//   LidarPlaneNormFactor::Evaluate     lio/src/liw/lidarFactor.cpp:18-51
//   CTLidarPlaneNormFactor::Evaluate   lio/src/liw/lidarFactor.cpp:59-120 (the reference's approximate slerp Jacobians are
//                                      reproduced as they are: tests/test_lio_oracle.py quantifies them)
//
// One thread per residual evaluates r and the 1 x 6 / 1 x 12 tangent Jacobian; a workgroup accumulates its J^T J, J^T r and
// cost in registers (thread-strided), reduces them in a fixed order (64-lane shuffles, then its waves through LDS) and writes
// one partial per workgroup; the host adds the partials in workgroup order. Thousands of identical tiny residuals on one or
// two poses: bandwidth- and latency-shaped, no MFMA ("genuinely bandwidth-shaped").
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "gfbe_device.h"

using namespace gfd;

namespace {

struct Qx { double x, y, z, w; };
__device__ __forceinline__ Qx qmulx(Qx a, Qx b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
          a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
__device__ __forceinline__ void qrotx(Qx q, double R[9]) {
  const double x = q.x, y = q.y, z = q.z, w = q.w;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
__device__ __forceinline__ Qx slerpx(Qx a, double t, Qx b) {   // Eigen::QuaternionBase::slerp
  const double one = 1.0 - 2.220446049250313e-16;
  const double d = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w, ad = fabs(d);
  double s0, s1;
  if (ad >= one) { s0 = 1.0 - t; s1 = t; }
  else { const double th = acos(ad), st = sin(th); s0 = sin((1.0 - t) * th) / st; s1 = sin(t * th) / st; }
  if (d < 0) s1 = -s1;
  return {s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z, s0 * a.w + s1 * b.w};
}
__device__ __forceinline__ void q_br(Qx q, double sgn, double M[9]) {   // bottom-right 3x3 of Qleft (+1) / Qright (-1)
  M[0] = q.w; M[1] = -sgn * q.z; M[2] = sgn * q.y; M[3] = sgn * q.z; M[4] = q.w; M[5] = -sgn * q.x; M[6] = -sgn * q.y; M[7] = sgn * q.x; M[8] = q.w;
}
__device__ __forceinline__ void inv3(const double A[9], double B[9]) {
  const double c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
  const double det = A[0] * c0 + A[1] * c1 + A[2] * c2;
  B[0] = c0 / det; B[1] = (A[2] * A[7] - A[1] * A[8]) / det; B[2] = (A[1] * A[5] - A[2] * A[4]) / det;
  B[3] = c1 / det; B[4] = (A[0] * A[8] - A[2] * A[6]) / det; B[5] = (A[2] * A[3] - A[0] * A[5]) / det;
  B[6] = c2 / det; B[7] = (A[1] * A[6] - A[0] * A[7]) / det; B[8] = (A[0] * A[4] - A[1] * A[3]) / det;
}
__device__ __forceinline__ void mm3(const double *A, const double *B, double *C) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j]; C[3 * i + j] = s; }
}

constexpr int LIO_THREADS = 256;
constexpr int LIO_ZERO_COPY_N = 8192;     // gfbe_lio_linearize: scans up to this many residuals are read from / written to pinned host memory by the kernel itself
constexpr int LIO_PART = 12 * 13 / 2 + 12 + 1;   // lower triangle of J^T J (78) + J^T r (12) + cost

template <int CT>
__global__ __launch_bounds__(LIO_THREADS) void k_lio(int n, const double *pts, const double *normals, const double *offsets, const double *alpha,
                                                     const double *weights, double sqrt_info, const double *pb, const double *pe, double *r_out,
                                                     double *J_out, double *part) {
  constexpr int DN = CT ? 12 : 6;
  const int t = threadIdx.x;
  double acc[LIO_PART];
#pragma unroll
  for (int q = 0; q < LIO_PART; q++) acc[q] = 0.0;
  const Qx qb = {pb[3], pb[4], pb[5], pb[6]};
  const Qx qe = {pe[3], pe[4], pe[5], pe[6]};
  for (int k = blockIdx.x * LIO_THREADS + t; k < n; k += gridDim.x * LIO_THREADS) {
    const double *p = pts + 3 * (size_t)k, *nv = normals + 3 * (size_t)k;
    const double wgt = weights[k];
    double Jk[DN], rk, R[9], al = 0.0;
    Qx qs = qb;
    double ts[3] = {pb[0], pb[1], pb[2]};
    if (CT) {
      al = alpha[k];
      const Qx s = slerpx(qb, al, qe);
      const double nn = sqrt(s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w);
      qs = {s.x / nn, s.y / nn, s.z / nn, s.w / nn};
      for (int a = 0; a < 3; a++) ts[a] = pb[a] * (1 - al) + pe[a] * al;
    }
    qrotx(qs, R);
    const double pw[3] = {R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + ts[0], R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + ts[1], R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + ts[2]};
    rk = sqrt_info * wgt * (nv[0] * pw[0] + nv[1] * pw[1] + nv[2] * pw[2] + offsets[k]);
    const double nR[3] = {nv[0] * R[0] + nv[1] * R[3] + nv[2] * R[6], nv[0] * R[1] + nv[1] * R[4] + nv[2] * R[7], nv[0] * R[2] + nv[1] * R[5] + nv[2] * R[8]};
    const double jrs[3] = {-wgt * (nR[1] * p[2] - nR[2] * p[1]), -wgt * (nR[2] * p[0] - nR[0] * p[2]), -wgt * (nR[0] * p[1] - nR[1] * p[0])};
    if (!CT) {
      for (int a = 0; a < 3; a++) { Jk[a] = sqrt_info * wgt * nv[a]; Jk[3 + a] = sqrt_info * jrs[a]; }
    } else {
      const Qx qbi = {-qb.x, -qb.y, -qb.z, qb.w};
      const Qx rd = qmulx(qbi, qe);
      const Qx rds = slerpx({0, 0, 0, 1}, al, rd);
      double Rds[9], Ql_s[9], Ql_d[9], Qr_s[9], Qr_d[9], inv[9], T1[9], Jb[9], Je[9];
      qrotx(rds, Rds);
      q_br(rds, +1, Ql_s); q_br(rd, +1, Ql_d); q_br(rds, -1, Qr_s); q_br(rd, -1, Qr_d);
      inv3(Ql_d, inv); mm3(Ql_s, inv, T1);
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
          double s = 0;
          for (int m = 0; m < 3; m++) s += Rds[3 * m + i] * (((m == j) ? 1.0 : 0.0) - al * T1[3 * m + j]);
          Jb[3 * i + j] = s;
        }
      inv3(Qr_d, inv); mm3(Qr_s, inv, T1);
      for (int q = 0; q < 9; q++) Je[q] = al * T1[q];
      for (int a = 0; a < 3; a++) {
        Jk[a] = sqrt_info * wgt * nv[a] * (1 - al);
        Jk[6 + a] = sqrt_info * wgt * nv[a] * al;
        Jk[3 + a] = sqrt_info * (jrs[0] * Jb[a] + jrs[1] * Jb[3 + a] + jrs[2] * Jb[6 + a]);
        Jk[9 + a] = sqrt_info * (jrs[0] * Je[a] + jrs[1] * Je[3 + a] + jrs[2] * Je[6 + a]);
      }
    }
    if (r_out) r_out[k] = rk;
    if (J_out) for (int a = 0; a < DN; a++) J_out[(size_t)DN * k + a] = Jk[a];
    int e = 0;
#pragma unroll
    for (int a = 0; a < DN; a++)
#pragma unroll
      for (int b = 0; b <= a; b++) acc[e++] += Jk[a] * Jk[b];
#pragma unroll
    for (int a = 0; a < DN; a++) acc[78 + a] += Jk[a] * rk;
    acc[90] += 0.5 * rk * rk;
  }
  // fixed-order reduction: 64-lane butterflies, then the waves through LDS
  __shared__ double red[LIO_THREADS / 64][LIO_PART];
#pragma unroll
  for (int q = 0; q < LIO_PART; q++) {
    double v = acc[q];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((t & 63) == 0) red[t >> 6][q] = v;
  }
  __syncthreads();
  if (t < LIO_PART) {
    double v = 0.0;
    for (int wv = 0; wv < LIO_THREADS / 64; wv++) v += red[wv][t];
    part[(size_t)blockIdx.x * LIO_PART + t] = v;
  }
}

}  // namespace

extern "C" gfbe_status gfbe_lio_linearize(gfbe_ctx *c, int32_t ct, int32_t n, const double *pts, const double *normals, const double *offsets,
                                          const double *alpha, const double *weights, double sqrt_info, const double *pose_begin,
                                          const double *pose_end, double *r, double *J, double *H, double *g, double *cost) {
  if (!c || n < 0 || !pose_begin || (n > 0 && (!pts || !normals || !offsets)) || (ct && (!alpha || !pose_end))) return GFBE_BAD_INPUT;
  if (ctx_device(c) < 0) return GFBE_NO_DEVICE;
  const int dn = ct ? 12 : 6;
  hipStream_t s = ctx_stream(c);
  // One host-to-device copy in, one device-to-host copy out, through the context's scratch and its pinned mirror (round 5: the call
  // made ten allocations and ten copies — 0.14 ms for a 2 000-point scan, which one CPU core evaluates in 0.06 ms):
  //   in  : pts [3 n] | normals [3 n] | offsets [n] | alpha [n] | weights [n] | pose_begin [7] | pose_end [7] (+ 2 of padding)
  //   out : partial sums [G][LIO_PART] | r [n] | J [n][dn]      (r, J only when asked for)
  const int G = std::max(1, std::min(256, (n + LIO_THREADS - 1) / LIO_THREADS));
  const size_t nn = (size_t)std::max(n, 1);
  const size_t in_d = 9 * nn + 16, out_d = (size_t)G * LIO_PART + (r ? nn : 0) + (J ? (size_t)dn * nn : 0);
  double *dev = (double *)ctx_scratch(c, sizeof(double) * (in_d + out_d));
  double *pin = (double *)ctx_scratch_pinned(c, sizeof(double) * (in_d + out_d));
  if (!dev || !pin) { ctx_set_error(c, "gfbe_lio_linearize: device / pinned allocation failed"); return GFBE_DEVICE_ERROR; }
  double *hp = pin, *hn = hp + 3 * nn, *ho = hn + 3 * nn, *ha = ho + nn, *hw = ha + nn, *hb = hw + nn, *he = hb + 8;
  if (n > 0) {
    std::memcpy(hp, pts, sizeof(double) * 3 * n); std::memcpy(hn, normals, sizeof(double) * 3 * n); std::memcpy(ho, offsets, sizeof(double) * n);
    if (ct) std::memcpy(ha, alpha, sizeof(double) * n); else std::memset(ha, 0, sizeof(double) * n);
    if (weights) std::memcpy(hw, weights, sizeof(double) * n); else std::fill(hw, hw + n, 1.0);
  }
  std::memcpy(hb, pose_begin, sizeof(double) * 7); std::memcpy(he, pose_end ? pose_end : pose_begin, sizeof(double) * 7);
  double *dp = dev, *dnv = dp + 3 * nn, *doff = dnv + 3 * nn, *dal = doff + nn, *dw = dal + nn, *dpb = dw + nn, *dpe = dpb + 8;
  double *dpart = dev + in_d, *dr = dpart + (size_t)G * LIO_PART, *dJ = dr + (r ? nn : 0);
  gfbe_status st = GFBE_OK;
  // A scan of the reference's size (lidarodom.cpp:929-1071 produces ~2 000 residuals per ICP iteration) is three dependent latencies —
  // copy in, launch, copy out — around 10 us of arithmetic: 0.055 ms, where one CPU core takes 0.044. Round 6: up to
  // LIO_ZERO_COPY_N residuals the kernel reads the pinned staging buffer across PCIe itself and writes its partial sums (and r, J) into
  // it — ONE launch and the wait for it, no copy command (the k_ingest_small pattern of gfbe_solve_window); larger scans keep the two DMA
  // copies, which stream faster than a kernel's loads across the bus.
  const bool zero_copy = n <= LIO_ZERO_COPY_N;
  if (zero_copy) {
    dp = pin; dnv = dp + 3 * nn; doff = dnv + 3 * nn; dal = doff + nn; dw = dal + nn; dpb = dw + nn; dpe = dpb + 8;
    dpart = pin + in_d; dr = dpart + (size_t)G * LIO_PART; dJ = dr + (r ? nn : 0);
  } else if (hipMemcpyAsync(dev, pin, sizeof(double) * in_d, hipMemcpyHostToDevice, s) != hipSuccess) st = GFBE_DEVICE_ERROR;
  if (st == GFBE_OK) {
    if (ct) hipLaunchKernelGGL(k_lio<1>, dim3(G), dim3(LIO_THREADS), 0, s, n, dp, dnv, doff, dal, dw, sqrt_info, dpb, dpe, r ? dr : nullptr, J ? dJ : nullptr, dpart);
    else hipLaunchKernelGGL(k_lio<0>, dim3(G), dim3(LIO_THREADS), 0, s, n, dp, dnv, doff, dal, dw, sqrt_info, dpb, dpe, r ? dr : nullptr, J ? dJ : nullptr, dpart);
    double *hout = pin + in_d;
    if (!zero_copy && hipMemcpyAsync(hout, dpart, sizeof(double) * out_d, hipMemcpyDeviceToHost, s) != hipSuccess) st = GFBE_DEVICE_ERROR;
    if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) st = GFBE_DEVICE_ERROR;
    if (st == GFBE_OK) {
      double tot[LIO_PART] = {0};
      for (int b = 0; b < G; b++) for (int q = 0; q < LIO_PART; q++) tot[q] += hout[(size_t)b * LIO_PART + q];   // workgroup order
      if (H) { int e = 0; for (int a = 0; a < dn; a++) for (int b = 0; b <= a; b++, e++) { H[a * dn + b] = tot[e]; H[b * dn + a] = tot[e]; } }
      if (g) for (int a = 0; a < dn; a++) g[a] = tot[78 + a];
      if (cost) *cost = tot[90];
      const double *hr = hout + (size_t)G * LIO_PART, *hJ = hr + (r ? nn : 0);
      if (r && n) std::memcpy(r, hr, sizeof(double) * n);
      if (J && n) std::memcpy(J, hJ, sizeof(double) * dn * n);
    }
  }
  if (st != GFBE_OK) ctx_set_error(c, "gfbe_lio_linearize: copy / launch failed");
  return st;
}

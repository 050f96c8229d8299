// gfbe_posegraph.hip — the global_fusion pose graph on the device (BASELINE configs[3]).
//
//   GlobalOptimization::optimize   global_fusion/src/globalOpt.cpp:107-236 (options :117-121)
//   RelativeRTError, TError        global_fusion/src/Factors.h:26-114
//
// One thread per pose evaluates the (at most two) RelativeRTError factors it takes part in with analytic tangent
// Jacobians (the reference differentiates automatically), adds its position fixes through the Huber corrector and
// writes ITS block row of the 6 x 6-block-tridiagonal normal equations — owner-computes, no atomics, bit-reproducible.
// The Levenberg-Marquardt system (Jacobi-scaled, diagonal clamp(diag) / radius) is solved by parallel block cyclic
// reduction: log2(n) sweeps in which every pose eliminates its two neighbours at the current stride (two 6 x 6 SPD
// solves + four 6 x 6 products per pose), instead of the length-n recurrence of a block Cholesky. The trust-region
// loop itself (Ceres 1.14 TrustRegionMinimizer + LevenbergMarquardtStrategy, 5 iterations) runs on the host: one graph,
// one scalar decision per iteration; the per-pose partial sums it needs come back in pose order and are summed serially
// (fixed order). HBM- / latency-bound small-matrix work: no MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "gfbe_device.h"

using namespace gfd;

namespace {

struct Qd { double w, x, y, z; };
__device__ __forceinline__ Qd qmul(Qd a, Qd b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Qd qinv(Qd a) { return {a.w, -a.x, -a.y, -a.z}; }

struct PgDev {
  int n, n_rel, n_fix;
  const int *rel_of;      // [n] index of the factor (i, i+1), or -1
  const int *fix_begin;   // [n+1] CSR of the fixes of pose i
  const double *rel_meas, *fix_meas;   // [n_rel][7], [n_fix][4] (sorted by pose)
  double t_var, q_var, delta;
};

// r(6), J(6 x 12) of RelativeRTError (Factors.h:59-100), columns dq_i t_i dq_j t_j
__device__ void rel_factor(const double *pi, const double *pj, const double *meas, double t_var, double q_var, double *r, double *J) {
  const Qd qi = {pi[3], pi[4], pi[5], pi[6]}, qj = {pj[3], pj[4], pj[5], pj[6]}, qm = {meas[3], meas[4], meas[5], meas[6]};
  const double nn = sqrt(qi.w * qi.w + qi.x * qi.x + qi.y * qi.y + qi.z * qi.z);   // QuaternionRotatePoint normalises
  const double w = qi.w / nn, x = qi.x / nn, y = qi.y / nn, z = qi.z / nn;
  const double Ri[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                        2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
  const double d[3] = {pj[0] - pi[0], pj[1] - pi[1], pj[2] - pi[2]};
  for (int a = 0; a < 3; a++) r[a] = (Ri[a] * d[0] + Ri[3 + a] * d[1] + Ri[6 + a] * d[2] - meas[a]) / t_var;
  const Qd A = qmul(qinv(qm), qinv(qi)), e = qmul(A, qj);
  r[3] = 2 * e.x / q_var; r[4] = 2 * e.y / q_var; r[5] = 2 * e.z / q_var;
  if (!J) return;
  for (int q = 0; q < 72; q++) J[q] = 0.0;
  const double dx[9] = {0, -d[2], d[1], d[2], 0, -d[0], -d[1], d[0], 0};
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += Ri[3 * k + a] * dx[3 * k + b];
      J[a * 12 + b] = 2 * s / t_var;
      J[a * 12 + 3 + b] = -Ri[3 * b + a] / t_var;
      J[a * 12 + 9 + b] = Ri[3 * b + a] / t_var;
    }
  const double La[16] = {A.w, -A.x, -A.y, -A.z, A.x, A.w, -A.z, A.y, A.y, A.z, A.w, -A.x, A.z, -A.y, A.x, A.w};
  const double Rj[16] = {qj.w, -qj.x, -qj.y, -qj.z, qj.x, qj.w, qj.z, -qj.y, qj.y, -qj.z, qj.w, qj.x, qj.z, qj.y, -qj.x, qj.w};
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += La[4 * (a + 1) + k] * Rj[4 * k + (b + 1)];
      J[(3 + a) * 12 + 6 + b] = 2 * s / q_var;
      J[(3 + a) * 12 + b] = -2 * s / q_var;
    }
}

__device__ double huber_corrector(double sq, double delta, double *s1, double *rs, double *asn) {
  const double b = delta * delta;
  double rho0, rho1, rho2;
  if (sq > b) { const double rr = sqrt(sq); rho0 = 2 * delta * rr - b; rho1 = fmax(1e-300, delta / rr); rho2 = -rho1 / (2 * sq); }
  else { rho0 = sq; rho1 = 1.0; rho2 = 0.0; }
  const double sqrt_rho1 = sqrt(rho1);
  if (sq == 0.0 || rho2 <= 0.0) { *s1 = sqrt_rho1; *rs = sqrt_rho1; *asn = 0.0; }
  else { const double D = 1.0 + 2.0 * sq * rho2 / rho1, alpha = 1.0 - sqrt(D); *s1 = sqrt_rho1; *rs = sqrt_rho1 / (1.0 - alpha); *asn = alpha / sq; }
  return 0.5 * rho0;
}

// Per pose i: cost share (the factor starting at i + the fixes of i) and, with Hd != nullptr, block row i of the normal
// equations: Hd[i] (6x6), Ho[i] = block (i+1, i) (written by pose i: its own factor), g[i].
__global__ __launch_bounds__(128) void k_pg_lin(PgDev P, const double *pose, double *cost_i, double *Hd, double *Ho, double *g,
                                                double *rel_r, double *rel_J, double *fix_r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  double hd[36], gg[6];
  for (int q = 0; q < 36; q++) hd[q] = 0.0;
  for (int q = 0; q < 6; q++) gg[q] = 0.0;
  double cost = 0.0;
  const int kp = i > 0 ? P.rel_of[i - 1] : -1, kn = P.rel_of[i];
  double r[6], J[72];
  if (kp >= 0 && Hd) {   // factor (i-1, i): this pose is "j"
    rel_factor(pose + 7 * (i - 1), pose + 7 * i, P.rel_meas + 7 * kp, P.t_var, P.q_var, r, J);
    for (int a = 0; a < 6; a++) {
      for (int b = 0; b < 6; b++) { double s = 0; for (int q = 0; q < 6; q++) s += J[q * 12 + 6 + a] * J[q * 12 + 6 + b]; hd[a * 6 + b] += s; }
      double s = 0; for (int q = 0; q < 6; q++) s += J[q * 12 + 6 + a] * r[q];
      gg[a] += s;
    }
  }
  if (kn >= 0) {         // factor (i, i+1): this pose is "i"; it owns the cost and the off-diagonal block
    rel_factor(pose + 7 * i, pose + 7 * (i + 1), P.rel_meas + 7 * kn, P.t_var, P.q_var, r, (Hd || rel_J) ? J : nullptr);
    for (int q = 0; q < 6; q++) cost += 0.5 * r[q] * r[q];
    if (rel_r) for (int q = 0; q < 6; q++) rel_r[6 * kn + q] = r[q];
    if (rel_J) for (int q = 0; q < 72; q++) rel_J[72 * kn + q] = J[q];
    if (Hd) {
      for (int a = 0; a < 6; a++) {
        for (int b = 0; b < 6; b++) {
          double sii = 0, sji = 0;
          for (int q = 0; q < 6; q++) { sii += J[q * 12 + a] * J[q * 12 + b]; sji += J[q * 12 + 6 + a] * J[q * 12 + b]; }
          hd[a * 6 + b] += sii;
          Ho[(size_t)i * 36 + a * 6 + b] = sji;
        }
        double s = 0; for (int q = 0; q < 6; q++) s += J[q * 12 + a] * r[q];
        gg[a] += s;
      }
    }
  } else if (Hd && i + 1 < P.n) {
    for (int q = 0; q < 36; q++) Ho[(size_t)i * 36 + q] = 0.0;
  }
  for (int k = P.fix_begin[i]; k < P.fix_begin[i + 1]; k++) {
    const double *m = P.fix_meas + 4 * k;
    double rr[3] = {(pose[7 * i] - m[0]) / m[3], (pose[7 * i + 1] - m[1]) / m[3], (pose[7 * i + 2] - m[2]) / m[3]};
    double s1, rs, asn;
    cost += huber_corrector(rr[0] * rr[0] + rr[1] * rr[1] + rr[2] * rr[2], P.delta, &s1, &rs, &asn);
    double Jc[9];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Jc[3 * a + b] = s1 * ((a == b ? 1.0 : 0.0) - asn * rr[a] * rr[b]) / m[3];
    for (int a = 0; a < 3; a++) rr[a] *= rs;
    if (fix_r) for (int a = 0; a < 3; a++) fix_r[3 * k + a] = rr[a];
    if (Hd) {
      for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) { double s = 0; for (int q = 0; q < 3; q++) s += Jc[3 * q + a] * Jc[3 * q + b]; hd[(3 + a) * 6 + 3 + b] += s; }
        double s = 0; for (int q = 0; q < 3; q++) s += Jc[3 * q + a] * rr[q];
        gg[3 + a] += s;
      }
    }
  }
  cost_i[i] = cost;
  if (Hd) {
    for (int q = 0; q < 36; q++) Hd[(size_t)i * 36 + q] = hd[q];
    for (int q = 0; q < 6; q++) g[(size_t)i * 6 + q] = gg[q];
  }
}

// Jacobi scale (iteration 0) and the scaled LM system in cyclic-reduction form:
//   A_i x_{i-1} + B_i x_i + C_i x_{i+1} = d_i,  B = S Hd S + D2 / radius,  A_i = S Ho_{i-1} S,  C_i = A_{i+1}^T,  d = -S g
__global__ __launch_bounds__(192) void k_pg_system(int n, const double *Hd, const double *Ho, const double *g, double *scale, int set_scale,
                                                   double *diag2, int keep_diag, double radius, double *B, double *d,
                                                   double *Bs /* unregularised S Hd S, for the model cost */, double *gmax_i,
                                                   double *d_pcr /* the copy the cyclic reduction consumes */, int *fail) {
  // six lanes per pose: lane a owns row a (scale factors of the pose's six dims are exchanged through LDS)
  __shared__ double ssc[192];
  const int gidx = blockIdx.x * 192 + threadIdx.x, i = gidx / 6, a = gidx - 6 * i;
  const bool on = i < n;
  double sa = 1.0, ga = 0.0;
  if (on) {
    if (set_scale) scale[(size_t)i * 6 + a] = 1.0 / (1.0 + sqrt(Hd[(size_t)i * 36 + a * 7]));
    sa = scale[(size_t)i * 6 + a];
    ga = g[(size_t)i * 6 + a];
  }
  ssc[threadIdx.x] = sa;
  __syncthreads();
  if (!on) return;
  const double *sp = ssc + (threadIdx.x - a);
  if (a == 0) {
    double gm = 0.0;
    for (int q = 0; q < 6; q++) gm = fmax(gm, fabs(g[(size_t)i * 6 + q]));
    gmax_i[i] = gm;
  }
  double diag = 0.0;
#pragma unroll
  for (int b = 0; b < 6; b++) {
    const double v = Hd[(size_t)i * 36 + a * 6 + b] * sa * sp[b];
    Bs[(size_t)i * 36 + a * 6 + b] = v;
    if (b != a) B[(size_t)i * 36 + a * 6 + b] = v;
    else diag = v;
  }
  const double dv = -sa * ga;
  d[(size_t)i * 6 + a] = dv;
  d_pcr[(size_t)i * 6 + a] = dv;
  if (!keep_diag) diag2[(size_t)i * 6 + a] = fmin(fmax(diag, 1e-6), 1e32);
  B[(size_t)i * 36 + a * 7] = diag + diag2[(size_t)i * 6 + a] / radius;
  if (gidx == 0) *fail = 0;
}
__global__ __launch_bounds__(192) void k_pg_system2(int n, const double *Ho, const double *scale, double *A, double *Cc, double *A_pcr, double *C_pcr) {
  const int gidx = blockIdx.x * 192 + threadIdx.x, i = gidx / 6, a = gidx - 6 * i;   // lane a owns row a of pose i
  if (i >= n) return;
  const double sa = scale[(size_t)i * 6 + a];
#pragma unroll
  for (int b = 0; b < 6; b++) {
    const double va = i > 0 ? Ho[(size_t)(i - 1) * 36 + a * 6 + b] * sa * scale[(size_t)(i - 1) * 6 + b] : 0.0;
    const double vc = i + 1 < n ? Ho[(size_t)i * 36 + b * 6 + a] * sa * scale[(size_t)(i + 1) * 6 + b] : 0.0;
    A[(size_t)i * 36 + a * 6 + b] = va; A_pcr[(size_t)i * 36 + a * 6 + b] = va;
    Cc[(size_t)i * 36 + a * 6 + b] = vc; C_pcr[(size_t)i * 36 + a * 6 + b] = vc;
  }
}

// X = B^-1 Y for an SPD 6 x 6 B (Cholesky), Y with nc columns (row-major 6 x nc); returns false if B is not SPD
__device__ bool spd_solve6(const double *Bm, double *Y, int nc) {
  double L[36];
  for (int c = 0; c < 6; c++) {
    double ds = Bm[c * 6 + c];
    for (int k = 0; k < c; k++) ds -= L[c * 6 + k] * L[c * 6 + k];
    if (!(ds > 0.0) || !isfinite(ds)) return false;
    const double lcc = sqrt(ds);
    L[c * 6 + c] = lcc;
    for (int a = c + 1; a < 6; a++) { double s = Bm[a * 6 + c]; for (int k = 0; k < c; k++) s -= L[a * 6 + k] * L[c * 6 + k]; L[a * 6 + c] = s / lcc; }
  }
  for (int j = 0; j < nc; j++) {
    for (int a = 0; a < 6; a++) { double s = Y[a * nc + j]; for (int k = 0; k < a; k++) s -= L[a * 6 + k] * Y[k * nc + j]; Y[a * nc + j] = s / L[a * 6 + a]; }
    for (int a = 5; a >= 0; a--) { double s = Y[a * nc + j]; for (int k = a + 1; k < 6; k++) s -= L[k * 6 + a] * Y[k * nc + j]; Y[a * nc + j] = s / L[a * 6 + a]; }
  }
  return true;
}

// One sweep of parallel block cyclic reduction at stride s (in -> out):
//   alpha = -A_i B_{i-s}^-1, gamma = -C_i B_{i+s}^-1
//   B' = B + alpha C_{i-s} + gamma A_{i+s};  d' = d + alpha d_{i-s} + gamma d_{i+s};  A' = alpha A_{i-s};  C' = gamma C_{i+s}
// Six lanes per pose, lane a owns row a of the outputs (and element a of d'): it solves ONE 6-vector system per neighbour
// (row a of A_i / C_i as the right-hand side; the neighbour's 6 x 6 Cholesky is recomputed by each of the six lanes) and forms
// its row of the products — the same operations per output element as one thread per pose (bit-identical), six times the
// parallelism on a kernel that runs at 79 waves for 5 000 poses, and 48-byte contiguous accesses per lane.
#define PCR_ROWS 32
__global__ __launch_bounds__(PCR_ROWS * 6) void k_pg_pcr(int n, int s, const double *A, const double *B, const double *Cc, const double *d,
                                                         double *A2, double *B2, double *C2, double *d2, int *fail) {
  const int g = blockIdx.x * (PCR_ROWS * 6) + threadIdx.x, i = g / 6, a = g - 6 * i;
  if (i >= n) return;
  double Bn[6], An[6], Cn[6], dn;
#pragma unroll
  for (int b = 0; b < 6; b++) { Bn[b] = B[(size_t)i * 36 + a * 6 + b]; An[b] = 0.0; Cn[b] = 0.0; }
  dn = d[(size_t)i * 6 + a];
  const int im = i - s, ip = i + s;
  if (im >= 0) {
    // column a of Y = B_{i-s}^-1 A_i^T, i.e. the solve with row a of A_i; alpha[a][k] = -Y[k][a]
    double y[6];
#pragma unroll
    for (int k = 0; k < 6; k++) y[k] = A[(size_t)i * 36 + a * 6 + k];
    if (!spd_solve6(B + (size_t)im * 36, y, 1)) { *fail = 1; return; }
#pragma unroll
    for (int b = 0; b < 6; b++) {
      double sb = 0, sa = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) { sb += y[k] * Cc[(size_t)im * 36 + k * 6 + b]; sa += y[k] * A[(size_t)im * 36 + k * 6 + b]; }
      Bn[b] -= sb;
      An[b] = -sa;
    }
    double sd = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) sd += y[k] * d[(size_t)im * 6 + k];
    dn -= sd;
  }
  if (ip < n) {
    double y[6];
#pragma unroll
    for (int k = 0; k < 6; k++) y[k] = Cc[(size_t)i * 36 + a * 6 + k];
    if (!spd_solve6(B + (size_t)ip * 36, y, 1)) { *fail = 1; return; }
#pragma unroll
    for (int b = 0; b < 6; b++) {
      double sb = 0, sc = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) { sb += y[k] * A[(size_t)ip * 36 + k * 6 + b]; sc += y[k] * Cc[(size_t)ip * 36 + k * 6 + b]; }
      Bn[b] -= sb;
      Cn[b] = -sc;
    }
    double sd = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) sd += y[k] * d[(size_t)ip * 6 + k];
    dn -= sd;
  }
#pragma unroll
  for (int b = 0; b < 6; b++) { B2[(size_t)i * 36 + a * 6 + b] = Bn[b]; A2[(size_t)i * 36 + a * 6 + b] = An[b]; C2[(size_t)i * 36 + a * 6 + b] = Cn[b]; }
  d2[(size_t)i * 6 + a] = dn;
}
__global__ __launch_bounds__(64) void k_pg_final(int n, const double *B, const double *d, double *y, int *fail) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double Y[6];
  for (int q = 0; q < 6; q++) Y[q] = d[(size_t)i * 6 + q];
  if (!spd_solve6(B + (size_t)i * 36, Y, 1)) { *fail = 1; return; }
  for (int q = 0; q < 6; q++) y[(size_t)i * 6 + q] = Y[q];
}

// model cost change share -(gs . y + 1/2 y^T Hs y) of pose i, the candidate x (+) S y, |step|^2 and |x|^2 shares
__global__ __launch_bounds__(128) void k_pg_candidate(int n, const double *Bs, const double *A, const double *Cc, const double *d, const double *y,
                                                      const double *scale, const double *x, double *cand, double *model_i, double *step2_i,
                                                      double *xn2_i) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double gy = 0, yHy = 0;
  for (int a = 0; a < 6; a++) {
    double s = 0;
    for (int b = 0; b < 6; b++) {
      s += Bs[(size_t)i * 36 + a * 6 + b] * y[(size_t)i * 6 + b];
      if (i > 0) s += A[(size_t)i * 36 + a * 6 + b] * y[(size_t)(i - 1) * 6 + b];
      if (i + 1 < n) s += Cc[(size_t)i * 36 + a * 6 + b] * y[(size_t)(i + 1) * 6 + b];
    }
    gy += -d[(size_t)i * 6 + a] * y[(size_t)i * 6 + a];
    yHy += y[(size_t)i * 6 + a] * s;
  }
  model_i[i] = -(gy + 0.5 * yHy);
  double d6[6];
  for (int a = 0; a < 6; a++) d6[a] = scale[(size_t)i * 6 + a] * y[(size_t)i * 6 + a];
  const double nrm = sqrt(d6[0] * d6[0] + d6[1] * d6[1] + d6[2] * d6[2]);
  Qd dq = {1.0, 0.0, 0.0, 0.0};
  if (nrm > 0.0) { const double sc = sin(nrm) / nrm; dq = {cos(nrm), sc * d6[0], sc * d6[1], sc * d6[2]}; }
  const double *xi = x + 7 * (size_t)i;
  const Qd q = qmul(dq, {xi[3], xi[4], xi[5], xi[6]});
  double out[7] = {xi[0] + d6[3], xi[1] + d6[4], xi[2] + d6[5], q.w, q.x, q.y, q.z};
  double s2 = 0, x2 = 0;
  for (int k = 0; k < 7; k++) { const double df = out[k] - xi[k]; s2 += df * df; x2 += out[k] * out[k]; cand[7 * (size_t)i + k] = out[k]; }
  step2_i[i] = s2; xn2_i[i] = x2;
}

// All device buffers of one call come out of ONE allocation (a bump allocator over a slab sized by a dry run):
// the solve makes ~30 buffers, and hipMalloc / hipFree cost more than the kernels at this problem size.
enum { PG_RESULT_BYTES = 64 };
struct PgBuffers {
  gfbe_ctx *c;
  char *slab = nullptr, *pin = nullptr;
  size_t cap = 0, used = 0, pin_cap = 0, pin_used = 0;
  bool dry = true;
  explicit PgBuffers(gfbe_ctx *ctx) : c(ctx) {}
  ~PgBuffers() { (void)hipStreamSynchronize(ctx_stream(c)); }
  // end of the dry run: take what was asked for from the context's grow-only scratch, restart. The host arrays travel through the
  // context's PINNED scratch (round 6: a copy from pageable memory is staged by the runtime, synchronously, piece by piece — five of
  // them at the head of every call); its last PG_RESULT_BYTES are where the reduction kernels leave what the host decides on.
  bool commit() {
    cap = used; used = 0; dry = false;
    pin_cap = pin_used; pin_used = 0;
    slab = (char *)ctx_scratch(c, std::max<size_t>(cap, 256));
    pin = (char *)ctx_scratch_pinned(c, pin_cap + PG_RESULT_BYTES);
    if (!slab || !pin) return false;
    (void)hipMemsetAsync(slab, 0, std::max<size_t>(cap, 256), ctx_stream(c));
    return true;
  }
  double *result() const { return (double *)(pin + pin_cap); }
  template <typename T>
  T *dev(size_t n, const T *h = nullptr) {
    const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
    const size_t at = used, pat = pin_used;
    used += bytes;
    if (h && n) pin_used += bytes;
    if (dry) return nullptr;
    T *q = (T *)(slab + at);
    if (h && n) {
      std::memcpy(pin + pat, h, n * sizeof(T));
      (void)hipMemcpyAsync(q, pin + pat, n * sizeof(T), hipMemcpyHostToDevice, ctx_stream(c));
    }
    return q;
  }
};

// Per-pose values -> up to three scalars on the device, 24 bytes back to the host instead of n doubles per sum. Two stages,
// fixed order: workgroup b reduces the segment [b * PGR_SEG, (b + 1) * PGR_SEG) (coalesced loads, thread t takes t, t + 256, ...,
// LDS tree), then one workgroup adds the segment partials in segment order. take_max: the first array is reduced with max
// (gradient infinity norm).
enum { PGR_SEG = 4096, PGR_MAXSEG = 1024 };
__global__ __launch_bounds__(256) void k_pg_reduce1(int n, const double *a, const double *b, const double *c3, const double *d4, int take_max, double *partial) {
  __shared__ double sh[4][256];
  const int t = threadIdx.x, i0 = blockIdx.x * PGR_SEG, i1 = min(n, i0 + PGR_SEG);
  double va = 0.0, vb = 0.0, vc = 0.0, vd = 0.0;
  for (int i = i0 + t; i < i1; i += 256) {
    va = take_max ? fmax(va, a[i]) : va + a[i];
    if (b) vb += b[i];
    if (c3) vc += c3[i];
    if (d4) vd += d4[i];
  }
  sh[0][t] = va; sh[1][t] = vb; sh[2][t] = vc; sh[3][t] = vd;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) {
      sh[0][t] = take_max ? fmax(sh[0][t], sh[0][t + o]) : sh[0][t] + sh[0][t + o];
      sh[1][t] += sh[1][t + o];
      sh[2][t] += sh[2][t + o];
      sh[3][t] += sh[3][t + o];
    }
    __syncthreads();
  }
  if (t < 4) partial[4 * blockIdx.x + t] = sh[t][0];
}
// (out: the PINNED result slot of the call — the kernel writes what the host decides on across PCIe itself: no copy command, and none
//  into pageable memory, between the reduction and the host's wait; out[4]: the factorisation's failure flag when one is passed)
__global__ __launch_bounds__(64) void k_pg_reduce2(int nseg, const double *partial, int take_max, double *out, const int *fail) {
  const int t = threadIdx.x;
  if (t == 4) out[4] = fail ? (double)*fail : 0.0;
  if (t >= 4) return;
  double v = 0.0;
  for (int q = 0; q < nseg; q++) { const double x = partial[4 * q + t]; v = (take_max && t == 0) ? fmax(v, x) : v + x; }
  out[t] = v;
}
// dscratch: 4 * PGR_MAXSEG doubles; res: the call's pinned result slot (PgBuffers::result)
void dev_reduce(gfbe_ctx *c, int n, const double *a, const double *b, const double *c3, const double *d4, bool take_max, double *dscratch, double *res,
                double out[4], const int *fail = nullptr, int *hfail = nullptr) {
  const int nseg = (n + PGR_SEG - 1) / PGR_SEG;
  hipLaunchKernelGGL(k_pg_reduce1, dim3(nseg), dim3(256), 0, ctx_stream(c), n, a, b, c3, d4, take_max ? 1 : 0, dscratch + 4);
  hipLaunchKernelGGL(k_pg_reduce2, dim3(1), dim3(64), 0, ctx_stream(c), nseg, dscratch + 4, take_max ? 1 : 0, res, fail);
  (void)hipStreamSynchronize(ctx_stream(c));
  for (int q = 0; q < 4; q++) out[q] = res[q];
  if (hfail) *hfail = res[4] != 0.0;
}
double host_sum(gfbe_ctx *c, const double *dptr, int n, double *dscratch, double *res, bool take_max = false) {
  double out[4];
  dev_reduce(c, n, dptr, nullptr, nullptr, nullptr, take_max, dscratch, res, out);
  return out[0];
}

gfbe_status pg_prepare(gfbe_ctx *c, int n, int n_rel, const int32_t *rel_i, int n_fix, const int32_t *fix_i, const double *fix_meas,
                       std::vector<int> &rel_of, std::vector<int> &fix_begin, std::vector<double> &fix_sorted, std::vector<int> &fix_order) {
  if (!c) return GFBE_BAD_INPUT;
  if (ctx_device(c) < 0) return GFBE_NO_DEVICE;
  if (n < 1 || n_rel < 0 || n_fix < 0 || n > PGR_SEG * PGR_MAXSEG) return GFBE_BAD_INPUT;   // (4.2 M poses: the reduction scratch)
  rel_of.assign(n, -1);
  for (int k = 0; k < n_rel; k++) {
    if (rel_i[k] < 0 || rel_i[k] + 1 >= n || rel_of[rel_i[k]] >= 0) { ctx_set_error(c, "gfbe_pg: relative factors must connect distinct consecutive poses (i, i+1)"); return GFBE_BAD_INPUT; }
    rel_of[rel_i[k]] = k;
  }
  fix_begin.assign(n + 1, 0);
  for (int k = 0; k < n_fix; k++) { if (fix_i[k] < 0 || fix_i[k] >= n) { ctx_set_error(c, "gfbe_pg: position fix on a pose out of range"); return GFBE_BAD_INPUT; } fix_begin[fix_i[k] + 1]++; }
  for (int i = 0; i < n; i++) fix_begin[i + 1] += fix_begin[i];
  std::vector<int> fill(fix_begin.begin(), fix_begin.end() - 1);
  fix_sorted.assign((size_t)4 * n_fix, 0.0); fix_order.assign(n_fix, 0);
  for (int k = 0; k < n_fix; k++) { const int p = fill[fix_i[k]]++; fix_order[p] = k; std::memcpy(&fix_sorted[4 * (size_t)p], fix_meas + 4 * (size_t)k, 4 * sizeof(double)); }
  return GFBE_OK;
}

}  // namespace

extern "C" {

gfbe_status gfbe_pg_eval(gfbe_ctx *c, int32_t n, const double *pose, int32_t n_rel, const int32_t *rel_i, const double *rel_meas, double t_var,
                         double q_var, int32_t n_fix, const int32_t *fix_i, const double *fix_meas, double delta, double *rel_r, double *rel_J,
                         double *fix_r, double *cost) {
  std::vector<int> rel_of, fix_begin, fix_order;
  std::vector<double> fix_sorted;
  gfbe_status st = pg_prepare(c, n, n_rel, rel_i, n_fix, fix_i, fix_meas, rel_of, fix_begin, fix_sorted, fix_order);
  if (st != GFBE_OK) return st;
  PgBuffers buf(c);
  PgDev P;
  double *dpose, *dcost, *dr, *dJ, *dfr, *Hd, *Ho, *g, *red3;
  for (int pass = 0; pass < 2; pass++) {
    P = {n, n_rel, n_fix, buf.dev<int>(n, rel_of.data()), buf.dev<int>(n + 1, fix_begin.data()), buf.dev<double>((size_t)7 * n_rel, rel_meas),
         buf.dev<double>((size_t)4 * n_fix, fix_sorted.data()), t_var, q_var, delta};
    dpose = buf.dev<double>((size_t)7 * n, pose); dcost = buf.dev<double>(n); dr = buf.dev<double>((size_t)6 * n_rel);
    dJ = buf.dev<double>((size_t)72 * n_rel); dfr = buf.dev<double>((size_t)3 * n_fix);
    Hd = buf.dev<double>((size_t)36 * n); Ho = buf.dev<double>((size_t)36 * n); g = buf.dev<double>((size_t)6 * n);
    red3 = buf.dev<double>(4 + 4 * (size_t)PGR_MAXSEG);
    if (pass == 0 && !buf.commit()) { ctx_set_error(c, "gfbe_pg_eval: device allocation failed"); return GFBE_DEVICE_ERROR; }
  }
  hipLaunchKernelGGL(k_pg_lin, dim3((n + 127) / 128), dim3(128), 0, ctx_stream(c), P, dpose, dcost, Hd, Ho, g, dr, dJ, dfr);
  const double total = host_sum(c, dcost, n, red3, buf.result());
  if (cost) *cost = total;
  if (rel_r && n_rel) (void)hipMemcpy(rel_r, dr, sizeof(double) * 6 * n_rel, hipMemcpyDeviceToHost);
  if (rel_J && n_rel) (void)hipMemcpy(rel_J, dJ, sizeof(double) * 72 * n_rel, hipMemcpyDeviceToHost);
  if (fix_r && n_fix) {
    std::vector<double> fr((size_t)3 * n_fix);
    (void)hipMemcpy(fr.data(), dfr, sizeof(double) * 3 * n_fix, hipMemcpyDeviceToHost);
    for (int p = 0; p < n_fix; p++) std::memcpy(fix_r + 3 * (size_t)fix_order[p], &fr[3 * (size_t)p], 3 * sizeof(double));
  }
  return hipGetLastError() == hipSuccess ? GFBE_OK : GFBE_DEVICE_ERROR;
}

gfbe_status gfbe_pg_solve(gfbe_ctx *c, int32_t n, const double *pose_in, int32_t n_rel, const int32_t *rel_i, const double *rel_meas, double t_var,
                          double q_var, int32_t n_fix, const int32_t *fix_i, const double *fix_meas, double delta, int32_t max_it,
                          double *pose_out, gfbe_summary *S) {
  std::vector<int> rel_of, fix_begin, fix_order;
  std::vector<double> fix_sorted;
  gfbe_status st = pg_prepare(c, n, n_rel, rel_i, n_fix, fix_i, fix_meas, rel_of, fix_begin, fix_sorted, fix_order);
  if (st != GFBE_OK) return st;
  if (!pose_in || !pose_out) return GFBE_BAD_INPUT;
  max_it = std::min(max_it, 15);
  hipStream_t s = ctx_stream(c);
  PgBuffers buf(c);
  PgDev P;
  double *x, *cand, *per, *per2, *per3, *per4, *Hd, *Ho, *g, *scale, *diag2, *Bs, *A0, *C0, *Ab[2], *Bb[2], *Cb[2], *db[2], *d0, *y;
  int *fail;
  double *red3;
  for (int pass = 0; pass < 2; pass++) {
    P = {n, n_rel, n_fix, buf.dev<int>(n, rel_of.data()), buf.dev<int>(n + 1, fix_begin.data()), buf.dev<double>((size_t)7 * n_rel, rel_meas),
         buf.dev<double>((size_t)4 * n_fix, fix_sorted.data()), t_var, q_var, delta};
    x = buf.dev<double>((size_t)7 * n, pose_in); cand = buf.dev<double>((size_t)7 * n);
    per = buf.dev<double>(n); per2 = buf.dev<double>(n); per3 = buf.dev<double>(n); per4 = buf.dev<double>(n);
    Hd = buf.dev<double>((size_t)36 * n); Ho = buf.dev<double>((size_t)36 * n); g = buf.dev<double>((size_t)6 * n);
    scale = buf.dev<double>((size_t)6 * n); diag2 = buf.dev<double>((size_t)6 * n); Bs = buf.dev<double>((size_t)36 * n);
    A0 = buf.dev<double>((size_t)36 * n); C0 = buf.dev<double>((size_t)36 * n);
    for (int q = 0; q < 2; q++) {
      Ab[q] = buf.dev<double>((size_t)36 * n); Bb[q] = buf.dev<double>((size_t)36 * n); Cb[q] = buf.dev<double>((size_t)36 * n);
      db[q] = buf.dev<double>((size_t)6 * n);
    }
    d0 = buf.dev<double>((size_t)6 * n); y = buf.dev<double>((size_t)6 * n);
    fail = buf.dev<int>(1);
    red3 = buf.dev<double>(4 + 4 * (size_t)PGR_MAXSEG);
    if (pass == 0 && !buf.commit()) { ctx_set_error(c, "gfbe_pg_solve: device allocation failed"); return GFBE_DEVICE_ERROR; }
  }
  const dim3 g128((n + 127) / 128), b128(128), g64((n + 63) / 64), b64(64);
  gfbe_summary sm;
  std::memset(&sm, 0, sizeof sm);
  hipLaunchKernelGGL(k_pg_lin, g128, b128, 0, s, P, x, per, Hd, Ho, g, (double *)nullptr, (double *)nullptr, (double *)nullptr);
  double cost = host_sum(c, per, n, red3, buf.result());
  sm.initial_cost = cost; sm.cost_history[0] = cost; sm.status = GFBE_NO_CONVERGENCE;
  double radius = 1e4, decrease = 2.0, x_norm;
  {
    std::vector<double> xh((size_t)7 * n);
    std::memcpy(xh.data(), pose_in, sizeof(double) * 7 * n);
    double s2 = 0; for (double e : xh) s2 += e * e;
    x_norm = std::sqrt(s2);
  }
  int invalid = 0, it = 0;
  bool reuse = false, have_scale = false;
  while (true) {
    if (it >= max_it) { sm.termination = 0; break; }
    // system at the current point (B includes the LM diagonal at the current radius)
    const dim3 g6((6 * n + 191) / 192), b6(192);
    hipLaunchKernelGGL(k_pg_system, g6, b6, 0, s, n, Hd, Ho, g, scale, have_scale ? 0 : 1, diag2, reuse ? 1 : 0, radius, Bb[0], d0, Bs, per4, db[0], fail);
    hipLaunchKernelGGL(k_pg_system2, g6, b6, 0, s, n, Ho, scale, A0, C0, Ab[0], Cb[0]);
    have_scale = true;
    // (the gradient norm of this point comes back together with the step's scalars — one host decision less per iteration;
    //  the step that was computed meanwhile is simply dropped when the gradient test ends the solve)
    if (radius < 1e-32) {
      const double gm = host_sum(c, per4, n, red3, buf.result(), true);
      if (gm <= 1e-10) { sm.termination = 3; sm.status = GFBE_OK; } else sm.termination = 4;
      break;
    }
    // parallel block cyclic reduction: log2(n) sweeps
    // (k_pg_system / k_pg_system2 also wrote the copies the reduction consumes and cleared the failure flag)
    int cur = 0;
    for (int stride = 1; stride < n; stride *= 2) {
      hipLaunchKernelGGL(k_pg_pcr, dim3((n + PCR_ROWS - 1) / PCR_ROWS), dim3(PCR_ROWS * 6), 0, s, n, stride, Ab[cur], Bb[cur], Cb[cur], db[cur], Ab[1 - cur], Bb[1 - cur], Cb[1 - cur], db[1 - cur], fail);
      cur = 1 - cur;
    }
    hipLaunchKernelGGL(k_pg_final, g64, b64, 0, s, n, Bb[cur], db[cur], y, fail);
    hipLaunchKernelGGL(k_pg_candidate, g128, b128, 0, s, n, Bs, A0, C0, d0, y, scale, x, cand, per, per2, per3);
    int hfail = 0;
    double r4[4];
    dev_reduce(c, n, per4, per, per2, per3, true, red3, buf.result(), r4, fail, &hfail);      // max |g|, model change, |step|^2, |candidate|^2; the failure flag
    if (r4[0] <= 1e-10) { sm.termination = 3; sm.status = GFBE_OK; break; }
    it++;
    const double model_change = r4[1];
    if (hfail || !(model_change > 0.0)) {
      sm.accepted[it] = 0; sm.cost_history[it] = cost;
      if (++invalid >= 5) { sm.termination = 4; sm.status = GFBE_NUMERICAL_FAILURE; break; }
      radius /= decrease; decrease *= 2; reuse = true;
      continue;
    }
    invalid = 0;
    const double step2 = r4[2], cand_x2 = r4[3];
    hipLaunchKernelGGL(k_pg_lin, g128, b128, 0, s, P, cand, per, (double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr,
                       (double *)nullptr, (double *)nullptr);
    const double cand_cost = host_sum(c, per, n, red3, buf.result());
    sm.cost_history[it] = cost;
    if (std::sqrt(step2) <= 1e-8 * (x_norm + 1e-8)) { sm.termination = 2; sm.status = GFBE_OK; break; }
    const double change = cost - cand_cost;
    if (std::fabs(change) <= 1e-6 * cost) { sm.termination = 1; sm.status = GFBE_OK; break; }
    const double rho = change / model_change;
    if (rho > 1e-3) {
      std::swap(x, cand);
      cost = cand_cost; x_norm = std::sqrt(cand_x2);
      sm.accepted[it] = 1; sm.num_successful++; sm.cost_history[it] = cost;
      radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3)));
      decrease = 2.0; reuse = false;
      hipLaunchKernelGGL(k_pg_lin, g128, b128, 0, s, P, x, per, Hd, Ho, g, (double *)nullptr, (double *)nullptr, (double *)nullptr);
    } else {
      sm.accepted[it] = 0;
      radius /= decrease; decrease *= 2; reuse = true;
    }
  }
  sm.iterations = it; sm.final_cost = cost; sm.final_radius = radius;
  (void)hipMemcpyAsync(pose_out, x, sizeof(double) * 7 * n, hipMemcpyDeviceToHost, s);
  (void)hipStreamSynchronize(s);
  if (S) *S = sm;
  if (hipGetLastError() != hipSuccess) return GFBE_DEVICE_ERROR;
  return sm.status == GFBE_NUMERICAL_FAILURE ? GFBE_NUMERICAL_FAILURE : GFBE_OK;
}

}  // extern "C"

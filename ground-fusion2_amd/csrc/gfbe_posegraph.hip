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
// loop itself (Ceres 1.14 TrustRegionMinimizer + LevenbergMarquardtStrategy, 5 iterations) runs ON THE DEVICE since round 6
// (PgState: the scalars of the loop, written by the last workgroup of the two reductions of an iteration; every kernel of a pass
// looks at its flags first): the host enqueues max_it passes and waits once — rounds 1-5 decided on the host, two
// synchronisations per iteration, which were most of the 5 000-pose graph's time. The candidate's pass linearises into the
// second set of (Hd, Ho, g), so an accepted step needs no launch of its own. Latency-bound small-matrix work: no MFMA.
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
__device__ __forceinline__ void pg_lin_body(const PgDev &P, const int i, const double *pose, double *cost_i, double *Hd, double *Ho, double *g,
                                            double *rel_r, double *rel_J, double *fix_r) {
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
__global__ __launch_bounds__(128) void k_pg_lin(PgDev P, const double *pose, double *cost_i, double *Hd, double *Ho, double *g,
                                                double *rel_r, double *rel_J, double *fix_r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  pg_lin_body(P, i, pose, cost_i, Hd, Ho, g, rel_r, rel_J, fix_r);
}

// ---- the Levenberg-Marquardt loop's state on the device (gfbe_pg_solve, round 6)
struct PgState {
  double cost, radius, decrease, x_norm, model_change, step2, cand_x2, initial_cost;
  int it, invalid, reuse, have_scale, done, termination, status, num_successful;
  int cur;         // which of the two pose arrays is x (the other one takes the candidate)
  int lb;          // which set of (Hd, Ho, g) was linearised at x (the other one takes the candidate's linearisation)
  int cand_on;     // this pass has a step to try (the decision after the solve)
  int pad;
  int accepted[16];
  double cost_history[16];
};
struct PgSets { double *pose[2], *Hd[2], *Ho[2], *g[2]; };
// cand = 0: the first linearisation (x into its set); 1: the candidate's cost AND linearisation, into the other set.
// Six lanes per pose: lanes 0 and 1 evaluate the pose's two relative factors — (i - 1, i) and (i, i + 1), one call site, side by side —
// into LDS, then lane a forms row a of the pose's blocks from them: every entry sees pg_lin_body's operations in its order (the same
// bits), 17 -> ~7 us per launch over 5 000 poses.
#define PGL_POSES 32
__global__ __launch_bounds__(PGL_POSES * 6) void k_pg_lin_st(PgDev P, const PgState *st, PgSets S, int cand, double *cost_i) {
  __shared__ double sJ[PGL_POSES][2][78];      // per pose and factor: J (6 x 12) | r (6)
  const int f_done = st->done, f_cand = st->cand_on, f_cur = st->cur, f_lb = st->lb;
  if (cand && (f_done || !f_cand)) return;
  const int gidx = blockIdx.x * (PGL_POSES * 6) + threadIdx.x, i = gidx / 6, a = gidx - 6 * i, pl = threadIdx.x / 6;
  const bool on = i < P.n;
  const int xb = cand ? 1 - f_cur : f_cur, sb = cand ? 1 - f_lb : f_lb;
  const double *pose = S.pose[xb];
  double *Hd = S.Hd[sb], *Ho = S.Ho[sb], *g = S.g[sb];
  const int kp = (on && i > 0) ? P.rel_of[i - 1] : -1, kn = on ? P.rel_of[i] : -1;
  if (on && a < 2) {
    const int k = a == 0 ? kp : kn, i0 = a == 0 ? i - 1 : i;
    if (k >= 0) {
      double r[6], J[72];
      rel_factor(pose + 7 * (size_t)i0, pose + 7 * (size_t)(i0 + 1), P.rel_meas + 7 * (size_t)k, P.t_var, P.q_var, r, J);
      for (int q = 0; q < 72; q++) sJ[pl][a][q] = J[q];
      for (int q = 0; q < 6; q++) sJ[pl][a][72 + q] = r[q];
    }
  }
  __syncthreads();
  if (!on) return;
  double hrow[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, ga = 0.0, cost = 0.0;
  if (kp >= 0) {         // factor (i - 1, i): this pose is "j"
    const double *J = sJ[pl][0], *r = J + 72;
    for (int b = 0; b < 6; b++) { double sv = 0; for (int q = 0; q < 6; q++) sv += J[q * 12 + 6 + a] * J[q * 12 + 6 + b]; hrow[b] += sv; }
    double sv = 0; for (int q = 0; q < 6; q++) sv += J[q * 12 + 6 + a] * r[q];
    ga += sv;
  }
  if (kn >= 0) {         // factor (i, i + 1): this pose is "i"; it owns the cost and the off-diagonal block
    const double *J = sJ[pl][1], *r = J + 72;
    if (a == 0) for (int q = 0; q < 6; q++) cost += 0.5 * r[q] * r[q];
    for (int b = 0; b < 6; b++) {
      double sii = 0, sji = 0;
      for (int q = 0; q < 6; q++) { sii += J[q * 12 + a] * J[q * 12 + b]; sji += J[q * 12 + 6 + a] * J[q * 12 + b]; }
      hrow[b] += sii;
      Ho[(size_t)i * 36 + a * 6 + b] = sji;
    }
    double sv = 0; for (int q = 0; q < 6; q++) sv += J[q * 12 + a] * r[q];
    ga += sv;
  } else if (i + 1 < P.n) {
    for (int b = 0; b < 6; b++) Ho[(size_t)i * 36 + a * 6 + b] = 0.0;
  }
  for (int k = P.fix_begin[i]; k < P.fix_begin[i + 1]; k++) {
    const double *m = P.fix_meas + 4 * k;
    double rr[3] = {(pose[7 * (size_t)i] - m[0]) / m[3], (pose[7 * (size_t)i + 1] - m[1]) / m[3], (pose[7 * (size_t)i + 2] - m[2]) / m[3]};
    double s1, rs, asn;
    const double hc = huber_corrector(rr[0] * rr[0] + rr[1] * rr[1] + rr[2] * rr[2], P.delta, &s1, &rs, &asn);
    if (a == 0) cost += hc;
    if (a >= 3) {        // rows 3 .. 5 (the position) take the fix: column aa = a - 3 of Jc, Jc[3 q + c] = s1 ((q == c) - asn rr[q] rr[c]) / m[3]
      const int aa = a - 3;
      const double rra = aa == 0 ? rr[0] : (aa == 1 ? rr[1] : rr[2]);
      double ca[3], Jc[9];
      for (int q = 0; q < 3; q++) ca[q] = s1 * ((q == aa ? 1.0 : 0.0) - asn * rr[q] * rra) / m[3];
      for (int q = 0; q < 3; q++) for (int b = 0; b < 3; b++) Jc[3 * q + b] = s1 * ((q == b ? 1.0 : 0.0) - asn * rr[q] * rr[b]) / m[3];
      for (int q = 0; q < 3; q++) rr[q] *= rs;
      for (int b = 0; b < 3; b++) { double sv = 0; for (int q = 0; q < 3; q++) sv += ca[q] * Jc[3 * q + b]; hrow[3 + b] += sv; }
      double sv = 0; for (int q = 0; q < 3; q++) sv += ca[q] * rr[q];
      ga += sv;
    }
  }
  if (a == 0) cost_i[i] = cost;
  for (int b = 0; b < 6; b++) Hd[(size_t)i * 36 + a * 6 + b] = hrow[b];
  g[(size_t)i * 6 + a] = ga;
}
// Column a of B^-1 for an SPD 6 x 6 B (row-major, e.g. in LDS): Cholesky B = L L^T with the reciprocals of the diagonal, then
// L z = e_a, L^T x = z. Straight-line code (a only enters as the right-hand side). Returns false if B is not SPD.
__device__ __forceinline__ bool spd_inv_col6(const double *Bm, const int a, double *x) {
  double L[6][6], ri[6];
#pragma unroll
  for (int c = 0; c < 6; c++) {
    double ds = Bm[c * 6 + c];
#pragma unroll
    for (int k = 0; k < c; k++) ds -= L[c][k] * L[c][k];
    if (!(ds > 0.0) || !isfinite(ds)) return false;
    const double lcc = sqrt(ds);
    ri[c] = 1.0 / lcc;
#pragma unroll
    for (int r = c + 1; r < 6; r++) {
      double sv = Bm[r * 6 + c];
#pragma unroll
      for (int k = 0; k < c; k++) sv -= L[r][k] * L[c][k];
      L[r][c] = sv * ri[c];
    }
  }
  double z[6];
#pragma unroll
  for (int r = 0; r < 6; r++) {
    double sv = r == a ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < r; k++) sv -= L[r][k] * z[k];
    z[r] = sv * ri[r];
  }
#pragma unroll
  for (int r = 5; r >= 0; r--) {
    double sv = z[r];
#pragma unroll
    for (int k = r + 1; k < 6; k++) sv -= L[k][r] * x[k];
    x[r] = sv * ri[r];
  }
  return true;
}
// The six lanes of a pose hold one row of its new diagonal block each: exchanged through LDS (sB: 36 doubles per pose of the
// workgroup; the lanes of a pose may sit in two waves: a block barrier, passed by every thread), then lane a forms column a of
// the inverse (= row a: symmetric) — the block's Cholesky is recomputed by each of the six lanes, ONCE per block and sweep.
__device__ __forceinline__ void pg_block_inverse(double *sB, const int pl, const int a, const bool on, const double *row, double *Binv_i, int *fail,
                                                 const int stride = 36) {
  if (on) {
#pragma unroll
    for (int b = 0; b < 6; b++) sB[pl * stride + a * 6 + b] = row[b];
  }
  __syncthreads();
  if (!on) return;
  double x[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (!spd_inv_col6(sB + pl * stride, a, x)) *fail = 1;
#pragma unroll
  for (int b = 0; b < 6; b++) Binv_i[a * 6 + b] = x[b];
}

// Jacobi scale (iteration 0) and the scaled LM system in cyclic-reduction form:
//   A_i x_{i-1} + B_i x_i + C_i x_{i+1} = d_i,  B = S Hd S + D2 / radius,  A_i = S Ho_{i-1} S,  C_i = A_{i+1}^T,  d = -S g
__global__ __launch_bounds__(192) void k_pg_system(int n, const PgState *st, PgSets S, double *scale,
                                                   double *diag2, double *B, double *d,
                                                   double *Bs /* unregularised S Hd S, for the model cost */, double *gmax_i,
                                                   double *d_pcr /* the copy the cyclic reduction consumes */, double *Binv /* the blocks' inverses, for the first sweep */,
                                                   int *fail) {
  // six lanes per pose: lane a owns row a (scale factors of the pose's six dims are exchanged through LDS)
  __shared__ double ssc[192];
  __shared__ double sB[32 * 36];
  if (st->done) return;
  const double *Hd = S.Hd[st->lb], *g = S.g[st->lb];
  const int set_scale = !st->have_scale, keep_diag = st->reuse;
  const double radius = st->radius;
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
  double brow[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (on) {
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
    brow[b] = v;
    if (b != a) B[(size_t)i * 36 + a * 6 + b] = v;
    else diag = v;
  }
  const double dv = -sa * ga;
  d[(size_t)i * 6 + a] = dv;
  d_pcr[(size_t)i * 6 + a] = dv;
  if (!keep_diag) diag2[(size_t)i * 6 + a] = fmin(fmax(diag, 1e-6), 1e32);
  brow[a] = diag + diag2[(size_t)i * 6 + a] / radius;
  B[(size_t)i * 36 + a * 7] = brow[a];
  }
  pg_block_inverse(sB, threadIdx.x / 6, a, on, brow, Binv + (size_t)(on ? i : 0) * 36, fail);
}
__global__ __launch_bounds__(192) void k_pg_system2(int n, const PgState *st, PgSets S, const double *scale, double *A, double *Cc, double *A_pcr, double *C_pcr) {
  const int gidx = blockIdx.x * 192 + threadIdx.x, i = gidx / 6, a = gidx - 6 * i;   // lane a owns row a of pose i
  if (i >= n || st->done) return;
  const double *Ho = S.Ho[st->lb];
  const double sa = scale[(size_t)i * 6 + a];
#pragma unroll
  for (int b = 0; b < 6; b++) {
    const double va = i > 0 ? Ho[(size_t)(i - 1) * 36 + a * 6 + b] * sa * scale[(size_t)(i - 1) * 6 + b] : 0.0;
    const double vc = i + 1 < n ? Ho[(size_t)i * 36 + b * 6 + a] * sa * scale[(size_t)(i + 1) * 6 + b] : 0.0;
    A[(size_t)i * 36 + a * 6 + b] = va; A_pcr[(size_t)i * 36 + a * 6 + b] = va;
    Cc[(size_t)i * 36 + a * 6 + b] = vc; C_pcr[(size_t)i * 36 + a * 6 + b] = vc;
  }
}

// One sweep of parallel block cyclic reduction at stride s (in -> out):
//   alpha = -A_i B_{i-s}^-1, gamma = -C_i B_{i+s}^-1
//   B' = B + alpha C_{i-s} + gamma A_{i+s};  d' = d + alpha d_{i-s} + gamma d_{i+s};  A' = alpha A_{i-s};  C' = gamma C_{i+s}
// Six lanes per pose, lane a owns row a of the outputs (and element a of d'). Round 6: the INVERSE of every diagonal block travels
// with it (Binv; k_pg_system forms the first ones), so that a sweep multiplies where rounds 1-5 solved: row a of alpha is row a of
// A_i times B_{i-s}^-1 — no factorisation on the way in —, and the new block's inverse is formed once, at the end of the sweep that
// produces it (pg_block_inverse): one 6 x 6 Cholesky per lane and sweep instead of two, no division inside the substitutions
// (10.9 -> ~5 us per sweep of the 5 000-pose graph, whose 65 sweeps were 59 % of the solve).
#define PCR_ROWS 32
enum { PCR_NB = 3 * 36 + 6 };      // per neighbour: Binv | C (left neighbour) or A (right) first ... see the staging below
__global__ __launch_bounds__(PCR_ROWS * 6) void k_pg_pcr(int n, int s, const double *A, const double *B, const double *Cc, const double *d, const double *Binv,
                                                         double *A2, double *B2, double *C2, double *d2, double *Binv2, int *fail, const PgState *st) {
  // the two neighbours' blocks [Binv | C | A | d] of every pose of the workgroup, staged by its six lanes together (lane a: row a of
  // each) — a lane used to load all 228 values of both neighbours itself, the same ones as its five partners
  __shared__ double sN[PCR_ROWS][2][PCR_NB];
  const int done = st->done;      // (requested with the operands below, looked at behind them: one level of dependent loads less per sweep)
  const int g = blockIdx.x * (PCR_ROWS * 6) + threadIdx.x, i = g / 6, a = g - 6 * i, pl = threadIdx.x / 6;
  const bool on = i < n;
  const int im = i - s, ip = i + s;
  const bool hm = on && im >= 0, hp = on && ip < n;
  double Bn[6], An[6], Cn[6], ra[6], rc[6], dn = 0.0;
#pragma unroll
  for (int b = 0; b < 6; b++) {
    Bn[b] = on ? B[(size_t)i * 36 + a * 6 + b] : 0.0; An[b] = 0.0; Cn[b] = 0.0;
    ra[b] = hm ? A[(size_t)i * 36 + a * 6 + b] : 0.0;
    rc[b] = hp ? Cc[(size_t)i * 36 + a * 6 + b] : 0.0;
  }
  if (on) dn = d[(size_t)i * 6 + a];
#pragma unroll
  for (int side = 0; side < 2; side++) {
    const int j = side == 0 ? im : ip;
    if (side == 0 ? hm : hp) {
      double *o = sN[pl][side];
#pragma unroll
      for (int b = 0; b < 6; b++) {
        o[a * 6 + b] = Binv[(size_t)j * 36 + a * 6 + b];
        o[36 + a * 6 + b] = Cc[(size_t)j * 36 + a * 6 + b];
        o[72 + a * 6 + b] = A[(size_t)j * 36 + a * 6 + b];
      }
      o[108 + a] = d[(size_t)j * 6 + a];
    }
  }
  __syncthreads();
  if (done) return;
  if (hm) {
    // y = B_{i-s}^-1 (row a of A_i)^T; alpha[a][k] = -y[k]
    const double *o = sN[pl][0];
    double y[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      double sv = 0;
#pragma unroll
      for (int m = 0; m < 6; m++) sv += o[k * 6 + m] * ra[m];
      y[k] = sv;
    }
#pragma unroll
    for (int b = 0; b < 6; b++) {
      double sb = 0, sa = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) { sb += y[k] * o[36 + k * 6 + b]; sa += y[k] * o[72 + k * 6 + b]; }
      Bn[b] -= sb;
      An[b] = -sa;
    }
    double sd = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) sd += y[k] * o[108 + k];
    dn -= sd;
  }
  if (hp) {
    const double *o = sN[pl][1];
    double y[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      double sv = 0;
#pragma unroll
      for (int m = 0; m < 6; m++) sv += o[k * 6 + m] * rc[m];
      y[k] = sv;
    }
#pragma unroll
    for (int b = 0; b < 6; b++) {
      double sb = 0, sc = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) { sb += y[k] * o[72 + k * 6 + b]; sc += y[k] * o[36 + k * 6 + b]; }
      Bn[b] -= sb;
      Cn[b] = -sc;
    }
    double sd = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) sd += y[k] * o[108 + k];
    dn -= sd;
  }
  if (on) {
#pragma unroll
    for (int b = 0; b < 6; b++) { B2[(size_t)i * 36 + a * 6 + b] = Bn[b]; A2[(size_t)i * 36 + a * 6 + b] = An[b]; C2[(size_t)i * 36 + a * 6 + b] = Cn[b]; }
    d2[(size_t)i * 6 + a] = dn;
  }
  __syncthreads();      // (every lane is done with the staged blocks: the first 36 doubles of a pose's slot take its new diagonal block)
  pg_block_inverse(&sN[0][0][0], pl, a, on, Bn, Binv2 + (size_t)(on ? i : 0) * 36, fail, 2 * PCR_NB);
}
// after the last sweep the blocks are decoupled: y_i = B_i^-1 d_i (six lanes per pose, lane a: element a)
__global__ __launch_bounds__(192) void k_pg_final(int n, const double *Binv, const double *d, double *y, const PgState *st) {
  const int g = blockIdx.x * 192 + threadIdx.x, i = g / 6, a = g - 6 * i;
  if (i >= n || st->done) return;
  double sv = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) sv += Binv[(size_t)i * 36 + a * 6 + k] * d[(size_t)i * 6 + k];
  y[(size_t)i * 6 + a] = sv;
}

// model cost change share -(gs . y + 1/2 y^T Hs y) of pose i, the candidate x (+) S y, |step|^2 and |x|^2 shares
__global__ __launch_bounds__(192) void k_pg_candidate(int n, const double *Bs, const double *A, const double *Cc, const double *d, const double *y,
                                                      const double *scale, const PgState *st, PgSets S, double *model_i, double *step2_i,
                                                      double *xn2_i) {
  // six lanes per pose: lane a forms row a of H y (the operations of one thread per pose, in its order), lane 0 adds the six shares up
  // in row order and retracts
  __shared__ double sgy[192], syh[192];
  if (st->done) return;
  const int gidx = blockIdx.x * 192 + threadIdx.x, i = gidx / 6, a = gidx - 6 * i;
  const double *x = S.pose[st->cur];
  double *cand = S.pose[1 - st->cur];
  if (i < n) {
    double s = 0;
    for (int b = 0; b < 6; b++) {
      s += Bs[(size_t)i * 36 + a * 6 + b] * y[(size_t)i * 6 + b];
      if (i > 0) s += A[(size_t)i * 36 + a * 6 + b] * y[(size_t)(i - 1) * 6 + b];
      if (i + 1 < n) s += Cc[(size_t)i * 36 + a * 6 + b] * y[(size_t)(i + 1) * 6 + b];
    }
    sgy[threadIdx.x] = -d[(size_t)i * 6 + a] * y[(size_t)i * 6 + a];
    syh[threadIdx.x] = y[(size_t)i * 6 + a] * s;
  }
  __syncthreads();
  if (i >= n || a != 0) return;
  double gy = 0, yHy = 0;
  for (int q = 0; q < 6; q++) { gy += sgy[threadIdx.x + q]; yHy += syh[threadIdx.x + q]; }
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
enum { PG_RESULT_BYTES = 1024 };      // (the pinned result slot: the reductions' scalars of gfbe_pg_eval, the final PgState of gfbe_pg_solve)
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
  // pinned host memory a kernel writes (device-accessible under the same pointer): results that cross PCIe as the kernel's own stores
  template <typename T>
  T *pinned(size_t n) {
    const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
    const size_t pat = pin_used;
    pin_used += bytes;
    return dry ? nullptr : (T *)(pin + pat);
  }
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
// The second stage of a reduction of gfbe_pg_solve, followed by what the host used to decide from its results (TrustRegionMinimizer with
// LevenbergMarquardtStrategy as globalOpt.cpp:117-121 configures it; the statements are the host loop's of rounds 1-5, in its order):
//   mode 0 — the first point's cost: the state of the loop is initialised;
//   mode 1 — after the solve and the candidate (max |g|, model change, |step|^2, |candidate|^2; the factorisation's failure flag): gradient
//            tolerance, the invalid step (LevenbergMarquardtStrategy::StepRejected through decrease), or "evaluate the candidate";
//   mode 2 — after the candidate's evaluation (its cost): parameter / function tolerance, acceptance, the radius update.
enum { PGD_THREADS = 1024, PGD_DIRECT_MAX = 65536 };
__global__ __launch_bounds__(PGD_THREADS) void k_pg_decide(int n, const double *a, const double *b, const double *c3, const double *d4, int nseg, const double *partial,
                                                           int take_max, PgState *st, int *fail, int mode, double x_norm0) {
  const int t = threadIdx.x;
  __shared__ double sh[4][PGD_THREADS / 64];
  __shared__ double r[4];
  const int f_done = mode != 0 ? st->done : 0, f_cand = st->cand_on;      // (requested with the values below, looked at behind them)
  if (partial) {      // the second stage behind k_pg_reduce1 (graphs beyond PGD_DIRECT_MAX poses)
    if (t < 4) {
      double v = 0.0;
      for (int q = 0; q < nseg; q++) { const double x = partial[4 * q + t]; v = (take_max && t == 0) ? fmax(v, x) : v + x; }
      r[t] = v;
    }
    __syncthreads();
  } else {            // the per-pose values themselves: thread t takes t, t + 1024, ... in order, then an LDS tree (fixed order)
    double va = 0.0, vb = 0.0, vc = 0.0, vd = 0.0;
    for (int i0 = t; i0 < n; i0 += 8 * PGD_THREADS) {      // (eight strides' values requested together: one memory round trip per 8192 poses)
      double xa[8], xb[8], xc[8], xd[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = min(i0 + u * PGD_THREADS, n - 1);
        xa[u] = a[i]; xb[u] = b ? b[i] : 0.0; xc[u] = c3 ? c3[i] : 0.0; xd[u] = d4 ? d4[i] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (i0 + u * PGD_THREADS < n) { va = take_max ? fmax(va, xa[u]) : va + xa[u]; vb += xb[u]; vc += xc[u]; vd += xd[u]; }
    }
    // the wave's 64 shares by a shuffle tree, the sixteen waves' in wave order (fixed order, one block barrier)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double oa = __shfl_down(va, o, 64), ob = __shfl_down(vb, o, 64), oc = __shfl_down(vc, o, 64), od = __shfl_down(vd, o, 64);
      va = take_max ? fmax(va, oa) : va + oa; vb += ob; vc += oc; vd += od;
    }
    if ((t & 63) == 0) { sh[0][t >> 6] = va; sh[1][t >> 6] = vb; sh[2][t >> 6] = vc; sh[3][t >> 6] = vd; }
    __syncthreads();
    if (t < 4) {
      double v = 0.0;
      for (int q = 0; q < PGD_THREADS / 64; q++) v = (take_max && t == 0) ? fmax(v, sh[t][q]) : v + sh[t][q];
      r[t] = v;
    }
    __syncthreads();
  }
  if (t != 0 || f_done || (mode == 2 && !f_cand)) return;
  PgState &s = *st;
  if (mode == 0) {
    s.cost = r[0]; s.initial_cost = r[0]; s.cost_history[0] = r[0];
    s.radius = 1e4; s.decrease = 2.0; s.x_norm = x_norm0;
    s.status = GFBE_NO_CONVERGENCE;
    return;
  }
  if (s.done) return;
  if (mode == 1) {
    s.have_scale = 1;
    s.cand_on = 0;
    if (s.radius < 1e-32) {      // (the host loop tested this before the solve: the step computed meanwhile is dropped)
      if (r[0] <= 1e-10) { s.termination = 3; s.status = GFBE_OK; } else s.termination = 4;
      s.done = 1;
      return;
    }
    if (r[0] <= 1e-10) { s.termination = 3; s.status = GFBE_OK; s.done = 1; return; }
    s.it++;
    const int it = s.it;
    const double model_change = r[1];
    const int failed = *fail;      // (raised by any block that was not positive definite — k_pg_system, k_pg_pcr — and cleared here, for the next pass)
    *fail = 0;
    if (failed || !(model_change > 0.0)) {
      s.accepted[it] = 0; s.cost_history[it] = s.cost;
      if (++s.invalid >= 5) { s.termination = 4; s.status = GFBE_NUMERICAL_FAILURE; s.done = 1; return; }
      s.radius /= s.decrease; s.decrease *= 2; s.reuse = 1;
      return;
    }
    s.invalid = 0;
    s.model_change = model_change; s.step2 = r[2]; s.cand_x2 = r[3];
    s.cand_on = 1;
    return;
  }
  if (!s.cand_on) return;
  const int it = s.it;
  const double cand_cost = r[0];
  s.cost_history[it] = s.cost;
  if (sqrt(s.step2) <= 1e-8 * (s.x_norm + 1e-8)) { s.termination = 2; s.status = GFBE_OK; s.done = 1; return; }
  const double change = s.cost - cand_cost;
  if (fabs(change) <= 1e-6 * s.cost) { s.termination = 1; s.status = GFBE_OK; s.done = 1; return; }
  const double rho = change / s.model_change;
  if (rho > 1e-3) {
    s.cur = 1 - s.cur; s.lb = 1 - s.lb;      // (the candidate's pass linearised into the other set)
    s.cost = cand_cost; s.x_norm = sqrt(s.cand_x2);
    s.accepted[it] = 1; s.num_successful++; s.cost_history[it] = cand_cost;
    s.radius = fmin(1e16, s.radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rho - 1.0, 3.0)));
    s.decrease = 2.0; s.reuse = 0;
  } else {
    s.accepted[it] = 0;
    s.radius /= s.decrease; s.decrease *= 2; s.reuse = 1;
  }
}
// the result: x and the loop's state into the call's pinned memory (the kernel's own stores cross PCIe: no copy command)
__global__ __launch_bounds__(256) void k_pg_finish(int n, const PgState *st, PgSets S, double *pose_out, PgState *st_out) {
  const double *x = S.pose[st->cur];
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)7 * n; e += (size_t)gridDim.x * 256) pose_out[e] = x[e];
  if (blockIdx.x == 0 && threadIdx.x == 0) *st_out = *st;
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
                          double *pose_out, gfbe_summary *S_out) {
  std::vector<int> rel_of, fix_begin, fix_order;
  std::vector<double> fix_sorted;
  gfbe_status st = pg_prepare(c, n, n_rel, rel_i, n_fix, fix_i, fix_meas, rel_of, fix_begin, fix_sorted, fix_order);
  if (st != GFBE_OK) return st;
  if (!pose_in || !pose_out) return GFBE_BAD_INPUT;
  max_it = std::min(max_it, 15);
  hipStream_t s = ctx_stream(c);
  static_assert(sizeof(PgState) <= PG_RESULT_BYTES, "the pinned result slot holds the loop's state");
  PgBuffers buf(c);
  PgDev P;
  PgSets S;
  double *per, *per2, *per3, *per4, *scale, *diag2, *Bs, *A0, *C0, *Ab[2], *Bb[2], *Cb[2], *db[2], *Ib[2], *d0, *y, *xo;
  PgState *dst;
  int *fail;
  double *red3;
  for (int pass = 0; pass < 2; pass++) {
    P = {n, n_rel, n_fix, buf.dev<int>(n, rel_of.data()), buf.dev<int>(n + 1, fix_begin.data()), buf.dev<double>((size_t)7 * n_rel, rel_meas),
         buf.dev<double>((size_t)4 * n_fix, fix_sorted.data()), t_var, q_var, delta};
    S.pose[0] = buf.dev<double>((size_t)7 * n, pose_in); S.pose[1] = buf.dev<double>((size_t)7 * n);
    per = buf.dev<double>(n); per2 = buf.dev<double>(n); per3 = buf.dev<double>(n); per4 = buf.dev<double>(n);
    for (int q = 0; q < 2; q++) { S.Hd[q] = buf.dev<double>((size_t)36 * n); S.Ho[q] = buf.dev<double>((size_t)36 * n); S.g[q] = buf.dev<double>((size_t)6 * n); }
    scale = buf.dev<double>((size_t)6 * n); diag2 = buf.dev<double>((size_t)6 * n); Bs = buf.dev<double>((size_t)36 * n);
    A0 = buf.dev<double>((size_t)36 * n); C0 = buf.dev<double>((size_t)36 * n);
    for (int q = 0; q < 2; q++) {
      Ab[q] = buf.dev<double>((size_t)36 * n); Bb[q] = buf.dev<double>((size_t)36 * n); Cb[q] = buf.dev<double>((size_t)36 * n);
      db[q] = buf.dev<double>((size_t)6 * n); Ib[q] = buf.dev<double>((size_t)36 * n);
    }
    d0 = buf.dev<double>((size_t)6 * n); y = buf.dev<double>((size_t)6 * n);
    fail = buf.dev<int>(1);
    dst = (PgState *)buf.dev<double>((sizeof(PgState) + 7) / 8);      // (cleared with the slab: cur = lb = 0, done = 0, ...)
    red3 = buf.dev<double>(4 + 4 * (size_t)PGR_MAXSEG);
    xo = buf.pinned<double>((size_t)7 * n);
    if (pass == 0 && !buf.commit()) { ctx_set_error(c, "gfbe_pg_solve: device allocation failed"); return GFBE_DEVICE_ERROR; }
  }
  const dim3 g6((6 * n + 191) / 192), b6(192);
  const int nseg = (n + PGR_SEG - 1) / PGR_SEG;
  double x_norm0;
  {
    double s2 = 0;
    for (size_t e = 0; e < (size_t)7 * n; e++) s2 += pose_in[e] * pose_in[e];
    x_norm0 = std::sqrt(s2);
  }
  auto reduce = [&](const double *a, const double *b, const double *c3, const double *d4, bool take_max, int mode) {
    const bool direct = n <= (int)PGD_DIRECT_MAX;      // (one launch: the reduction and what follows from it)
    if (!direct) hipLaunchKernelGGL(k_pg_reduce1, dim3(nseg), dim3(256), 0, s, n, a, b, c3, d4, take_max ? 1 : 0, red3 + 4);
    hipLaunchKernelGGL(k_pg_decide, dim3(1), dim3(PGD_THREADS), 0, s, n, a, b, c3, d4, nseg, direct ? (const double *)nullptr : (const double *)(red3 + 4), take_max ? 1 : 0,
                       dst, fail, mode, x_norm0);
  };
  hipLaunchKernelGGL(k_pg_lin_st, g6, b6, 0, s, P, dst, S, 0, per);
  reduce(per, nullptr, nullptr, nullptr, false, 0);
  // max_it passes, enqueued blindly: every kernel of a pass returns at once when the loop has ended (PgState::done), the candidate's
  // evaluation also when the pass has no step to try
  for (int pass = 0; pass < max_it; pass++) {
    // system at the current point (B includes the LM diagonal at the current radius; the copies the reduction consumes; the failure flag cleared)
    hipLaunchKernelGGL(k_pg_system, g6, b6, 0, s, n, dst, S, scale, diag2, Bb[0], d0, Bs, per4, db[0], Ib[0], fail);
    hipLaunchKernelGGL(k_pg_system2, g6, b6, 0, s, n, dst, S, scale, A0, C0, Ab[0], Cb[0]);
    // parallel block cyclic reduction: log2(n) sweeps
    int cur = 0;
    for (int stride = 1; stride < n; stride *= 2) {
      hipLaunchKernelGGL(k_pg_pcr, dim3((n + PCR_ROWS - 1) / PCR_ROWS), dim3(PCR_ROWS * 6), 0, s, n, stride, Ab[cur], Bb[cur], Cb[cur], db[cur], Ib[cur], Ab[1 - cur], Bb[1 - cur], Cb[1 - cur], db[1 - cur], Ib[1 - cur], fail, dst);
      cur = 1 - cur;
    }
    hipLaunchKernelGGL(k_pg_final, g6, b6, 0, s, n, Ib[cur], db[cur], y, dst);
    hipLaunchKernelGGL(k_pg_candidate, g6, b6, 0, s, n, Bs, A0, C0, d0, y, scale, dst, S, per, per2, per3);
    reduce(per4, per, per2, per3, true, 1);      // max |g|, model change, |step|^2, |candidate|^2; the failure flag
    hipLaunchKernelGGL(k_pg_lin_st, g6, b6, 0, s, P, dst, S, 1, per);      // the candidate's cost — and its linearisation, should it be accepted
    reduce(per, nullptr, nullptr, nullptr, false, 2);
  }
  PgState *hst = (PgState *)buf.result();
  hipLaunchKernelGGL(k_pg_finish, dim3(std::min(256, (7 * n + 255) / 256)), dim3(256), 0, s, n, dst, S, xo, hst);
  if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) { ctx_set_error(c, "gfbe_pg_solve: device error"); return GFBE_DEVICE_ERROR; }
  std::memcpy(pose_out, xo, sizeof(double) * 7 * n);
  gfbe_summary sm;
  std::memset(&sm, 0, sizeof sm);
  sm.status = hst->status; sm.termination = hst->done ? hst->termination : 0;
  sm.iterations = hst->it; sm.num_successful = hst->num_successful;
  sm.initial_cost = hst->initial_cost; sm.final_cost = hst->cost; sm.final_radius = hst->radius;
  for (int q = 0; q < 16; q++) { sm.cost_history[q] = hst->cost_history[q]; sm.accepted[q] = (uint8_t)hst->accepted[q]; }
  if (S_out) *S_out = sm;
  return sm.status == GFBE_NUMERICAL_FAILURE ? GFBE_NUMERICAL_FAILURE : GFBE_OK;
}

}  // extern "C"

// oracle/gfo_posegraph.cpp — TEST INFRASTRUCTURE ONLY (CPU oracle).
// The global_fusion pose graph restated:
//   GlobalOptimization::optimize     global_fusion/src/globalOpt.cpp:107-236 (problem set-up, options :117-121)
//   RelativeRTError, TError          global_fusion/src/Factors.h:26-114
// and what ceres::Solve (Ceres 1.14, NOT vendored: PARITY UNPINNED) does with those options: TrustRegionMinimizer with
// LevenbergMarquardtStrategy (radius 1e4, diagonal clamp [1e-6, 1e32], radius / max(1/3, 1 - (2 rho - 1)^3) on success,
// radius / decrease_factor with decrease_factor doubling on failure), Jacobi scaling 1 / (1 + sqrt(diag)),
// HuberLoss(1.0) on the position fixes through the Corrector, QuaternionParameterization. The reference differentiates
// the functors automatically; the tangent Jacobians below are derived by hand (left perturbation q <- [1, d] * q) and
// pinned against central differences in tests/test_posegraph_oracle.py. Linear solver: block-tridiagonal Cholesky
// (the reference's SPARSE_NORMAL_CHOLESKY solves the same normal equations).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "gfo_api.h"

namespace {
struct Q { double w, x, y, z; };
inline Q qmul(Q a, Q b) { return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                                  a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w}; }
inline Q qinv(Q a) { return {a.w, -a.x, -a.y, -a.z}; }
inline void qrot(Q q, double R[9]) {   // unit quaternion assumed after normalisation (QuaternionRotatePoint normalises)
  const double n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  const double w = q.w / n, x = q.x / n, y = q.y / n, z = q.z / n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
// left / right quaternion product matrices (w first): a*b = L(a) b = R(b) a
inline void Lmat(Q a, double M[16]) { const double m[16] = {a.w, -a.x, -a.y, -a.z, a.x, a.w, -a.z, a.y, a.y, a.z, a.w, -a.x, a.z, -a.y, a.x, a.w}; std::memcpy(M, m, sizeof m); }
inline void Rmat(Q b, double M[16]) { const double m[16] = {b.w, -b.x, -b.y, -b.z, b.x, b.w, b.z, -b.y, b.y, -b.z, b.w, b.x, b.z, b.y, -b.x, b.w}; std::memcpy(M, m, sizeof m); }

struct PG {
  int n, n_rel, n_fix;
  const int32_t *rel_i, *fix_i;
  const double *rel_meas, *fix_meas;
  double t_var, q_var, delta;
};

// one RelativeRTError: r(6), J(6 x 12) columns dq_i t_i dq_j t_j
void rel_factor(const double *pi, const double *pj, const double *meas, double t_var, double q_var, double *r, double *J) {
  const Q qi = {pi[3], pi[4], pi[5], pi[6]}, qj = {pj[3], pj[4], pj[5], pj[6]}, qm = {meas[3], meas[4], meas[5], meas[6]};
  double Ri[9];
  qrot(qi, Ri);
  const double d[3] = {pj[0] - pi[0], pj[1] - pi[1], pj[2] - pi[2]};
  for (int a = 0; a < 3; a++) r[a] = (Ri[a] * d[0] + Ri[3 + a] * d[1] + Ri[6 + a] * d[2] - meas[a]) / t_var;   // R_i^T d
  const Q A = qmul(qinv(qm), qinv(qi)), e = qmul(A, qj);
  r[3] = 2 * e.x / q_var; r[4] = 2 * e.y / q_var; r[5] = 2 * e.z / q_var;
  if (!J) return;
  std::memset(J, 0, sizeof(double) * 72);
  // d r_t / d dq_i = 2 R_i^T [d]x / t_var ; d r_t / d t_i = -R_i^T / t_var ; d r_t / d t_j = R_i^T / t_var
  const double dx[9] = {0, -d[2], d[1], d[2], 0, -d[0], -d[1], d[0], 0};
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += Ri[3 * k + a] * dx[3 * k + b];
      J[a * 12 + b] = 2 * s / t_var;
      J[a * 12 + 3 + b] = -Ri[3 * b + a] / t_var;
      J[a * 12 + 9 + b] = Ri[3 * b + a] / t_var;
    }
  // d e / d dq_j = [L(A) R(q_j)](:, 1:4); d e / d dq_i = -the same
  double La[16], Rj[16];
  Lmat(A, La); Rmat(qj, Rj);
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += La[4 * (a + 1) + k] * Rj[4 * k + (b + 1)];
      J[(3 + a) * 12 + 6 + b] = 2 * s / q_var;
      J[(3 + a) * 12 + b] = -2 * s / q_var;
    }
}

struct Lin {
  std::vector<double> Hd, Ho, g;   // diagonal blocks [n][36], sub-diagonal blocks (i+1, i) [n-1][36], gradient [n][6]
  double cost;
};

double huber_corrector(double sq, double delta, double *s1, double *rs, double *asn) {
  // ceres::HuberLoss + Corrector (marginalization_factor.cpp:46-77 is the citable restatement)
  const double b = delta * delta;
  double rho0, rho1, rho2;
  if (sq > b) { const double rr = std::sqrt(sq); rho0 = 2 * delta * rr - b; rho1 = std::max(1e-300, delta / rr); rho2 = -rho1 / (2 * sq); }
  else { rho0 = sq; rho1 = 1.0; rho2 = 0.0; }
  const double sqrt_rho1 = std::sqrt(rho1);
  if (sq == 0.0 || rho2 <= 0.0) { *s1 = sqrt_rho1; *rs = sqrt_rho1; *asn = 0.0; }
  else {
    const double D = 1.0 + 2.0 * sq * rho2 / rho1, alpha = 1.0 - std::sqrt(D);
    *s1 = sqrt_rho1; *rs = sqrt_rho1 / (1.0 - alpha); *asn = alpha / sq;
  }
  return 0.5 * rho0;
}

double evaluate(const PG &P, const double *pose, Lin *L, double *rel_r, double *rel_J, double *fix_r) {
  const int n = P.n;
  if (L) { L->Hd.assign((size_t)n * 36, 0.0); L->Ho.assign((size_t)std::max(n - 1, 0) * 36, 0.0); L->g.assign((size_t)n * 6, 0.0); }
  double cost = 0.0;
  for (int k = 0; k < P.n_rel; k++) {
    const int i = P.rel_i[k], j = i + 1;
    double r[6], J[72];
    rel_factor(pose + 7 * i, pose + 7 * j, P.rel_meas + 7 * k, P.t_var, P.q_var, r, (L || rel_J) ? J : nullptr);
    for (int a = 0; a < 6; a++) cost += 0.5 * r[a] * r[a];
    if (rel_r) std::memcpy(rel_r + 6 * k, r, sizeof r);
    if (rel_J) std::memcpy(rel_J + 72 * k, J, sizeof J);
    if (L) {
      for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) {
          double sii = 0, sjj = 0, sji = 0;
          for (int q = 0; q < 6; q++) { sii += J[q * 12 + a] * J[q * 12 + b]; sjj += J[q * 12 + 6 + a] * J[q * 12 + 6 + b]; sji += J[q * 12 + 6 + a] * J[q * 12 + b]; }
          L->Hd[(size_t)i * 36 + a * 6 + b] += sii; L->Hd[(size_t)j * 36 + a * 6 + b] += sjj; L->Ho[(size_t)i * 36 + a * 6 + b] += sji;
        }
      for (int a = 0; a < 6; a++) {
        double gi = 0, gj = 0;
        for (int q = 0; q < 6; q++) { gi += J[q * 12 + a] * r[q]; gj += J[q * 12 + 6 + a] * r[q]; }
        L->g[(size_t)i * 6 + a] += gi; L->g[(size_t)j * 6 + a] += gj;
      }
    }
  }
  for (int k = 0; k < P.n_fix; k++) {
    const int i = P.fix_i[k];
    const double *m = P.fix_meas + 4 * k;
    double r[3] = {(pose[7 * i] - m[0]) / m[3], (pose[7 * i + 1] - m[1]) / m[3], (pose[7 * i + 2] - m[2]) / m[3]};
    double s1, rs, asn;
    cost += huber_corrector(r[0] * r[0] + r[1] * r[1] + r[2] * r[2], P.delta, &s1, &rs, &asn);
    // J = I / var; corrected J = s1 (J - asn r r^T J), corrected r = rs r
    double Jc[9];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Jc[3 * a + b] = s1 * ((a == b ? 1.0 : 0.0) - asn * r[a] * r[b]) / m[3];
    for (int a = 0; a < 3; a++) r[a] *= rs;
    if (fix_r) std::memcpy(fix_r + 3 * k, r, sizeof r);
    if (L) {
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
          double s = 0;
          for (int q = 0; q < 3; q++) s += Jc[3 * q + a] * Jc[3 * q + b];
          L->Hd[(size_t)i * 36 + (3 + a) * 6 + 3 + b] += s;
        }
      for (int a = 0; a < 3; a++) { double s = 0; for (int q = 0; q < 3; q++) s += Jc[3 * q + a] * r[q]; L->g[(size_t)i * 6 + 3 + a] += s; }
    }
  }
  if (L) L->cost = cost;
  return cost;
}

// (D + T) y = b for the block-tridiagonal SPD matrix: diagonal blocks Ad [n][36], sub-diagonal blocks Ao (i+1, i).
bool block_tridiag_solve(int n, std::vector<double> Ad, const std::vector<double> &Ao, std::vector<double> b, std::vector<double> &y) {
  // forward: L_ii L_ii^T = Ad_i - W_{i} W_{i}^T, W_{i+1} = Ao_i L_ii^-T
  std::vector<double> W((size_t)std::max(n - 1, 0) * 36, 0.0);
  for (int i = 0; i < n; i++) {
    double *A = &Ad[(size_t)i * 36];
    if (i > 0) {
      const double *Wp = &W[(size_t)(i - 1) * 36];
      for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) { double s = 0; for (int k = 0; k < 6; k++) s += Wp[a * 6 + k] * Wp[c * 6 + k]; A[a * 6 + c] -= s; }
      for (int a = 0; a < 6; a++) { double s = 0; for (int k = 0; k < 6; k++) s += Wp[a * 6 + k] * b[(size_t)(i - 1) * 6 + k]; b[(size_t)i * 6 + a] -= s; }
    }
    for (int c = 0; c < 6; c++) {   // Cholesky in place (lower)
      double dsum = A[c * 6 + c];
      for (int k = 0; k < c; k++) dsum -= A[c * 6 + k] * A[c * 6 + k];
      if (!(dsum > 0.0) || !std::isfinite(dsum)) return false;
      const double lcc = std::sqrt(dsum);
      A[c * 6 + c] = lcc;
      for (int a = c + 1; a < 6; a++) { double s = A[a * 6 + c]; for (int k = 0; k < c; k++) s -= A[a * 6 + k] * A[c * 6 + k]; A[a * 6 + c] = s / lcc; }
    }
    // z_i = L^-1 b_i
    for (int a = 0; a < 6; a++) { double s = b[(size_t)i * 6 + a]; for (int k = 0; k < a; k++) s -= A[a * 6 + k] * b[(size_t)i * 6 + k]; b[(size_t)i * 6 + a] = s / A[a * 6 + a]; }
    if (i + 1 < n) {   // W = Ao L^-T  (rows of Ao solved against L)
      double *Wn = &W[(size_t)i * 36];
      for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) { double s = Ao[(size_t)i * 36 + r * 6 + c]; for (int k = 0; k < c; k++) s -= Wn[r * 6 + k] * A[c * 6 + k]; Wn[r * 6 + c] = s / A[c * 6 + c]; }
    }
  }
  y.assign((size_t)n * 6, 0.0);
  for (int i = n - 1; i >= 0; i--) {
    const double *A = &Ad[(size_t)i * 36];
    double rhs[6];
    for (int a = 0; a < 6; a++) rhs[a] = b[(size_t)i * 6 + a];
    if (i + 1 < n) { const double *Wn = &W[(size_t)i * 36]; for (int a = 0; a < 6; a++) { double s = 0; for (int k = 0; k < 6; k++) s += Wn[k * 6 + a] * y[(size_t)(i + 1) * 6 + k]; rhs[a] -= s; } }
    for (int a = 5; a >= 0; a--) { double s = rhs[a]; for (int k = a + 1; k < 6; k++) s -= A[k * 6 + a] * y[(size_t)i * 6 + k]; y[(size_t)i * 6 + a] = s / A[a * 6 + a]; }
  }
  return true;
}

void plus(const double *x, const double *d6, double *out) {   // QuaternionParameterization::Plus + identity on t
  const double nrm = std::sqrt(d6[0] * d6[0] + d6[1] * d6[1] + d6[2] * d6[2]);
  Q dq;
  if (nrm > 0.0) { const double sc = std::sin(nrm) / nrm; dq = {std::cos(nrm), sc * d6[0], sc * d6[1], sc * d6[2]}; }
  else dq = {1.0, 0.0, 0.0, 0.0};
  const Q q = qmul(dq, {x[3], x[4], x[5], x[6]});
  out[0] = x[0] + d6[3]; out[1] = x[1] + d6[4]; out[2] = x[2] + d6[5];
  out[3] = q.w; out[4] = q.x; out[5] = q.y; out[6] = q.z;
}
}  // namespace

extern "C" {

int32_t gfo_pg_eval(void *, int32_t n, const double *pose, int32_t n_rel, const int32_t *rel_i, const double *rel_meas, double t_var,
                    double q_var, int32_t n_fix, const int32_t *fix_i, const double *fix_meas, double delta, double *rel_r, double *rel_J,
                    double *fix_r, double *cost) {
  const PG P = {n, n_rel, n_fix, rel_i, fix_i, rel_meas, fix_meas, t_var, q_var, delta};
  for (int k = 0; k < n_rel; k++) if (rel_i[k] < 0 || rel_i[k] + 1 >= n) return GFBE_BAD_INPUT;
  for (int k = 0; k < n_fix; k++) if (fix_i[k] < 0 || fix_i[k] >= n) return GFBE_BAD_INPUT;
  const double c = evaluate(P, pose, nullptr, rel_r, rel_J, fix_r);
  if (cost) *cost = c;
  return GFBE_OK;
}

int32_t gfo_pg_solve(void *, int32_t n, const double *pose_in, int32_t n_rel, const int32_t *rel_i, const double *rel_meas, double t_var,
                     double q_var, int32_t n_fix, const int32_t *fix_i, const double *fix_meas, double delta, int32_t max_it,
                     double *pose_out, gfbe_summary *S) {
  const PG P = {n, n_rel, n_fix, rel_i, fix_i, rel_meas, fix_meas, t_var, q_var, delta};
  for (int k = 0; k < n_rel; k++) if (rel_i[k] < 0 || rel_i[k] + 1 >= n) return GFBE_BAD_INPUT;
  for (int k = 0; k < n_fix; k++) if (fix_i[k] < 0 || fix_i[k] >= n) return GFBE_BAD_INPUT;
  max_it = std::min(max_it, 15);
  std::vector<double> x(pose_in, pose_in + (size_t)7 * n), cand((size_t)7 * n);
  gfbe_summary sm;
  std::memset(&sm, 0, sizeof sm);
  Lin L;
  double cost = evaluate(P, x.data(), &L, nullptr, nullptr, nullptr);
  sm.initial_cost = cost; sm.cost_history[0] = cost; sm.status = GFBE_NO_CONVERGENCE;
  std::vector<double> scale((size_t)6 * n, 1.0), y;
  for (int i = 0; i < n; i++) for (int a = 0; a < 6; a++) scale[(size_t)i * 6 + a] = 1.0 / (1.0 + std::sqrt(L.Hd[(size_t)i * 36 + a * 7]));
  double radius = 1e4, decrease = 2.0;
  auto xnorm = [&](const std::vector<double> &v) { double s = 0; for (double e : v) s += e * e; return std::sqrt(s); };
  double x_norm = xnorm(x);
  int invalid = 0, it = 0;
  bool reuse = false;
  std::vector<double> diag2((size_t)6 * n);
  while (true) {
    if (it >= max_it) { sm.termination = 0; break; }
    double gmax = 0; for (double e : L.g) gmax = std::max(gmax, std::fabs(e));
    if (gmax <= 1e-10) { sm.termination = 3; sm.status = GFBE_OK; break; }
    if (radius < 1e-32) { sm.termination = 4; break; }
    it++;
    // scaled system: Hs = S H S, gs = S g; LM diagonal D^2 / radius with D^2 = clamp(diag(Hs), 1e-6, 1e32) (kept on a rejected step)
    std::vector<double> Ad((size_t)n * 36), Ao((size_t)std::max(n - 1, 0) * 36), rhs((size_t)n * 6);
    for (int i = 0; i < n; i++)
      for (int a = 0; a < 6; a++) {
        for (int b = 0; b < 6; b++) Ad[(size_t)i * 36 + a * 6 + b] = L.Hd[(size_t)i * 36 + a * 6 + b] * scale[(size_t)i * 6 + a] * scale[(size_t)i * 6 + b];
        rhs[(size_t)i * 6 + a] = -scale[(size_t)i * 6 + a] * L.g[(size_t)i * 6 + a];
      }
    for (int i = 0; i + 1 < n; i++)
      for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) Ao[(size_t)i * 36 + a * 6 + b] = L.Ho[(size_t)i * 36 + a * 6 + b] * scale[(size_t)(i + 1) * 6 + a] * scale[(size_t)i * 6 + b];
    if (!reuse) for (int i = 0; i < n; i++) for (int a = 0; a < 6; a++) diag2[(size_t)i * 6 + a] = std::min(std::max(Ad[(size_t)i * 36 + a * 7], 1e-6), 1e32);
    std::vector<double> Areg = Ad;
    for (int i = 0; i < n; i++) for (int a = 0; a < 6; a++) Areg[(size_t)i * 36 + a * 7] += diag2[(size_t)i * 6 + a] / radius;
    const bool ok = block_tridiag_solve(n, Areg, Ao, rhs, y);
    double model_change = 0.0;
    if (ok) {   // -(gs . y + 1/2 y^T Hs y)
      double gy = 0, yHy = 0;
      for (int i = 0; i < n; i++) {
        double Hy[6];
        for (int a = 0; a < 6; a++) {
          double s = 0;
          for (int b = 0; b < 6; b++) s += Ad[(size_t)i * 36 + a * 6 + b] * y[(size_t)i * 6 + b];
          if (i > 0) for (int b = 0; b < 6; b++) s += Ao[(size_t)(i - 1) * 36 + a * 6 + b] * y[(size_t)(i - 1) * 6 + b];
          if (i + 1 < n) for (int b = 0; b < 6; b++) s += Ao[(size_t)i * 36 + b * 6 + a] * y[(size_t)(i + 1) * 6 + b];
          Hy[a] = s;
        }
        for (int a = 0; a < 6; a++) { gy += -rhs[(size_t)i * 6 + a] * y[(size_t)i * 6 + a]; yHy += y[(size_t)i * 6 + a] * Hy[a]; }
      }
      model_change = -(gy + 0.5 * yHy);
    }
    if (!ok || !(model_change > 0.0)) {   // invalid step
      sm.accepted[it] = 0; sm.cost_history[it] = cost;
      if (++invalid >= 5) { sm.termination = 4; sm.status = GFBE_NUMERICAL_FAILURE; break; }
      radius /= decrease; decrease *= 2; reuse = true;
      continue;
    }
    invalid = 0;
    double step2 = 0;
    for (int i = 0; i < n; i++) {
      double d6[6];
      for (int a = 0; a < 6; a++) d6[a] = scale[(size_t)i * 6 + a] * y[(size_t)i * 6 + a];
      plus(&x[(size_t)7 * i], d6, &cand[(size_t)7 * i]);
      for (int k = 0; k < 7; k++) { const double df = cand[(size_t)7 * i + k] - x[(size_t)7 * i + k]; step2 += df * df; }
    }
    const double cand_cost = evaluate(P, cand.data(), nullptr, nullptr, nullptr, nullptr);
    sm.cost_history[it] = cost;
    if (std::sqrt(step2) <= 1e-8 * (x_norm + 1e-8)) { sm.termination = 2; sm.status = GFBE_OK; break; }
    const double change = cost - cand_cost;
    if (std::fabs(change) <= 1e-6 * cost) { sm.termination = 1; sm.status = GFBE_OK; break; }
    const double rho = change / model_change;
    if (rho > 1e-3) {
      x = cand; cost = cand_cost; x_norm = xnorm(x);
      sm.accepted[it] = 1; sm.num_successful++; sm.cost_history[it] = cost;
      radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3)));
      decrease = 2.0; reuse = false;
      evaluate(P, x.data(), &L, nullptr, nullptr, nullptr);
    } else {
      sm.accepted[it] = 0;
      radius /= decrease; decrease *= 2; reuse = true;
    }
  }
  sm.iterations = it; sm.final_cost = cost; sm.final_radius = radius;
  std::memcpy(pose_out, x.data(), sizeof(double) * 7 * n);
  if (S) *S = sm;
  return sm.status == GFBE_NUMERICAL_FAILURE ? GFBE_NUMERICAL_FAILURE : GFBE_OK;
}

}  // extern "C"

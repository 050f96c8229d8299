// oracle/gfo_lio.cpp — TEST INFRASTRUCTURE ONLY (CPU oracle).
// LiDAR point-to-plane factors restated:
//   LidarPlaneNormFactor::Evaluate     lio/src/liw/lidarFactor.cpp:18-51
//   CTLidarPlaneNormFactor::Evaluate   lio/src/liw/lidarFactor.cpp:59-120 (Eigen's Quaternion::slerp restated; Qleft / Qright
//                                      of lio_utils.h:127-144; the 3 x 3 inverse by cofactors)
// The reference checks these analytic Jacobians against automatic differentiation of PointToPlaneFunctor at 1e-6 on a fixed
// input (lio/src/apps/test_analytic_factor.cpp:56-134): tests/test_lio_oracle.py replays that input.
#include <cmath>
#include <cstring>

#include "gfo_api.h"

namespace {
struct Qx { double x, y, z, w; };
inline Qx qmul(Qx a, Qx b) { return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
                                     a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z}; }
inline void qrot(Qx q, double R[9]) {
  const double x = q.x, y = q.y, z = q.z, w = q.w;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
inline Qx slerp(Qx a, double t, Qx b) {   // Eigen::QuaternionBase::slerp
  const double one = 1.0 - 2.220446049250313e-16;
  const double d = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w, ad = std::fabs(d);
  double s0, s1;
  if (ad >= one) { s0 = 1.0 - t; s1 = t; }
  else { const double th = std::acos(ad), st = std::sin(th); s0 = std::sin((1.0 - t) * th) / st; s1 = std::sin(t * th) / st; }
  if (d < 0) s1 = -s1;
  return {s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z, s0 * a.w + s1 * b.w};
}
inline Qx normalized(Qx q) { const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w); return {q.x / n, q.y / n, q.z / n, q.w / n}; }
inline void q_br(Qx q, double sgn, double M[9]) {   // bottom-right 3x3 of Qleft (sgn +1) / Qright (sgn -1): w I +- [v]x
  M[0] = q.w; M[1] = -sgn * q.z; M[2] = sgn * q.y; M[3] = sgn * q.z; M[4] = q.w; M[5] = -sgn * q.x; M[6] = -sgn * q.y; M[7] = sgn * q.x; M[8] = q.w;
}
inline void inv3(const double A[9], double B[9]) {
  const double c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
  const double det = A[0] * c0 + A[1] * c1 + A[2] * c2;
  B[0] = c0 / det; B[1] = (A[2] * A[7] - A[1] * A[8]) / det; B[2] = (A[1] * A[5] - A[2] * A[4]) / det;
  B[3] = c1 / det; B[4] = (A[0] * A[8] - A[2] * A[6]) / det; B[5] = (A[2] * A[3] - A[0] * A[5]) / det;
  B[6] = c2 / det; B[7] = (A[1] * A[6] - A[0] * A[7]) / det; B[8] = (A[0] * A[4] - A[1] * A[3]) / det;
}
inline void mm(const double *A, const double *B, double *C) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j]; C[3 * i + j] = s; } }
}  // namespace

extern "C" int32_t gfo_lio_linearize(void *, int32_t ct, int32_t n, const double *pts, const double *normals, const double *offsets,
                                     const double *alpha, const double *weights, double sqrt_info, const double *pb, const double *pe,
                                     double *r, double *J, double *H, double *g, double *cost) {
  const int dnum = ct ? 12 : 6;
  if (H) std::memset(H, 0, sizeof(double) * dnum * dnum);
  if (g) std::memset(g, 0, sizeof(double) * dnum);
  double c = 0.0;
  const Qx qb = {pb[3], pb[4], pb[5], pb[6]};
  Qx qe = qb;
  if (ct) qe = {pe[3], pe[4], pe[5], pe[6]};
  for (int k = 0; k < n; k++) {
    const double *p = pts + 3 * k, *nv = normals + 3 * k;
    const double wgt = weights ? weights[k] : 1.0;
    double Jk[12] = {0}, rk;
    if (!ct) {
      double R[9];
      qrot(qb, R);
      const double pw[3] = {R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + pb[0], R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + pb[1], R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + pb[2]};
      rk = sqrt_info * wgt * (nv[0] * pw[0] + nv[1] * pw[1] + nv[2] * pw[2] + offsets[k]);
      // J_t = sqrt_info w n^T ; J_theta = -sqrt_info w n^T R [p]x
      const double nR[3] = {nv[0] * R[0] + nv[1] * R[3] + nv[2] * R[6], nv[0] * R[1] + nv[1] * R[4] + nv[2] * R[7], nv[0] * R[2] + nv[1] * R[5] + nv[2] * R[8]};
      for (int a = 0; a < 3; a++) Jk[a] = sqrt_info * wgt * nv[a];
      Jk[3] = -sqrt_info * wgt * (nR[1] * p[2] - nR[2] * p[1]);     // (nR^T [p]x)_0 = nR1 p2 - nR2 p1 ... see skew below
      Jk[4] = -sqrt_info * wgt * (nR[2] * p[0] - nR[0] * p[2]);
      Jk[5] = -sqrt_info * wgt * (nR[0] * p[1] - nR[1] * p[0]);
    } else {
      const double al = alpha[k];
      const Qx qs = normalized(slerp(qb, al, qe));
      double R[9];
      qrot(qs, R);
      const double ts[3] = {pb[0] * (1 - al) + pe[0] * al, pb[1] * (1 - al) + pe[1] * al, pb[2] * (1 - al) + pe[2] * al};
      const double pw[3] = {R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + ts[0], R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + ts[1], R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + ts[2]};
      rk = sqrt_info * wgt * (nv[0] * pw[0] + nv[1] * pw[1] + nv[2] * pw[2] + offsets[k]);
      const double nR[3] = {nv[0] * R[0] + nv[1] * R[3] + nv[2] * R[6], nv[0] * R[1] + nv[1] * R[4] + nv[2] * R[7], nv[0] * R[2] + nv[1] * R[5] + nv[2] * R[8]};
      const double jrs[3] = {-wgt * (nR[1] * p[2] - nR[2] * p[1]), -wgt * (nR[2] * p[0] - nR[0] * p[2]), -wgt * (nR[0] * p[1] - nR[1] * p[0])};
      const Qx qbi = {-qb.x, -qb.y, -qb.z, qb.w};
      const Qx rd = qmul(qbi, qe);                                  // rot_begin.inverse() * rot_end (unit inputs)
      const Qx ident = {0, 0, 0, 1};
      const Qx rds = slerp(ident, al, rd);
      double Rds[9], Ql_s[9], Ql_d[9], Qr_s[9], Qr_d[9], inv[9], T1[9], T2[9], Jb[9], Je[9];
      qrot(rds, Rds);
      q_br(rds, +1, Ql_s); q_br(rd, +1, Ql_d); q_br(rds, -1, Qr_s); q_br(rd, -1, Qr_d);
      inv3(Ql_d, inv); mm(Ql_s, inv, T1);
      for (int q = 0; q < 9; q++) T2[q] = ((q % 4 == 0) ? 1.0 : 0.0) - al * T1[q];
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int m = 0; m < 3; m++) s += Rds[3 * m + i] * T2[3 * m + j]; Jb[3 * i + j] = s; }   // Rds^T (I - a Ql_s Ql_d^-1)
      inv3(Qr_d, inv); mm(Qr_s, inv, T1);
      for (int q = 0; q < 9; q++) Je[q] = al * T1[q];
      for (int a = 0; a < 3; a++) {
        Jk[a] = sqrt_info * wgt * nv[a] * (1 - al);
        Jk[6 + a] = sqrt_info * wgt * nv[a] * al;
        Jk[3 + a] = sqrt_info * (jrs[0] * Jb[a] + jrs[1] * Jb[3 + a] + jrs[2] * Jb[6 + a]);
        Jk[9 + a] = sqrt_info * (jrs[0] * Je[a] + jrs[1] * Je[3 + a] + jrs[2] * Je[6 + a]);
      }
    }
    c += 0.5 * rk * rk;
    if (r) r[k] = rk;
    if (J) std::memcpy(J + (size_t)dnum * k, Jk, sizeof(double) * dnum);
    if (H) for (int a = 0; a < dnum; a++) for (int b = 0; b < dnum; b++) H[a * dnum + b] += Jk[a] * Jk[b];
    if (g) for (int a = 0; a < dnum; a++) g[a] += Jk[a] * rk;
  }
  if (cost) *cost = c;
  return GFBE_OK;
}

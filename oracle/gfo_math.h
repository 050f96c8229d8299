// oracle/gfo_math.h — TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
//
// Dependency-free FP64 restatement of the small SO(3)/quaternion toolbox the reference's hot path
// uses (Eigen/Sophus are not available in this container):
//   Utility::deltaQ / skewSymmetric / Qleft / Qright / R2ypr / ypr2R   VE/utility/utility.h:23-120
//   Sophus::SO3d::exp / log                                            GF/lio/thirdparty/sophus/so3.hpp
//   Sophus::rightJacobianSO3 / rightJacobianInvSO3                     VE/utility/sophus_utils.hpp:154-236
//   Eigen::Quaterniond product / toRotationMatrix / Quaterniond(Matrix3d) (Eigen 3.3.7, published algorithm)
// (VE = Ground-Fusion++/vins_estimator/src). Parity with Eigen/Ceres is UNPINNED: nothing in the
// reference tests these; the functions are pinned by tests/test_oracle_*.py (numpy re-derivation,
// algebraic identities, central differences).
#pragma once
#include <cmath>
#include <cstring>

namespace gfo {

struct V3 { double x, y, z; };
struct Q4 { double x, y, z, w; };      // Hamilton, storage order x y z w (estimator.cpp:2345-2348)
struct M3 { double m[3][3]; };

inline V3 v3(double x, double y, double z) { return {x, y, z}; }
inline V3 v3(const double *p) { return {p[0], p[1], p[2]}; }
inline Q4 q4(const double *p) { return {p[0], p[1], p[2], p[3]}; }   // from [x y z w]
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, double s) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm2(V3 a) { return dot(a, a); }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }
inline double get(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

inline M3 zero3() { M3 r; std::memset(&r, 0, sizeof r); return r; }
inline M3 eye3() { M3 r = zero3(); r.m[0][0] = r.m[1][1] = r.m[2][2] = 1.0; return r; }
inline M3 diag3(double a, double b, double c) { M3 r = zero3(); r.m[0][0] = a; r.m[1][1] = b; r.m[2][2] = c; return r; }
inline M3 operator*(const M3 &a, const M3 &b) {
  M3 r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
    r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
inline V3 operator*(const M3 &a, V3 v) {
  return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z,
          a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
          a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
inline M3 operator*(double s, const M3 &a) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = s * a.m[i][j]; return r; }
inline M3 operator+(const M3 &a, const M3 &b) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + b.m[i][j]; return r; }
inline M3 operator-(const M3 &a, const M3 &b) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] - b.m[i][j]; return r; }
inline M3 operator-(const M3 &a) { return (-1.0) * a; }
inline M3 T(const M3 &a) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i]; return r; }

// Utility::skewSymmetric, utility.h:39-47
inline M3 skew(V3 q) {
  M3 r = zero3();
  r.m[0][1] = -q.z; r.m[0][2] = q.y;
  r.m[1][0] = q.z;  r.m[1][2] = -q.x;
  r.m[2][0] = -q.y; r.m[2][1] = q.x;
  return r;
}

// Eigen quaternion product (Hamilton).
inline Q4 operator*(Q4 a, Q4 b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x,
          a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
// Eigen's Quaternion::inverse() divides the conjugate by the squared norm.
inline Q4 inv(Q4 q) {
  double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  return {-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2};
}
inline Q4 conj(Q4 q) { return {-q.x, -q.y, -q.z, q.w}; }
inline Q4 normalized(Q4 q) {
  double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return {q.x / n, q.y / n, q.z / n, q.w / n};
}
inline V3 vec(Q4 q) { return {q.x, q.y, q.z}; }

// Eigen QuaternionBase::toRotationMatrix (no normalisation).
inline M3 rot(Q4 q) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  M3 r;
  r.m[0][0] = 1 - (tyy + tzz); r.m[0][1] = txy - twz;       r.m[0][2] = txz + twy;
  r.m[1][0] = txy + twz;       r.m[1][1] = 1 - (txx + tzz); r.m[1][2] = tyz - twx;
  r.m[2][0] = txz - twy;       r.m[2][1] = tyz + twx;       r.m[2][2] = 1 - (txx + tyy);
  return r;
}
// q * v as Eigen does it (rotation of a vector by a quaternion).
inline V3 rotv(Q4 q, V3 v) { return rot(q) * v; }

// Eigen Quaterniond(Matrix3d): trace branch / largest-diagonal branch. Needed because
// vector2double() rebuilds every quaternion from Rs[i] (estimator.cpp:2344), which fixes its sign.
inline Q4 quat_from_rot(const M3 &R) {
  double q[4];  // x y z w
  double t = R.m[0][0] + R.m[1][1] + R.m[2][2];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R.m[2][1] - R.m[1][2]) * t;
    q[1] = (R.m[0][2] - R.m[2][0]) * t;
    q[2] = (R.m[1][0] - R.m[0][1]) * t;
  } else {
    int i = 0;
    if (R.m[1][1] > R.m[0][0]) i = 1;
    if (R.m[2][2] > R.m[i][i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R.m[i][i] - R.m[j][j] - R.m[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R.m[k][j] - R.m[j][k]) * t;
    q[j] = (R.m[j][i] + R.m[i][j]) * t;
    q[k] = (R.m[k][i] + R.m[i][k]) * t;
  }
  return {q[0], q[1], q[2], q[3]};
}

// Utility::deltaQ, utility.h:23-36: [1, theta/2] normalised.
inline Q4 deltaQ(V3 theta) {
  Q4 dq = {theta.x / 2.0, theta.y / 2.0, theta.z / 2.0, 1.0};
  return normalized(dq);
}

// Utility::Qleft / Qright (utility.h:59-76); 4x4 in (w, x, y, z) order. Only the bottom-right
// 3x3 corner is ever used by the factors: w*I + skew(v)  resp.  w*I - skew(v).
inline M3 Qleft_br(Q4 q) { return q.w * eye3() + skew(vec(q)); }
inline M3 Qright_br(Q4 q) { return q.w * eye3() - skew(vec(q)); }

// Utility::R2ypr (degrees!) utility.h:78-94 and ypr2R utility.h:96-120.
inline V3 R2ypr(const M3 &R) {
  V3 n = {R.m[0][0], R.m[1][0], R.m[2][0]};
  V3 o = {R.m[0][1], R.m[1][1], R.m[2][1]};
  V3 a = {R.m[0][2], R.m[1][2], R.m[2][2]};
  double y = std::atan2(n.y, n.x);
  double p = std::atan2(-n.z, n.x * std::cos(y) + n.y * std::sin(y));
  double r = std::atan2(a.x * std::sin(y) - a.y * std::cos(y), -o.x * std::sin(y) + o.y * std::cos(y));
  return {y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0};
}
inline M3 ypr2R(V3 ypr) {
  double y = ypr.x / 180.0 * M_PI, p = ypr.y / 180.0 * M_PI, r = ypr.z / 180.0 * M_PI;
  M3 Rz = zero3(), Ry = zero3(), Rx = zero3();
  Rz.m[0][0] = std::cos(y); Rz.m[0][1] = -std::sin(y); Rz.m[1][0] = std::sin(y); Rz.m[1][1] = std::cos(y); Rz.m[2][2] = 1;
  Ry.m[0][0] = std::cos(p); Ry.m[0][2] = std::sin(p); Ry.m[1][1] = 1; Ry.m[2][0] = -std::sin(p); Ry.m[2][2] = std::cos(p);
  Rx.m[0][0] = 1; Rx.m[1][1] = std::cos(r); Rx.m[1][2] = -std::sin(r); Rx.m[2][1] = std::sin(r); Rx.m[2][2] = std::cos(r);
  return Rz * Ry * Rx;
}

// Sophus constants: epsilon = 1e-10 (double), epsilonSqrt = 1e-5 (common.hpp:117-121).
constexpr double kSophusEps = 1e-10;

// Sophus::SO3d::exp (so3.hpp expAndTheta).
inline Q4 so3_exp(V3 w) {
  double th2 = norm2(w), imag, real;
  if (th2 < kSophusEps * kSophusEps) {
    double th4 = th2 * th2;
    imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
    real = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
  } else {
    double th = std::sqrt(th2), h = 0.5 * th;
    imag = std::sin(h) / th;
    real = std::cos(h);
  }
  return {imag * w.x, imag * w.y, imag * w.z, real};
}
// Sophus::SO3d::log (so3.hpp logAndTheta), atan-based.
inline V3 so3_log(Q4 q) {
  double n2 = q.x * q.x + q.y * q.y + q.z * q.z, w = q.w, f;
  if (n2 < kSophusEps * kSophusEps) {
    f = 2.0 / w - (2.0 / 3.0) * n2 / (w * w * w);
  } else {
    double n = std::sqrt(n2);
    if (std::fabs(w) < kSophusEps) f = (w > 0 ? M_PI : -M_PI) / n;
    else f = 2.0 * std::atan(n / w) / n;
  }
  return {f * q.x, f * q.y, f * q.z};
}

// Sophus::rightJacobianSO3, sophus_utils.hpp:154-183.
inline M3 right_jac(V3 phi) {
  double n2 = norm2(phi);
  M3 h = skew(phi), h2 = h * h, J = eye3();
  if (n2 > kSophusEps) {
    double n = std::sqrt(n2), n3 = n2 * n;
    J = J - ((1 - std::cos(n)) / n2) * h;
    J = J + ((n - std::sin(n)) / n3) * h2;
  } else {
    J = J - 0.5 * h;
    J = J + (1.0 / 6.0) * h2;
  }
  return J;
}
// Sophus::rightJacobianInvSO3, sophus_utils.hpp:194-236.
inline M3 right_jac_inv(V3 phi) {
  double n2 = norm2(phi);
  M3 h = skew(phi), h2 = h * h, J = eye3() + 0.5 * h;
  if (n2 > kSophusEps) {
    double n = std::sqrt(n2);
    if (n < M_PI - 1e-5) J = J + (1.0 / n2 - (1.0 + std::cos(n)) / (2.0 * n * std::sin(n))) * h2;
    else J = J + (1.0 / (M_PI * M_PI)) * h2;
  } else {
    J = J + (1.0 / 12.0) * h2;
  }
  return J;
}

// ---- small dense helpers on row-major arrays -------------------------------------------------
// C(n x p) = A(n x m) * B(m x p)
inline void matmul(const double *A, const double *B, double *C, int n, int m, int p) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < p; j++) {
      double s = 0;
      for (int k = 0; k < m; k++) s += A[i * m + k] * B[k * p + j];
      C[i * p + j] = s;
    }
}
inline void set_block(double *A, int lda, int r0, int c0, const M3 &B) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[(r0 + i) * lda + c0 + j] = B.m[i][j];
}
inline M3 get_block(const double *A, int lda, int r0, int c0) {
  M3 B; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) B.m[i][j] = A[(r0 + i) * lda + c0 + j]; return B;
}

// Eigen's Matrix::inverse() for sizes > 4 is PartialPivLU; restated (row-major, in/out separate).
// Returns false on an exactly-zero pivot.
bool lu_inverse(const double *A, double *Ainv, int n);
// Eigen::LLT: lower Cholesky A = L L^T. Returns false if a pivot <= 0 (Eigen: NumericalIssue).
bool llt_lower(const double *A, double *L, int n);
// sqrt_info = LLT(cov^-1).matrixL().transpose()  (imu_factor.h:73, wheel_factor.h:85); row-major n x n.
bool sqrt_info_from_cov(const double *cov, double *sqrt_info, int n);
// Symmetric eigen-decomposition A = V diag(w) V^T (cyclic Jacobi; stands in for
// Eigen::SelfAdjointEigenSolver, marginalization_factor.cpp:279,294). w ascending, V column j = vector j.
void sym_eig(const double *A, int n, double *w, double *V);

}  // namespace gfo

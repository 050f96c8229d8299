// gfbe_math.h — FP64 SO(3)/quaternion device toolbox for the HIP back end (gfx950).
//
// Everything here is __host__ __device__ so the factor arithmetic can also be exercised by the
// host-compiled unit harness in tests/ (tests/host_shim.cpp); the product only ever calls it from
// kernels. Semantics follow the reference's helpers:
//   Utility::deltaQ / skewSymmetric / Qleft / Qright / R2ypr / ypr2R
//       Ground-Fusion++/vins_estimator/src/utility/utility.h:23-120
//   Sophus::SO3d::exp / log            (Sophus; vendored copy Ground-Fusion++/lio/thirdparty/sophus/so3.hpp)
//   Sophus::rightJacobianSO3 / rightJacobianInvSO3
//       Ground-Fusion++/vins_estimator/src/utility/sophus_utils.hpp:154-236
// Quaternions are stored x,y,z,w (estimator.cpp:2345-2348).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define GF_HD __host__ __device__ __forceinline__

namespace gfd {

struct vec3 {
  double v[3];
  GF_HD double &operator[](int i) { return v[i]; }
  GF_HD const double &operator[](int i) const { return v[i]; }
};
struct quat {  // x y z w
  double x, y, z, w;
};
struct mat3 {
  double m[9];  // row-major
  GF_HD double &operator()(int r, int c) { return m[3 * r + c]; }
  GF_HD const double &operator()(int r, int c) const { return m[3 * r + c]; }
};

GF_HD vec3 mk3(double a, double b, double c) { vec3 r; r[0] = a; r[1] = b; r[2] = c; return r; }
GF_HD vec3 ld3(const double *p) { return mk3(p[0], p[1], p[2]); }
GF_HD quat ldq(const double *p) { quat q; q.x = p[0]; q.y = p[1]; q.z = p[2]; q.w = p[3]; return q; }
GF_HD vec3 add(const vec3 &a, const vec3 &b) { return mk3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
GF_HD vec3 sub(const vec3 &a, const vec3 &b) { return mk3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
GF_HD vec3 scl(double s, const vec3 &a) { return mk3(s * a[0], s * a[1], s * a[2]); }
GF_HD vec3 neg(const vec3 &a) { return mk3(-a[0], -a[1], -a[2]); }
GF_HD double dot3(const vec3 &a, const vec3 &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

GF_HD mat3 ident3() { mat3 r; for (int i = 0; i < 9; i++) r.m[i] = 0.0; r.m[0] = r.m[4] = r.m[8] = 1.0; return r; }
GF_HD mat3 diagm(double a, double b, double c) { mat3 r; for (int i = 0; i < 9; i++) r.m[i] = 0.0; r.m[0] = a; r.m[4] = b; r.m[8] = c; return r; }
GF_HD mat3 mul(const mat3 &a, const mat3 &b) {
  mat3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j);
  return r;
}
// a^T * b
GF_HD mat3 tmul(const mat3 &a, const mat3 &b) {
  mat3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r(i, j) = a(0, i) * b(0, j) + a(1, i) * b(1, j) + a(2, i) * b(2, j);
  return r;
}
GF_HD vec3 mv(const mat3 &a, const vec3 &x) {
  return mk3(a(0, 0) * x[0] + a(0, 1) * x[1] + a(0, 2) * x[2], a(1, 0) * x[0] + a(1, 1) * x[1] + a(1, 2) * x[2],
             a(2, 0) * x[0] + a(2, 1) * x[1] + a(2, 2) * x[2]);
}
// a^T * x
GF_HD vec3 tmv(const mat3 &a, const vec3 &x) {
  return mk3(a(0, 0) * x[0] + a(1, 0) * x[1] + a(2, 0) * x[2], a(0, 1) * x[0] + a(1, 1) * x[1] + a(2, 1) * x[2],
             a(0, 2) * x[0] + a(1, 2) * x[1] + a(2, 2) * x[2]);
}
GF_HD mat3 transp(const mat3 &a) { mat3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = a(j, i); return r; }
GF_HD mat3 madd(const mat3 &a, const mat3 &b) { mat3 r; for (int i = 0; i < 9; i++) r.m[i] = a.m[i] + b.m[i]; return r; }
GF_HD mat3 msub(const mat3 &a, const mat3 &b) { mat3 r; for (int i = 0; i < 9; i++) r.m[i] = a.m[i] - b.m[i]; return r; }
GF_HD mat3 mscl(double s, const mat3 &a) { mat3 r; for (int i = 0; i < 9; i++) r.m[i] = s * a.m[i]; return r; }
GF_HD mat3 mneg(const mat3 &a) { return mscl(-1.0, a); }

// [v]x  (utility.h:39-47)
GF_HD mat3 hat(const vec3 &v) {
  mat3 r;
  r.m[0] = 0.0;   r.m[1] = -v[2]; r.m[2] = v[1];
  r.m[3] = v[2];  r.m[4] = 0.0;   r.m[5] = -v[0];
  r.m[6] = -v[1]; r.m[7] = v[0];  r.m[8] = 0.0;
  return r;
}

// Hamilton product.
GF_HD quat qmul(const quat &a, const quat &b) {
  quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
// conjugate / |q|^2  (what Eigen's inverse() returns; inputs are not assumed unit).
GF_HD quat qinv(const quat &q) {
  const double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  quat r; r.x = -q.x / n2; r.y = -q.y / n2; r.z = -q.z / n2; r.w = q.w / n2; return r;
}
GF_HD quat qnormalize(const quat &q) {
  const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  quat r; r.x = q.x / n; r.y = q.y / n; r.z = q.z / n; r.w = q.w / n; return r;
}
GF_HD vec3 qvec(const quat &q) { return mk3(q.x, q.y, q.z); }

// Rotation matrix of a (not necessarily unit) quaternion, Eigen's toRotationMatrix formula.
GF_HD mat3 qrot(const quat &q) {
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  mat3 r;
  r.m[0] = 1.0 - (tyy + tzz); r.m[1] = txy - twz;         r.m[2] = txz + twy;
  r.m[3] = txy + twz;         r.m[4] = 1.0 - (txx + tzz); r.m[5] = tyz - twx;
  r.m[6] = txz - twy;         r.m[7] = tyz + twx;         r.m[8] = 1.0 - (txx + tyy);
  return r;
}

// Rotation matrix -> quaternion with Eigen's branch structure (trace first, then the largest
// diagonal entry): fixes the sign convention vector2double() produces (estimator.cpp:2344).
GF_HD quat rot2quat(const mat3 &R) {
  double q[4];
  double t = R(0, 0) + R(1, 1) + R(2, 2);
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R(2, 1) - R(1, 2)) * t;
    q[1] = (R(0, 2) - R(2, 0)) * t;
    q[2] = (R(1, 0) - R(0, 1)) * t;
  } else {
    int i = 0;
    if (R(1, 1) > R(0, 0)) i = 1;
    if (R(2, 2) > R(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R(k, j) - R(j, k)) * t;
    q[j] = (R(j, i) + R(i, j)) * t;
    q[k] = (R(k, i) + R(i, k)) * t;
  }
  quat r; r.x = q[0]; r.y = q[1]; r.z = q[2]; r.w = q[3]; return r;
}

// utility.h:23-36
GF_HD quat small_rot(const vec3 &theta) {
  quat d; d.x = 0.5 * theta[0]; d.y = 0.5 * theta[1]; d.z = 0.5 * theta[2]; d.w = 1.0;
  return qnormalize(d);
}

// bottom-right 3x3 of Qleft(q) = w I + [v]x and of Qright(q) = w I - [v]x (utility.h:59-76)
GF_HD mat3 qleft3(const quat &q) { return madd(mscl(q.w, ident3()), hat(qvec(q))); }
GF_HD mat3 qright3(const quat &q) { return msub(mscl(q.w, ident3()), hat(qvec(q))); }
// bottom-right 3x3 of the 4x4 product Qleft(a) * Qright(b):  -va vb^T + (wa I + [va]x)(wb I - [vb]x)
GF_HD mat3 qleft_qright3(const quat &a, const quat &b) {
  mat3 r = mul(qleft3(a), qright3(b));
  const vec3 va = qvec(a), vb = qvec(b);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) -= va[i] * vb[j];
  return r;
}

// utility.h:78-120 — degrees.
GF_HD vec3 rot_to_ypr_deg(const mat3 &R) {
  const double nx = R(0, 0), ny = R(1, 0), nz = R(2, 0);
  const double ox = R(0, 1), oy = R(1, 1);
  const double ax = R(0, 2), ay = R(1, 2);
  const double y = atan2(ny, nx);
  const double p = atan2(-nz, nx * cos(y) + ny * sin(y));
  const double r = atan2(ax * sin(y) - ay * cos(y), -ox * sin(y) + oy * cos(y));
  return mk3(y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0);
}
GF_HD mat3 yaw_rot_deg(double yaw_deg) {   // ypr2R(yaw, 0, 0) = Rz * I * I
  const double y = yaw_deg / 180.0 * M_PI;
  mat3 r = ident3();
  r(0, 0) = cos(y); r(0, 1) = -sin(y); r(1, 0) = sin(y); r(1, 1) = cos(y);
  return r;
}

#define GF_SOPHUS_EPS 1e-10

GF_HD quat so3exp(const vec3 &w) {
  const double th2 = dot3(w, w);
  double im, re;
  if (th2 < GF_SOPHUS_EPS * GF_SOPHUS_EPS) {
    const double th4 = th2 * th2;
    im = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
    re = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
  } else {
    const double th = sqrt(th2);
    im = sin(0.5 * th) / th;
    re = cos(0.5 * th);
  }
  quat q; q.x = im * w[0]; q.y = im * w[1]; q.z = im * w[2]; q.w = re; return q;
}
GF_HD vec3 so3log(const quat &q) {
  const double n2 = q.x * q.x + q.y * q.y + q.z * q.z;
  double f;
  if (n2 < GF_SOPHUS_EPS * GF_SOPHUS_EPS) {
    f = 2.0 / q.w - (2.0 / 3.0) * n2 / (q.w * q.w * q.w);
  } else {
    const double n = sqrt(n2);
    if (fabs(q.w) < GF_SOPHUS_EPS) f = (q.w > 0.0 ? M_PI : -M_PI) / n;
    else f = 2.0 * atan(n / q.w) / n;
  }
  return mk3(f * q.x, f * q.y, f * q.z);
}
GF_HD mat3 jr_so3(const vec3 &phi) {
  const double n2 = dot3(phi, phi);
  const mat3 h = hat(phi), h2 = mul(h, h);
  double c1, c2;
  if (n2 > GF_SOPHUS_EPS) {
    const double n = sqrt(n2);
    c1 = (1.0 - cos(n)) / n2;
    c2 = (n - sin(n)) / (n2 * n);
  } else {
    c1 = 0.5; c2 = 1.0 / 6.0;
  }
  return madd(msub(ident3(), mscl(c1, h)), mscl(c2, h2));
}
GF_HD mat3 jr_inv_so3(const vec3 &phi) {
  const double n2 = dot3(phi, phi);
  const mat3 h = hat(phi), h2 = mul(h, h);
  double c2;
  if (n2 > GF_SOPHUS_EPS) {
    const double n = sqrt(n2);
    if (n < M_PI - 1e-5) c2 = 1.0 / n2 - (1.0 + cos(n)) / (2.0 * n * sin(n));
    else c2 = 1.0 / (M_PI * M_PI);
  } else {
    c2 = 1.0 / 12.0;
  }
  return madd(madd(ident3(), mscl(0.5, h)), mscl(c2, h2));
}

}  // namespace gfd

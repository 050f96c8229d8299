// oracle/gfo_optional.cpp — TEST INFRASTRUCTURE ONLY (CPU oracle). PARITY UNPINNED (no reference goldens; pinned by the
// numerical-derivative and hand-computed checks of tests/test_optional_oracle.py).
// The optional in-window factors, evaluation only:
//   PlaneFactor::Evaluate                          factor/plane_factor.h:25-122 (blocks pose_i 7, ex_wheel 7, plane_R 4, plane_Z 1)
//   PoseAnchorFactor::Evaluate                     factor/pose_anchor_factor.cpp:8-32 (sqrt_info 120, pose_anchor_factor.h:19)
//   OrientationSubsetParameterization::Plus        factor/orientation_subset_parameterization.cpp:27-45
// Jacobians are returned in TANGENT space (what Ceres builds from the factor's global Jacobian and the parameterization's
// [I; 0] ComputeJacobian: the leading 6 / 3 columns of every block).
#include <cmath>
#include <cstring>

#include "gfo_api.h"
#include "gfo_math.h"

using namespace gfo;

extern "C" int32_t gfo_plane_eval(void *, int32_t n, const double *pose, const double *ex_wheel, const double *plane_R, double plane_Z,
                                  const double *noise_inv, double *r, double *J, double *cost) {
  const V3 tio = v3(ex_wheel);
  const M3 Rio = rot(q4(ex_wheel + 3)), Rpw = rot(q4(plane_R));
  const V3 e3 = v3(0, 0, 1);
  const V3 up_p = T(Rpw) * e3;                 // the plane normal seen from the world frame
  double c = 0.0;
  for (int k = 0; k < n; k++) {
    const V3 Pi = v3(pose + 7 * k);
    const M3 Ri = rot(q4(pose + 7 * k + 3));
    const V3 up_b = T(Ri) * up_p, up_o = T(Rio) * up_b;        // ... from the body, from the wheel odometer frame
    const V3 lever = Pi + Ri * tio;                              // odometer origin in the world
    const double res[3] = {noise_inv[0] * up_o.x, noise_inv[1] * up_o.y, noise_inv[2] * (plane_Z + (Rpw * lever).z)};
    for (int i = 0; i < 3; i++) c += 0.5 * res[i] * res[i];
    if (r) std::memcpy(r + 3 * k, res, sizeof res);
    if (!J) continue;
    double *Jk = J + (size_t)k * 48;
    std::memset(Jk, 0, sizeof(double) * 48);
    const M3 A = T(Rio) * skew(up_b);        // d up_o / d theta_i
    const M3 B = skew(up_o);                 // d up_o / d theta_io
    const M3 Cq = T(Rio) * (T(Ri) * skew(up_p));   // d up_o / d theta_pw
    const M3 RpwRi = Rpw * Ri;
    const M3 D = RpwRi * skew(tio), E = Rpw * skew(lever);
    for (int row = 0; row < 2; row++)
      for (int j = 0; j < 3; j++) {
        Jk[row * 16 + 3 + j] = noise_inv[row] * A.m[row][j];
        Jk[row * 16 + 9 + j] = noise_inv[row] * B.m[row][j];
        Jk[row * 16 + 12 + j] = noise_inv[row] * Cq.m[row][j];
      }
    for (int j = 0; j < 3; j++) {
      Jk[2 * 16 + j] = noise_inv[2] * Rpw.m[2][j];
      Jk[2 * 16 + 3 + j] = -noise_inv[2] * D.m[2][j];
      Jk[2 * 16 + 6 + j] = noise_inv[2] * RpwRi.m[2][j];
      Jk[2 * 16 + 12 + j] = -noise_inv[2] * E.m[2][j];
    }
    Jk[2 * 16 + 15] = noise_inv[2];
  }
  if (cost) *cost = c;
  return GFBE_OK;
}

extern "C" int32_t gfo_anchor_eval(void *, int32_t n, const double *pose, const double *anchor, double sqrt_info, double *r, double *J,
                                   double *cost) {
  double c = 0.0;
  for (int k = 0; k < n; k++) {
    const double *x = pose + 7 * k, *a = anchor + 7 * k;
    const Q4 qa_inv = inv(q4(a + 3));                           // Eigen: conjugate / squared norm
    const V3 dv = vec(q4(x + 3) * qa_inv);
    const double res[6] = {sqrt_info * (x[0] - a[0]), sqrt_info * (x[1] - a[1]), sqrt_info * (x[2] - a[2]),
                           sqrt_info * 2.0 * dv.x, sqrt_info * 2.0 * dv.y, sqrt_info * 2.0 * dv.z};
    for (int i = 0; i < 6; i++) c += 0.5 * res[i] * res[i];
    if (r) std::memcpy(r + 6 * k, res, sizeof res);
    if (!J) continue;
    double *Jk = J + (size_t)k * 36;
    std::memset(Jk, 0, sizeof(double) * 36);
    // the reference scales the WHOLE Jacobian by 2 sqrt_info (pose_anchor_factor.cpp:29): the position block is 2 sqrt_info I
    // although the residual is sqrt_info (p - p_a); the rotation block is built from the anchor alone. Reproduced as it is.
    const double s = 2.0 * sqrt_info;
    for (int i = 0; i < 3; i++) Jk[i * 6 + i] = s;
    const M3 Jq = Qright_br(qa_inv);         // [[w z -y], [-z w x], [y -x w]] of the inverted anchor
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Jk[(3 + i) * 6 + 3 + j] = s * Jq.m[i][j];
  }
  if (cost) *cost = c;
  return GFBE_OK;
}

extern "C" void gfo_orientation_subset_plus(const double *q, const double *delta, const uint8_t *constant, double *out) {
  const V3 d = v3(constant[0] ? 0.0 : delta[0], constant[1] ? 0.0 : delta[1], constant[2] ? 0.0 : delta[2]);
  const Q4 p = normalized(q4(q) * deltaQ(d));
  out[0] = p.x; out[1] = p.y; out[2] = p.z; out[3] = p.w;
}

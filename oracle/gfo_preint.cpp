// oracle/gfo_preint.cpp — TEST INFRASTRUCTURE ONLY (CPU oracle).
// Mid-point pre-integration, restated from
//   IntegrationBase::{push_back,propagate,midPointIntegration}       VE/factor/integration_base.h:39-167
//   WheelIntegrationBase::{push_back,propagate,midPointIntegration}  VE/factor/wheel_integration_base.h:41-178
#include "gfo_api.h"
#include "gfo_math.h"
#include <vector>
#include <cstring>

namespace gfo {

static void mat_ABt_add(const double *A, const double *B, double *C, int n, int m, int p) {
  // C(n x p) += A(n x m) * B(p x m)^T
  for (int i = 0; i < n; i++)
    for (int j = 0; j < p; j++) {
      double s = 0;
      for (int k = 0; k < m; k++) s += A[i * m + k] * B[j * m + k];
      C[i * p + j] += s;
    }
}

void preintegrate_imu(int n_samples, const double *samples, const double *first_acc_gyr,
                      const double *lin_ba_bg, const double noise4[4], gfbe_imu_preint *out) {
  V3 acc_0 = v3(first_acc_gyr), gyr_0 = v3(first_acc_gyr + 3);
  V3 ba = v3(lin_ba_bg), bg = v3(lin_ba_bg + 3);
  V3 dp = {0, 0, 0}, dv = {0, 0, 0};
  Q4 dq = {0, 0, 0, 1};
  double sum_dt = 0;
  std::vector<double> jac(225, 0.0), cov(225, 0.0);
  for (int i = 0; i < 15; i++) jac[i * 15 + i] = 1.0;
  double N[18];   // diagonal of the 18x18 noise (integration_base.h:30-36)
  const double an2 = noise4[0] * noise4[0], gn2 = noise4[1] * noise4[1], aw2 = noise4[2] * noise4[2], gw2 = noise4[3] * noise4[3];
  for (int k = 0; k < 3; k++) { N[k] = an2; N[3 + k] = gn2; N[6 + k] = an2; N[9 + k] = gn2; N[12 + k] = aw2; N[15 + k] = gw2; }

  for (int s = 0; s < n_samples; s++) {
    const double dt = samples[7 * s];
    V3 acc_1 = v3(samples + 7 * s + 1), gyr_1 = v3(samples + 7 * s + 4);
    // midPointIntegration :72-80
    V3 un_acc_0 = rotv(dq, acc_0 - ba);
    V3 un_gyr = 0.5 * (gyr_0 + gyr_1) - bg;
    Q4 rq = dq * Q4{un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2, 1.0};
    V3 un_acc_1 = rotv(rq, acc_1 - ba);
    V3 un_acc = 0.5 * (un_acc_0 + un_acc_1);
    V3 rp = dp + dt * dv + (0.5 * dt * dt) * un_acc;
    V3 rv = dv + dt * un_acc;
    // jacobian / covariance :83-134
    V3 w_x = un_gyr, a_0_x = acc_0 - ba, a_1_x = acc_1 - ba;
    M3 R_w_x = skew(w_x), R_a_0_x = skew(a_0_x), R_a_1_x = skew(a_1_x);
    M3 Rd = rot(dq), Rr = rot(rq), I = eye3();
    M3 ImW = I - dt * R_w_x;
    double F[225], V[15 * 18];
    std::memset(F, 0, sizeof F); std::memset(V, 0, sizeof V);
    set_block(F, 15, 0, 0, I);
    set_block(F, 15, 0, 3, (-0.25 * dt * dt) * (Rd * R_a_0_x) + (-0.25 * dt * dt) * (Rr * R_a_1_x * ImW));
    set_block(F, 15, 0, 6, dt * I);
    set_block(F, 15, 0, 9, (-0.25 * dt * dt) * (Rd + Rr));
    set_block(F, 15, 0, 12, (-0.25 * dt * dt * -dt) * (Rr * R_a_1_x));
    set_block(F, 15, 3, 3, ImW);
    set_block(F, 15, 3, 12, (-dt) * I);
    set_block(F, 15, 6, 3, (-0.5 * dt) * (Rd * R_a_0_x) + (-0.5 * dt) * (Rr * R_a_1_x * ImW));
    set_block(F, 15, 6, 6, I);
    set_block(F, 15, 6, 9, (-0.5 * dt) * (Rd + Rr));
    set_block(F, 15, 6, 12, (-0.5 * dt * -dt) * (Rr * R_a_1_x));
    set_block(F, 15, 9, 9, I);
    set_block(F, 15, 12, 12, I);
    M3 v03 = (0.25 * dt * dt * 0.5 * dt) * (-(Rr * R_a_1_x));
    M3 v63 = (0.5 * dt * 0.5 * dt) * (-(Rr * R_a_1_x));
    set_block(V, 18, 0, 0, (0.25 * dt * dt) * Rd);
    set_block(V, 18, 0, 3, v03);
    set_block(V, 18, 0, 6, (0.25 * dt * dt) * Rr);
    set_block(V, 18, 0, 9, v03);
    set_block(V, 18, 3, 3, (0.5 * dt) * I);
    set_block(V, 18, 3, 9, (0.5 * dt) * I);
    set_block(V, 18, 6, 0, (0.5 * dt) * Rd);
    set_block(V, 18, 6, 3, v63);
    set_block(V, 18, 6, 6, (0.5 * dt) * Rr);
    set_block(V, 18, 6, 9, v63);
    set_block(V, 18, 9, 12, dt * I);
    set_block(V, 18, 12, 15, dt * I);
    double tmp[225], ncov[225], VN[15 * 18];
    matmul(F, jac.data(), tmp, 15, 15, 15);                 // jacobian = F * jacobian
    std::memcpy(jac.data(), tmp, sizeof tmp);
    matmul(F, cov.data(), tmp, 15, 15, 15);                 // F * cov
    std::memset(ncov, 0, sizeof ncov);
    mat_ABt_add(tmp, F, ncov, 15, 15, 15);                  // (F cov) F^T
    for (int i = 0; i < 15; i++) for (int k = 0; k < 18; k++) VN[i * 18 + k] = V[i * 18 + k] * N[k];
    mat_ABt_add(VN, V, ncov, 15, 18, 15);                   // + V N V^T
    std::memcpy(cov.data(), ncov, sizeof ncov);
    // propagate :152-166
    dp = rp; dv = rv; dq = normalized(rq);
    sum_dt += dt;
    acc_0 = acc_1; gyr_0 = gyr_1;
  }
  out->sum_dt = sum_dt;
  out->delta_p[0] = dp.x; out->delta_p[1] = dp.y; out->delta_p[2] = dp.z;
  out->delta_q[0] = dq.x; out->delta_q[1] = dq.y; out->delta_q[2] = dq.z; out->delta_q[3] = dq.w;
  out->delta_v[0] = dv.x; out->delta_v[1] = dv.y; out->delta_v[2] = dv.z;
  for (int k = 0; k < 3; k++) { out->linearized_ba[k] = lin_ba_bg[k]; out->linearized_bg[k] = lin_ba_bg[3 + k]; }
  std::memcpy(out->jacobian, jac.data(), sizeof out->jacobian);
  std::memcpy(out->covariance, cov.data(), sizeof out->covariance);
}

void preintegrate_wheel(int n_samples, const double *samples, const double *first_vel_gyr,
                        const double *lin_s_td, const double noise2[2], gfbe_wheel_preint *out) {
  V3 vel_0 = v3(first_vel_gyr), gyr_0 = v3(first_vel_gyr + 3);
  const double lsx = lin_s_td[0], lsy = lin_s_td[1], lsw = lin_s_td[2];
  V3 dp = {0, 0, 0};
  Q4 dq = {0, 0, 0, 1};
  double sum_dt = 0;
  double jac[18], cov[36];
  std::memset(jac, 0, sizeof jac); std::memset(cov, 0, sizeof cov);
  const double vn2 = noise2[0] * noise2[0], gn2 = noise2[1] * noise2[1];
  double N[12];
  for (int k = 0; k < 3; k++) { N[k] = vn2; N[3 + k] = gn2; N[6 + k] = vn2; N[9 + k] = gn2; }
  M3 sv = diag3(lsx, lsy, 1.0);
  V3 vel_1 = vel_0, gyr_1 = gyr_0;
  for (int s = 0; s < n_samples; s++) {
    const double dt = samples[7 * s];
    vel_1 = v3(samples + 7 * s + 1); gyr_1 = v3(samples + 7 * s + 4);
    // midPointIntegration :80-87
    V3 un_vel_0 = rotv(dq, sv * vel_0);
    V3 un_gyr = (0.5 * lsw) * (gyr_0 + gyr_1);
    Q4 ddq = {un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2, 1.0};
    Q4 rq = dq * ddq;
    V3 un_vel_1 = rotv(rq, sv * vel_1);
    V3 rp = dp + dt * (0.5 * (un_vel_0 + un_vel_1));
    // jacobian / covariance :95-142
    V3 vel_0_x = sv * vel_0, vel_1_x = sv * vel_1;
    M3 R_vel_0_x = skew(vel_0_x), R_vel_1_x = skew(vel_1_x);
    M3 Rd = rot(dq), Rr = rot(rq), Rdd = rot(ddq);
    double F[36], V[6 * 12];
    std::memset(F, 0, sizeof F); std::memset(V, 0, sizeof V);
    set_block(F, 6, 0, 0, eye3());
    set_block(F, 6, 0, 3, (-0.5 * dt) * (Rd * R_vel_0_x + Rr * R_vel_1_x * T(Rdd)));
    set_block(F, 6, 3, 3, T(Rdd));
    M3 Jr = right_jac(dt * un_gyr);
    M3 v03 = (-0.25 * dt * dt) * (Rr * R_vel_1_x * Jr);
    set_block(V, 12, 0, 0, (0.5 * dt) * (Rd * sv));
    set_block(V, 12, 0, 3, v03);
    set_block(V, 12, 0, 6, (0.5 * dt) * (Rr * sv));
    set_block(V, 12, 0, 9, v03);
    set_block(V, 12, 3, 3, (0.5 * lsw * dt) * Jr);
    set_block(V, 12, 3, 9, (0.5 * lsw * dt) * Jr);
    // intrinsic-Jacobian recursion :134-139
    M3 I1 = diag3(1, 0, 0), I2 = diag3(0, 1, 0);
    auto getc = [&](int r0, int c) { return v3(jac[(r0) * 3 + c], jac[(r0 + 1) * 3 + c], jac[(r0 + 2) * 3 + c]); };
    auto setc = [&](int r0, int c, V3 v) { jac[r0 * 3 + c] = v.x; jac[(r0 + 1) * 3 + c] = v.y; jac[(r0 + 2) * 3 + c] = v.z; };
    setc(0, 0, getc(0, 0) + (0.5 * dt) * (Rd * (I1 * vel_0) + Rr * (I1 * vel_1)));
    setc(0, 1, getc(0, 1) + (0.5 * dt) * (Rd * (I2 * vel_0) + Rr * (I2 * vel_1)));
    V3 dr_dsw_last = getc(3, 2);
    setc(3, 2, dr_dsw_last + Jr * ((0.5 * dt) * (gyr_0 + gyr_1)));
    setc(0, 2, getc(0, 2) + (0.5 * dt) * (Rd * (skew(dr_dsw_last) * (sv * vel_0)) + Rr * (skew(getc(3, 2)) * (sv * vel_1))));
    double tmp[36], ncov[36], VN[6 * 12];
    matmul(F, cov, tmp, 6, 6, 6);
    std::memset(ncov, 0, sizeof ncov);
    mat_ABt_add(tmp, F, ncov, 6, 6, 6);
    for (int i = 0; i < 6; i++) for (int k = 0; k < 12; k++) VN[i * 12 + k] = V[i * 12 + k] * N[k];
    mat_ABt_add(VN, V, ncov, 6, 12, 6);
    std::memcpy(cov, ncov, sizeof ncov);
    // propagate :165-177
    dp = rp; dq = normalized(rq);
    sum_dt += dt;
    vel_0 = vel_1; gyr_0 = gyr_1;
  }
  std::memset(out, 0, sizeof *out);
  out->sum_dt = sum_dt;
  out->delta_p[0] = dp.x; out->delta_p[1] = dp.y; out->delta_p[2] = dp.z;
  out->delta_q[0] = dq.x; out->delta_q[1] = dq.y; out->delta_q[2] = dq.z; out->delta_q[3] = dq.w;
  out->linearized_sx = lsx; out->linearized_sy = lsy; out->linearized_sw = lsw; out->linearized_td = lin_s_td[3];
  for (int k = 0; k < 3; k++) {
    out->linearized_vel[k] = first_vel_gyr[k]; out->linearized_gyr[k] = first_vel_gyr[3 + k];
    out->vel_1[k] = get(vel_1, k); out->gyr_1[k] = get(gyr_1, k);
  }
  std::memcpy(out->jacobian, jac, sizeof jac);
  std::memcpy(out->covariance, cov, sizeof cov);
}

}  // namespace gfo

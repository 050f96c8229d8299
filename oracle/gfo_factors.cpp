// oracle/gfo_factors.cpp — TEST INFRASTRUCTURE ONLY (CPU oracle). See gfo_factors.h for the
// reference file:line each function restates. Parity with the reference is UNPINNED by reference
// tests (there are none); pinned here by central differences + a numpy re-derivation.
#include "gfo_factors.h"
#include <cstring>
#include <limits>
#include <algorithm>

namespace gfo {

const double *block_ptr(const gfbe_state &st, int id) {
  if (id >= GFBE_BLK_POSE0 && id < GFBE_BLK_POSE0 + GFBE_NFRAMES) return st.para_Pose[id - GFBE_BLK_POSE0];
  if (id >= GFBE_BLK_SB0 && id < GFBE_BLK_SB0 + GFBE_NFRAMES) return st.para_SpeedBias[id - GFBE_BLK_SB0];
  switch (id) {
    case GFBE_BLK_EX_CAM: return st.para_Ex_Pose;
    case GFBE_BLK_EX_WHEEL: return st.para_Ex_Pose_wheel;
    case GFBE_BLK_SX: return &st.para_Ix_wheel[0];
    case GFBE_BLK_SY: return &st.para_Ix_wheel[1];
    case GFBE_BLK_SW: return &st.para_Ix_wheel[2];
    case GFBE_BLK_TD: return &st.para_Td;
    case GFBE_BLK_TD_WHEEL: return &st.para_Td_wheel;
    case GFBE_BLK_PLANE_R: return st.para_plane_R;
    case GFBE_BLK_PLANE_Z: return &st.para_plane_Z;
    case GFBE_BLK_ANC_ECEF: return st.gnss.anc_ecef;
    case GFBE_BLK_YAW_ENU: return &st.gnss.yaw_enu_local;
  }
  if (id >= GFBE_BLK_RCV_DT0 && id < GFBE_BLK_RCV_DDT0) return &st.gnss.rcv_dt[0][0] + (id - GFBE_BLK_RCV_DT0);
  if (id >= GFBE_BLK_RCV_DDT0 && id < GFBE_BLK_COUNT) return &st.gnss.rcv_ddt[id - GFBE_BLK_RCV_DDT0];
  return nullptr;
}
double *block_ptr(gfbe_state &st, int id) { return const_cast<double *>(block_ptr(const_cast<const gfbe_state &>(st), id)); }
int block_global_size(int id) {
  if (id < GFBE_BLK_SB0) return 7;
  if (id < GFBE_BLK_EX_CAM) return 9;
  if (id == GFBE_BLK_EX_CAM || id == GFBE_BLK_EX_WHEEL) return 7;
  if (id == GFBE_BLK_PLANE_R) return 4;   // (local size 4 as well: MarginalizationInfo::localSize only knows size-7 manifolds,
                                          //  marginalization_factor.cpp:140-143; the solve leaves the 4th tangent slot inactive)
  if (id == GFBE_BLK_ANC_ECEF) return 3;
  return 1;
}
int block_local_size(int id) { int g = block_global_size(id); return g == 7 ? 6 : g; }

// ---------------------------------------------------------------------------------------------
// Visual: projectionTwoFrameOneCamFactor.cpp:43-151 (non-UNIT_SPHERE branch; parameters.h:26 has
// UNIT_SPHERE_ERROR commented out).
// ---------------------------------------------------------------------------------------------
void eval_visual(const double *pose_i, const double *pose_j, const double *ex, double inv_dep, double td,
                 const double *pts_i, const double *pts_j, const double *vel_i, const double *vel_j,
                 double td_i, double td_j, double sqrt_info, double *r, double *J) {
  V3 Pi = v3(pose_i), Pj = v3(pose_j), tic = v3(ex);
  Q4 Qi = q4(pose_i + 3), Qj = q4(pose_j + 3), qic = q4(ex + 3);
  V3 vi = {vel_i[0], vel_i[1], 0.0}, vj = {vel_j[0], vel_j[1], 0.0};
  V3 pi_td = v3(pts_i) - (td - td_i) * vi;                 // :60
  V3 pj_td = v3(pts_j) - (td - td_j) * vj;                 // :61
  V3 p_ci = (1.0 / inv_dep) * pi_td;                       // :62
  M3 Ri = rot(Qi), Rj = rot(Qj), ric = rot(qic);
  V3 p_bi = ric * p_ci + tic;                              // :63
  V3 p_w = Ri * p_bi + Pi;                                 // :64
  V3 p_bj = T(Rj) * (p_w - Pj);                            // :65
  V3 p_cj = T(ric) * (p_bj - tic);                         // :66
  double dep = p_cj.z;
  r[0] = sqrt_info * (p_cj.x / dep - pj_td.x);             // :72-76
  r[1] = sqrt_info * (p_cj.y / dep - pj_td.y);
  if (!J) return;
  double red[2][3] = {{sqrt_info / dep, 0.0, -sqrt_info * p_cj.x / (dep * dep)},   // :97-100
                      {0.0, sqrt_info / dep, -sqrt_info * p_cj.y / (dep * dep)}};
  M3 A = T(ric) * T(Rj);
  M3 ji_p = A;                                             // :106
  M3 ji_r = -(A * Ri * skew(p_bi));                        // :107
  M3 jj_p = -A;                                            // :118
  M3 jj_r = T(ric) * skew(p_bj);                           // :119
  M3 je_p = T(ric) * (T(Rj) * Ri - eye3());                // :128
  M3 tmp_r = A * Ri * ric;                                 // :129
  M3 je_r = -(tmp_r * skew(p_ci)) + skew(tmp_r * p_ci) +
            skew(T(ric) * (T(Rj) * (Ri * tic + Pi - Pj) - tic));   // :130-131
  V3 jl = (-1.0 / (inv_dep * inv_dep)) * (tmp_r * pi_td);  // :139
  V3 jt = (-1.0 / inv_dep) * (tmp_r * vi);                 // :144
  const M3 *blk[6] = {&ji_p, &ji_r, &jj_p, &jj_r, &je_p, &je_r};
  for (int row = 0; row < 2; row++) {
    double *Jr = J + row * 20;
    for (int b = 0; b < 6; b++)
      for (int c = 0; c < 3; c++)
        Jr[b * 3 + c] = red[row][0] * blk[b]->m[0][c] + red[row][1] * blk[b]->m[1][c] + red[row][2] * blk[b]->m[2][c];
    Jr[18] = red[row][0] * jl.x + red[row][1] * jl.y + red[row][2] * jl.z;
    Jr[19] = red[row][0] * jt.x + red[row][1] * jt.y + red[row][2] * jt.z + sqrt_info * (row == 0 ? vj.x : vj.y);  // :145
  }
}

// ---------------------------------------------------------------------------------------------
// IMU: imu_factor.h:28-191, integration_base.h:169-195.
// ---------------------------------------------------------------------------------------------
namespace {
struct M4 { double m[4][4]; };
// Utility::Qleft / Qright, utility.h:59-76, in (w,x,y,z) order.
M4 Qleft4(Q4 q) {
  M4 r; M3 br = q.w * eye3() + skew(vec(q));
  r.m[0][0] = q.w; r.m[0][1] = -q.x; r.m[0][2] = -q.y; r.m[0][3] = -q.z;
  r.m[1][0] = q.x; r.m[2][0] = q.y; r.m[3][0] = q.z;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[1 + i][1 + j] = br.m[i][j];
  return r;
}
M4 Qright4(Q4 p) {
  M4 r; M3 br = p.w * eye3() - skew(vec(p));
  r.m[0][0] = p.w; r.m[0][1] = -p.x; r.m[0][2] = -p.y; r.m[0][3] = -p.z;
  r.m[1][0] = p.x; r.m[2][0] = p.y; r.m[3][0] = p.z;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[1 + i][1 + j] = br.m[i][j];
  return r;
}
M3 br3(const M4 &a) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[1 + i][1 + j]; return r; }
M4 mul4(const M4 &a, const M4 &b) {
  M4 r;
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += a.m[i][k] * b.m[k][j]; r.m[i][j] = s; }
  return r;
}
}  // namespace

void eval_imu(const gfbe_imu_preint &pre, const double *sqrt_info, double g_norm,
              const double *pose_i, const double *sb_i, const double *pose_j, const double *sb_j,
              double *r, double *J) {
  V3 Pi = v3(pose_i), Pj = v3(pose_j);
  Q4 Qi = q4(pose_i + 3), Qj = q4(pose_j + 3);
  V3 Vi = v3(sb_i), Bai = v3(sb_i + 3), Bgi = v3(sb_i + 6);
  V3 Vj = v3(sb_j), Baj = v3(sb_j + 3), Bgj = v3(sb_j + 6);
  const double dt = pre.sum_dt;
  V3 G = {0.0, 0.0, g_norm};
  M3 dp_dba = get_block(pre.jacobian, 15, 0, 9), dp_dbg = get_block(pre.jacobian, 15, 0, 12);
  M3 dq_dbg = get_block(pre.jacobian, 15, 3, 12);
  M3 dv_dba = get_block(pre.jacobian, 15, 6, 9), dv_dbg = get_block(pre.jacobian, 15, 6, 12);
  V3 dba = Bai - v3(pre.linearized_ba), dbg = Bgi - v3(pre.linearized_bg);
  Q4 dq = q4(pre.delta_q);
  Q4 cq = dq * deltaQ(dq_dbg * dbg);                                   // integration_base.h:188
  V3 cv = v3(pre.delta_v) + dv_dba * dba + dv_dbg * dbg;
  V3 cp = v3(pre.delta_p) + dp_dba * dba + dp_dbg * dbg;
  Q4 Qi_inv = inv(Qi);
  M3 RiT = rot(Qi_inv);
  V3 a_p = RiT * (0.5 * dt * dt * G + Pj - Pi - dt * Vi);
  V3 a_v = RiT * (dt * G + Vj - Vi);
  double raw[15];
  V3 rp = a_p - cp;                                                    // :191
  V3 rq = 2.0 * vec(inv(cq) * (Qi_inv * Qj));                          // :192
  V3 rv = a_v - cv;                                                    // :193
  V3 rba = Baj - Bai, rbg = Bgj - Bgi;
  const V3 parts[5] = {rp, rq, rv, rba, rbg};
  for (int b = 0; b < 5; b++) { raw[3 * b] = parts[b].x; raw[3 * b + 1] = parts[b].y; raw[3 * b + 2] = parts[b].z; }
  matmul(sqrt_info, raw, r, 15, 15, 1);                                // imu_factor.h:75
  if (!J) return;
  double Jraw[15 * 30];
  std::memset(Jraw, 0, sizeof Jraw);
  // pose_i  (imu_factor.h:103-118), tangent columns 0..5
  set_block(Jraw, 30, 0, 0, -RiT);
  set_block(Jraw, 30, 0, 3, skew(a_p));
  set_block(Jraw, 30, 3, 3, -br3(mul4(Qleft4(inv(Qj) * Qi), Qright4(cq))));
  set_block(Jraw, 30, 6, 3, skew(a_v));
  // sb_i (imu_factor.h:129-155), columns 6..14
  set_block(Jraw, 30, 0, 6, (-dt) * RiT);
  set_block(Jraw, 30, 0, 9, -dp_dba);
  set_block(Jraw, 30, 0, 12, -dp_dbg);
  set_block(Jraw, 30, 3, 12, -(br3(Qleft4(inv(Qj) * Qi * dq)) * dq_dbg));   // uncorrected delta_q, :137
  set_block(Jraw, 30, 6, 6, -RiT);
  set_block(Jraw, 30, 6, 9, -dv_dba);
  set_block(Jraw, 30, 6, 12, -dv_dbg);
  set_block(Jraw, 30, 9, 9, -eye3());
  set_block(Jraw, 30, 12, 12, -eye3());
  // pose_j (imu_factor.h:157-173), columns 15..20
  set_block(Jraw, 30, 0, 15, RiT);
  set_block(Jraw, 30, 3, 18, br3(Qleft4(inv(cq) * Qi_inv * Qj)));
  // sb_j (imu_factor.h:174-187), columns 21..29
  set_block(Jraw, 30, 6, 21, RiT);
  set_block(Jraw, 30, 9, 24, eye3());
  set_block(Jraw, 30, 12, 27, eye3());
  matmul(sqrt_info, Jraw, J, 15, 15, 30);
}

// ---------------------------------------------------------------------------------------------
// Wheel: wheel_factor.h:28-247, wheel_integration_base.h:180-219.
// ---------------------------------------------------------------------------------------------
void eval_wheel(const gfbe_wheel_preint &pre, const double *sqrt_info,
                const double *pose_i, const double *pose_j, const double *ex_wheel,
                double sx, double sy, double sw, double td, double *r, double *J) {
  V3 Pi = v3(pose_i), Pj = v3(pose_j), tio = v3(ex_wheel);
  Q4 Qi = q4(pose_i + 3), Qj = q4(pose_j + 3), qio = q4(ex_wheel + 3);
  auto col = [&](int r0, int c) { return v3(pre.jacobian[(r0 + 0) * 3 + c], pre.jacobian[(r0 + 1) * 3 + c], pre.jacobian[(r0 + 2) * 3 + c]); };
  V3 dp_dsx = col(0, 0), dp_dsy = col(0, 1), dp_dsw = col(0, 2), dq_dsw = col(3, 2);
  const double dsx = sx - pre.linearized_sx, dsy = sy - pre.linearized_sy, dsw = sw - pre.linearized_sw;
  M3 sv = diag3(sx, sy, 1.0);
  M3 Ri = rot(Qi), Rj = rot(Qj), rio = rot(qio);
  V3 lin_vel = v3(pre.linearized_vel), lin_gyr = v3(pre.linearized_gyr), vel_1 = v3(pre.vel_1), gyr_1 = v3(pre.gyr_1);
  // wheel_integration_base.h:201-206
  V3 cp = v3(pre.delta_p) + dsx * dp_dsx + dsy * dp_dsy + dsw * dp_dsw;
  Q4 cq = normalized(q4(pre.delta_q)) * so3_exp(dsw * dq_dsw);
  const double dtd = td - pre.linearized_td;
  Q4 e_fw = so3_exp((sw * dtd) * lin_gyr);
  Q4 q_time = e_fw * cq * so3_exp((-sw * dtd) * gyr_1);
  M3 Rcq = rot(cq);
  V3 p_time = rot(e_fw) * (sv * (dtd * lin_vel) + cp - Rcq * (sv * (dtd * vel_1)));
  M3 RiRioT = T(Ri * rio);
  V3 world_d = Rj * tio + Pj - Ri * tio - Pi;
  V3 rp = RiRioT * world_d - p_time;                                             // :211
  V3 rq = so3_log(normalized(inv(q_time) * inv(Qi * qio) * Qj * qio));           // :212
  double raw[6] = {rp.x, rp.y, rp.z, rq.x, rq.y, rq.z};
  matmul(sqrt_info, raw, r, 6, 6, 1);                                            // wheel_factor.h:88
  if (!J) return;
  double Jraw[6 * 22];
  std::memset(Jraw, 0, sizeof Jraw);
  M3 Jr_inv = right_jac_inv(rq);                                                 // :106-108
  V3 drdsw = (sw - pre.linearized_sw) * dq_dsw;
  M3 Jr_drdsw = right_jac(drdsw);                                                // :110-112
  // pose_i, :117-143
  set_block(Jraw, 22, 0, 0, -RiRioT);
  set_block(Jraw, 22, 0, 3, RiRioT * (Ri * skew(tio)) + T(rio) * skew(T(Ri) * world_d));
  set_block(Jraw, 22, 3, 3, -(Jr_inv * rot(inv(Qj * qio) * Qi)));
  // pose_j, :145-164
  set_block(Jraw, 22, 0, 6, RiRioT);
  set_block(Jraw, 22, 0, 9, -(rot(inv(Qi * qio) * Qj) * skew(tio)));
  set_block(Jraw, 22, 3, 9, Jr_inv * rot(inv(qio)));
  // extrinsic, :165-180
  set_block(Jraw, 22, 0, 12, RiRioT * (Rj - Ri));
  set_block(Jraw, 22, 0, 15, skew(RiRioT * world_d));
  set_block(Jraw, 22, 3, 15, Jr_inv * (eye3() - rot(inv(Qj * qio) * Qi * qio)));
  // intrinsics and td, :181-243
  V3 fw = (sw * dtd) * lin_gyr, fv = sv * (dtd * lin_vel), bv = sv * (dtd * vel_1), bw = (sw * dtd) * gyr_1;
  M3 Jrtd = right_jac(fw), Jr_mtd = right_jac(-fw);
  M3 I1 = diag3(1, 0, 0), I2 = diag3(0, 1, 0);
  M3 Efv = rot(so3_exp(fv)), Efw = rot(so3_exp(fw));
  V3 c_sx = -(Efv * (I1 * (dtd * lin_vel) + dp_dsx - Rcq * (I1 * (dtd * vel_1))));       // :199 (exp of fv: as written)
  V3 c_sy = -(Efv * (I2 * (dtd * lin_vel) + dp_dsy - Rcq * (I2 * (dtd * vel_1))));       // :211
  V3 inner = fv + cp - Rcq * bv;
  V3 c_sw_p = -(Efw * (dp_dsw - Rcq * (skew(Jr_drdsw * dq_dsw) * (sv * (dtd * vel_1))) +
                       skew(Jrtd * (dtd * lin_gyr)) * inner));                           // :223
  M3 Emr = rot(so3_exp(-rq)), Ebw = rot(so3_exp(bw)), RcqT = rot(inv(cq));
  V3 c_sw_r = -(Jr_inv * (Emr * (Ebw * (RcqT * (Jrtd * (dtd * lin_gyr)) + Jr_drdsw * dq_dsw))));   // :225
  V3 c_td_p = -(Efw * (sv * lin_vel - Rcq * (sv * vel_1) + skew(Jrtd * (sw * lin_gyr)) * inner));  // :236
  V3 c_td_r = -(Jr_inv * (Emr * (Ebw * (RcqT * (Jrtd * (sw * lin_gyr))) - Jr_mtd * (sw * gyr_1)))); // :237
  auto put = [&](int c, V3 p, V3 q, bool has_q) {
    Jraw[0 * 22 + c] = p.x; Jraw[1 * 22 + c] = p.y; Jraw[2 * 22 + c] = p.z;
    if (has_q) { Jraw[3 * 22 + c] = q.x; Jraw[4 * 22 + c] = q.y; Jraw[5 * 22 + c] = q.z; }
  };
  put(18, c_sx, v3(0, 0, 0), false);
  put(19, c_sy, v3(0, 0, 0), false);
  put(20, c_sw_p, c_sw_r, true);
  put(21, c_td_p, c_td_r, true);
  matmul(sqrt_info, Jraw, J, 6, 6, 22);
}

// ---------------------------------------------------------------------------------------------
// Prior: marginalization_factor.cpp:344-392.
// ---------------------------------------------------------------------------------------------
void prior_dx(const gfbe_prior &pr, const gfbe_state &st, double *dx) {
  int xoff = 0;
  for (int b = 0; b < pr.n_blocks; b++) {
    const int size = pr.block_size[b], idx = pr.block_idx[b];
    const double *x = block_ptr(st, pr.block_id[b]);
    const double *x0 = pr.x0 + xoff;
    if (size != 7) {
      for (int k = 0; k < size; k++) dx[idx + k] = x[k] - x0[k];
    } else {
      for (int k = 0; k < 3; k++) dx[idx + k] = x[k] - x0[k];
      Q4 d = inv(q4(x0 + 3)) * q4(x + 3);                 // :369
      double sgn = (d.w >= 0) ? 2.0 : -2.0;               // :370-373 (NaN w also flips)
      if (!(d.w >= 0)) sgn = -2.0;
      dx[idx + 3] = sgn * d.x; dx[idx + 4] = sgn * d.y; dx[idx + 5] = sgn * d.z;
    }
    xoff += size;
  }
}
void eval_prior(const gfbe_prior &pr, const gfbe_state &st, double *r) {
  double dx[GFBE_DENSE_DIM];
  prior_dx(pr, st, dx);
  const int n = pr.n;
  for (int i = 0; i < n; i++) {
    double s = pr.r0[i];
    const double *row = pr.J0 + (size_t)i * n;
    for (int k = 0; k < n; k++) s += row[k] * dx[k];
    r[i] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// Loss: ceres::HuberLoss::Evaluate + Corrector as re-implemented at marginalization_factor.cpp:46-77.
// ---------------------------------------------------------------------------------------------
void huber(double s, double delta, double rho[3]) {
  const double b = delta * delta;
  if (s > b) {
    const double rt = std::sqrt(s);
    rho[0] = 2.0 * delta * rt - b;
    rho[1] = std::max(std::numeric_limits<double>::min(), delta / rt);
    rho[2] = -rho[1] / (2.0 * s);
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}
double robustify(double *r, double *J, int nr, int nc, double delta) {
  double s = 0;
  for (int i = 0; i < nr; i++) s += r[i] * r[i];
  double rho[3];
  huber(s, delta, rho);
  const double sqrt_rho1 = std::sqrt(rho[1]);
  double residual_scaling, alpha_sq_norm;
  if (s == 0.0 || rho[2] <= 0.0) {
    residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0;
  } else {
    const double D = 1.0 + 2.0 * s * rho[2] / rho[1];
    const double alpha = 1.0 - std::sqrt(D);
    residual_scaling = sqrt_rho1 / (1 - alpha);
    alpha_sq_norm = alpha / s;
  }
  if (J) {
    for (int c = 0; c < nc; c++) {
      double rtj = 0;
      for (int i = 0; i < nr; i++) rtj += r[i] * J[i * nc + c];
      for (int i = 0; i < nr; i++) J[i * nc + c] = sqrt_rho1 * (J[i * nc + c] - alpha_sq_norm * r[i] * rtj);
    }
  }
  for (int i = 0; i < nr; i++) r[i] *= residual_scaling;
  return 0.5 * rho[0];
}

}  // namespace gfo

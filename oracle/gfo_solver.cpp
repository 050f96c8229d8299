// oracle/gfo_solver.cpp — TEST INFRASTRUCTURE ONLY (CPU oracle).
//
// Restates the numerical core of Estimator::optimization():
//   problem construction            VE/estimator/estimator.cpp:2956-3358
//   ceres::Solve (DENSE_SCHUR + traditional DOGLEG + HuberLoss, Jacobi scaling, 1 thread)
//                                   VE/estimator/estimator.cpp:3364-3379
//     -- Ceres 1.14 is NOT in /root/reference (README.md:66-72). What is restated here is its
//        published algorithm: TrustRegionMinimizer (iteration loop, step acceptance, tolerances),
//        DoglegStrategy (diagonal D = sqrt(clamp(diag(J'J),1e-6,1e32)), Cauchy point, mu-regularised
//        Gauss-Newton solve with mu in [1e-8,1], traditional dogleg interpolation, radius update),
//        SchurEliminator on the 1-D inverse-depth blocks, Corrector. PARITY UNPINNED.
//   double2vector re-anchoring      VE/estimator/estimator.cpp:2501-2555
//   marginalisation                 VE/estimator/estimator.cpp:3394-3693,
//                                   VE/factor/marginalization_factor.cpp:119-330
// (VE = Ground-Fusion++/vins_estimator/src)
#include "gfo_api.h"
#include "gfo_factors.h"
#include <vector>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <cmath>
#include <algorithm>
#include <limits>

// the optional in-window factors (gfo_optional.cpp)
extern "C" int32_t gfo_plane_eval(void *, int32_t n, const double *pose, const double *ex_wheel, const double *plane_R, double plane_Z,
                                  const double *noise_inv, double *r, double *J, double *cost);
extern "C" int32_t gfo_anchor_eval(void *, int32_t n, const double *pose, const double *anchor, double sqrt_info, double *r, double *J,
                                   double *cost);
extern "C" void gfo_orientation_subset_plus(const double *q, const double *delta, const uint8_t *constant, double *out);

namespace gfo {

// ---- tangent layout of the dense block (oracle's own; the product documents its layout in DESIGN.md)
enum { NV = 73, ND = GFBE_DENSE_DIM };
static inline int T_POSE(int k) { return 6 * k; }
enum { T_EX = 66, T_TD = 72 };
static inline int T_SB(int k) { return 73 + 9 * k; }
enum { T_EXW = 172, T_SX = 178, T_SY = 179, T_SW = 180, T_TDW = 181, T_PLR = 182, T_PLZ = 186, T_ANC = 187, T_YAW = 190 };
static inline int T_DT(int i, int k) { return 191 + 4 * i + k; }    // para_rcv_dt + 4 i + k
static inline int T_DDT(int i) { return 235 + i; }                  // para_rcv_ddt + i

static int tan_off(int id) {
  if (id < GFBE_BLK_SB0) return T_POSE(id);
  if (id < GFBE_BLK_EX_CAM) return T_SB(id - GFBE_BLK_SB0);
  switch (id) {
    case GFBE_BLK_EX_CAM: return T_EX;
    case GFBE_BLK_EX_WHEEL: return T_EXW;
    case GFBE_BLK_SX: return T_SX;
    case GFBE_BLK_SY: return T_SY;
    case GFBE_BLK_SW: return T_SW;
    case GFBE_BLK_TD: return T_TD;
    case GFBE_BLK_TD_WHEEL: return T_TDW;
    case GFBE_BLK_PLANE_R: return T_PLR;
    case GFBE_BLK_PLANE_Z: return T_PLZ;
    case GFBE_BLK_ANC_ECEF: return T_ANC;
    case GFBE_BLK_YAW_ENU: return T_YAW;
  }
  if (id >= GFBE_BLK_RCV_DT0 && id < GFBE_BLK_RCV_DDT0) return T_DT(0, id - GFBE_BLK_RCV_DT0);
  if (id >= GFBE_BLK_RCV_DDT0 && id < GFBE_BLK_COUNT) return T_DDT(id - GFBE_BLK_RCV_DDT0);
  return -1;
}

struct Problem {
  const gfbe_window *win;
  gfbe_options opt;
  int L;
  std::vector<double> imu_sqrt;    // n_imu * 225
  std::vector<double> wheel_sqrt;  // n_wheel * 36
  std::vector<double> Hprior;      // n*n  J0'J0 (constant over the solve)
  bool has_prior;
  bool gnss_factors;               // gnss_ready && !lowspeed (estimator.cpp:3239): the GNSS residual blocks are in the problem
  bool blk_used[GFBE_BLK_COUNT];
  bool blk_free[GFBE_BLK_COUNT];   // in the reduced program (not constant, touched by a factor)
  bool act[ND];                    // per tangent dim
  std::vector<uint8_t> lm_free;    // landmark in the reduced program
  bool ok;
};

struct Lin {
  std::vector<double> H, g, Hll, gl, Hpl;
  double cost;
  void init(int L) { H.assign(ND * ND, 0.0); g.assign(ND, 0.0); Hll.assign(L, 0.0); gl.assign(L, 0.0); Hpl.assign((size_t)L * NV, 0.0); cost = 0; }
};

static bool setup(Problem &P, const gfbe_window *win, const gfbe_options *opt) {
  P.win = win; P.opt = *opt; P.L = win->n_feature; P.ok = true;
  P.imu_sqrt.resize((size_t)win->n_imu * 225);
  for (int k = 0; k < win->n_imu; k++)                      // imu_factor.h:73 (once per window: SURVEY App. A.5)
    if (!sqrt_info_from_cov(win->imu[k].covariance, &P.imu_sqrt[(size_t)k * 225], 15)) P.ok = false;
  P.wheel_sqrt.resize((size_t)win->n_wheel * 36);
  for (int k = 0; k < win->n_wheel; k++)                    // wheel_factor.h:85
    if (!sqrt_info_from_cov(win->wheel[k].covariance, &P.wheel_sqrt[(size_t)k * 36], 6)) P.ok = false;
  P.has_prior = win->prior && win->prior->valid && win->prior->n > 0;
  if (P.has_prior) {
    const int n = win->prior->n;
    P.Hprior.assign((size_t)n * n, 0.0);
    const double *J0 = win->prior->J0;
    for (int r = 0; r < n; r++)
      for (int i = 0; i < n; i++) {
        const double a = J0[(size_t)r * n + i];
        if (a == 0.0) continue;
        for (int j = 0; j < n; j++) P.Hprior[(size_t)i * n + j] += a * J0[(size_t)r * n + j];
      }
  }
  // Which blocks does any residual touch? (Ceres drops parameter blocks without residuals.)
  for (int b = 0; b < GFBE_BLK_COUNT; b++) P.blk_used[b] = false;
  for (int k = 0; k < win->n_imu; k++) {
    int i = win->imu_frame[k];
    P.blk_used[GFBE_BLK_POSE0 + i] = P.blk_used[GFBE_BLK_SB0 + i] = true;
    P.blk_used[GFBE_BLK_POSE0 + i + 1] = P.blk_used[GFBE_BLK_SB0 + i + 1] = true;
  }
  for (int k = 0; k < win->n_wheel; k++) {
    int i = win->wheel_frame[k];
    P.blk_used[GFBE_BLK_POSE0 + i] = P.blk_used[GFBE_BLK_POSE0 + i + 1] = true;
    P.blk_used[GFBE_BLK_EX_WHEEL] = P.blk_used[GFBE_BLK_SX] = P.blk_used[GFBE_BLK_SY] = P.blk_used[GFBE_BLK_SW] = P.blk_used[GFBE_BLK_TD_WHEEL] = true;
  }
  P.lm_free.assign(P.L, 0);
  for (int k = 0; k < win->vis.n_factor; k++) {
    P.blk_used[GFBE_BLK_POSE0 + win->vis.imu_i[k]] = P.blk_used[GFBE_BLK_POSE0 + win->vis.imu_j[k]] = true;
    P.blk_used[GFBE_BLK_EX_CAM] = P.blk_used[GFBE_BLK_TD] = true;
    int l = win->vis.feature_index[k];
    if (l < 0 || l >= P.L) return false;
    P.lm_free[l] = win->feature_const ? !win->feature_const[l] : 1;
  }
  if (P.has_prior)
    for (int b = 0; b < win->prior->n_blocks; b++) P.blk_used[win->prior->block_id[b]] = true;
  if (win->lio.n > 0) {
    if (win->lio.frame < 0 || win->lio.frame > win->frame_count) return false;
    P.blk_used[GFBE_BLK_POSE0 + win->lio.frame] = true;
  }
  if (win->use_plane) {       // one PlaneFactor per pose i < frame_count (estimator.cpp:3214-3220)
    for (int i = 0; i < win->frame_count; i++) P.blk_used[GFBE_BLK_POSE0 + i] = true;
    if (win->frame_count > 0) P.blk_used[GFBE_BLK_EX_WHEEL] = P.blk_used[GFBE_BLK_PLANE_R] = P.blk_used[GFBE_BLK_PLANE_Z] = true;
  }
  if (win->use_anchor) P.blk_used[GFBE_BLK_POSE0] = true;
  // GNSS (estimator.cpp:2965-3002, 3239-3291): lowspeed from the window's velocities, then the blocks the factors touch
  P.gnss_factors = false;
  if (win->gnss_ready) {
    double ax = 0.0, ay = 0.0;
    for (int i = 0; i <= GFBE_WINDOW_SIZE; i++) { ax += std::fabs(win->state.para_SpeedBias[i][0]); ay += std::fabs(win->state.para_SpeedBias[i][1]); }
    ax /= GFBE_WINDOW_SIZE + 1; ay /= GFBE_WINDOW_SIZE + 1;
    P.gnss_factors = !(std::sqrt(ax * ax + ay * ay) < 0.3);
    if (win->n_gnss < 0 || (win->n_gnss > 0 && !win->gnss_obs)) return false;
    for (int k = 0; k < win->n_gnss; k++) {
      const gfbe_gnss_obs &ob = win->gnss_obs[k];
      if (ob.frame < 0 || ob.frame > GFBE_WINDOW_SIZE || ob.lower_idx < 0 || ob.lower_idx >= GFBE_WINDOW_SIZE || ob.sys_idx < 0 || ob.sys_idx > 3 ||
          !(ob.pr_uura > 0.0) || !(ob.dp_uura > 0.0)) return false;
    }
  }
  if (P.gnss_factors) {
    for (int k = 0; k < win->n_gnss; k++) {
      const gfbe_gnss_obs &ob = win->gnss_obs[k];
      P.blk_used[GFBE_BLK_POSE0 + ob.lower_idx] = P.blk_used[GFBE_BLK_SB0 + ob.lower_idx] = true;
      P.blk_used[GFBE_BLK_POSE0 + ob.lower_idx + 1] = P.blk_used[GFBE_BLK_SB0 + ob.lower_idx + 1] = true;
      P.blk_used[GFBE_BLK_RCV_DT0 + 4 * ob.frame + ob.sys_idx] = P.blk_used[GFBE_BLK_RCV_DDT0 + ob.frame] = true;
      P.blk_used[GFBE_BLK_YAW_ENU] = P.blk_used[GFBE_BLK_ANC_ECEF] = true;
    }
    for (int b = GFBE_BLK_RCV_DT0; b < GFBE_BLK_COUNT; b++) P.blk_used[b] = true;     // DtDdtFactor / DdtSmoothFactor chains
  }
  for (int b = 0; b < GFBE_BLK_COUNT; b++) {
    bool c;
    if (b < GFBE_BLK_SB0) c = win->pose_const[b] || b > win->frame_count;
    else if (b < GFBE_BLK_EX_CAM) c = win->sb_const[b - GFBE_BLK_SB0] || (b - GFBE_BLK_SB0) > win->frame_count;
    else if (b == GFBE_BLK_EX_CAM) c = win->ex_cam_const;
    else if (b == GFBE_BLK_EX_WHEEL) c = win->ex_wheel_const;
    else if (b == GFBE_BLK_TD) c = win->td_const;
    else if (b == GFBE_BLK_TD_WHEEL) c = win->td_wheel_const;
    else if (b == GFBE_BLK_PLANE_R || b == GFBE_BLK_PLANE_Z) c = win->plane_const;
    else if (b == GFBE_BLK_YAW_ENU) c = win->gnss_ready != 0;          // estimator.cpp:2991
    else if (b == GFBE_BLK_ANC_ECEF || b >= GFBE_BLK_RCV_DT0) c = false;
    else c = win->ix_wheel_const;
    P.blk_free[b] = P.blk_used[b] && !c;
  }
  for (int d = 0; d < ND; d++) P.act[d] = false;
  for (int b = 0; b < GFBE_BLK_COUNT; b++)
    if (P.blk_free[b]) for (int k = 0; k < block_local_size(b); k++) P.act[tan_off(b) + k] = true;
  P.act[T_PLR + 3] = false;   // the quaternion's 4th slot exists in the prior only (OrientationSubsetParameterization: 3 tangent dims)
  return P.ok;
}

// Accumulate J'J, J'r of one residual block whose columns map to tangent offsets.
static void accum(Lin &lin, const double *r, const double *J, int nr, int nc, const int *colmap) {
  for (int a = 0; a < nc; a++) {
    const int ga = colmap[a];
    if (ga < 0) continue;
    double gr = 0;
    for (int i = 0; i < nr; i++) gr += J[i * nc + a] * r[i];
    lin.g[ga] += gr;
    for (int b = 0; b < nc; b++) {
      const int gb = colmap[b];
      if (gb < 0) continue;
      double s = 0;
      for (int i = 0; i < nr; i++) s += J[i * nc + a] * J[i * nc + b];
      lin.H[(size_t)ga * ND + gb] += s;
    }
  }
}

// The GNSS residual blocks (estimator.cpp:3239-3291; only_frame0: the marginalisation set of :3462-3496). `emit` receives every
// block as (r, J, rows, cols, tangent dim of each column).
template <class Emit>
static void gnss_blocks(const gfbe_window &w, const gfbe_state &st, bool with_jac, bool only_frame0, Emit emit) {
  const gfbe_gnss_state &g = st.gnss;
  for (int k = 0; k < w.n_gnss; k++) {
    const gfbe_gnss_obs &ob = w.gnss_obs[k];
    if (only_frame0 && ob.frame != 0) continue;
    const int lo = only_frame0 ? 0 : ob.lower_idx;
    double r[2], J[36];
    eval_gnss_psr_dopp(ob, w.gnss_iono, st.para_Pose[lo], st.para_SpeedBias[lo], st.para_Pose[lo + 1], st.para_SpeedBias[lo + 1],
                       g.rcv_dt[ob.frame][ob.sys_idx], g.rcv_ddt[ob.frame], g.yaw_enu_local, g.anc_ecef, r, with_jac ? J : nullptr);
    int map[18];
    for (int q = 0; q < 3; q++) { map[q] = T_POSE(lo) + q; map[3 + q] = T_SB(lo) + q; map[6 + q] = T_POSE(lo + 1) + q; map[9 + q] = T_SB(lo + 1) + q; map[15 + q] = T_ANC + q; }
    map[12] = T_DT(ob.frame, ob.sys_idx); map[13] = T_DDT(ob.frame); map[14] = T_YAW;
    emit(r, J, 2, 18, map);
  }
  const int ni = only_frame0 ? 1 : GFBE_WINDOW_SIZE;
  for (int k = 0; k < 4; k++)                      // DtDdtFactor (gnss_dt_ddt_factor.cpp:3-34), dt_info_coeff = 50
    for (int i = 0; i < ni; i++) {
      const double dt = w.gnss_frame_dt[i];
      double r[1] = {(g.rcv_dt[i + 1][k] - g.rcv_dt[i][k] - 0.5 * (g.rcv_ddt[i] + g.rcv_ddt[i + 1]) * dt) * 50.0};
      double J[4] = {-50.0, 50.0, -0.5 * dt * 50.0, -0.5 * dt * 50.0};
      int map[4] = {T_DT(i, k), T_DT(i + 1, k), T_DDT(i), T_DDT(i + 1)};
      emit(r, J, 1, 4, map);
    }
  for (int i = 0; i < ni; i++) {                   // DdtSmoothFactor (gnss_ddt_smooth_factor.cpp:3-22)
    double r[1] = {(g.rcv_ddt[i] - g.rcv_ddt[i + 1]) * w.gnss_ddt_weight};
    double J[2] = {w.gnss_ddt_weight, -w.gnss_ddt_weight};
    int map[2] = {T_DDT(i), T_DDT(i + 1)};
    emit(r, J, 1, 2, map);
  }
}

// Evaluate everything at (st, lam). with_jac=false: cost only.
static double evaluate(const Problem &P, const gfbe_state &st, const double *lam, Lin *lin) {
  const gfbe_window &w = *P.win;
  double cost = 0;
  if (lin) lin->init(P.L);
  // --- prior (estimator.cpp:3163-3169), no loss
  if (P.has_prior) {
    const gfbe_prior &pr = *w.prior;
    const int n = pr.n;
    std::vector<double> r(n);
    eval_prior(pr, st, r.data());
    for (int i = 0; i < n; i++) cost += 0.5 * r[i] * r[i];
    if (lin) {
      std::vector<int> map(n, -1);
      for (int b = 0; b < pr.n_blocks; b++)
        for (int k = 0; k < block_local_size(pr.block_id[b]); k++) map[pr.block_idx[b] + k] = tan_off(pr.block_id[b]) + k;
      for (int i = 0; i < n; i++) {
        double s = 0;
        for (int k = 0; k < n; k++) s += pr.J0[(size_t)k * n + i] * r[k];
        lin->g[map[i]] += s;
        for (int j = 0; j < n; j++) lin->H[(size_t)map[i] * ND + map[j]] += P.Hprior[(size_t)i * n + j];
      }
    }
  }
  // --- IMU (estimator.cpp:3170-3180), no loss
  for (int k = 0; k < w.n_imu; k++) {
    const int i = w.imu_frame[k], j = i + 1;
    double r[15], J[15 * 30];
    eval_imu(w.imu[k], &P.imu_sqrt[(size_t)k * 225], P.opt.g_norm, st.para_Pose[i], st.para_SpeedBias[i],
             st.para_Pose[j], st.para_SpeedBias[j], r, lin ? J : nullptr);
    for (int q = 0; q < 15; q++) cost += 0.5 * r[q] * r[q];
    if (lin) {
      int map[30];
      for (int q = 0; q < 6; q++) { map[q] = T_POSE(i) + q; map[15 + q] = T_POSE(j) + q; }
      for (int q = 0; q < 9; q++) { map[6 + q] = T_SB(i) + q; map[21 + q] = T_SB(j) + q; }
      accum(*lin, r, J, 15, 30, map);
    }
  }
  // --- wheel (estimator.cpp:3181-3212), no loss
  for (int k = 0; k < w.n_wheel; k++) {
    const int i = w.wheel_frame[k], j = i + 1;
    double r[6], J[6 * 22];
    eval_wheel(w.wheel[k], &P.wheel_sqrt[(size_t)k * 36], st.para_Pose[i], st.para_Pose[j], st.para_Ex_Pose_wheel,
               st.para_Ix_wheel[0], st.para_Ix_wheel[1], st.para_Ix_wheel[2], st.para_Td_wheel, r, lin ? J : nullptr);
    for (int q = 0; q < 6; q++) cost += 0.5 * r[q] * r[q];
    if (lin) {
      int map[22];
      for (int q = 0; q < 6; q++) { map[q] = T_POSE(i) + q; map[6 + q] = T_POSE(j) + q; map[12 + q] = T_EXW + q; }
      map[18] = T_SX; map[19] = T_SY; map[20] = T_SW; map[21] = T_TDW;
      accum(*lin, r, J, 6, 22, map);
    }
  }
  // --- visual (estimator.cpp:3326-3358), HuberLoss(1.0)
  const gfbe_visual &v = w.vis;
  for (int k = 0; k < v.n_factor; k++) {
    const int i = v.imu_i[k], j = v.imu_j[k], l = v.feature_index[k];
    double r[2], J[40];
    eval_visual(st.para_Pose[i], st.para_Pose[j], st.para_Ex_Pose, lam[l], st.para_Td, v.pts_i + 3 * k, v.pts_j + 3 * k,
                v.vel_i + 2 * k, v.vel_j + 2 * k, v.td_i[k], v.td_j[k], P.opt.vis_sqrt_info, r, lin ? J : nullptr);
    cost += robustify(r, lin ? J : nullptr, 2, 20, P.opt.huber_delta);
    if (lin) {
      int map[20];
      for (int q = 0; q < 6; q++) { map[q] = T_POSE(i) + q; map[6 + q] = T_POSE(j) + q; map[12 + q] = T_EX + q; }
      map[18] = -1; map[19] = T_TD;
      accum(*lin, r, J, 2, 20, map);
      if (P.lm_free[l]) {
        const double w0 = J[18], w1 = J[38];
        lin->Hll[l] += w0 * w0 + w1 * w1;
        lin->gl[l] += w0 * r[0] + w1 * r[1];
        double *h = &lin->Hpl[(size_t)l * NV];
        for (int a = 0; a < 20; a++) if (map[a] >= 0) h[map[a]] += J[a] * w0 + J[20 + a] * w1;
      }
    }
  }
  // --- LiDAR point-to-plane factors on one pose (the joint LIO + VIO solve; LidarPlaneNormFactor, lidarFactor.cpp:18-51;
  //     HuberLoss as in lidarodom.cpp:539)
  for (int k = 0; k < w.lio.n; k++) {
    const double *x = st.para_Pose[w.lio.frame], *p = w.lio.pts + 3 * k, *nv = w.lio.normals + 3 * k;
    const double sw = w.lio.sqrt_info * (w.lio.weights ? w.lio.weights[k] : 1.0);
    const M3 R = rot(q4(x + 3));
    const V3 pw = R * v3(p) + v3(x), nR = T(R) * v3(nv);
    double r[1] = {sw * (dot(v3(nv), pw) + w.lio.offsets[k])};
    const V3 jr = cross(nR, v3(p));            // n^T R [p]x = (R^T n x p)^T
    double J[6] = {sw * nv[0], sw * nv[1], sw * nv[2], -sw * jr.x, -sw * jr.y, -sw * jr.z};
    if (w.lio.huber_delta > 0) cost += robustify(r, lin ? J : nullptr, 1, 6, w.lio.huber_delta);
    else cost += 0.5 * r[0] * r[0];
    if (lin) {
      int map[6];
      for (int q = 0; q < 6; q++) map[q] = T_POSE(w.lio.frame) + q;
      accum(*lin, r, J, 1, 6, map);
    }
  }
  // --- PlaneFactor per pose i < frame_count (estimator.cpp:3214-3220; plane_factor.h:25-122), no loss
  if (w.use_plane) {
    for (int i = 0; i < w.frame_count; i++) {
      double r[3], J[48];
      gfo_plane_eval(nullptr, 1, st.para_Pose[i], st.para_Ex_Pose_wheel, st.para_plane_R, st.para_plane_Z, w.plane_noise_inv, r, lin ? J : nullptr, nullptr);
      for (int q = 0; q < 3; q++) cost += 0.5 * r[q] * r[q];
      if (lin) {
        int map[16];
        for (int q = 0; q < 6; q++) { map[q] = T_POSE(i) + q; map[6 + q] = T_EXW + q; }
        for (int q = 0; q < 3; q++) map[12 + q] = T_PLR + q;
        map[15] = T_PLZ;
        accum(*lin, r, J, 3, 16, map);
      }
    }
  }
  // --- PoseAnchorFactor on Pose[0] (estimator.cpp:3004-3012; pose_anchor_factor.cpp:8-32), no loss
  if (w.use_anchor) {
    double r[6], J[36];
    gfo_anchor_eval(nullptr, 1, st.para_Pose[0], w.anchor_pose, w.anchor_sqrt_info, r, lin ? J : nullptr, nullptr);
    for (int q = 0; q < 6; q++) cost += 0.5 * r[q] * r[q];
    if (lin) {
      int map[6];
      for (int q = 0; q < 6; q++) map[q] = T_POSE(0) + q;
      accum(*lin, r, J, 6, 6, map);
    }
  }
  // --- GNSS: GnssPsrDoppFactor per observation, DtDdtFactor / DdtSmoothFactor chains (estimator.cpp:3239-3291), no loss
  if (P.gnss_factors)
    gnss_blocks(w, st, lin != nullptr, false, [&](const double *r, const double *J, int nr, int nc, const int *map) {
      for (int q = 0; q < nr; q++) cost += 0.5 * r[q] * r[q];
      if (lin) accum(*lin, r, J, nr, nc, map);
    });
  if (lin) {
    // Remove constant / unused dims from the reduced program.
    for (int a = 0; a < ND; a++)
      if (!P.act[a]) {
        lin->g[a] = 0;
        for (int b = 0; b < ND; b++) lin->H[(size_t)a * ND + b] = lin->H[(size_t)b * ND + a] = 0;
      }
    for (int l = 0; l < P.L; l++) {
      double *h = &lin->Hpl[(size_t)l * NV];
      for (int a = 0; a < NV; a++) if (!P.act[a]) h[a] = 0;
    }
    lin->cost = cost;
  }
  return cost;
}

// x (+) delta on every free block (PoseLocalParameterization::Plus, pose_local_parameterization.cpp:12-27;
// PoseSubsetParameterization::Plus masks, pose_subset_parameterization.cpp:27-55).
static void plus(const Problem &P, const gfbe_state &x, const double *lam, const double *dp, const double *dl,
                 gfbe_state &y, double *lam_y) {
  y = x;
  for (int b = 0; b < GFBE_BLK_COUNT; b++) {
    if (!P.blk_free[b]) continue;
    const double *xb = block_ptr(x, b);
    double *yb = block_ptr(y, b);
    const int off = tan_off(b), gs = block_global_size(b);
    if (gs == 7) {
      double d[6];
      for (int k = 0; k < 6; k++) d[k] = dp[off + k];
      const uint8_t *mask = (b == GFBE_BLK_EX_CAM) ? P.win->ex_cam_mask : (b == GFBE_BLK_EX_WHEEL ? P.win->ex_wheel_mask : nullptr);
      if (mask) for (int k = 0; k < 6; k++) if (mask[k]) d[k] = 0;
      for (int k = 0; k < 3; k++) yb[k] = xb[k] + d[k];
      Q4 q = normalized(q4(xb + 3) * deltaQ(v3(d + 3)));
      yb[3] = q.x; yb[4] = q.y; yb[5] = q.z; yb[6] = q.w;
    } else if (gs == 4) {   // para_plane_R: OrientationSubsetParameterization({2}) (estimator.cpp:3122; .cpp:27-45)
      const uint8_t constant[3] = {0, 0, 1};
      gfo_orientation_subset_plus(xb, dp + off, constant, yb);
    } else {
      for (int k = 0; k < gs; k++) yb[k] = xb[k] + dp[off + k];
    }
  }
  for (int l = 0; l < P.L; l++) lam_y[l] = P.lm_free[l] ? lam[l] + dl[l] : lam[l];
}

static double ambient_norm2_diff(const Problem &P, const gfbe_state &a, const double *la, const gfbe_state *b, const double *lb) {
  double s = 0;
  for (int id = 0; id < GFBE_BLK_COUNT; id++) {
    if (!P.blk_free[id]) continue;
    const double *pa = block_ptr(a, id);
    const double *pb = b ? block_ptr(*b, id) : nullptr;
    for (int k = 0; k < block_global_size(id); k++) { double d = pa[k] - (pb ? pb[k] : 0.0); s += d * d; }
  }
  for (int l = 0; l < P.L; l++) if (P.lm_free[l]) { double d = la[l] - (lb ? lb[l] : 0.0); s += d * d; }
  return s;
}

// Dense Cholesky solve S y = rhs on the active dims (compacted: the inactive rows would be identity rows that touch nothing).
// false on failure.
static bool chol_solve(std::vector<double> &S, const std::vector<double> &rhs, const bool *act, double *y) {
  int idx[ND], n = 0;
  for (int a = 0; a < ND; a++) { y[a] = 0.0; if (act[a]) idx[n++] = a; }
  std::vector<double> C((size_t)n * n), z(n), yc(n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) C[(size_t)i * n + j] = S[(size_t)idx[i] * ND + idx[j]];
  // in-place lower Cholesky (Eigen::LLT as used by Ceres 1.14 DenseSchurComplementSolver)
  for (int j = 0; j < n; j++) {
    double d = C[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= C[(size_t)j * n + k] * C[(size_t)j * n + k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    d = std::sqrt(d);
    C[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = C[(size_t)i * n + j];
      const double *Li = &C[(size_t)i * n], *Lj = &C[(size_t)j * n];
      for (int k = 0; k < j; k++) s -= Li[k] * Lj[k];
      C[(size_t)i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; i++) { double s = rhs[idx[i]]; for (int k = 0; k < i; k++) s -= C[(size_t)i * n + k] * z[k]; z[i] = s / C[(size_t)i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { double s = z[i]; for (int k = i + 1; k < n; k++) s -= C[(size_t)k * n + i] * yc[k]; yc[i] = s / C[(size_t)i * n + i]; }
  for (int i = 0; i < n; i++) { if (!std::isfinite(yc[i])) return false; y[idx[i]] = yc[i]; }
  return true;
}

struct Solution {
  gfbe_state x;
  std::vector<double> lam;
  gfbe_summary sum;
};

static void solve(const Problem &P, Solution &sol) {
  const gfbe_window &w = *P.win;
  const gfbe_options &o = P.opt;
  const int L = P.L;
  gfbe_summary &S = sol.sum;
  std::memset(&S, 0, sizeof S);
  const auto t_begin = std::chrono::steady_clock::now();      // (Ceres' clock starts before the first evaluation)
  sol.x = w.state;
  sol.lam.assign(w.para_Feature, w.para_Feature + L);
  Lin lin;
  double cost = evaluate(P, sol.x, sol.lam.data(), &lin);
  S.initial_cost = S.final_cost = cost;
  S.cost_history[0] = cost;
  // Jacobi scaling, computed once at iteration 0 (TrustRegionMinimizer::IterationZero).
  std::vector<double> sp(ND, 1.0), sl(L, 1.0);
  if (o.jacobi_scaling) {
    for (int a = 0; a < ND; a++) if (P.act[a]) sp[a] = 1.0 / (1.0 + std::sqrt(lin.H[(size_t)a * ND + a]));
    for (int l = 0; l < L; l++) if (P.lm_free[l]) sl[l] = 1.0 / (1.0 + std::sqrt(lin.Hll[l]));
  }
  const double min_diag = 1e-6, max_diag = 1e32, min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0;
  double radius = o.initial_trust_region_radius, mu = min_mu;
  bool reuse = false;
  double x_norm = std::sqrt(ambient_norm2_diff(P, sol.x, sol.lam.data(), nullptr, nullptr));
  int invalid_steps = 0;
  S.termination = 0;
  S.status = GFBE_NO_CONVERGENCE;
  // per-linearisation dogleg data
  std::vector<double> Dp(ND, 1.0), Dl(L, 1.0), gts(ND, 0.0), glts(L, 0.0), yp(ND, 0.0), yl(L, 0.0), vp(ND, 0.0), vl(L, 0.0);
  double G2 = 0, N2 = 0, gy = 0, vHv = 0, vHy = 0, yHy = 0, alpha = 0;
  auto grad_max = [&]() { double m = 0; for (int a = 0; a < ND; a++) if (P.act[a]) m = std::max(m, std::fabs(lin.g[a])); for (int l = 0; l < L; l++) if (P.lm_free[l]) m = std::max(m, std::fabs(lin.gl[l])); return m; };
  std::vector<double> St((size_t)ND * ND), rhs(ND), dp(ND), dl(L), lam_c(L);
  gfbe_state xc;
  int it = 0;
  while (true) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (it >= std::min(o.max_num_iterations, 15)) { S.termination = 0; break; }   // summary arrays hold 16 entries
    // Solver::Options::max_solver_time_in_seconds (estimator.cpp:3369-3376: SOLVER_TIME): checked between iterations like
    // TrustRegionMinimizer does; termination NO_CONVERGENCE. 0 = no cap (the deterministic setting every parity test uses)
    if (o.max_solver_time_in_seconds > 0.0 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() >= o.max_solver_time_in_seconds) { S.termination = 0; break; }
    if (grad_max() <= o.gradient_tolerance) { S.termination = 3; S.status = GFBE_OK; break; }
    if (radius < 1e-32) { S.termination = 4; break; }
    it++;
    S.iterations = it;
    if (!reuse) {
      // scaled system: Ht = s H s, gt = s g; D = sqrt(clamp(diag(Ht)))
      G2 = 0;
      for (int a = 0; a < ND; a++) {
        if (!P.act[a]) { Dp[a] = 1; gts[a] = 0; vp[a] = 0; continue; }
        double d2 = std::min(std::max(sp[a] * sp[a] * lin.H[(size_t)a * ND + a], min_diag), max_diag);
        Dp[a] = std::sqrt(d2); gts[a] = sp[a] * lin.g[a]; vp[a] = gts[a] / d2; G2 += gts[a] * gts[a] / d2;
      }
      for (int l = 0; l < L; l++) {
        if (!P.lm_free[l]) { Dl[l] = 1; glts[l] = 0; vl[l] = 0; continue; }
        double d2 = std::min(std::max(sl[l] * sl[l] * lin.Hll[l], min_diag), max_diag);
        Dl[l] = std::sqrt(d2); glts[l] = sl[l] * lin.gl[l]; vl[l] = glts[l] / d2; G2 += glts[l] * glts[l] / d2;
      }
      // Gauss-Newton step with the mu-regularised Schur solve (DoglegStrategy::ComputeGaussNewtonStep)
      bool solved = false;
      // fault injection (tests only): the first factorisation of iteration gfbe_options.test_fail_chol_iter is declared failed,
      // so that the mu-retry path (DoglegStrategy: mu *= 10 until the linear solver succeeds) is exercised — Gauss-Newton
      // systems with mu >= 1e-8 practically never fail on their own
      int inject = (o.test_fail_chol_iter > 0 && o.test_fail_chol_iter == it) ? std::max(o.test_fail_chol_count, 1) : 0;   // consecutive failing attempts
      while (mu < max_mu) {
        for (int a = 0; a < ND; a++) {
          for (int b = 0; b < ND; b++) St[(size_t)a * ND + b] = sp[a] * sp[b] * lin.H[(size_t)a * ND + b];
          St[(size_t)a * ND + a] += mu * Dp[a] * Dp[a];
          rhs[a] = gts[a];
        }
        for (int l = 0; l < L; l++) {
          if (!P.lm_free[l]) continue;
          const double *h = &lin.Hpl[(size_t)l * NV];
          const double hll = sl[l] * sl[l] * lin.Hll[l] + mu * Dl[l] * Dl[l];
          const double inv = 1.0 / hll;
          int idx[NV]; double hv[NV]; int nz = 0;
          for (int a = 0; a < NV; a++) if (h[a] != 0.0) { idx[nz] = a; hv[nz] = sl[l] * sp[a] * h[a]; nz++; }
          for (int p = 0; p < nz; p++) {
            const double f = hv[p] * inv;
            rhs[idx[p]] -= f * glts[l];
            double *row = &St[(size_t)idx[p] * ND];
            for (int q = 0; q < nz; q++) row[idx[q]] -= f * hv[q];
          }
        }
        bool chol_ok = chol_solve(St, rhs, P.act, yp.data());
        if (inject > 0) { chol_ok = false; inject--; }
        if (chol_ok) {
          bool fin = true;
          for (int l = 0; l < L; l++) {
            if (!P.lm_free[l]) { yl[l] = 0; continue; }
            const double *h = &lin.Hpl[(size_t)l * NV];
            double s = glts[l];
            for (int a = 0; a < NV; a++) if (h[a] != 0.0) s -= sl[l] * sp[a] * h[a] * yp[a];
            yl[l] = s / (sl[l] * sl[l] * lin.Hll[l] + mu * Dl[l] * Dl[l]);
            if (!std::isfinite(yl[l])) fin = false;
          }
          if (fin) { solved = true; break; }
        }
        mu *= mu_inc;
      }
      if (!solved) { S.termination = 4; S.status = GFBE_NUMERICAL_FAILURE; break; }
      // Gram scalars of v = gt/D^2 and y in the scaled space
      auto quad = [&](const std::vector<double> &ap, const std::vector<double> &al, const std::vector<double> &bp, const std::vector<double> &bl) {
        double s = 0;
        for (int a = 0; a < ND; a++) {
          if (!P.act[a] || ap[a] == 0.0) continue;
          const double *row = &lin.H[(size_t)a * ND];
          double t = 0;
          for (int b = 0; b < ND; b++) t += sp[b] * row[b] * bp[b];
          s += sp[a] * ap[a] * t;
        }
        for (int l = 0; l < L; l++) {
          if (!P.lm_free[l]) continue;
          const double *h = &lin.Hpl[(size_t)l * NV];
          double ha = 0, hb = 0;
          for (int a = 0; a < NV; a++) if (h[a] != 0.0) { ha += sp[a] * h[a] * ap[a]; hb += sp[a] * h[a] * bp[a]; }
          s += sl[l] * (al[l] * hb + bl[l] * ha) + sl[l] * sl[l] * lin.Hll[l] * al[l] * bl[l];
        }
        return s;
      };
      vHv = quad(vp, vl, vp, vl); vHy = quad(vp, vl, yp, yl); yHy = quad(yp, yl, yp, yl);
      alpha = G2 / vHv;
      N2 = 0; gy = 0;
      for (int a = 0; a < ND; a++) if (P.act[a]) { N2 += Dp[a] * Dp[a] * yp[a] * yp[a]; gy += gts[a] * yp[a]; }
      for (int l = 0; l < L; l++) if (P.lm_free[l]) { N2 += Dl[l] * Dl[l] * yl[l] * yl[l]; gy += glts[l] * yl[l]; }
      reuse = true;
    }
    // DoglegStrategy::ComputeTraditionalDoglegStep; step (Jacobi-scaled space) = c1 v + c2 y
    const double g_norm = std::sqrt(G2), gn_norm = std::sqrt(N2);
    double c1, c2, step_norm;
    if (gn_norm <= radius) { c1 = 0; c2 = -1; step_norm = gn_norm; }
    else if (g_norm * alpha >= radius) { c1 = -radius / g_norm; c2 = 0; step_norm = radius; }
    else {
      const double b_dot_a = alpha * gy;
      const double a_sq = (alpha * g_norm) * (alpha * g_norm);
      const double bma = a_sq - 2 * b_dot_a + N2;
      const double c = b_dot_a - a_sq;
      const double d = std::sqrt(c * c + bma * (radius * radius - a_sq));
      const double beta = (c <= 0) ? (d - c) / bma : (radius * radius - a_sq) / (d + c);
      c1 = -alpha * (1.0 - beta); c2 = -beta;
      step_norm = std::sqrt(std::max(0.0, c1 * c1 * G2 + 2 * c1 * c2 * gy + c2 * c2 * N2));
    }
    const double model_change = -(c1 * G2 + c2 * gy) - 0.5 * (c1 * c1 * vHv + 2 * c1 * c2 * vHy + c2 * c2 * yHy);
    if (!(model_change > 0.0)) {          // invalid step (TrustRegionMinimizer::HandleInvalidStep)
      S.accepted[it] = 0; S.cost_history[it] = cost;
      if (++invalid_steps >= 5) { S.termination = 4; S.status = GFBE_NUMERICAL_FAILURE; break; }
      mu *= mu_inc; reuse = false;
      continue;
    }
    invalid_steps = 0;
    for (int a = 0; a < ND; a++) dp[a] = P.act[a] ? sp[a] * (c1 * vp[a] + c2 * yp[a]) : 0.0;
    for (int l = 0; l < L; l++) dl[l] = P.lm_free[l] ? sl[l] * (c1 * vl[l] + c2 * yl[l]) : 0.0;
    plus(P, sol.x, sol.lam.data(), dp.data(), dl.data(), xc, lam_c.data());
    double cand_cost = evaluate(P, xc, lam_c.data(), nullptr);
    if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
    const double step_amb = std::sqrt(ambient_norm2_diff(P, sol.x, sol.lam.data(), &xc, lam_c.data()));
    S.cost_history[it] = cost;
    if (step_amb <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) { S.termination = 2; S.status = GFBE_OK; break; }
    const double cost_change = cost - cand_cost;
    if (std::fabs(cost_change) <= o.function_tolerance * cost) { S.termination = 1; S.status = GFBE_OK; break; }
    const double quality = cost_change / model_change;
    if (quality > o.min_relative_decrease) {
      sol.x = xc; sol.lam = lam_c; cost = cand_cost;
      x_norm = std::sqrt(ambient_norm2_diff(P, sol.x, sol.lam.data(), nullptr, nullptr));
      evaluate(P, sol.x, sol.lam.data(), &lin);
      S.accepted[it] = 1; S.num_successful++;
      S.cost_history[it] = cost;
      if (quality < 0.25) radius *= 0.5;                                  // DoglegStrategy::StepAccepted
      if (quality > 0.75) radius = std::max(radius, 3.0 * step_norm);
      mu = std::max(min_mu, 2.0 * mu / mu_inc);
      reuse = false;
    } else {
      S.accepted[it] = 0;
      radius *= 0.5; reuse = true;                                        // DoglegStrategy::StepRejected
    }
  }
  S.final_cost = cost;
  S.final_radius = radius;
}

// double2vector()'s gauge fix followed by vector2double() (estimator.cpp:2501-2555, 2341-2362).
static void reanchor(const gfbe_state &before, const gfbe_state &after, int frame_count, gfbe_state &out) {
  out = after;
  M3 R0 = rot(q4(before.para_Pose[0] + 3));
  V3 origin_R0 = R2ypr(R0), origin_P0 = v3(before.para_Pose[0]);
  M3 R00 = rot(q4(after.para_Pose[0] + 3));
  V3 origin_R00 = R2ypr(R00);
  const double y_diff = origin_R0.x - origin_R00.x;
  M3 rot_diff = ypr2R(v3(y_diff, 0, 0));
  if (std::fabs(std::fabs(origin_R0.y) - 90) < 1.0 || std::fabs(std::fabs(origin_R00.y) - 90) < 1.0)
    rot_diff = R0 * T(R00);
  (void)frame_count;
  for (int i = 0; i < GFBE_NFRAMES; i++) {
    M3 Ri = rot_diff * rot(normalized(q4(after.para_Pose[i] + 3)));
    V3 Pi = rot_diff * (v3(after.para_Pose[i]) - v3(after.para_Pose[0])) + origin_P0;
    V3 Vi = rot_diff * v3(after.para_SpeedBias[i]);
    Q4 q = quat_from_rot(Ri);
    double *p = out.para_Pose[i];
    p[0] = Pi.x; p[1] = Pi.y; p[2] = Pi.z; p[3] = q.x; p[4] = q.y; p[5] = q.z; p[6] = q.w;
    out.para_SpeedBias[i][0] = Vi.x; out.para_SpeedBias[i][1] = Vi.y; out.para_SpeedBias[i][2] = Vi.z;
  }
  // extrinsics go through R as well: ric = q.toRotationMatrix() then Quaterniond{ric} (estimator.cpp:2575-2579, 2373)
  Q4 qe = quat_from_rot(rot(q4(after.para_Ex_Pose + 3)));
  out.para_Ex_Pose[3] = qe.x; out.para_Ex_Pose[4] = qe.y; out.para_Ex_Pose[5] = qe.z; out.para_Ex_Pose[6] = qe.w;
  Q4 qw = quat_from_rot(rot(normalized(q4(after.para_Ex_Pose_wheel + 3))));
  out.para_Ex_Pose_wheel[3] = qw.x; out.para_Ex_Pose_wheel[4] = qw.y; out.para_Ex_Pose_wheel[5] = qw.z; out.para_Ex_Pose_wheel[6] = qw.w;
}

// ---------------------------------------------------------------------------------------------
// Marginalisation (dense, as the reference does it).
// ---------------------------------------------------------------------------------------------
static int marginalize(const Problem &P, const gfbe_state &st, const double *lam, int flag, gfbe_prior *out,
                       double *A_out, double *b_out) {
  const gfbe_window &w = *P.win;
  const gfbe_options &o = P.opt;
  out->valid = 0; out->n = 0; out->n_blocks = 0;
  // Parameter blocks involved: dense ids + landmark ids (GFBE_BLK_COUNT + l).
  std::vector<int> drop, keep;
  bool touched[GFBE_BLK_COUNT];
  for (int b = 0; b < GFBE_BLK_COUNT; b++) touched[b] = false;
  std::vector<int> lm0;   // landmarks with start_frame == 0, ascending feature_index
  const bool has_prior = P.has_prior;
  if (has_prior) for (int b = 0; b < w.prior->n_blocks; b++) touched[w.prior->block_id[b]] = true;
  bool use_imu0 = false, use_wheel0 = false;
  int imu0 = -1, wheel0 = -1;
  if (flag == GFBE_MARGIN_OLD) {
    for (int k = 0; k < w.n_imu; k++) if (w.imu_frame[k] == 0 && w.imu[k].sum_dt < 10.0) { use_imu0 = true; imu0 = k; }
    for (int k = 0; k < w.n_wheel; k++) if (w.wheel_frame[k] == 0 && w.wheel[k].sum_dt < 10.0) { use_wheel0 = true; wheel0 = k; }
    if (use_imu0) touched[GFBE_BLK_POSE0] = touched[GFBE_BLK_SB0] = touched[GFBE_BLK_POSE0 + 1] = touched[GFBE_BLK_SB0 + 1] = true;
    if (use_wheel0) touched[GFBE_BLK_POSE0] = touched[GFBE_BLK_POSE0 + 1] = touched[GFBE_BLK_EX_WHEEL] = touched[GFBE_BLK_SX] = touched[GFBE_BLK_SY] = touched[GFBE_BLK_SW] = touched[GFBE_BLK_TD_WHEEL] = true;
    if (w.use_plane && w.frame_count > 0)     // estimator.cpp:3441-3448: the PlaneFactor of frame 0, drop set {Pose[0]}
      touched[GFBE_BLK_POSE0] = touched[GFBE_BLK_EX_WHEEL] = touched[GFBE_BLK_PLANE_R] = touched[GFBE_BLK_PLANE_Z] = true;
    if (w.gnss_ready) {                          // estimator.cpp:3459-3496: the GNSS factors of frame 0
      for (int k = 0; k < w.n_gnss; k++) {
        if (w.gnss_obs[k].frame != 0) continue;
        touched[GFBE_BLK_POSE0] = touched[GFBE_BLK_SB0] = touched[GFBE_BLK_POSE0 + 1] = touched[GFBE_BLK_SB0 + 1] = true;
        touched[GFBE_BLK_RCV_DT0 + w.gnss_obs[k].sys_idx] = touched[GFBE_BLK_RCV_DDT0] = touched[GFBE_BLK_YAW_ENU] = touched[GFBE_BLK_ANC_ECEF] = true;
      }
      for (int k = 0; k < 4; k++) touched[GFBE_BLK_RCV_DT0 + k] = touched[GFBE_BLK_RCV_DT0 + 4 + k] = true;
      touched[GFBE_BLK_RCV_DDT0] = touched[GFBE_BLK_RCV_DDT0 + 1] = true;
    }
    std::vector<uint8_t> seen(P.L, 0);
    for (int k = 0; k < w.vis.n_factor; k++) {
      if (w.vis.imu_i[k] != 0) continue;
      touched[GFBE_BLK_POSE0] = touched[GFBE_BLK_POSE0 + w.vis.imu_j[k]] = touched[GFBE_BLK_EX_CAM] = touched[GFBE_BLK_TD] = true;
      int l = w.vis.feature_index[k];
      if (!seen[l]) { seen[l] = 1; lm0.push_back(l); }
    }
    std::sort(lm0.begin(), lm0.end());
    // drop set: Pose[0], SpeedBias[0] (prior: estimator.cpp:3405-3407; IMU {0,1}: :3421; wheel {0}: :3434; visual {0,3}: :3526)
    if (touched[GFBE_BLK_POSE0]) drop.push_back(GFBE_BLK_POSE0);
    if (touched[GFBE_BLK_SB0]) drop.push_back(GFBE_BLK_SB0);
    if (w.gnss_ready) {                          // drop sets {0, 1, 4, 5}, {0, 2}, {0} (:3477, :3487, :3494): rcv_dt[0][k], rcv_ddt[0]
      for (int k = 0; k < 4; k++) drop.push_back(GFBE_BLK_RCV_DT0 + k);
      drop.push_back(GFBE_BLK_RCV_DDT0);
    }
    for (int l : lm0) drop.push_back(GFBE_BLK_COUNT + l);
  } else {
    // MARGIN_SECOND_NEW: only when the prior touches Pose[WINDOW_SIZE-1] (estimator.cpp:3600-3601)
    if (!has_prior && w.prior && !w.prior->valid) {
      // estimator.cpp:3622-3632: last_marginalization_info exists but is NOT valid (a marginalisation with nothing to drop,
      // marginalization_factor.cpp:204-210, still lists its parameter blocks) and lists Pose[WINDOW_SIZE-1]: the only residual is a
      // PoseAnchorFactor on Pose[0] anchored at Pose[0] itself, drop set {Pose[0]} — m = 6, nothing kept: the new info is VALID
      // and empty (no blocks, n = 0), i.e. the invalid prior is replaced by no prior at all
      bool lists = false;
      for (int b = 0; b < w.prior->n_blocks; b++) lists |= w.prior->block_id[b] == GFBE_BLK_POSE0 + GFBE_WINDOW_SIZE - 1;
      if (lists) { out->valid = 1; out->n = 0; out->n_blocks = 0; return 0; }
    }
    if (!has_prior || !touched[GFBE_BLK_POSE0 + GFBE_WINDOW_SIZE - 1]) return 1;   // nothing to do: keep old prior
    drop.push_back(GFBE_BLK_POSE0 + GFBE_WINDOW_SIZE - 1);
  }
  std::vector<int> idx_of(GFBE_BLK_COUNT + P.L, -1);
  int pos = 0;
  for (int id : drop) { idx_of[id] = pos; pos += (id < GFBE_BLK_COUNT) ? block_local_size(id) : 1; }
  int m = pos;
  for (int b = 0; b < GFBE_BLK_COUNT; b++)
    if (touched[b] && idx_of[b] < 0) { keep.push_back(b); idx_of[b] = pos; pos += block_local_size(b); }
  const int n = pos - m;
  if (m == 0) { out->valid = 0; return 2; }       // marginalization_factor.cpp:205-210
  std::vector<double> A((size_t)pos * pos, 0.0), b(pos, 0.0);
  auto add = [&](const double *r, const double *J, int nr, int nc, const int *colmap) {
    for (int a = 0; a < nc; a++) {
      if (colmap[a] < 0) continue;
      double gr = 0;
      for (int i = 0; i < nr; i++) gr += J[i * nc + a] * r[i];
      b[colmap[a]] += gr;
      for (int c = 0; c < nc; c++) {
        if (colmap[c] < 0) continue;
        double s = 0;
        for (int i = 0; i < nr; i++) s += J[i * nc + a] * J[i * nc + c];
        A[(size_t)colmap[a] * pos + colmap[c]] += s;
      }
    }
  };
  if (has_prior) {
    const gfbe_prior &pr = *w.prior;
    const int pn = pr.n;
    std::vector<double> r(pn);
    eval_prior(pr, st, r.data());
    std::vector<int> map(pn, -1);
    for (int q = 0; q < pr.n_blocks; q++)
      for (int k = 0; k < block_local_size(pr.block_id[q]); k++) map[pr.block_idx[q] + k] = idx_of[pr.block_id[q]] + k;
    add(r.data(), pr.J0, pn, pn, map.data());
  }
  if (use_imu0) {
    double r[15], J[15 * 30]; int map[30];
    eval_imu(w.imu[imu0], &P.imu_sqrt[(size_t)imu0 * 225], o.g_norm, st.para_Pose[0], st.para_SpeedBias[0], st.para_Pose[1], st.para_SpeedBias[1], r, J);
    for (int q = 0; q < 6; q++) { map[q] = idx_of[GFBE_BLK_POSE0] + q; map[15 + q] = idx_of[GFBE_BLK_POSE0 + 1] + q; }
    for (int q = 0; q < 9; q++) { map[6 + q] = idx_of[GFBE_BLK_SB0] + q; map[21 + q] = idx_of[GFBE_BLK_SB0 + 1] + q; }
    add(r, J, 15, 30, map);
  }
  if (use_wheel0) {
    double r[6], J[6 * 22]; int map[22];
    eval_wheel(w.wheel[wheel0], &P.wheel_sqrt[(size_t)wheel0 * 36], st.para_Pose[0], st.para_Pose[1], st.para_Ex_Pose_wheel,
               st.para_Ix_wheel[0], st.para_Ix_wheel[1], st.para_Ix_wheel[2], st.para_Td_wheel, r, J);
    for (int q = 0; q < 6; q++) { map[q] = idx_of[GFBE_BLK_POSE0] + q; map[6 + q] = idx_of[GFBE_BLK_POSE0 + 1] + q; map[12 + q] = idx_of[GFBE_BLK_EX_WHEEL] + q; }
    map[18] = idx_of[GFBE_BLK_SX]; map[19] = idx_of[GFBE_BLK_SY]; map[20] = idx_of[GFBE_BLK_SW]; map[21] = idx_of[GFBE_BLK_TD_WHEEL];
    add(r, J, 6, 22, map);
  }
  if (flag == GFBE_MARGIN_OLD && w.use_plane && w.frame_count > 0) {
    double r[3], J[48]; int map[16];
    gfo_plane_eval(nullptr, 1, st.para_Pose[0], st.para_Ex_Pose_wheel, st.para_plane_R, st.para_plane_Z, w.plane_noise_inv, r, J, nullptr);
    for (int q = 0; q < 6; q++) { map[q] = idx_of[GFBE_BLK_POSE0] + q; map[6 + q] = idx_of[GFBE_BLK_EX_WHEEL] + q; }
    for (int q = 0; q < 3; q++) map[12 + q] = idx_of[GFBE_BLK_PLANE_R] + q;     // (the factor's 3 x 4 block has a zero 4th column)
    map[15] = idx_of[GFBE_BLK_PLANE_Z];
    add(r, J, 3, 16, map);
  }
  if (flag == GFBE_MARGIN_OLD && w.gnss_ready)
    gnss_blocks(w, st, true, true, [&](const double *r, const double *J, int nr, int nc, const int *tmap) {
      int map[18];
      for (int a = 0; a < nc; a++) {     // tangent dim -> (block, offset) -> position in the marginalisation's ordering
        map[a] = -1;
        for (int b2 = 0; b2 < GFBE_BLK_COUNT && map[a] < 0; b2++)
          if (tmap[a] >= tan_off(b2) && tmap[a] < tan_off(b2) + block_local_size(b2)) map[a] = idx_of[b2] + (tmap[a] - tan_off(b2));
      }
      add(r, J, nr, nc, map);
    });
  if (flag == GFBE_MARGIN_OLD) {
    const gfbe_visual &v = w.vis;
    for (int k = 0; k < v.n_factor; k++) {
      if (v.imu_i[k] != 0) continue;
      const int j = v.imu_j[k], l = v.feature_index[k];
      double r[2], J[40]; int map[20];
      eval_visual(st.para_Pose[0], st.para_Pose[j], st.para_Ex_Pose, lam[l], st.para_Td, v.pts_i + 3 * k, v.pts_j + 3 * k,
                  v.vel_i + 2 * k, v.vel_j + 2 * k, v.td_i[k], v.td_j[k], o.vis_sqrt_info, r, J);
      robustify(r, J, 2, 20, o.huber_delta);       // ResidualBlockInfo::Evaluate with loss_function
      for (int q = 0; q < 6; q++) { map[q] = idx_of[GFBE_BLK_POSE0] + q; map[6 + q] = idx_of[GFBE_BLK_POSE0 + j] + q; map[12 + q] = idx_of[GFBE_BLK_EX_CAM] + q; }
      map[18] = idx_of[GFBE_BLK_COUNT + l]; map[19] = idx_of[GFBE_BLK_TD];
      add(r, J, 2, 20, map);
    }
  }
  if (o.marg_sqrt == 1 && !lm0.empty()) {
    // NOT the reference's construction (its Amm is eigen-decomposed whole, below): the product's order of elimination (DESIGN.md
    // section 6, always on in the product; here tied to marg_sqrt = 1 = "the product's algorithm", for bench.py's like-for-like CPU
    // leg). The landmarks of the drop set are 1-D blocks coupled to the dense dims only: eliminate each with lambda > eps by its own
    // rank-1 term (a landmark with lambda <= eps is dropped, as the thresholded pseudo-inverse would), then hand the (m - L0 + n)
    // system to the code below. The same Schur complement whenever every eigenvalue of Amm exceeds eps.
    const int nl = (int)lm0.size(), md = m - nl, pr = md + n;
    std::vector<int> nz;
    for (int k = 0; k < nl; k++) {
      const int q = md + k;
      const double lq = A[(size_t)q * pos + q];
      if (!(lq > o.marg_eps)) continue;
      nz.clear();
      for (int i = 0; i < pos; i++) if ((i < md || i >= m) && (A[(size_t)q * pos + i] != 0.0 || A[(size_t)i * pos + q] != 0.0)) nz.push_back(i);
      for (int i : nz) {
        const double f = A[(size_t)i * pos + q] / lq;
        b[i] -= f * b[q];
        for (int j : nz) A[(size_t)i * pos + j] -= f * A[(size_t)q * pos + j];
      }
    }
    std::vector<double> A2((size_t)pr * pr), b2(pr);
    auto src = [&](int i) { return i < md ? i : i + nl; };
    for (int i = 0; i < pr; i++) { b2[i] = b[src(i)]; for (int j = 0; j < pr; j++) A2[(size_t)i * pr + j] = A[(size_t)src(i) * pos + src(j)]; }
    A.swap(A2); b.swap(b2);
    // (from here on the landmarks are gone: m, pos and the kept blocks' offsets shrink by L0)
    for (size_t q = 0; q < idx_of.size(); q++) if (idx_of[q] >= m) idx_of[q] -= nl;
    m = md; pos = pr;
  }
  // marginalization_factor.cpp:278-292
  std::vector<double> Amm((size_t)m * m), wv(m), V((size_t)m * m), Amm_inv((size_t)m * m, 0.0);
  for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Amm[(size_t)i * m + j] = 0.5 * (A[(size_t)i * pos + j] + A[(size_t)j * pos + i]);
  sym_eig(Amm.data(), m, wv.data(), V.data());
  for (int k = 0; k < m; k++) {
    if (!(wv[k] > o.marg_eps)) continue;
    const double iw = 1.0 / wv[k];
    for (int i = 0; i < m; i++) { const double f = V[(size_t)i * m + k] * iw; if (f == 0.0) continue; for (int j = 0; j < m; j++) Amm_inv[(size_t)i * m + j] += f * V[(size_t)j * m + k]; }
  }
  // T = Arm * Amm_inv (n x m); A' = Arr - T Amr ; b' = brr - T bmm
  std::vector<double> Tm((size_t)n * m, 0.0), Ap((size_t)n * n), bp(n);
  for (int i = 0; i < n; i++)
    for (int k = 0; k < m; k++) {
      const double a = A[(size_t)(m + i) * pos + k];
      if (a == 0.0) continue;
      for (int j = 0; j < m; j++) Tm[(size_t)i * m + j] += a * Amm_inv[(size_t)k * m + j];
    }
  for (int i = 0; i < n; i++) {
    double sb = b[m + i];
    for (int k = 0; k < m; k++) sb -= Tm[(size_t)i * m + k] * b[k];
    bp[i] = sb;
    for (int j = 0; j < n; j++) {
      double s = A[(size_t)(m + i) * pos + m + j];
      for (int k = 0; k < m; k++) s -= Tm[(size_t)i * m + k] * A[(size_t)k * pos + m + j];
      Ap[(size_t)i * n + j] = s;
    }
  }
  if (A_out) std::memcpy(A_out, Ap.data(), sizeof(double) * n * n);
  if (b_out) std::memcpy(b_out, bp.data(), sizeof(double) * n);
  if (o.marg_sqrt == 1) {
    // NOT the reference's construction: the product's default square root (gfbe_options.marg_sqrt = 1), restated
    // here so that bench.py can time both legs of its CPU / GPU ratio on the SAME algorithm. Diagonally pivoted LDL^T of
    // the symmetrised A' with pivots > eps:  A' ~= sum_k d_k l_k l_k^T (l_k = column p_k of the remaining matrix / d_k),
    // J0 row k = sqrt(d_k) l_k^T,  r0[k] = beta_k / sqrt(d_k) with beta_k the p_k-th entry of the right-hand side after the
    // same eliminations (forward substitution folded in): J0^T J0 = A'+, J0^T r0 = b'+ like the eigen form. Rows past the rank: 0.
    std::vector<double> M((size_t)n * n), rhs(bp);
    std::vector<char> gone(n, 0);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) M[(size_t)i * n + j] = 0.5 * (Ap[(size_t)i * n + j] + Ap[(size_t)j * n + i]);
    for (size_t e = 0; e < (size_t)n * n; e++) out->J0[e] = 0.0;
    for (int k = 0; k < n; k++) out->r0[k] = 0.0;
    for (int k = 0; k < n; k++) {
      int p = -1;
      for (int i = 0; i < n; i++) if (!gone[i] && (p < 0 || M[(size_t)i * n + i] > M[(size_t)p * n + p])) p = i;   // (first of equal maxima)
      if (p < 0 || !(M[(size_t)p * n + p] > o.marg_eps)) break;
      const double dk = M[(size_t)p * n + p], sd = std::sqrt(dk);
      std::vector<double> l(n, 0.0);
      for (int i = 0; i < n; i++) if (!gone[i]) l[i] = M[(size_t)i * n + p] / dk;
      l[p] = 1.0;
      const double beta = rhs[p];
      for (int i = 0; i < n; i++) {
        if (gone[i] || i == p) continue;
        rhs[i] -= l[i] * beta;
        for (int j = 0; j < n; j++) if (!gone[j] && j != p) M[(size_t)i * n + j] -= l[i] * dk * l[j];
      }
      for (int i = 0; i < n; i++) out->J0[(size_t)k * n + i] = sd * l[i];
      out->r0[k] = beta / sd;
      gone[p] = 1;
    }
  } else {
  // :294-302  J0 = diag(sqrt(S)) V^T, r0 = diag(1/sqrt(S)) V^T b
  std::vector<double> w2(n), V2((size_t)n * n);
  sym_eig(Ap.data(), n, w2.data(), V2.data());
  for (int k = 0; k < n; k++) {
    const double Sk = (w2[k] > o.marg_eps) ? w2[k] : 0.0;
    const double Sinv = (w2[k] > o.marg_eps) ? 1.0 / w2[k] : 0.0;
    const double ss = std::sqrt(Sk), si = std::sqrt(Sinv);
    double vb = 0;
    for (int i = 0; i < n; i++) { out->J0[(size_t)k * n + i] = ss * V2[(size_t)i * n + k]; vb += V2[(size_t)i * n + k] * bp[i]; }
    out->r0[k] = si * vb;
  }
  }
  // getParameterBlocks + addr_shift (estimator.cpp:3561-3590 / 3644-3687)
  out->valid = 1; out->n = n; out->n_blocks = (int)keep.size();
  int xoff = 0;
  for (size_t q = 0; q < keep.size(); q++) {
    const int id = keep[q];
    int nid = id;
    const bool is_dt = id >= GFBE_BLK_RCV_DT0 && id < GFBE_BLK_RCV_DDT0, is_ddt = id >= GFBE_BLK_RCV_DDT0;
    if (flag == GFBE_MARGIN_OLD) {                                                      // slot i -> i-1 (pose, speed-bias, receiver clock)
      if (id < GFBE_BLK_EX_CAM || is_ddt) nid = id - 1;
      if (is_dt) nid = id - 4;
    } else {
      if (id == GFBE_BLK_POSE0 + GFBE_WINDOW_SIZE || id == GFBE_BLK_SB0 + GFBE_WINDOW_SIZE || id == GFBE_BLK_RCV_DDT0 + GFBE_WINDOW_SIZE) nid = id - 1;
      if (is_dt && id >= GFBE_BLK_RCV_DT0 + 4 * GFBE_WINDOW_SIZE) nid = id - 4;
    }
    out->block_id[q] = nid;
    out->block_size[q] = block_global_size(id);
    out->block_idx[q] = idx_of[id] - m;
    std::memcpy(out->x0 + xoff, block_ptr(st, id), sizeof(double) * block_global_size(id));
    xoff += block_global_size(id);
  }
  return 0;
}

}  // namespace gfo

using namespace gfo;

extern "C" {

void gfo_default_options(gfbe_options *o) {
  o->struct_size = (int32_t)sizeof(gfbe_options); o->speculative_linearization = 0; o->merge_lin_schur = 0;   // (device options; the second is meaningless on the CPU)
  o->max_num_iterations = 8; o->huber_delta = 1.0; o->vis_sqrt_info = 600.0 / 1.5; o->g_norm = 9.7944;
  o->initial_trust_region_radius = 1e4; o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8; o->min_relative_decrease = 1e-3; o->jacobi_scaling = 1; o->marg_eps = 1e-8;
  o->marg_sqrt = 0;   // the reference's eigen-decomposition square root (1: the product's pivoted LDL^T, for bench.py's like-for-like CPU leg)
  o->use_graph = 0;   // (device options; meaningless on the CPU)
  o->split_batch = 1;
  o->max_solver_time_in_seconds = 0.0; o->host_threads = 0;
  o->solve_kernel = 0; o->test_fail_chol_iter = 0; o->test_fail_chol_count = 1; o->sharded_mu_retries = 1;
}

int32_t gfo_sqrt_info(const double *cov, double *out, int32_t n) { return sqrt_info_from_cov(cov, out, n) ? 0 : 1; }
void gfo_sym_eig(const double *A, int32_t n, double *w, double *V) { sym_eig(A, n, w, V); }

int32_t gfo_preintegrate_imu(int32_t n, const int32_t *off, const double *samples, const double *first, const double *lin,
                             const double noise[4], gfbe_imu_preint *out) {
  for (int k = 0; k < n; k++) preintegrate_imu(off[k + 1] - off[k], samples + 7 * (size_t)off[k], first + 6 * k, lin + 6 * k, noise, out + k);
  return 0;
}
int32_t gfo_preintegrate_wheel(int32_t n, const int32_t *off, const double *samples, const double *first, const double *lin,
                               const double noise[2], gfbe_wheel_preint *out) {
  for (int k = 0; k < n; k++) preintegrate_wheel(off[k + 1] - off[k], samples + 7 * (size_t)off[k], first + 6 * k, lin + 4 * k, noise, out + k);
  return 0;
}

int32_t gfo_eval_factors(const gfbe_options *opt, const gfbe_window *w, int32_t robust, double *vis_r, double *vis_J,
                         double *imu_r, double *imu_J, double *wheel_r, double *wheel_J, double *prior_r, double *cost) {
  Problem P;
  if (!setup(P, w, opt)) return GFBE_BAD_INPUT;
  const gfbe_state &st = w->state;
  if (vis_r) {
    const gfbe_visual &v = w->vis;
    for (int k = 0; k < v.n_factor; k++) {
      double r[2], J[40];
      eval_visual(st.para_Pose[v.imu_i[k]], st.para_Pose[v.imu_j[k]], st.para_Ex_Pose, w->para_Feature[v.feature_index[k]], st.para_Td,
                  v.pts_i + 3 * k, v.pts_j + 3 * k, v.vel_i + 2 * k, v.vel_j + 2 * k, v.td_i[k], v.td_j[k], opt->vis_sqrt_info, r, J);
      if (robust) robustify(r, J, 2, 20, opt->huber_delta);
      std::memcpy(vis_r + 2 * (size_t)k, r, sizeof r);
      if (vis_J) std::memcpy(vis_J + 40 * (size_t)k, J, sizeof J);
    }
  }
  if (imu_r)
    for (int k = 0; k < w->n_imu; k++) {
      const int i = w->imu_frame[k];
      eval_imu(w->imu[k], &P.imu_sqrt[(size_t)k * 225], opt->g_norm, st.para_Pose[i], st.para_SpeedBias[i], st.para_Pose[i + 1], st.para_SpeedBias[i + 1],
               imu_r + 15 * k, imu_J ? imu_J + 450 * k : nullptr);
    }
  if (wheel_r)
    for (int k = 0; k < w->n_wheel; k++) {
      const int i = w->wheel_frame[k];
      eval_wheel(w->wheel[k], &P.wheel_sqrt[(size_t)k * 36], st.para_Pose[i], st.para_Pose[i + 1], st.para_Ex_Pose_wheel,
                 st.para_Ix_wheel[0], st.para_Ix_wheel[1], st.para_Ix_wheel[2], st.para_Td_wheel, wheel_r + 6 * k, wheel_J ? wheel_J + 132 * k : nullptr);
    }
  if (prior_r && P.has_prior) eval_prior(*w->prior, st, prior_r);
  if (cost) *cost = evaluate(P, st, w->para_Feature, nullptr);
  return GFBE_OK;
}

int32_t gfo_linearize(const gfbe_options *opt, const gfbe_window *w, double *H, double *g, double *Hll, double *gl, double *Hpl, double *cost) {
  Problem P;
  if (!setup(P, w, opt)) return GFBE_BAD_INPUT;
  Lin lin;
  evaluate(P, w->state, w->para_Feature, &lin);
  if (H) std::memcpy(H, lin.H.data(), sizeof(double) * ND * ND);
  if (g) std::memcpy(g, lin.g.data(), sizeof(double) * ND);
  if (Hll) std::memcpy(Hll, lin.Hll.data(), sizeof(double) * P.L);
  if (gl) std::memcpy(gl, lin.gl.data(), sizeof(double) * P.L);
  if (Hpl) std::memcpy(Hpl, lin.Hpl.data(), sizeof(double) * (size_t)P.L * NV);
  if (cost) *cost = lin.cost;
  return GFBE_OK;
}

int32_t gfo_reanchor(const gfbe_state *before, const gfbe_state *after, gfbe_state *out) {
  reanchor(*before, *after, GFBE_WINDOW_SIZE, *out);
  return 0;
}

int32_t gfo_marginalize(const gfbe_options *opt, const gfbe_window *w, int32_t flag, gfbe_prior *out, double *A_out, double *b_out) {
  Problem P;
  if (!setup(P, w, opt)) return GFBE_BAD_INPUT;
  return marginalize(P, w->state, w->para_Feature, flag, out, A_out, b_out);
}

int32_t gfo_solve_window(const gfbe_options *opt, const gfbe_window *w, int32_t margin_flag, gfbe_state *out_state,
                         double *out_feature, gfbe_prior *prior_out, gfbe_summary *summary) {
  Problem P;
  if (!setup(P, w, opt)) return GFBE_BAD_INPUT;
  Solution sol;
  solve(P, sol);
  if (sol.sum.status == GFBE_NUMERICAL_FAILURE) { if (summary) *summary = sol.sum; return GFBE_NUMERICAL_FAILURE; }
  if (std::isfinite(sol.x.gnss.yaw_enu_local) && std::fabs(sol.x.gnss.yaw_enu_local) < 1e6) {   // estimator.cpp:3383-3386 (absurd values left alone)
    while (sol.x.gnss.yaw_enu_local > M_PI) sol.x.gnss.yaw_enu_local -= 2.0 * M_PI;
    while (sol.x.gnss.yaw_enu_local < -M_PI) sol.x.gnss.yaw_enu_local += 2.0 * M_PI;
  }
  gfbe_state anchored;
  reanchor(w->state, sol.x, w->frame_count, anchored);
  if (margin_flag != GFBE_MARGIN_NONE && prior_out && w->frame_count == GFBE_WINDOW_SIZE) {
    const int rc = marginalize(P, anchored, sol.lam.data(), margin_flag, prior_out, nullptr, nullptr);
    if (rc == 1 && P.has_prior) {      // estimator.cpp:3600-3601 not satisfied: last_marginalization_info stays
      const gfbe_prior &pr = *w->prior;
      prior_out->valid = pr.valid; prior_out->n = pr.n; prior_out->n_blocks = pr.n_blocks;
      int xo = 0;
      for (int q = 0; q < pr.n_blocks; q++) { prior_out->block_id[q] = pr.block_id[q]; prior_out->block_size[q] = pr.block_size[q]; prior_out->block_idx[q] = pr.block_idx[q]; xo += pr.block_size[q]; }
      std::memcpy(prior_out->x0, pr.x0, sizeof(double) * xo);
      std::memcpy(prior_out->J0, pr.J0, sizeof(double) * pr.n * pr.n);
      std::memcpy(prior_out->r0, pr.r0, sizeof(double) * pr.n);
    }
  }
  *out_state = anchored;
  if (out_feature) std::memcpy(out_feature, sol.lam.data(), sizeof(double) * P.L);
  if (summary) *summary = sol.sum;
  return sol.sum.status;
}

}  // extern "C"

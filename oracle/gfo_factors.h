// oracle/gfo_factors.h — TEST INFRASTRUCTURE ONLY (CPU oracle).
// Restates the four residual models of the hot path with analytic TANGENT-space Jacobians
// (the first 6 columns of each 7-wide global pose block, pose_local_parameterization.cpp:27-34):
//   visual  VE/factor/projectionTwoFrameOneCamFactor.cpp:43-151
//   IMU     VE/factor/imu_factor.h:28-191 + integration_base.h:169-195
//   wheel   VE/factor/wheel_factor.h:28-247 + wheel_integration_base.h:180-219
//   prior   VE/factor/marginalization_factor.cpp:344-392
//   loss corrector  VE/factor/marginalization_factor.cpp:46-77 (Ceres' Corrector, HuberLoss)
#pragma once
#include "../include/gfbe.h"
#include "gfo_math.h"

namespace gfo {

// r[2]; J[2][20] row-major, columns pose_i(6) pose_j(6) ex(6) lambda(1) td(1). J may be null.
void eval_visual(const double *pose_i, const double *pose_j, const double *ex, double inv_dep, double td,
                 const double *pts_i, const double *pts_j, const double *vel_i, const double *vel_j,
                 double td_i, double td_j, double sqrt_info, double *r, double *J);

// r[15]; J[15][30] columns pose_i(6) sb_i(9) pose_j(6) sb_j(9). sqrt_info 15x15 row-major (precomputed).
void eval_imu(const gfbe_imu_preint &pre, const double *sqrt_info, double g_norm,
              const double *pose_i, const double *sb_i, const double *pose_j, const double *sb_j,
              double *r, double *J);

// r[6]; J[6][22] columns pose_i(6) pose_j(6) ex_wheel(6) sx sy sw td_wheel.
void eval_wheel(const gfbe_wheel_preint &pre, const double *sqrt_info,
                const double *pose_i, const double *pose_j, const double *ex_wheel,
                double sx, double sy, double sw, double td, double *r, double *J);

// dx (tangent, n) of the prior's kept blocks w.r.t. x0, then r = r0 + J0 dx.
void prior_dx(const gfbe_prior &pr, const gfbe_state &st, double *dx);
void eval_prior(const gfbe_prior &pr, const gfbe_state &st, double *r);

// Huber rho (ceres::HuberLoss::Evaluate) and the Corrector scaling.
void huber(double s, double delta, double rho[3]);
// Applies the corrector in place to r[nr], J[nr][nc]; returns 0.5*rho(s) (the block's cost).
double robustify(double *r, double *J, int nr, int nc, double delta);

// Pointer to the global block storage of GFBE_BLK_* id inside a state.
const double *block_ptr(const gfbe_state &st, int id);
double *block_ptr(gfbe_state &st, int id);
int block_global_size(int id);
int block_local_size(int id);

// GnssPsrDoppFactor::Evaluate for one observation (gfo_gnss.cpp). r[2]; J[2][18] or null: columns P_i(3) V_i(3) P_j(3) V_j(3)
// rcv_dt rcv_ddt yaw_enu_local anc_ecef(3).
void eval_gnss_psr_dopp(const gfbe_gnss_obs &o, const double *iono, const double *pi, const double *vi, const double *pj, const double *vj,
                        double rcv_dt, double rcv_ddt, double yaw, const double *anc, double *r, double *J);

}  // namespace gfo

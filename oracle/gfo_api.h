// oracle/gfo_api.h — TEST INFRASTRUCTURE ONLY.
//
// CPU oracle of the sliding-window back end: a dependency-free FP64 restatement of what
// Estimator::optimization() computes (Ground-Fusion++/vins_estimator/src/estimator/estimator.cpp:2951-3698)
// including the parts that live in Ceres Solver 1.14 (NOT vendored in the reference: README.md:66-72).
//
// *** PARITY UNPINNED ***  The reference holds no golden vectors or known-answer tests for this
// path and cannot be compiled here (no Eigen/Ceres/ROS). This oracle is pinned
// by: central-difference Jacobian checks, an independent numpy re-derivation of every residual,
// algebraic invariants of the marginalisation, and convergence to ground truth on noise-free
// synthetic windows (tests/test_oracle_*.py; fixtures in tests/golden/).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
// It shares nothing with the product except the public C header include/gfbe.h (struct layouts).
#pragma once
#include "../include/gfbe.h"

namespace gfo {
void preintegrate_imu(int n_samples, const double *samples, const double *first_acc_gyr,
                      const double *lin_ba_bg, const double noise4[4], gfbe_imu_preint *out);
void preintegrate_wheel(int n_samples, const double *samples, const double *first_vel_gyr,
                        const double *lin_s_td, const double noise2[2], gfbe_wheel_preint *out);
}

extern "C" {
// Same signatures as the gfbe_* product entry points, CPU implementation.
void gfo_default_options(gfbe_options *opt);
int32_t gfo_feature_count(const gfbe_feature_list *fl);
int32_t gfo_visual_factor_count(const gfbe_feature_list *fl, int32_t only_start_frame0);
int32_t gfo_build_visual_factors(const gfbe_feature_list *fl, int32_t only_start_frame0,
                                 int32_t *feature_index, int32_t *imu_i, int32_t *imu_j,
                                 double *pts_i, double *pts_j, double *vel_i, double *vel_j,
                                 double *td_i, double *td_j, double *para_Feature, uint8_t *feature_const);
void gfo_set_depth(const gfbe_feature_list *fl, const double *para_Feature, double *estimated_depth,
                   int32_t *solve_flag);
int32_t gfo_eval_factors(const gfbe_options *opt, const gfbe_window *win, int32_t robustify,
                         double *vis_r, double *vis_J, double *imu_r, double *imu_J,
                         double *wheel_r, double *wheel_J, double *prior_r, double *cost);
int32_t gfo_preintegrate_imu(int32_t n_interval, const int32_t *offset, const double *samples,
                             const double *first_acc_gyr, const double *lin_ba_bg, const double noise[4],
                             gfbe_imu_preint *out);
int32_t gfo_preintegrate_wheel(int32_t n_interval, const int32_t *offset, const double *samples,
                               const double *first_vel_gyr, const double *lin_sx_sy_sw_td,
                               const double noise[2], gfbe_wheel_preint *out);
int32_t gfo_solve_window(const gfbe_options *opt, const gfbe_window *win, int32_t margin_flag,
                         gfbe_state *out_state, double *out_feature, gfbe_prior *prior_out,
                         gfbe_summary *summary);
// Pieces, exposed for the invariant tests:
//   state <- double2vector(vector) re-anchoring only (estimator.cpp:2501-2555 + 2341-2348)
int32_t gfo_reanchor(const gfbe_state *before_solve, const gfbe_state *after_solve, gfbe_state *out);
//   marginalisation only, at the given (already re-anchored) state. Also returns, when non-NULL,
//   the Schur-complemented information A'(n*n), b'(n) before the eigen square root.
int32_t gfo_marginalize(const gfbe_options *opt, const gfbe_window *win, int32_t margin_flag,
                        gfbe_prior *prior_out, double *A_out, double *b_out);
//   full normal equations of one linearisation in the oracle's tangent layout:
//   H (182*182), g (182), per-landmark Hll[L], gl[L], Hpl[L*73] (visual dims 0..72), cost.
int32_t gfo_linearize(const gfbe_options *opt, const gfbe_window *win, double *H, double *g,
                      double *Hll, double *gl, double *Hpl, double *cost);
//   Eigen-stand-in helpers for tests
int32_t gfo_sqrt_info(const double *cov, double *out, int32_t n);
void gfo_sym_eig(const double *A, int32_t n, double *w, double *V);
}

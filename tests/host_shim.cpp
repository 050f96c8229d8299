// tests/host_shim.cpp — TEST HARNESS ONLY. Compiles the product's __host__ __device__ factor
// arithmetic (ground-fusion2_amd/csrc/gfbe_factors.h) for the HOST so that `-m "not gpu"` tests
// can pin it against the oracle in a container without a GPU. Never loaded by the package: the
// product path calls these functions only from HIP kernels.
#include "../ground-fusion2_amd/csrc/gfbe_factors.h"
#include <cstring>
#include <vector>

using namespace gfd;

extern "C" int shim_eval_factors(const gfbe_options *opt, const gfbe_window *w, int robust, double *vis_r, double *vis_J,
                                 double *imu_r, double *imu_J, double *wheel_r, double *wheel_J, double *prior_r) {
  const gfbe_state &st = w->state;
  const PoseRT Ex = make_pose(st.para_Ex_Pose);
  const gfbe_visual &v = w->vis;
  for (int k = 0; k < v.n_factor; k++) {
    const PoseRT Fi = make_pose(st.para_Pose[v.imu_i[k]]), Fj = make_pose(st.para_Pose[v.imu_j[k]]);
    double r[2], Ji[12], Jj[12], Je[12], Jl[2], Jt[2];
    const PairConst pc = make_pair_const(Fi, Fj, Ex);      // the form the kernels use
    visual_eval_pc<true>(pc, w->para_Feature[v.feature_index[k]], st.para_Td, v.pts_i[3 * k], v.pts_i[3 * k + 1],
                         v.pts_i[3 * k + 2], v.pts_j[3 * k], v.pts_j[3 * k + 1], v.vel_i[2 * k], v.vel_i[2 * k + 1],
                         v.vel_j[2 * k], v.vel_j[2 * k + 1], v.td_i[k], v.td_j[k], opt->vis_sqrt_info, r, Ji, Jj, Je, Jl, Jt);
    {   // and the direct restatement of the reference's formulas must agree with it
      double r2[2], Ji2[12], Jj2[12], Je2[12], Jl2[2], Jt2[2];
      visual_eval<true>(Fi, Fj, Ex, w->para_Feature[v.feature_index[k]], st.para_Td, v.pts_i[3 * k], v.pts_i[3 * k + 1],
                        v.pts_i[3 * k + 2], v.pts_j[3 * k], v.pts_j[3 * k + 1], v.vel_i[2 * k], v.vel_i[2 * k + 1],
                        v.vel_j[2 * k], v.vel_j[2 * k + 1], v.td_i[k], v.td_j[k], opt->vis_sqrt_info, r2, Ji2, Jj2, Je2, Jl2, Jt2);
      double scale = 1.0, err = 0.0;
      for (int q = 0; q < 12; q++) { scale = fmax(scale, fabs(Ji2[q])); scale = fmax(scale, fabs(Je2[q])); }
      for (int q = 0; q < 12; q++) { err = fmax(err, fabs(Ji[q] - Ji2[q])); err = fmax(err, fabs(Jj[q] - Jj2[q])); err = fmax(err, fabs(Je[q] - Je2[q])); }
      for (int q = 0; q < 2; q++) { err = fmax(err, fabs(Jl[q] - Jl2[q]) / fmax(1.0, fabs(Jl2[q]))); err = fmax(err, fabs(Jt[q] - Jt2[q])); err = fmax(err, fabs(r[q] - r2[q])); }
      if (err > 1e-11 * scale) return 2;
    }
    if (robust) {
      double s1, rs, asn;
      corrector(r[0] * r[0] + r[1] * r[1], opt->huber_delta, &s1, &rs, &asn);
      correct_cols(Ji, Ji + 6, 6, r[0], r[1], s1, asn);
      correct_cols(Jj, Jj + 6, 6, r[0], r[1], s1, asn);
      correct_cols(Je, Je + 6, 6, r[0], r[1], s1, asn);
      correct_cols(Jl, Jl + 1, 1, r[0], r[1], s1, asn);
      correct_cols(Jt, Jt + 1, 1, r[0], r[1], s1, asn);
      r[0] *= rs; r[1] *= rs;
    }
    {   // the fused form the linearisation kernels run (visual_lin: corrector folded into `reduce`, explicit FMAs) against the
        // block-by-block form above, and its reduced variant (constant extrinsic / td) against the full one, bit for bit
      double r3[2], Ji3[12], Jj3[12], Je3[12], Jl3[2], Jt3[2], r4[2], Ji4[12], Jj4[12], Je4[1], Jl4[2], Jt4[2];
      const double delta = robust ? opt->huber_delta : 1e150;
      const double c3 = visual_lin<true, true>(pc, w->para_Feature[v.feature_index[k]], st.para_Td, v.pts_i[3 * k], v.pts_i[3 * k + 1],
                                               v.pts_i[3 * k + 2], v.pts_j[3 * k], v.pts_j[3 * k + 1], v.vel_i[2 * k], v.vel_i[2 * k + 1],
                                               v.vel_j[2 * k], v.vel_j[2 * k + 1], v.td_i[k], v.td_j[k], opt->vis_sqrt_info, delta, r3, Ji3, Jj3, Je3, Jl3, Jt3);
      const double c4 = visual_lin<true, false>(pc, w->para_Feature[v.feature_index[k]], st.para_Td, v.pts_i[3 * k], v.pts_i[3 * k + 1],
                                                v.pts_i[3 * k + 2], v.pts_j[3 * k], v.pts_j[3 * k + 1], v.vel_i[2 * k], v.vel_i[2 * k + 1],
                                                v.vel_j[2 * k], v.vel_j[2 * k + 1], v.td_i[k], v.td_j[k], opt->vis_sqrt_info, delta, r4, Ji4, Jj4, Je4, Jl4, Jt4);
      double rc[2], cc;
      cc = visual_lin<false, true>(pc, w->para_Feature[v.feature_index[k]], st.para_Td, v.pts_i[3 * k], v.pts_i[3 * k + 1],
                                   v.pts_i[3 * k + 2], v.pts_j[3 * k], v.pts_j[3 * k + 1], v.vel_i[2 * k], v.vel_i[2 * k + 1],
                                   v.vel_j[2 * k], v.vel_j[2 * k + 1], v.td_i[k], v.td_j[k], opt->vis_sqrt_info, delta, rc, nullptr, nullptr, nullptr, nullptr, nullptr);
      if (c3 != c4 || c3 != cc) return 3;                       // the candidate-cost pass sees the same cost as the linearisation
      double scale = 1.0, err = 0.0;
      for (int q = 0; q < 12; q++) { scale = fmax(scale, fabs(Ji[q])); scale = fmax(scale, fabs(Je[q])); }
      for (int q = 0; q < 12; q++) {
        err = fmax(err, fabs(Ji[q] - Ji3[q])); err = fmax(err, fabs(Jj[q] - Jj3[q])); err = fmax(err, fabs(Je[q] - Je3[q]));
        if (Ji3[q] != Ji4[q] || Jj3[q] != Jj4[q]) return 4;
      }
      for (int q = 0; q < 2; q++) {
        err = fmax(err, fabs(Jl[q] - Jl3[q]) / fmax(1.0, fabs(Jl[q]))); err = fmax(err, fabs(Jt[q] - Jt3[q])); err = fmax(err, fabs(r[q] - r3[q]));
        if (Jl3[q] != Jl4[q] || r3[q] != r4[q]) return 4;
      }
      if (err > 1e-11 * scale) return 5;
      // what leaves the shim is the kernels' form
      for (int q = 0; q < 12; q++) { Ji[q] = Ji3[q]; Jj[q] = Jj3[q]; Je[q] = Je3[q]; }
      for (int q = 0; q < 2; q++) { Jl[q] = Jl3[q]; Jt[q] = Jt3[q]; r[q] = r3[q]; }
    }
    vis_r[2 * k] = r[0]; vis_r[2 * k + 1] = r[1];
    double *J = vis_J + 40 * (size_t)k;
    for (int row = 0; row < 2; row++) {
      for (int c = 0; c < 6; c++) { J[row * 20 + c] = Ji[row * 6 + c]; J[row * 20 + 6 + c] = Jj[row * 6 + c]; J[row * 20 + 12 + c] = Je[row * 6 + c]; }
      J[row * 20 + 18] = Jl[row]; J[row * 20 + 19] = Jt[row];
    }
  }
  for (int k = 0; k < w->n_imu; k++) {
    const int i = w->imu_frame[k];
    double S[225], work[450], raw[15], Jraw[450];
    if (sqrt_info_from_cov(w->imu[k].covariance, 15, S, work)) return 1;
    std::memset(Jraw, 0, sizeof Jraw);
    imu_raw(&w->imu[k], opt->g_norm, st.para_Pose[i], st.para_SpeedBias[i], st.para_Pose[i + 1], st.para_SpeedBias[i + 1], raw, Jraw);
    for (int a = 0; a < 15; a++) {
      double s = 0; for (int b = 0; b < 15; b++) s += S[a * 15 + b] * raw[b];
      imu_r[15 * k + a] = s;
      for (int c = 0; c < 30; c++) { double t = 0; for (int b = 0; b < 15; b++) t += S[a * 15 + b] * Jraw[b * 30 + c]; imu_J[450 * k + a * 30 + c] = t; }
    }
  }
  for (int k = 0; k < w->n_wheel; k++) {
    const int i = w->wheel_frame[k];
    double S[36], work[72], raw[6], Jraw[132];
    if (sqrt_info_from_cov(w->wheel[k].covariance, 6, S, work)) return 1;
    std::memset(Jraw, 0, sizeof Jraw);
    wheel_raw(&w->wheel[k], st.para_Pose[i], st.para_Pose[i + 1], st.para_Ex_Pose_wheel, st.para_Ix_wheel[0], st.para_Ix_wheel[1],
              st.para_Ix_wheel[2], st.para_Td_wheel, raw, Jraw);
    for (int a = 0; a < 6; a++) {
      double s = 0; for (int b = 0; b < 6; b++) s += S[a * 6 + b] * raw[b];
      wheel_r[6 * k + a] = s;
      for (int c = 0; c < 22; c++) { double t = 0; for (int b = 0; b < 6; b++) t += S[a * 6 + b] * Jraw[b * 22 + c]; wheel_J[132 * k + a * 22 + c] = t; }
    }
  }
  (void)prior_r;
  return 0;
}

// the GNSS factor arithmetic of csrc/gfbe_gnss.h, the loop k_gnss runs one thread per factor
#include "../ground-fusion2_amd/csrc/gfbe_gnss.h"
extern "C" int shim_gnss_eval(int n_obs, const gfbe_gnss_obs *obs, const double *iono, const gfbe_state *st, const gfbe_gnss_state *g,
                              const double *frame_dt, double ddt_weight, double *r_obs, double *J_obs, double *r_dt_ddt, double *r_smooth) {
  for (int k = 0; k < n_obs; k++) {
    const gfbe_gnss_obs &o = obs[k];
    gnss_psr_dopp_eval(o, iono, st->para_Pose[o.lower_idx], st->para_SpeedBias[o.lower_idx], st->para_Pose[o.lower_idx + 1],
                       st->para_SpeedBias[o.lower_idx + 1], g->rcv_dt[o.frame][o.sys_idx], g->rcv_ddt[o.frame], g->yaw_enu_local, g->anc_ecef,
                       r_obs + 2 * k, J_obs + 36 * k);
  }
  for (int sys = 0; sys < 4; sys++)
    for (int i = 0; i < GFBE_WINDOW_SIZE; i++)
      r_dt_ddt[sys * GFBE_WINDOW_SIZE + i] = gnss_dt_ddt_res(g->rcv_dt[i][sys], g->rcv_dt[i + 1][sys], g->rcv_ddt[i], g->rcv_ddt[i + 1], frame_dt[i]);
  for (int i = 0; i < GFBE_WINDOW_SIZE; i++) r_smooth[i] = gnss_ddt_smooth_res(g->rcv_ddt[i], g->rcv_ddt[i + 1], ddt_weight);
  return 0;
}

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
      {   // the form the throughput kernels sum for a window with constant extrinsic and td (round 4: visual_lin_y) — every Jacobian
          // block as G times a pair / landmark constant: [J_i J_j] = [G | G [e]x] T, the landmark row from d = G^T w
        const PoseRT F0 = make_pose(st.para_Pose[0]);
        const FrameConst fc = make_frame_const(Fi, Ex, F0);
        const double lamv = w->para_Feature[v.feature_index[k]], inv_l = 1.0 / lamv, dti = st.para_Td - v.td_i[k];
        const double cx = __builtin_fma(-dti, v.vel_i[2 * k], v.pts_i[3 * k]) * inv_l, cy = __builtin_fma(-dti, v.vel_i[2 * k + 1], v.pts_i[3 * k + 1]) * inv_l,
                     cz = v.pts_i[3 * k + 2] * inv_l;
        vec3 f, e;
        for (int a = 0; a < 3; a++) { f[a] = __builtin_fma(fc.W(a, 0), cx, __builtin_fma(fc.W(a, 1), cy, fc.W(a, 2) * cz)); e[a] = f[a] + fc.wt[a]; }
        double r5[2], g0[3], g1[3], Jl5[2];
        const double c5 = visual_lin_y(pc, cx, cy, cz, f, inv_l, st.para_Td, v.pts_j[3 * k], v.pts_j[3 * k + 1], v.vel_j[2 * k], v.vel_j[2 * k + 1], v.td_j[k],
                                       opt->vis_sqrt_info, delta, r5, g0, g1, Jl5);
        if (c5 != c3 || r5[0] != r3[0] || r5[1] != r3[1]) return 6;          // the same cost and residual, bit for bit
        // Y = [G | G [x]x], x = P_w - P_0;  J_i = Y T_i, J_j = -Y T_j with T_f = [ I  [P_f - P_0]x R_f ; 0  -R_f ]
        const vec3 x = add(e, fc.dPc);
        const vec3 y0 = cross3(mk3(g0[0], g0[1], g0[2]), x), y1 = cross3(mk3(g1[0], g1[1], g1[2]), x);
        const mat3 DRi = hat_mul(fc.dPc, fc.R), DRj = hat_mul(sub(Fj.t, F0.t), pc.Rj);
        double e2 = 0.0;
        for (int c = 0; c < 3; c++) {
          const double gi[2] = {g0[c], g1[c]};
          for (int h = 0; h < 2; h++) {
            const vec3 &yh = h ? y1 : y0;
            const double *gh = h ? g1 : g0;
            e2 = fmax(e2, fabs(gi[h] - Ji3[6 * h + c]));                                                   // d/dP_i = G
            e2 = fmax(e2, fabs(-gi[h] - Jj3[6 * h + c]));                                                  // d/dP_j = -G
            const double ji = gh[0] * DRi(0, c) + gh[1] * DRi(1, c) + gh[2] * DRi(2, c) - (yh[0] * fc.R(0, c) + yh[1] * fc.R(1, c) + yh[2] * fc.R(2, c));
            const double jj = -(gh[0] * DRj(0, c) + gh[1] * DRj(1, c) + gh[2] * DRj(2, c)) + yh[0] * pc.Rj(0, c) + yh[1] * pc.Rj(1, c) + yh[2] * pc.Rj(2, c);
            e2 = fmax(e2, fabs(ji - Ji3[6 * h + 3 + c]));
            e2 = fmax(e2, fabs(jj - Jj3[6 * h + 3 + c]));
          }
        }
        for (int q = 0; q < 2; q++) e2 = fmax(e2, fabs(Jl5[q] - Jl3[q]) / fmax(1.0, fabs(Jl3[q])) * scale);
        // landmark row: pose i [ d ; Ri^T (e x d) ], pose j [ -d ; Rj^T (d x (e + P_i - P_j)) ] against J^T w of the blocks
        vec3 dv;
        for (int q = 0; q < 3; q++) dv[q] = __builtin_fma(g0[q], Jl5[0], g1[q] * Jl5[1]);
        const vec3 ti = cross3(e, dv), tj = cross3(dv, add(e, pc.dP));
        double hscale = 1.0;
        for (int q = 0; q < 6; q++) hscale = fmax(hscale, fmax(fabs(Ji3[q] * Jl3[0] + Ji3[6 + q] * Jl3[1]), fabs(Jj3[q] * Jl3[0] + Jj3[6 + q] * Jl3[1])));
        for (int q = 0; q < 3; q++) {
          const double hci = fc.R(0, q) * ti[0] + fc.R(1, q) * ti[1] + fc.R(2, q) * ti[2], hpj = pc.Rj(0, q) * tj[0] + pc.Rj(1, q) * tj[1] + pc.Rj(2, q) * tj[2];
          if (fabs(dv[q] - (Ji3[q] * Jl3[0] + Ji3[6 + q] * Jl3[1])) > 1e-10 * hscale) return 7;
          if (fabs(hci - (Ji3[3 + q] * Jl3[0] + Ji3[9 + q] * Jl3[1])) > 1e-10 * hscale) return 7;
          if (fabs(-dv[q] - (Jj3[q] * Jl3[0] + Jj3[6 + q] * Jl3[1])) > 1e-10 * hscale) return 7;
          if (fabs(hpj - (Jj3[3 + q] * Jl3[0] + Jj3[9 + q] * Jl3[1])) > 1e-10 * hscale) return 7;
        }
        if (e2 > 1e-10 * scale) return 8;
      }
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
    for (int nparts = 2; nparts <= 8; nparts *= 2) {   // the Jacobian's items dealt over the waves of a workgroup: the same bits
      double raw2[15], Jsplit[450];
      std::memset(Jsplit, 0, sizeof Jsplit);
      for (int part = nparts - 1; part >= 0; part--)
        imu_raw(&w->imu[k], opt->g_norm, st.para_Pose[i], st.para_SpeedBias[i], st.para_Pose[i + 1], st.para_SpeedBias[i + 1], raw2, Jsplit,
                1, part, nparts);
      if (std::memcmp(Jsplit, Jraw, sizeof Jraw) || std::memcmp(raw2, raw, sizeof raw)) return 9;
    }
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
    for (int nparts = 2; nparts <= 8; nparts *= 2) {   // the Jacobian dealt over the waves of a workgroup (dense_body<true>): the same bits
      double raw2[6], Jsplit[132];
      std::memset(Jsplit, 0, sizeof Jsplit);
      for (int part = nparts - 1; part >= 0; part--)
        wheel_raw(&w->wheel[k], st.para_Pose[i], st.para_Pose[i + 1], st.para_Ex_Pose_wheel, st.para_Ix_wheel[0], st.para_Ix_wheel[1],
                  st.para_Ix_wheel[2], st.para_Td_wheel, raw2, Jsplit, 1, part, nparts);
      if (std::memcmp(Jsplit, Jraw, sizeof Jraw) || std::memcmp(raw2, raw, sizeof raw)) return 9;
    }
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

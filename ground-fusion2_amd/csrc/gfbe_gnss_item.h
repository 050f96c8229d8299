// gfbe_gnss_item.h — the GNSS factors of a window at the CANDIDATE of a trust-region iteration as one work item (round 6): k_gnss's
// mode 1 (the candidate's cost) and its sub 1 (the speculative pass: J, r and the cost at the candidate into the other set of outputs),
// statement for statement, callable from any workgroup of at least 64 threads. k_lin_small runs it as one more workgroup beside the
// visual tiles and the inertial / wheel / prior items of a small batch's candidate pass, so that a single GNSS window no longer pays
// a launch of its own for it (~25 us per iteration) and keeps the accepting tail of that launch (k_accept: ~7 us).
//   GnssPsrDoppFactor / DtDdtFactor / DdtSmoothFactor as estimator.cpp:3239-3291 adds them; the arithmetic is gfbe_gnss.h.
// The block sum below is gfbe_devutil.h's: per-wave shuffle trees, the waves in order — with at most 256 observations every thread
// holds at most one, the waves a 512-thread workgroup would have beyond the fourth add exact zeros, and the cost has k_gnss's bits
// (launch_lin_small only takes the item for windows of up to GN_ITEM_MAX_OBS observations).
#pragma once
#include "gfbe_devutil.h"
#include "gfbe_gnss.h"

namespace gfd {

enum { GN_ITEM_MAX_OBS = 256, GN_ITEM_NCLK = 5 * GFBE_WINDOW_SIZE };

// d: the view the pass writes (k_lin_small: lin_view of the other set for the speculative pass, the batch itself for the cost pass);
// linearise: evaluate J too and keep J, r (sub 1) — or the cost alone into slot 1 (mode 1)
__device__ __forceinline__ void gnss_candidate_item(const BatchDev &d, const int w, const bool linearise, double *red /* >= 16 doubles of LDS */) {
  const WinDesc &ds = d.desc[w];
  const WinCtl &c = d.ctl[w];
  if (!ds.gnss_ready || !ds.gnss_factors || c.done || !c.have_step) return;      // (workgroup-uniform)
  const int t = threadIdx.x, nth = blockDim.x;
  const double *X = d.x + ((size_t)w * 2 + (1 - c.cur)) * NA;
  const gfbe_gnss_obs *obs = d.gnss_obs + ds.gnss_off;
  double *Jw = d.gnss_J + (size_t)ds.gnss_off * 36, *rw = d.gnss_r + (size_t)ds.gnss_off * 2;
  const int n_obs = ds.n_gnss;
  double cost = 0.0;
  for (int k = t; k < n_obs; k += nth) {
    const gfbe_gnss_obs o = obs[k];
    const int lw = o.lower_idx;
    double r[2], J[36];
    gnss_psr_dopp_eval(o, ds.gnss_has_iono ? ds.gnss_iono : nullptr, X + A_POSE(lw), X + A_SB(lw), X + A_POSE(lw + 1), X + A_SB(lw + 1),
                       X[A_DT + 4 * o.frame + o.sys_idx], X[A_DDT + o.frame], X[A_YAW], X + A_ANC, r, linearise ? J : nullptr);
    cost += 0.5 * r[0] * r[0] + 0.5 * r[1] * r[1];
    if (linearise) {
      rw[2 * k] = r[0]; rw[2 * k + 1] = r[1];
      for (int q = 0; q < 36; q++) Jw[(size_t)36 * k + q] = J[q];
    }
  }
  if (t < GN_ITEM_NCLK) {
    double r = 0.0;
    if (t < 4 * GFBE_WINDOW_SIZE) {
      const int k = t / GFBE_WINDOW_SIZE, i = t % GFBE_WINDOW_SIZE;
      r = gnss_dt_ddt_res(X[A_DT + 4 * i + k], X[A_DT + 4 * (i + 1) + k], X[A_DDT + i], X[A_DDT + i + 1], ds.gnss_frame_dt[i]);
    } else {
      const int i = t - 4 * GFBE_WINDOW_SIZE;
      r = gnss_ddt_smooth_res(X[A_DDT + i], X[A_DDT + i + 1], ds.gnss_ddt_weight);
    }
    cost += 0.5 * r * r;
  }
  cost = block_sum(cost, red);
  if (t == 0) d.gnss_cost[(size_t)w * 2 + (linearise ? 0 : 1)] = cost;
}

}  // namespace gfd

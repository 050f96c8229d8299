// gfbe_gnss_solve.hip — the GNSS factors inside the window solve and the marginalisation.
//   problem set-up        Ground-Fusion++/vins_estimator/src/estimator/estimator.cpp:2965-3002 (blocks, the lowspeed gate)
//   residual blocks       estimator.cpp:3239-3291 (GnssPsrDoppFactor per observation, DtDdtFactor / DdtSmoothFactor chains)
//   marginalisation set   estimator.cpp:3459-3496 (the factors of frame 0, drop sets {0, 1, 4, 5}, {0, 2}, {0})
// The observations are evaluated one per thread (the arithmetic is gfbe_gnss.h, shared with gfbe_gnss_eval) and their 2 x 18
// Jacobians staged in LDS. Every entry of the normal equations the GNSS factors reach is then summed by ONE owner thread over the
// observations of the frames that can hold both dims, in observation order: no atomics, the same bits on every run and for every
// shape of the launch. A window's entries are dealt over gridDim.y workgroups (round 6: one window alone on the device has 255 idle
// CUs beside it — every workgroup stages the window's Jacobians and owns a sixteenth of the 7874 entries; rounds 3-5 summed 20 x 20
// cell partials first, a wave per cell, and then the entries over the cells, all on one CU: 52 us per launch). The clock factors are
// linear with constant Jacobians and are added in closed form. (A window with more observations than fit in LDS sums from the global
// copy of the Jacobians, the same operations in the same order.)
#include "gfbe_devutil.h"
#include <algorithm>

#include "gfbe_gnss.h"

namespace gfd {

// ---- compact list of the tangent dims the GNSS factors reach in the solve (GN_C = 124; the yaw is constant there)
__device__ __forceinline__ int gn_tan(int c) {
  if (c < 33) return T_POSE(c / 3) + c % 3;                      // position of pose c / 3
  if (c < 66) return T_SB((c - 33) / 3) + (c - 33) % 3;          // velocity of speed-bias (c - 33) / 3
  if (c < 110) return T_DT(0, c - 66);
  if (c < 121) return T_DDT(c - 110);
  return T_ANC + (c - 121);
}
// frames whose observations can reach compact dim c (an observation of frame i interpolates between poses lower_idx and
// lower_idx + 1 with lower_idx in {i - 1, i}: estimator.cpp:3253-3259)
__device__ __forceinline__ void gn_frames(int c, int &lo, int &hi) {
  if (c < 66) { const int f = (c < 33 ? c : c - 33) / 3; lo = max(f - 1, 0); hi = min(f + 1, NF - 1); }
  else if (c < 110) lo = hi = (c - 66) / 4;
  else if (c < 121) lo = hi = c - 110;
  else { lo = 0; hi = NF - 1; }
}
// column of compact dim c in the 2 x 18 Jacobian of an observation (frame fr, lower_idx lw, constellation sys), or -1
__device__ __forceinline__ int gn_col(int c, int fr, int lw, int sys) {
  if (c < 33) { const int f = c / 3, q = c % 3; return f == lw ? q : (f == lw + 1 ? 6 + q : -1); }
  if (c < 66) { const int f = (c - 33) / 3, q = (c - 33) % 3; return f == lw ? 3 + q : (f == lw + 1 ? 9 + q : -1); }
  if (c < 110) return (c - 66) == 4 * fr + sys ? 12 : -1;
  if (c < 121) return (c - 110) == fr ? 13 : -1;
  return 15 + (c - 121);
}
// coefficient of compact dim c (a clock dim, 66 <= c < 121) in DtDdtFactor (i, k): [-50, 50, -25 dt, -25 dt] over
// rcv_dt[i][k], rcv_dt[i + 1][k], rcv_ddt[i], rcv_ddt[i + 1] (gnss_dt_ddt_factor.cpp:3-34)
__device__ __forceinline__ double gn_dtddt_coef(int c, int i, int k, double dt) {
  if (c < 110) { const int f = (c - 66) >> 2, kk = (c - 66) & 3; return kk != k ? 0.0 : (f == i ? -50.0 : (f == i + 1 ? 50.0 : 0.0)); }
  const int f = c - 110;
  return (f == i || f == i + 1) ? -25.0 * dt : 0.0;
}
// ... and in DdtSmoothFactor (i): [w, -w] over rcv_ddt[i], rcv_ddt[i + 1] (gnss_ddt_smooth_factor.cpp:3-22)
__device__ __forceinline__ double gn_smooth_coef(int c, int i, double wgt) {
  if (c < 110) return 0.0;
  const int f = c - 110;
  return f == i ? wgt : (f == i + 1 ? -wgt : 0.0);
}

// ---- the marginalisation set's local dims (GN_M = 26): P0 V0 P1 V1 | rcv_dt[0][4] rcv_ddt[0] | rcv_dt[1][4] rcv_ddt[1] | yaw | anc
__device__ __forceinline__ int gm_col(int la, int sys) {       // column in the Jacobian of a frame-0 observation (lower_idx 0), or -1
  if (la < 12) return la;
  if (la < 16) return (la - 12) == sys ? 12 : -1;
  if (la == 16) return 13;
  if (la < 22) return -1;
  if (la == 22) return 14;
  return 15 + (la - 23);
}
__device__ __forceinline__ double gm_dtddt_coef(int la, int k, double dt) {
  if (la >= 12 && la < 16) return (la - 12) == k ? -50.0 : 0.0;
  if (la >= 17 && la < 21) return (la - 17) == k ? 50.0 : 0.0;
  return (la == 16 || la == 21) ? -25.0 * dt : 0.0;
}
__device__ __forceinline__ double gm_smooth_coef(int la, double wgt) { return la == 16 ? wgt : (la == 21 ? -wgt : 0.0); }

// 512 threads: the sums below are issue-bound integer / LDS work (a lone wave issues one instruction every ~5 cycles), two waves
// per SIMD halve that; the per-observation evaluation keeps its 256 VGPRs.
#define GN_THREADS 512
enum { GN_NCLK = 5 * GFBE_WINDOW_SIZE,     // 40 DtDdtFactors (constellation-major, the reference's insertion order) + 10 DdtSmoothFactors
       GN_ROW = 38,                        // doubles per staged observation: J (2 x 18) | r (2)
       GN_LDS_OBS = 320,                   // observations a window can stage in LDS (95 KB); a larger window sums from the global copy (L2)
       GN_MAX_GROUPS = 16 };               // workgroups a window's entries are dealt over (small batches)

// sub (mode 0; BatchDev::spec, round 5): 0 — evaluate the factors at the current state and add their J^T J / J^T r into H / g; 1 — the
// speculative pass: evaluate AT THE CANDIDATE into the set of outputs that is not the current one (per-observation J, r and the cost:
// what k_accept reads), nothing is added; 2 — the iteration after an accepted speculative pass: add the sums of the current set's
// J, r (the evaluation they came from ran at this very state). 1 + 2 perform sub 0's operations in its order.
__global__ __launch_bounds__(GN_THREADS) void k_gnss(BatchDev d0, int mode, unsigned lds_obs, int sub) {
  const int w = blockIdx.x, t = threadIdx.x, grp = blockIdx.y, ngrp = gridDim.y;     // (ngrp > 1: mode 0, sub 0 / 2 only)
  const WinDesc &ds = d0.desc[w];
  if (!ds.gnss_ready) return;
  const WinCtl &c = d0.ctl[w];
  const bool ev_only = mode == 0 && sub == 1, sum_only = mode == 0 && sub == 2;
  if (mode == 0 && !ev_only && (!ds.gnss_factors || c.done || c.reuse)) return;
  if ((mode == 1 || ev_only) && (!ds.gnss_factors || c.done || !c.have_step)) return;
  if (mode == 2 && ds.frame_count < GFBE_WINDOW_SIZE) return;
  const BatchDev d = mode == 0 ? lin_view(d0, ev_only ? 1 - c.lb : c.lb) : d0;
  const double *X = mode == 2 ? d.xout + (size_t)w * NA : d.x + ((size_t)w * 2 + ((mode == 1 || ev_only) ? 1 - c.cur : c.cur)) * NA;
  __shared__ double red[16], clk_r[GN_NCLK], s_dt[GFBE_WINDOW_SIZE];
  __shared__ int s_fb[NF + 1];
  __shared__ unsigned char s_act[GN_C];      // (the descriptor lives in global memory: what the sums below consult per entry is staged once)
  if (t < GFBE_WINDOW_SIZE) s_dt[t] = ds.gnss_frame_dt[t];
  if (t <= NF) s_fb[t] = ds.gnss_frame_begin[t];
  if (t < GN_C) s_act[t] = ds.act[gn_tan(t)];
  double *stamp = d.timing + (size_t)d.B * 32 + 8 * mode;     // phase stamps of window 0 (diagnostics: gfbe_debug_timing(batch, B))
#define GSTAMP(i) do { if (w == 0 && t == 0 && grp == 0) stamp[i] = (double)wall_clock64(); } while (0)
  GSTAMP(0);
  // staged copy of the Jacobians / residuals and of (frame, lower_idx, constellation) per observation: the sums below walk them
  // once per entry, and a single window has no other wave to hide a global load behind (measured on one window: 400 us per
  // launch from L2)
  extern __shared__ __attribute__((aligned(16))) double gn_lds[];
  const bool staged = ds.n_gnss <= (int)lds_obs;
  double *sJ = gn_lds;
  int *sMeta = (int *)(sJ + (size_t)lds_obs * GN_ROW);
  const gfbe_gnss_obs *obs = d.gnss_obs + ds.gnss_off;
  double *Jw = d.gnss_J + (size_t)ds.gnss_off * 36, *rw = d.gnss_r + (size_t)ds.gnss_off * 2;
  // MARGIN_OLD takes the observations of frame 0 (first in the frame-sorted list) between poses 0 and 1, and the clock factors of
  // the first interval
  const int n_obs = mode == 2 ? ds.gnss_frame_begin[1] : ds.n_gnss, n_int = mode == 2 ? 1 : GFBE_WINDOW_SIZE;
  double cost = 0.0;
  if (sum_only) {      // the evaluation ran in the last iteration's candidate pass: its J, r back into the staging (coalesced: 36 | 2 doubles per observation)
    if (staged) {
      for (int e = t; e < 36 * n_obs; e += GN_THREADS) { const int k = e / 36; sJ[k * GN_ROW + (e - 36 * k)] = Jw[e]; }
      for (int e = t; e < 2 * n_obs; e += GN_THREADS) sJ[(e >> 1) * GN_ROW + 36 + (e & 1)] = rw[e];
      for (int k = t; k < n_obs; k += GN_THREADS) sMeta[k] = obs[k].frame | (obs[k].lower_idx << 8) | (obs[k].sys_idx << 16);
    }
  } else
  for (int k = t; k < n_obs; k += GN_THREADS) {
    const gfbe_gnss_obs o = obs[k];
    const int lw = mode == 2 ? 0 : o.lower_idx;
    double r[2], J[36];
    gnss_psr_dopp_eval(o, ds.gnss_has_iono ? ds.gnss_iono : nullptr, X + A_POSE(lw), X + A_SB(lw), X + A_POSE(lw + 1), X + A_SB(lw + 1),
                       X[A_DT + 4 * o.frame + o.sys_idx], X[A_DDT + o.frame], X[A_YAW], X + A_ANC, r, mode == 1 ? nullptr : J);
    cost += 0.5 * r[0] * r[0] + 0.5 * r[1] * r[1];
    if (mode != 1) {
      if (grp == 0) {      // (every workgroup of the window evaluates — the same instructions on the same inputs —, the first one keeps the results)
        rw[2 * k] = r[0]; rw[2 * k + 1] = r[1];
        for (int q = 0; q < 36; q++) Jw[(size_t)36 * k + q] = J[q];
      }
      if (staged) {
        for (int q = 0; q < 36; q++) sJ[k * GN_ROW + q] = J[q];
        sJ[k * GN_ROW + 36] = r[0]; sJ[k * GN_ROW + 37] = r[1];
        sMeta[k] = o.frame | (lw << 8) | (o.sys_idx << 16);
      }
    }
  }
  if (t < GN_NCLK) {
    double r = 0.0;
    if (t < 4 * GFBE_WINDOW_SIZE) {
      const int k = t / GFBE_WINDOW_SIZE, i = t % GFBE_WINDOW_SIZE;
      if (i < n_int) r = gnss_dt_ddt_res(X[A_DT + 4 * i + k], X[A_DT + 4 * (i + 1) + k], X[A_DDT + i], X[A_DDT + i + 1], ds.gnss_frame_dt[i]);
    } else {
      const int i = t - 4 * GFBE_WINDOW_SIZE;
      if (i < n_int) r = gnss_ddt_smooth_res(X[A_DDT + i], X[A_DDT + i + 1], ds.gnss_ddt_weight);
    }
    clk_r[t] = r;
    cost += 0.5 * r * r;
  }
  GSTAMP(1);
  cost = block_sum(cost, red);      // (two block barriers: clk_r and this workgroup's Jw / rw are visible to every thread afterwards)
  if (mode == 1) { if (t == 0) d.gnss_cost[(size_t)w * 2 + 1] = cost; return; }
  const double wgt = ds.gnss_ddt_weight;
  if (mode == 2) {
    double *part = d.gnss_marg + (size_t)w * GN_MPART;
    for (int e = t; e < GN_M * GN_M + GN_M; e += GN_THREADS) {
      const bool isg = e >= GN_M * GN_M;
      const int la = isg ? e - GN_M * GN_M : e / GN_M, lb = isg ? 0 : e % GN_M;
      double s = 0.0;
      for (int k = 0; k < n_obs; k++) {
        const int sys = staged ? (sMeta[k] >> 16) : obs[k].sys_idx, ca = gm_col(la, sys), cb = isg ? 0 : gm_col(lb, sys);
        if (ca < 0 || cb < 0) continue;
        if (staged) {
          const double *J = sJ + k * GN_ROW;
          s += isg ? J[ca] * J[36] + J[18 + ca] * J[37] : J[ca] * J[cb] + J[18 + ca] * J[18 + cb];
        } else {
          const double *J = Jw + (size_t)36 * k;
          s += isg ? J[ca] * rw[2 * k] + J[18 + ca] * rw[2 * k + 1] : J[ca] * J[cb] + J[18 + ca] * J[18 + cb];
        }
      }
      for (int k = 0; k < 4; k++) {
        const double a = gm_dtddt_coef(la, k, ds.gnss_frame_dt[0]);
        s += a * (isg ? clk_r[k * GFBE_WINDOW_SIZE] : gm_dtddt_coef(lb, k, ds.gnss_frame_dt[0]));
      }
      s += gm_smooth_coef(la, wgt) * (isg ? clk_r[4 * GFBE_WINDOW_SIZE] : gm_smooth_coef(lb, wgt));
      part[e] = s;
    }
    if (t == 0) part[GN_MPART - 2] = cost;
    return;
  }
  if (t == 0 && !sum_only && grp == 0) d.gnss_cost[(size_t)w * 2] = cost;
  if (ev_only) return;
  if (d.rank != 0) return;      // landmark sharding: like the inertial / wheel / prior factors, added once
  double *H = d.H + (size_t)w * ND * ND, *g = d.g + (size_t)w * ND;
  GSTAMP(3);
  for (int e = grp * GN_THREADS + t; e < GN_C * (GN_C + 1) / 2 + GN_C; e += ngrp * GN_THREADS) {
    const bool isg = e >= GN_C * (GN_C + 1) / 2;
    int ca, cb;
    if (isg) { ca = e - GN_C * (GN_C + 1) / 2; cb = ca; } else tri_decode(e, ca, cb);
    if (!s_act[ca] || !s_act[cb]) continue;
    const int ta = gn_tan(ca), tb = gn_tan(cb);
    int fa0, fa1, fb0, fb1;
    gn_frames(ca, fa0, fa1);
    gn_frames(cb, fb0, fb1);
    const int f0 = max(fa0, fb0), f1 = min(fa1, fb1);
    double s = 0.0;
    if (f0 <= f1)      // the observations of frames f0 .. f1 that hold both dims, in observation order
      for (int k = s_fb[f0]; k < s_fb[f1 + 1]; k++) {
        int fr, lw, sys;
        if (staged) { const int mt = sMeta[k]; fr = mt & 255; lw = (mt >> 8) & 255; sys = mt >> 16; }
        else { const gfbe_gnss_obs &o = obs[k]; fr = o.frame; lw = o.lower_idx; sys = o.sys_idx; }
        const int ja = gn_col(ca, fr, lw, sys), jb = isg ? 0 : gn_col(cb, fr, lw, sys);
        if (ja < 0 || jb < 0) continue;
        if (staged) {
          const double *J = sJ + k * GN_ROW;
          s += isg ? J[ja] * J[36] + J[18 + ja] * J[37] : J[ja] * J[jb] + J[18 + ja] * J[18 + jb];
        } else {
          const double *J = Jw + (size_t)36 * k;
          s += isg ? J[ja] * rw[2 * k] + J[18 + ja] * rw[2 * k + 1] : J[ja] * J[jb] + J[18 + ja] * J[18 + jb];
        }
      }
    if (ca >= 66 && ca < 121 && cb >= 66 && cb < 121) {
      // clock factors: interval i couples frames i and i + 1 — only i in [max(fa, fb) - 1, min(fa, fb)] can hold both dims;
      // constellation-major, then interval order, like the reference inserts them
      const int fca = ca < 110 ? (ca - 66) >> 2 : ca - 110, fcb = cb < 110 ? (cb - 66) >> 2 : cb - 110;
      const int i0 = max(max(fca, fcb) - 1, 0), i1 = min(min(fca, fcb), GFBE_WINDOW_SIZE - 1);
      for (int k = 0; k < 4; k++)
        for (int i = i0; i <= i1; i++) {
          const double a = gn_dtddt_coef(ca, i, k, s_dt[i]);
          if (a != 0.0) s += a * (isg ? clk_r[k * GFBE_WINDOW_SIZE + i] : gn_dtddt_coef(cb, i, k, s_dt[i]));
        }
      for (int i = i0; i <= i1; i++) {
        const double a = gn_smooth_coef(ca, i, wgt);
        if (a != 0.0) s += a * (isg ? clk_r[4 * GFBE_WINDOW_SIZE + i] : gn_smooth_coef(cb, i, wgt));
      }
    }
    if (isg) g[ta] += s;
    else H[(size_t)max(ta, tb) * ND + min(ta, tb)] += s;      // (H holds its lower triangle; every entry has one owner thread)
  }
  GSTAMP(2);
#undef GSTAMP
}

static size_t gnss_lds_bytes(unsigned n_obs) { return sizeof(double) * ((size_t)n_obs * GN_ROW) + sizeof(int) * (size_t)n_obs; }
hipError_t gnss_init_device() {   // per device, from gfbe_create (see kernels_init_device)
  return hipFuncSetAttribute((const void *)k_gnss, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gnss_lds_bytes(GN_LDS_OBS));
}
void launch_gnss(const BatchDev &d, int mode, hipStream_t s, int sub) {
  if (!d.any_gnss) return;
  // LDS for the batch's largest window (mode 1, the candidate cost, stages nothing)
  const unsigned n = mode == 1 ? 0u : (unsigned)std::min(d.gnss_max_obs, (int)GN_LDS_OBS);
  // the sums of a small batch: a window's entries over up to 16 workgroups (~1024 workgroups of entries at most; a window that cannot
  // stage its observations keeps ONE workgroup: the others would read the global copy while the first one writes it)
  int groups = 1;
  if (mode == 0 && sub != 1 && d.gnss_max_obs <= (int)GN_LDS_OBS) groups = std::max(1, std::min((int)GN_MAX_GROUPS, 64 / std::max(d.B, 1)));
  hipLaunchKernelGGL(k_gnss, dim3(d.B, groups), dim3(GN_THREADS), mode == 1 ? 0 : gnss_lds_bytes(n), s, d, mode, n, mode == 0 ? sub : 0);
}

}  // namespace gfd

/* Minimal C caller of the drop-in boundary (include/gfbe.h only — no C++, no Python): one 11-frame window with a handful of
 * hand-made visual factors, solved by gfbe_solve_window. Shows the call sequence a maintainer wires into
 * Estimator::optimization() and proves that the header is plain C.
 *   cc -std=c99 -I include examples/gfbe_minimal.c -L ground-fusion2_amd/csrc -lgfbe -Wl,-rpath,$PWD/ground-fusion2_amd/csrc -lm
 * Exit code 0: solved (GPU present) or GFBE_NO_DEVICE reported loudly (no GPU: there is no CPU fallback). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gfbe.h"

#define NLM 12

int main(void) {
  gfbe_options opt;
  gfbe_default_options(&opt);
  gfbe_ctx *ctx = NULL;
  gfbe_status rc = gfbe_create(&ctx, 0, &opt);
  if (rc == GFBE_NO_DEVICE || rc == GFBE_DEVICE_ERROR) {
    printf("gfbe_create: status %d (%s) — no usable GPU, and no CPU fallback by design\n", (int)rc, rc == GFBE_NO_DEVICE ? "GFBE_NO_DEVICE" : "GFBE_DEVICE_ERROR");
    return 0;
  }
  if (rc != GFBE_OK) { printf("gfbe_create failed: %d\n", (int)rc); return 1; }
  printf("%s\n", gfbe_version());

  /* a camera moving along x, looking along z; body frame = camera frame (identity extrinsic) */
  gfbe_window w;
  memset(&w, 0, sizeof w);
  w.frame_count = GFBE_WINDOW_SIZE;
  for (int i = 0; i < GFBE_NFRAMES; i++) {
    w.state.para_Pose[i][0] = 0.1 * i + 0.003 * ((i * 7) % 5 - 2);   /* truth 0.1 i, slightly perturbed */
    w.state.para_Pose[i][6] = 1.0;                                      /* q = (0, 0, 0, 1) */
  }
  w.state.para_Ex_Pose[6] = 1.0;
  w.state.para_Ex_Pose_wheel[6] = 1.0;
  w.state.para_Ix_wheel[0] = w.state.para_Ix_wheel[1] = w.state.para_Ix_wheel[2] = 1.0;
  w.ex_cam_const = w.ex_wheel_const = w.ix_wheel_const = w.td_const = w.td_wheel_const = 1;
  w.pose_const[0] = 1;                        /* gauge: without IMU factors frame 0 is held */

  /* NLM landmarks seen from frame 0 in all 11 frames: K = 10 factors each (imu_i = 0, imu_j = 1..10) */
  enum { K = NLM * GFBE_WINDOW_SIZE };
  static int32_t idx[K], ii[K], jj[K];
  static double pi_[3 * K], pj_[3 * K], vi[2 * K], vj[2 * K], tdi[K], tdj[K], lam[NLM], out_lam[NLM];
  int k = 0;
  for (int l = 0; l < NLM; l++) {
    const double X = -1.0 + 0.3 * l, Y = 0.2 * (l % 3) - 0.2, Z = 4.0 + 0.5 * (l % 4);
    lam[l] = 1.0 / (Z * 1.1);                /* 10 % depth error */
    for (int j = 1; j <= GFBE_WINDOW_SIZE; j++, k++) {
      idx[k] = l; ii[k] = 0; jj[k] = j;
      pi_[3 * k] = X / Z; pi_[3 * k + 1] = Y / Z; pi_[3 * k + 2] = 1.0;
      pj_[3 * k] = (X - 0.1 * j) / Z; pj_[3 * k + 1] = Y / Z; pj_[3 * k + 2] = 1.0;
    }
  }
  w.n_feature = NLM; w.para_Feature = lam;
  w.vis.n_factor = K; w.vis.feature_index = idx; w.vis.imu_i = ii; w.vis.imu_j = jj;
  w.vis.pts_i = pi_; w.vis.pts_j = pj_; w.vis.vel_i = vi; w.vis.vel_j = vj; w.vis.td_i = tdi; w.vis.td_j = tdj;

  gfbe_state out;
  gfbe_summary sum;
  rc = gfbe_solve_window(ctx, &w, GFBE_MARGIN_NONE, &out, out_lam, NULL, &sum);
  if (rc > GFBE_NO_CONVERGENCE) { printf("gfbe_solve_window failed: %d (%s)\n", (int)rc, gfbe_last_error(ctx)); gfbe_destroy(ctx); return 1; }
  double err = 0.0;
  for (int i = 0; i < GFBE_NFRAMES; i++) err = fmax(err, fabs(out.para_Pose[i][0] - 0.1 * i));
  printf("status %d, %d iterations, cost %.3e -> %.3e, max |x - truth| %.2e m, first inverse depth %.4f (truth %.4f)\n", (int)rc, (int)sum.iterations,
         sum.initial_cost, sum.final_cost, err, out_lam[0], 1.0 / 4.0);
  gfbe_destroy(ctx);
  return (sum.final_cost <= sum.initial_cost) ? 0 : 1;
}

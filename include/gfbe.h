/*
 * gfbe.h — C ABI of the MI355X-native sliding-window visual-inertial-wheel back end.
 *
 * This is the drop-in boundary for ONE hot path of sjtuyinjie/Ground-Fusion2:
 *   Ground-Fusion++/vins_estimator/src/estimator/estimator.cpp:2951-3698  (Estimator::optimization())
 * A maintainer binds these entry points from Estimator::optimization();
 * everything else in the reference (ROS node, front end, LIO, mapping) is untouched.
 *
 * Conventions (identical to the reference's parameter blocks, estimator.h:230-238,336-342):
 *   pose block       = [px py pz qx qy qz qw]            (estimator.cpp:2341-2348)
 *   speed-bias block = [vx vy vz bax bay baz bgx bgy bgz] (estimator.cpp:2352-2362)
 *   quaternions are x,y,z,w inside every block.
 *   tangent order per pose = (dP, dtheta) with right perturbation q <- q * [1, dtheta/2]
 *   (pose_local_parameterization.cpp:12-36).
 * All floating point is FP64; all indices are int32. Plain pointers + sizes only: no C++/torch
 * types cross this boundary. Memory passed in is caller-owned host memory unless a function
 * name says "_dev". One gfbe_ctx per Estimator; a ctx is thread-compatible (one call at a time).
 */
#ifndef GFBE_H_
#define GFBE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFBE_WINDOW_SIZE 10 /* parameters.h:24 */
#define GFBE_NFRAMES 11     /* WINDOW_SIZE + 1 */
#define GFBE_MAX_OBS 11     /* a landmark has at most one observation per frame */
#define GFBE_DENSE_DIM 246  /* tangent dims of the non-landmark blocks: 11*6 + 11*9 + 6 + 6 + 3 + 1 + 1 + 4 + 1 of the ground
                               plane (para_plane_R is marginalised as a 4-D block without a manifold, marginalization_factor.cpp:140-143:
                               four prior columns) = 187, + 3 + 1 + 44 + 11 of the GNSS blocks (anc_ecef, yaw_enu_local, rcv_dt, rcv_ddt) */
#define GFBE_CORE_DIM 187   /* the tangent dims without the GNSS blocks */
#define GFBE_MAX_PRIOR_BLOCKS 32

typedef enum gfbe_status {
  GFBE_OK = 0,
  GFBE_NO_CONVERGENCE = 1,   /* ran out of iterations (normal for an 8-iteration budget) */
  GFBE_NUMERICAL_FAILURE = 2,/* linear solve failed for every mu < max_mu, or NaN */
  GFBE_BAD_INPUT = 3,
  GFBE_DEVICE_ERROR = 4,
  GFBE_NO_DEVICE = 5         /* HIP back end asked for, no GPU / kernels missing: fails loudly */
} gfbe_status;

/* Symbolic parameter-block ids. They replace the raw double* addresses the reference uses as
 * block identity (marginalization_factor.cpp:104-116, estimator.cpp:3561-3588 addr_shift). */
enum {
  GFBE_BLK_POSE0 = 0,   /* .. GFBE_BLK_POSE0+10  para_Pose[i]      size 7 */
  GFBE_BLK_SB0 = 11,    /* .. GFBE_BLK_SB0+10    para_SpeedBias[i] size 9 */
  GFBE_BLK_EX_CAM = 22, /* para_Ex_Pose[0]       size 7 */
  GFBE_BLK_EX_WHEEL = 23, /* para_Ex_Pose_wheel[0] size 7 */
  GFBE_BLK_SX = 24, GFBE_BLK_SY = 25, GFBE_BLK_SW = 26, /* para_Ix_s{x,y,w}_wheel size 1 */
  GFBE_BLK_TD = 27,     /* para_Td       size 1 */
  GFBE_BLK_TD_WHEEL = 28, /* para_Td_wheel size 1 */
  GFBE_BLK_PLANE_R = 29,  /* para_plane_R  size 4 (quaternion x y z w, OrientationSubsetParameterization({2}): estimator.cpp:3120-3123) */
  GFBE_BLK_PLANE_Z = 30,  /* para_plane_Z  size 1 */
  /* GNSS blocks (estimator.h:306-309, estimator.cpp:2965-3002; only in the problem when gfbe_window.gnss_ready) */
  GFBE_BLK_ANC_ECEF = 31, /* para_anc_ecef       size 3 */
  GFBE_BLK_YAW_ENU = 32,  /* para_yaw_enu_local  size 1 — always SetParameterBlockConstant (estimator.cpp:2991) */
  GFBE_BLK_RCV_DT0 = 33,  /* .. +43: para_rcv_dt + 4 i + k (frame i, constellation k)  size 1 */
  GFBE_BLK_RCV_DDT0 = 77, /* .. +10: para_rcv_ddt + i                                  size 1 */
  GFBE_BLK_COUNT = 88
};

/* One GNSS observation of the window with everything GnssPsrDoppFactor's constructor derives from it (see f2 below). */
typedef struct gfbe_gnss_obs {
  double sv_pos[3], sv_vel[3];   /* satellite ECEF position / velocity at transmission time (:20-35) */
  double svdt, svddt, tgd;       /* satellite clock bias (s), drift (s/s), group delay (s) */
  double pr_uura, dp_uura;       /* pseudo-range / Doppler deviation scalings (:24-26, :38-40) */
  double psr, dopp;              /* obs->psr[freq_idx] (m), obs->dopp[freq_idx] (Hz) */
  double wavelength;             /* LIGHT_SPEED / freq (m) */
  double ratio;                  /* ts_ratio (estimator.cpp:3262): weight of frame lower_idx against lower_idx + 1 */
  double doy, tow;               /* time2doy(obs->time), time2gpst(obs->time): the atmosphere models' time arguments */
  int32_t frame;                 /* i of gnss_meas_buf[i]: uses rcv_dt[i][sys_idx] and rcv_ddt[i] */
  int32_t lower_idx;             /* the factor connects Pose / SpeedBias lower_idx and lower_idx + 1 */
  int32_t sys_idx;               /* gnss_comm::sys2idx: 0 GPS, 1 GLO, 2 GAL, 3 BDS */
  int32_t _pad;
} gfbe_gnss_obs;

typedef struct gfbe_gnss_state {   /* estimator.h: para_rcv_dt, para_rcv_ddt, para_yaw_enu_local, para_anc_ecef */
  double rcv_dt[GFBE_WINDOW_SIZE + 1][4];
  double rcv_ddt[GFBE_WINDOW_SIZE + 1];
  double yaw_enu_local;
  double anc_ecef[3];
} gfbe_gnss_state;

/* The dense ("camera side") parameter blocks of one window — what vector2double() fills
 * (estimator.cpp:2337-2414) and double2vector() reads back (estimator.cpp:2501-2630). */
typedef struct gfbe_state {
  double para_Pose[GFBE_NFRAMES][7];
  double para_SpeedBias[GFBE_NFRAMES][9];
  double para_Ex_Pose[7];        /* camera 0 (the only one optimization() wires: estimator.cpp:3351) */
  double para_Ex_Pose_wheel[7];
  double para_Ix_wheel[3];       /* sx, sy, sw */
  double para_Td;
  double para_Td_wheel;
  double para_plane_R[4];        /* ground plane in the world: rotation (x y z w) and height (estimator.h:233-236); used with use_plane */
  double para_plane_Z;
  gfbe_gnss_state gnss;          /* para_rcv_dt, para_rcv_ddt, para_yaw_enu_local, para_anc_ecef; used with gnss_ready */
} gfbe_state; /* 259 doubles */

/* What IMUFactor reads from IntegrationBase (imu_factor.h:69-90, integration_base.h:169-195). */
typedef struct gfbe_imu_preint {
  double sum_dt;
  double delta_p[3];
  double delta_q[4]; /* x y z w */
  double delta_v[3];
  double linearized_ba[3];
  double linearized_bg[3];
  double jacobian[225];   /* 15x15 row-major, order P,R,V,BA,BG (parameters.h:166-173) */
  double covariance[225]; /* 15x15 row-major */
} gfbe_imu_preint; /* 467 doubles */

/* What WheelFactor reads from WheelIntegrationBase (wheel_factor.h:80-243,
 * wheel_integration_base.h:180-219). */
typedef struct gfbe_wheel_preint {
  double sum_dt;
  double delta_p[3];
  double delta_q[4]; /* x y z w */
  double linearized_sx, linearized_sy, linearized_sw, linearized_td;
  double linearized_vel[3], linearized_gyr[3]; /* first sample of the interval */
  double vel_1[3], gyr_1[3];                   /* last sample of the interval */
  double jacobian[18];   /* 6x3 row-major: d(p,theta)/d(sx,sy,sw) */
  double covariance[36]; /* 6x6 row-major */
} gfbe_wheel_preint; /* 78 doubles */

/* Marginalisation prior = MarginalizationInfo's result fields
 * (marginalization_factor.h:67-81): linearized_jacobians, linearized_residuals,
 * keep_block_{size,idx,data}; block identity is symbolic (block_id) instead of an address. */
typedef struct gfbe_prior {
  int32_t valid;    /* MarginalizationInfo::valid */
  int32_t n;        /* tangent dimension (rows == cols of J0) */
  int32_t n_blocks;
  int32_t block_id[GFBE_MAX_PRIOR_BLOCKS];   /* GFBE_BLK_* the block is attached to NOW (after addr_shift) */
  int32_t block_size[GFBE_MAX_PRIOR_BLOCKS]; /* global size: 7, 9 or 1 */
  int32_t block_idx[GFBE_MAX_PRIOR_BLOCKS];  /* tangent offset in [0,n)  (keep_block_idx - m) */
  double x0[GFBE_NFRAMES * 16 + 32];         /* keep_block_data, concatenated in block order */
  double *J0;       /* n*n row-major, caller-owned, capacity >= GFBE_DENSE_DIM^2 */
  double *r0;       /* n,             caller-owned, capacity >= GFBE_DENSE_DIM   */
} gfbe_prior;

/* Visual factor list: one entry per ProjectionTwoFrameOneCamFactor that optimization() would
 * `new` (estimator.cpp:3330-3358). Produced bit-exactly by gfbe_build_visual_factors(). */
typedef struct gfbe_visual {
  int32_t n_factor;             /* K */
  const int32_t *feature_index; /* [K] index into para_Feature */
  const int32_t *imu_i;         /* [K] start_frame of the landmark */
  const int32_t *imu_j;         /* [K] observing frame, != imu_i */
  const double *pts_i;          /* [K][3] first observation (normalised coords) */
  const double *pts_j;          /* [K][3] */
  const double *vel_i;          /* [K][2] */
  const double *vel_j;          /* [K][2] */
  const double *td_i;           /* [K] cur_td of the first observation */
  const double *td_j;           /* [K] */
} gfbe_visual;

/* f4 (BASELINE configs[4]): LiDAR point-to-plane factors attached to ONE window pose — the
 * joint LIO + VIO solve, a capability the reference does not have (its LIO solves them alone, lidarodom.cpp:538-581).
 * Factor k: residual = sqrt_info * weight[k] * (normal[k] . (R p[k] + t) + offset[k]) on pose `frame` = [t | q]
 * (LidarPlaneNormFactor::Evaluate, lio/src/liw/lidarFactor.cpp:18-51; points already in the IMU frame), robustified by
 * ceres::HuberLoss(huber_delta) as the LIO does (lidarodom.cpp:539: 0.5); huber_delta <= 0: no loss. n == 0: no block. */
typedef struct gfbe_lio_block {
  int32_t n;
  int32_t frame;                /* window pose the scan belongs to (the newest: frame_count) */
  const double *pts;            /* [n][3] */
  const double *normals;        /* [n][3] */
  const double *offsets;        /* [n] */
  const double *weights;        /* [n] or NULL (= 1) */
  double sqrt_info;
  double huber_delta;
} gfbe_lio_block;

/* One call of optimization(): everything it reads. */
typedef struct gfbe_window {
  int32_t frame_count;          /* poses 0..frame_count are in the problem (== WINDOW_SIZE when full) */
  gfbe_state state;             /* in: vector2double() output */
  int32_t n_feature;            /* L = FeatureManager::getFeatureCount() */
  const double *para_Feature;   /* [L] inverse depths (getDepthVector, feature_manager.cpp:286-302) */
  const uint8_t *feature_const; /* [L] 1 => SetParameterBlockConstant (estimate_flag==1, estimator.cpp:3352) */
  /* SetParameterBlockConstant decisions (estimator.cpp:3022,3059,3101,3114-3116,3159-3161,3302-3303) */
  uint8_t pose_const[GFBE_NFRAMES];
  uint8_t sb_const[GFBE_NFRAMES];
  uint8_t ex_cam_const, ex_wheel_const, ix_wheel_const, td_const, td_wheel_const;
  /* PoseSubsetParameterization constancy masks (pose_subset_parameterization.cpp:11-45): tangent
   * components zeroed in Plus only (the Jacobian stays unmasked — reference quirk, reproduced). */
  uint8_t ex_cam_mask[6], ex_wheel_mask[6];
  /* Optional in-window factors (estimator.cpp:3120-3136, 3214-3228, 3004-3012; all shipped yamls: plane 0, gnss_enable 0):
   *   use_plane   one PlaneFactor (plane_factor.h:25-122) per window pose i < frame_count on {Pose[i], Ex_Pose_wheel, plane_R,
   *               plane_Z}; the MARGIN_OLD set takes the one of frame 0 (estimator.cpp:3441-3448). plane_const = the
   *               SetParameterBlockConstant decision of :3126-3135.
   *   use_anchor  one PoseAnchorFactor (pose_anchor_factor.cpp:8-32) on Pose[0] (the `first_optimization && GNSS_ENABLE`
   *               block, :3004-3012). */
  uint8_t use_plane, plane_const, use_anchor;
  int32_t n_imu;                /* IMU factors; factor k links frames imu_frame[k], imu_frame[k]+1 */
  const int32_t *imu_frame;
  const gfbe_imu_preint *imu;
  int32_t n_wheel;
  const int32_t *wheel_frame;
  const gfbe_wheel_preint *wheel;
  gfbe_visual vis;
  const gfbe_prior *prior;      /* NULL or !valid => no MarginalizationFactor */
  gfbe_lio_block lio;           /* optional LiDAR factors on one pose (n = 0: none) */
  double plane_noise_inv[3];    /* PITCH_N_INV, ROLL_N_INV, ZPW_N_INV (parameters.cpp:340-345) */
  double anchor_pose[7];        /* PoseAnchorFactor's anchor_value (para_Pose[0] at the first optimisation) */
  double anchor_sqrt_info;      /* 120 in the reference (pose_anchor_factor.h:19) */
  /* GNSS inside the window (estimator.cpp:2965-3002, 3239-3291, 3462-3496; gnss_enable: 0 in every shipped yaml).
   *   gnss_ready  the blocks para_yaw_enu_local (held constant, :2991), para_anc_ecef, para_rcv_dt[11][4], para_rcv_ddt[11] join the
   *               problem. Unless the window moves too slowly — mean |Vs[i].xy| over the window below 0.3 m/s at the start of the
   *               call, `lowspeed`, :2969-2984, computed by the library from `state` — one GnssPsrDoppFactor per observation, one
   *               DtDdtFactor per constellation and frame interval and one DdtSmoothFactor per interval are added (:3239-3291).
   *               MARGIN_OLD takes the factors of frame 0 with the drop sets of :3462-3496 whether the window is slow or not.
   *   gnss_obs    the n_gnss observations of gnss_meas_buf[0 .. WINDOW_SIZE] in the reference's insertion order (frame-major);
   *               `frame`, `lower_idx`, `ratio`, `sys_idx` as :3250-3263 computes them (frame 0 always has lower_idx 0).
   *   gnss_iono   latest_gnss_iono_params (8 Klobuchar parameters) or NULL; gnss_frame_dt[i] = Headers[i + 1] - Headers[i];
   *               gnss_ddt_weight = GNSS_DDT_WEIGHT. The GNSS state travels in `state.gnss` and comes back in out_state->gnss. */
  int32_t gnss_ready;
  int32_t n_gnss;
  const gfbe_gnss_obs *gnss_obs;
  const double *gnss_iono;
  double gnss_frame_dt[GFBE_WINDOW_SIZE];
  double gnss_ddt_weight;
} gfbe_window;

typedef struct gfbe_options {
  /* sizeof(gfbe_options) of the header the CALLER was compiled against, written by gfbe_default_options (always start from it).
   * gfbe_create refuses a struct whose size is not the library's own (GFBE_BAD_INPUT, "options ABI mismatch"): the struct has grown
   * by trailing fields from round to round, and a caller built against an older header would otherwise hand over a shorter struct
   * whose missing fields the library reads from whatever follows it. gfbe_options_size() returns the library's. */
  int32_t struct_size;
  int32_t max_num_iterations;   /* NUM_ITERATIONS = 8 (m3dgr.yaml:109) */
  double huber_delta;           /* HuberLoss(1.0)  (estimator.cpp:2959) */
  double vis_sqrt_info;         /* FOCAL_LENGTH/1.5 = 400 (estimator.cpp:193) */
  double g_norm;                /* G = (0,0,g_norm) (parameters.cpp:223; m3dgr.yaml:117) */
  /* Ceres 1.14 Solver::Options defaults the reference leaves untouched (estimator.cpp:3364-3376) */
  double initial_trust_region_radius; /* 1e4 */
  double function_tolerance;          /* 1e-6 */
  double gradient_tolerance;          /* 1e-10 */
  double parameter_tolerance;         /* 1e-8 */
  double min_relative_decrease;       /* 1e-3 */
  int32_t jacobi_scaling;             /* 1 */
  double marg_eps;                    /* eps = 1e-8, marginalization_factor.h:70 */
  /* Square root of the marginalised information A' (marginalization_factor.cpp:294-302):
   *   0  eigen-decomposition, S > eps thresholding — the reference's construction
   *   1  diagonally pivoted LDL^T with pivots > eps — same J0^T J0 / J0^T r0 up to O(n*eps) absolute
   *      (1e-14 relative to |A'|); the cheaper one on the device: 0.09 ms of a 1.15 ms call, where mode 0 — Householder
   *      tridiagonalisation + divide & conquer since round 6 — takes 0.64 ms of a 1.72 ms call. Default.
   *   The two square roots carry the same information J0^T J0, J0^T r0; they do NOT share the prior's constant term |r0|^2 (the
   *   smallest kept eigenvalues of A' amplify b'): a caller that logs costs sees a constant offset between the modes (and between
   *   mode 0 and the reference's own Eigen build), never a different step. */
  int32_t marg_sqrt;
  /* 1: gfbe_batch_solve replays its fixed kernel sequence as a hipGraph from the third call on a batch (first
   * call eager, second captured); 0 (default): eager launches — on ROCm 7.2 / MI355X the replay measured 2.50 vs
   * 2.54 ms for one window and 3 % slower at 256 windows. Ignored while profiling / landmark sharding. */
  int32_t use_graph;
  /* Parts a batch of >= 128 windows is split into by gfbe_batch_upload; gfbe_batch_solve runs the parts side by side,
   * each on its own pair of streams (kernels of different stages share the GPU). 1 (default): four parts for batches of
   * >= 2048 windows, one below (measured round 3: 1024 windows 62.7k solves/s whole, 61.2k as four parts, 56.7k as two;
   * 4096 windows 67.2k as four parts, 65.0k whole); n >= 2: exactly n parts. Results per window are unchanged (every window is independent). 0: one launch sequence for the
   * whole batch. */
  int32_t split_batch;
  /* Solver::Options::max_solver_time_in_seconds (estimator.cpp:3369-3376: SOLVER_TIME = 0.04 s, x 4/5 before a MARGIN_OLD).
   * Checked ON THE DEVICE at the start of every trust-region iteration against the device's wall clock (no host
   * synchronisation inside a solve); a window over budget stops with termination 0 / GFBE_NO_CONVERGENCE like Ceres'
   * "maximum solver time reached". 0 (default) = no cap: results then depend on the inputs only. A binding that wants the
   * reference's behaviour sets 0.04 (or 0.032); at 2 ms per solve it never triggers. */
  double max_solver_time_in_seconds;
  /* Host threads gfbe_batch_upload / gfbe_batch_download pack and unpack windows with (one window per task).
   * 0 (default) = the library's choice per job: the packing passes of an upload min(hardware threads, 24) — 48 on a host with 128 or more
   * hardware threads —, the unpacking of a download min(hardware threads, 16); 1 = the calling thread only. */
  int32_t host_threads;
  /* Factorisation of the reduced system (the DENSE_SCHUR linear solve of estimator.cpp:3364-3379 after the landmark elimination):
   *   0 (default)  the speed-bias blocks are eliminated as a chain of 9 x 9 blocks before the dense pose / extrinsic part is
   *                factorised (~70 KB of LDS, two workgroups per CU). Needs the structure every factor of the reference has:
   *                a prior that keeps a speed-bias block other than SpeedBias[0] switches its batch to 1 by itself.
   *                Batches below 32 windows — the reference's call pattern is ONE — eliminate the chain from both ends at once
   *                (eight waves, one workgroup per CU: 6 sequential block steps instead of 11), larger ones from one end (four waves, two
   *                workgroups per CU: throughput). Same step to rounding.
   *   1            one blocked factorisation of the whole reduced system (160 KB of LDS). Same step to rounding.
   *   2 / 3        (tests, measurements) the one-ended / the two-ended chain kernel whatever the batch size.
   * A batch with GNSS blocks (gfbe_window.gnss_enable; 246 tangent dims instead of 187) takes the one-ended chain kernel with the
   * 58 GNSS dims as dense columns when every window's dense part fits nine 16-wide tile columns (<= 142 dims), and the blocked
   * out-of-LDS factorisation of rounds 3-5 otherwise;
   *   4            (tests, measurements) that blocked factorisation for every batch with GNSS blocks. Same step to rounding. */
  int32_t solve_kernel;
  /* TEST HOOK (0 = off, the default; never set it in production): the first factorisation of trust-region iteration
   * `test_fail_chol_iter` is declared failed, so that DoglegStrategy's mu retry — which well-posed windows never take —
   * can be exercised (tests/test_gpu_branches.py). */
  int32_t test_fail_chol_iter;
  /* TEST HOOK: how many consecutive factorisation attempts of that iteration fail (default 1; 0 is taken as 1). */
  int32_t test_fail_chol_count;
  /* Landmark sharding (gfbe_set_allreduce) only: how many times a window whose reduced system failed to factorise is retried with
   * mu x 10 (DoglegStrategy::ComputeGaussNewtonStep retries up to max_mu = 1, i.e. 8 times from min_mu = 1e-8; the unsharded kernels
   * do so inside the solve kernel). Every retry is one more [E rebuild | all-reduce | factorisation] triple in the FIXED launch
   * sequence of every linearisation, taken or not: default 1 (well-posed windows never take even that one); 8 reproduces Ceres;
   * 0: no retry — a failed factorisation ends the solve as a failed linear solve, and every linearisation carries one all-reduce
   * less (three per trust-region iteration instead of four: the packed system and the two scalar exchanges). */
  int32_t sharded_mu_retries;
  /* Batches without an all-reduce hook: 1 (default) — the pass that evaluates the candidate of a trust-region
   * iteration LINEARISES there (every iteration but the last of a solve), into a second set of the linearisation's outputs; an accepted
   * step makes it the current set, a rejected one leaves the old linearisation in place (DoglegStrategy's reuse), and the next
   * iteration starts at the landmark elimination. The evaluations are TrustRegionMinimizer's own, in its order — the candidate's
   * cost, then residuals + Jacobians at the accepted point: the same state —, so nothing a solve returns changes by a bit
   * (tests/test_gpu_speculative.py); what goes is one evaluation pass per iteration: the cost-only pass of the visual and the dense
   * factors (8192 resident windows: +10 % solves/s; a single window: one launch less per iteration, -3.5 %). Costs the second set:
   * ~1.5 MB per 2k-landmark window. 0 — cost pass and linearisation separately, as in rounds 1-4. */
  int32_t speculative_linearization;
  /* Throughput batches (>= 32 windows) whose windows all hold the camera extrinsic and td constant, without an all-reduce hook:
   * 1 — the visual factors are evaluated AND the landmarks eliminated by one kernel (k_linschur: one workgroup per window and group of
   * start frames; a landmark tile's rows go from the evaluating lanes' registers into the LDS panel the matrix cores multiply, and are
   * not re-read from HBM by a second kernel); 0 (default) — k_vis<0> + k_schur. Same quantities, the per-landmark sums added in
   * another order (last bits). Built in round 6 because the two kernels exchange ~1.4 GB per 2048 windows; MEASURED SLOWER on MI355X
   * (per 512 windows: 281 us at the first iteration against 133 + 143, 325 us at a candidate against 137 + 142; 8192 resident windows:
   * 97k against 114k solves/s, profiles/r6_linschur_ab.txt): neither kernel was bound by that traffic, and one workgroup that
   * alternates between the two phases behind three barriers per tile overlaps worse than two kernels that each fill the device.
   * Kept as an option with its tests (tests/test_gpu_linschur.py); DESIGN.md section 4. */
  int32_t merge_lin_schur;
} gfbe_options;

typedef struct gfbe_summary {
  int32_t status;            /* gfbe_status */
  int32_t iterations;        /* trust-region iterations run (excluding iteration 0) */
  int32_t num_successful;    /* accepted steps */
  int32_t termination;       /* 0 max-iter, 1 function tol, 2 parameter tol, 3 gradient tol, 4 failure */
  double initial_cost;
  double final_cost;
  double final_radius;
  double cost_history[16];   /* cost after each iteration (accepted or not), [0] = initial */
  uint8_t accepted[16];      /* accept/reject per iteration */
  /* Phase times from the device's wall clock (100 MHz), per window: first kernel of the solve -> re-anchored state, and
   * re-anchored state -> prior written (0 when no marginalisation ran). In a batch all windows advance together, so these
   * are the batch's phase times as seen by this window. */
  double ms_solve, ms_marginalize;
  /* Host <-> device bytes of this window: packed upload (landmarks, factors, pre-integrations, prior) and result download. */
  double bytes_uploaded, bytes_downloaded;
} gfbe_summary;

enum { GFBE_MARGIN_OLD = 0, GFBE_MARGIN_SECOND_NEW = 1, GFBE_MARGIN_NONE = 2 };

/* ------------------------------------------------------------------------------------------
 * Context
 * ------------------------------------------------------------------------------------------ */
typedef struct gfbe_ctx gfbe_ctx;

void gfbe_default_options(gfbe_options *opt);
/* sizeof(gfbe_options) as the LIBRARY was built (gfbe_options.struct_size must equal it). */
int32_t gfbe_options_size(void);

/* device < 0: host-only context (bookkeeping functions only; every compute entry point returns
 * GFBE_NO_DEVICE). device >= 0: HIP device ordinal; fails with GFBE_NO_DEVICE if absent.
 * The solver drives up to eight HIP streams (two solver lanes with a side stream each, upload, download, the caller's): set
 * GPU_MAX_HW_QUEUES=8 in the process environment before HIP initialises — the runtime's default of four makes uploads queue
 * behind solves. The library does not touch the environment; gfbe_create returns GFBE_OK and leaves a note in gfbe_create_note
 * (never in gfbe_last_error, which only ever holds the cause of a failing call) when the variable is unset or below 8.
 * (gfbe_batch_upload leaves a note there too when it had to switch speculative_linearization off for a batch whose second set of
 * outputs did not fit the device.) */
gfbe_status gfbe_create(gfbe_ctx **ctx, int device, const gfbe_options *opt);
/* Frees the context and the device memory it caches. Batches (gfbe_batch_free) and feature tables (gfbe_ftab_destroy) made with
 * the context must be released BEFORE it. The HIP streams the context created are kept in a process-wide pool for the next
 * gfbe_create on the same device (a later context then runs on the hardware queues the first one had); they live until the process ends. */
void gfbe_destroy(gfbe_ctx *ctx);
const char *gfbe_last_error(const gfbe_ctx *ctx);
/* Informational remarks of gfbe_create ("" when there are none); the reference has no counterpart (ROS_WARN at start-up). */
const char *gfbe_create_note(const gfbe_ctx *ctx);
const char *gfbe_version(void);
/* Launch kernels on this hipStream_t (e.g. torch's current stream). NULL = the ctx's own stream. */
gfbe_status gfbe_set_stream(gfbe_ctx *ctx, void *hip_stream);

/* ------------------------------------------------------------------------------------------
 * a13  Landmark bookkeeping (host, integer, bit-exact):
 *   FeatureManager::getFeatureCount   feature_manager.cpp:43-55
 *   FeatureManager::getDepthVector    feature_manager.cpp:286-302
 *   visual-factor loop                estimator.cpp:3326-3358 (solve) and 3498-3531 (marginalise)
 * Input is the std::list<FeaturePerId> flattened in list order: feature f has n_obs[f]
 * observations starting at frame start_frame[f]; obs rows are [x y z u v vx vy] (7 doubles) plus
 * cur_td, stored contiguously at obs_offset[f].
 * ------------------------------------------------------------------------------------------ */
typedef struct gfbe_feature_list {
  int32_t n;                    /* features in the list (all, including used_num < 4) */
  const int32_t *start_frame;   /* [n] */
  const int32_t *n_obs;         /* [n] feature_per_frame.size() */
  const int32_t *obs_offset;    /* [n] first row of this feature in obs / obs_td */
  const double *obs;            /* [sum n_obs][7]  x y z u v vx vy */
  const double *obs_td;         /* [sum n_obs]     cur_td */
  const double *estimated_depth;/* [n] */
  const int32_t *estimate_flag; /* [n] 1 => depth came from the RGB-D image */
} gfbe_feature_list;

/* Returns the number of landmarks L (getFeatureCount). */
int32_t gfbe_feature_count(const gfbe_feature_list *fl);
/* Counts factors: total (solve) or only landmarks with start_frame == 0 (marginalise). */
int32_t gfbe_visual_factor_count(const gfbe_feature_list *fl, int32_t only_start_frame0);
/* Fills caller-allocated arrays (capacity from the two functions above); returns K written.
 * para_Feature/feature_const have capacity L. */
int32_t gfbe_build_visual_factors(const gfbe_feature_list *fl, int32_t only_start_frame0,
                                  int32_t *feature_index, int32_t *imu_i, int32_t *imu_j,
                                  double *pts_i, double *pts_j, double *vel_i, double *vel_j,
                                  double *td_i, double *td_j,
                                  double *para_Feature, uint8_t *feature_const);
/* FeatureManager::setDepth (feature_manager.cpp:249-267): solve_flag[f] = 0 untouched,
 * 1 ok, 2 failed (negative depth); estimated_depth updated for landmarks with used_num >= 4. */
void gfbe_set_depth(const gfbe_feature_list *fl, const double *para_Feature,
                    double *estimated_depth, int32_t *solve_flag);

/* ------------------------------------------------------------------------------------------
 * f1  FeatureManager / slideWindow operations on device-resident feature tables
 *     (SURVEY.md section 8f rank 1 — the step either side of the solve):
 *       FeatureManager::addFeatureCheckParallax   feature_manager.cpp:57-116 (+ compensatedParallax2 :978-1011)
 *       setDepth / removeFailures / clearDepth / getDepthVector   :249-302
 *       triangulate :669-724, triangulateWithDepth :726-799
 *       removeOutlier :801-816, removeBackShiftDepth :818-856, removeBack :858-874, removeFront :914-934
 *       Estimator::slideWindow / slideWindowNew / slideWindowOld   estimator.cpp:3700-3899
 *       Estimator::outliersRejection :3971-4028, movingConsistencyCheckW :4030-4074
 * A gfbe_ftab holds W independent tables in HBM (one per window; every call applies the same operation to
 * all W tables, per-table arguments are arrays of length W). A table is FeatureManager's
 * std::list<FeaturePerId> in insertion order — the order that defines para_Feature[k]
 * (getFeatureCount / getDepthVector) — with at most WINDOW_SIZE+1 observations per feature, each
 * [x y z u v vx vy depth] + cur_td (FeaturePerFrame, feature_manager.h:30-64). Erasures keep the order.
 * Integer contents are bit-exact against the CPU oracle; depths to the tolerance stated in the tests.
 * Pose arguments: [P(3) | R(9, row-major)] per frame; extrinsic: [tic(3) | ric(9, row-major)].
 * ------------------------------------------------------------------------------------------ */
typedef struct gfbe_ftab gfbe_ftab;
typedef struct gfbe_ftab_options {
  double init_depth;       /* INIT_DEPTH = 5.0 (parameters.cpp:484) */
  double min_parallax;     /* MIN_PARALLAX = keyframe_parallax / FOCAL_LENGTH = 10 / 600 (parameters.cpp:351-352, m3dgr.yaml:110) */
  double focal_length;     /* FOCAL_LENGTH = 600 (parameters.h:23) in the parallax / outlier thresholds */
  double depth_threshold;  /* triangulateWithDepth: RGB-D depths in [0.1, depth_threshold] are trusted (parameters.cpp:178; yaml) */
} gfbe_ftab_options;
void gfbe_ftab_default_options(gfbe_ftab_options *opt);
/* feature_capacity <= 16384 features per table.
 * Every table operation runs in order on the context's stream. The ones without outputs (triangulate, set_depth, clear_depth,
 * remove_*) copy their arguments and return without waiting for the device; the ones that hand something back (add_frame,
 * check_outliers, get_depth_vector, size, download, gfbe_batch_upload_tables) wait for what was enqueued before them. Capacity /
 * observation-count overflow is raised by gfbe_ftab_add_frame (GFBE_BAD_INPUT, sticky).
 * A device error of a deferred operation (its kernel has not run when the call returns) is reported by the NEXT table call that
 * synchronises — usually add_frame or check_outliers — and gfbe_last_error then names that call, not the operation that faulted.
 * gfbe_ftab_create leaves *out null and frees what it had allocated when any of its allocations fails. */
gfbe_status gfbe_ftab_create(gfbe_ctx *ctx, int32_t n_tables, int32_t feature_capacity,
                             const gfbe_ftab_options *opt, gfbe_ftab **out);
void gfbe_ftab_destroy(gfbe_ctx *ctx, gfbe_ftab *t);
/* addFeatureCheckParallax. Table w receives the features [offset[w], offset[w+1]) — ascending feature_id, the
 * iteration order of the reference's std::map — with observation rows obs8 [x y z u v vx vy depth].
 * Outputs per table: keyframe (1 = the function returns true = MARGIN_OLD), counters
 * [last_track_num, new_feature_num, long_track_num], avg_parallax (last_average_parallax). */
gfbe_status gfbe_ftab_add_frame(gfbe_ctx *ctx, gfbe_ftab *t, const int32_t *frame_count, const int32_t *offset,
                                const int32_t *feature_id, const double *obs8, const double *td,
                                int32_t *keyframe, int32_t *counters, double *avg_parallax);
/* removeBackShiftDepth(marg_R, marg_P, new_R, new_P); *_PR = [W][12]. */
gfbe_status gfbe_ftab_remove_back_shift_depth(gfbe_ctx *ctx, gfbe_ftab *t, const double *marg_PR, const double *new_PR);
gfbe_status gfbe_ftab_remove_back(gfbe_ctx *ctx, gfbe_ftab *t);
gfbe_status gfbe_ftab_remove_front(gfbe_ctx *ctx, gfbe_ftab *t, const int32_t *frame_count);
/* removeOutlier: ids of table w are [offset[w], offset[w+1]). */
gfbe_status gfbe_ftab_remove_outlier(gfbe_ctx *ctx, gfbe_ftab *t, const int32_t *offset, const int32_t *ids);
gfbe_status gfbe_ftab_remove_failures(gfbe_ctx *ctx, gfbe_ftab *t);
gfbe_status gfbe_ftab_clear_depth(gfbe_ctx *ctx, gfbe_ftab *t);
/* setDepth(x) / getDepthVector(): para_Feature of table w starts at offset[w] (capacity offset[w+1]-offset[w]);
 * count[w] = getFeatureCount(). */
gfbe_status gfbe_ftab_set_depth(gfbe_ctx *ctx, gfbe_ftab *t, const int32_t *offset, const double *para_Feature);
gfbe_status gfbe_ftab_get_depth_vector(gfbe_ctx *ctx, gfbe_ftab *t, const int32_t *offset, double *para_Feature,
                                       int32_t *count);
/* triangulate (with_depth = 0) / triangulateWithDepth (1). poses [W][11][12], tic_ric [W][12]. */
gfbe_status gfbe_ftab_triangulate(gfbe_ctx *ctx, gfbe_ftab *t, const double *poses, const double *tic_ric,
                                  int32_t with_depth);
/* mode 0: outliersRejection, 1: movingConsistencyCheckW. ids_out of table w start at offset[w] (ascending, the
 * iteration order of the reference's std::set); count_out[w] ids are written. */
gfbe_status gfbe_ftab_check_outliers(gfbe_ctx *ctx, gfbe_ftab *t, const double *poses, const double *tic_ric,
                                     int32_t mode, const int32_t *offset, int32_t *ids_out, int32_t *count_out);
gfbe_status gfbe_ftab_size(gfbe_ctx *ctx, gfbe_ftab *t, int32_t *n_features);
/* Snapshot of table w in list order; obs8 [n][11][8], obs_td [n][11] (rows >= n_obs are zero). Any pointer may be NULL. */
gfbe_status gfbe_ftab_download(gfbe_ctx *ctx, gfbe_ftab *t, int32_t w, int32_t *feature_id, int32_t *start_frame,
                               int32_t *n_obs, double *obs8, double *obs_td, double *estimated_depth,
                               int32_t *estimate_flag, int32_t *solve_flag);
struct gfbe_batch;
/* gfbe_batch_upload with the visual part of window w read from table w of `t` ON THE DEVICE (landmarks = features with
 * >= 4 observations in list order, para_Feature[k] = 1 / estimated_depth, constant when estimate_flag == 1:
 * estimator.cpp:3330-3358): wins[w]->vis, n_feature, para_Feature and feature_const are ignored; state, pre-integrations,
 * prior and the constant flags come from wins[w]. gfbe_batch_download returns the features in list order, ready for
 * gfbe_ftab_set_depth. n_windows <= the number of tables. */
gfbe_status gfbe_batch_upload_tables(gfbe_ctx *ctx, gfbe_ftab *t, int32_t n_windows, const gfbe_window *const *win,
                                     struct gfbe_batch **out);
/* Landmarks of window w of a batch (= the length of out_feature[w] in gfbe_batch_download; getFeatureCount(),
 * feature_manager.cpp:56-68); -1 for a bad index. */
int32_t gfbe_batch_feature_count(const struct gfbe_batch *batch, int32_t w);
/* slideWindow()'s shift of the state blocks (estimator.cpp:3700-3858): MARGIN_OLD moves frames 1..WINDOW_SIZE to
 * 0..WINDOW_SIZE-1 and leaves the newest duplicated at WINDOW_SIZE; MARGIN_SECOND_NEW copies frame WINDOW_SIZE
 * onto WINDOW_SIZE-1. Host function (the caller owns the state). */
void gfbe_slide_window_state(gfbe_state *state, int32_t margin_flag);

/* ------------------------------------------------------------------------------------------
 * f3  global_fusion pose graph (BASELINE configs[3]):
 *       GlobalOptimization::optimize     global_fusion/src/globalOpt.cpp:107-236
 *       RelativeRTError / TError         global_fusion/src/Factors.h:26-114 (ceres::AutoDiffCostFunction in the
 *                                        reference; analytic tangent Jacobians here)
 * A chain of n poses [t(3) | q(w,x,y,z)] (the quaternion order of globalOpt.cpp:46-47, NOT the x,y,z,w of the VIO
 * blocks), one RelativeRTError between consecutive poses (rel_i[k], rel_i[k] + 1) with measurement [t(3) | q(wxyz)]
 * and the reference's constant weights 1 / t_var, 1 / q_var (0.1, 0.01: globalOpt.cpp:171-173), and Huber-robustified
 * (delta 1.0) position fixes TError [x y z var] on a subset of the poses (GPS / AprilTag, :177-185). Solver = what
 * ceres::Solve does with the reference's options (:117-121): Levenberg-Marquardt trust region, Jacobi scaling,
 * max_num_iterations 5, quaternions on ceres::QuaternionParameterization (x <- [cos|d|, sin|d|/|d| d] * x) — the normal
 * equations are block-tridiagonal (6 x 6 blocks), factorised by a block Cholesky recurrence instead of
 * SPARSE_NORMAL_CHOLESKY's general sparse factorisation.
 * gfbe_pg_eval: residuals and tangent Jacobians at `pose` (columns: dq_i(3) t_i(3) dq_j(3) t_j(3)); cost includes the
 * Huber loss of the fixes. Any output pointer may be NULL.
 * ------------------------------------------------------------------------------------------ */
gfbe_status gfbe_pg_eval(gfbe_ctx *ctx, int32_t n_poses, const double *pose, int32_t n_rel, const int32_t *rel_i,
                         const double *rel_meas, double t_var, double q_var, int32_t n_fix, const int32_t *fix_i,
                         const double *fix_meas, double huber_delta, double *rel_r, double *rel_J, double *fix_r,
                         double *cost);
gfbe_status gfbe_pg_solve(gfbe_ctx *ctx, int32_t n_poses, const double *pose_in, int32_t n_rel, const int32_t *rel_i,
                          const double *rel_meas, double t_var, double q_var, int32_t n_fix, const int32_t *fix_i,
                          const double *fix_meas, double huber_delta, int32_t max_num_iterations, double *pose_out,
                          gfbe_summary *summary);

/* ------------------------------------------------------------------------------------------
 * f4  LIO scan residuals (BASELINE configs[4]): point-to-plane factors of the LiDAR
 *     odometry, evaluated and reduced to normal equations on the device (the voxel neighbour search that produces
 *     the planes stays on the CPU):
 *       LidarPlaneNormFactor::Evaluate     lio/src/liw/lidarFactor.cpp:18-51   (ct = 0: one pose [t | q(x,y,z,w)])
 *       CTLidarPlaneNormFactor::Evaluate   lio/src/liw/lidarFactor.cpp:59-120  (ct = 1: begin and end pose, the point
 *                                          is taken at slerp(alpha) / lerp(alpha) between them)
 *     residual = sqrt_info * weight * (n . (R p + t) + offset); tangent = RotationParameterization (q * deltaQ(d),
 *     poseParameterization.cpp:31-42). Jacobian columns: ct = 0: [t(3) | theta(3)]; ct = 1: [t_b | theta_b | t_e | theta_e]
 *     (the parameter-block order of the factor). pts are the points the factor stores (point_body / raw_keypoint,
 *     already in the IMU frame). Outputs (any may be NULL): r [n], J [n][6 or 12], H [d][d] = J^T J, g [d] = J^T r,
 *     cost = 1/2 sum r^2.
 * ------------------------------------------------------------------------------------------ */
gfbe_status gfbe_lio_linearize(gfbe_ctx *ctx, int32_t ct, int32_t n, const double *pts, const double *normals,
                               const double *offsets, const double *alpha, const double *weights, double sqrt_info,
                               const double *pose_begin, const double *pose_end, double *r, double *J, double *H,
                               double *g, double *cost);

/* ------------------------------------------------------------------------------------------
 * f2  Optional in-window factors. PlaneFactor and PoseAnchorFactor run INSIDE
 *     gfbe_solve_window / gfbe_batch_solve when gfbe_window.use_plane / use_anchor are set (see gfbe_window); the
 *     functions below evaluate them stand-alone. The GNSS factors run inside the solve and MARGIN_OLD as well when
 *     gfbe_window.gnss_ready is set: the problem then grows by rcv_dt[11][4], rcv_ddt[11], yaw_enu_local and anc_ecef
 *     (59 more tangent dimensions, GFBE_DENSE_DIM = 246); such batches factorise their reduced system from global memory
 *     (k_solve_big) instead of the LDS-resident kernels. Every shipped yaml sets gnss_enable: 0.
 *       PlaneFactor::Evaluate        factor/plane_factor.h:25-122  — n factors sharing ex_wheel [p | q(x,y,z,w)],
 *           plane_R [q(x,y,z,w)] and plane_Z (estimator.cpp:3214-3220: one factor per window pose);
 *           noise_inv = {PITCH_N_INV, ROLL_N_INV, ZPW_N_INV} (parameters.cpp:340-345).
 *           r [n][3]; J [n][3][16], tangent columns = pose_i(6) ex_wheel(6) plane_R(3) plane_Z(1).
 *       PoseAnchorFactor::Evaluate   factor/pose_anchor_factor.cpp:8-32 — sqrt_info = 120 in the reference
 *           (pose_anchor_factor.h:19). r [n][6]; J [n][6][6] (tangent of the pose). The reference's Jacobian is 2 sqrt_info
 *           times [I 0; 0 Qright(q_anchor^-1)] — twice the derivative of its own position residual; reproduced as it is.
 *       OrientationSubsetParameterization::Plus   factor/orientation_subset_parameterization.cpp:27-45 (host function):
 *           out = normalize(q * deltaQ(delta with the constant components zeroed)); plane_R uses constant = {0, 0, 1}
 *           (estimator.cpp:3122).
 *     r, J, cost may be NULL. cost = 1/2 sum r^2.
 * ------------------------------------------------------------------------------------------ */
gfbe_status gfbe_plane_eval(gfbe_ctx *ctx, int32_t n, const double *pose /*[n][7]*/, const double *ex_wheel /*[7]*/,
                            const double *plane_R /*[4]*/, double plane_Z, const double *noise_inv /*[3]*/, double *r,
                            double *J, double *cost);
gfbe_status gfbe_anchor_eval(gfbe_ctx *ctx, int32_t n, const double *pose /*[n][7]*/, const double *anchor /*[n][7]*/,
                             double sqrt_info, double *r, double *J, double *cost);
void gfbe_orientation_subset_plus(const double *q /*[4] x y z w*/, const double *delta /*[3]*/,
                                  const uint8_t *constant /*[3]*/, double *out /*[4]*/);

/* ------------------------------------------------------------------------------------------
 * f2  GNSS factors of the window (estimator.cpp:3239-3291), evaluated stand-alone on the device (inspection / parity API; inside
 *     the solve they are driven by gfbe_window.gnss_*):
 *       GnssPsrDoppFactor::Evaluate   factor/gnss_psr_dopp_factor.cpp:50-208   (2 rows: pseudo-range, Doppler)
 *       DtDdtFactor::Evaluate         factor/gnss_dt_ddt_factor.cpp:3-34       (receiver clock bias / drift chain)
 *       DdtSmoothFactor::Evaluate     factor/gnss_ddt_smooth_factor.cpp:3-22
 *     together with the gnss_comm functions the pseudo-range factor calls per evaluation (gnss_comm/src/gnss_utility.cpp:
 *     ecef2geo :347, ecef2rotation :757, sat_azel :762, calculate_trop_delay :841 — Saastamoinen + Niell mapping,
 *     calculate_ion_delay :865 — Klobuchar). What the factor's CONSTRUCTOR derives from the observation and the
 *     broadcast ephemeris (gnss_psr_dopp_factor.cpp:3-47: eph2pos / geph2pos / eph2svdt at the transmission time, the
 *     group delay, the URA scalings) is front-end work and crosses the boundary precomputed in gfbe_gnss_obs.
 * ------------------------------------------------------------------------------------------ */
/* (gfbe_gnss_obs and gfbe_gnss_state are defined with gfbe_state near the top of this header) */

/* One thread per factor. iono: the 8 Klobuchar parameters (latest_gnss_iono_params) or NULL (no ionosphere term).
 * frame_dt[i] = Headers[i+1] - Headers[i]. Outputs (any may be NULL):
 *   r_obs [n_obs][2]; J_obs [n_obs][2][18] with columns P_lower(3) V_lower(3) P_upper(3) V_upper(3) rcv_dt rcv_ddt yaw_enu_local
 *   anc_ecef(3) — the non-zero parts of the reference's 2 x {7, 9, 7, 9, 1, 1, 1, 3} blocks; like the reference the Jacobian leaves
 *   out the atmosphere and Sagnac derivatives and approximates d/d anc_ecef;
 *   r_dt_ddt [4][GFBE_WINDOW_SIZE] (k-major, the reference's insertion order; Jacobian = {-50, 50, -25 dt, -25 dt});
 *   r_smooth [GFBE_WINDOW_SIZE] (Jacobian = {w, -w});
 *   cost = 1/2 sum r^2 over the three families, summed in the reference's insertion order. */
gfbe_status gfbe_gnss_eval(gfbe_ctx *ctx, int32_t n_obs, const gfbe_gnss_obs *obs, const double *iono /*[8] or NULL*/,
                           const gfbe_state *state, const gfbe_gnss_state *gnss, const double *frame_dt /*[WINDOW_SIZE]*/,
                           double ddt_weight, double *r_obs, double *J_obs, double *r_dt_ddt, double *r_smooth, double *cost);

/* ------------------------------------------------------------------------------------------
 * a4/a5/a7/a9/a10  Factor evaluation on the device, block-CSR output (parity / inspection API).
 * Each evaluates residuals and TANGENT-space Jacobian blocks at the window's current state,
 * exactly what ceres::CostFunction::Evaluate + the manifold lift produce:
 *   ProjectionTwoFrameOneCamFactor::Evaluate  projectionTwoFrameOneCamFactor.cpp:43-151
 *   IMUFactor::Evaluate                       imu_factor.h:28-191
 *   WheelFactor::Evaluate                     wheel_factor.h:28-247
 *   MarginalizationFactor::Evaluate           marginalization_factor.cpp:344-392
 * robustify != 0 additionally applies the loss corrector of ResidualBlockInfo::Evaluate
 * (marginalization_factor.cpp:46-77) to the visual blocks.
 * Output layouts (host, row-major):
 *   vis_r [K][2]; vis_J [K][2][20]  columns = pose_i(6) pose_j(6) ex_cam(6) lambda(1) td(1)
 *   imu_r [n_imu][15]; imu_J [n_imu][15][30]  columns = pose_i(6) sb_i(9) pose_j(6) sb_j(9)
 *   wheel_r [n_wheel][6]; wheel_J [n_wheel][6][22] columns = pose_i(6) pose_j(6) ex_wheel(6) sx sy sw td_wheel
 *   prior_r [n]; prior_J is J0 itself (constant), not returned.
 * Any output pointer may be NULL to skip that factor family.
 * ------------------------------------------------------------------------------------------ */
gfbe_status gfbe_eval_factors(gfbe_ctx *ctx, const gfbe_window *win, int32_t robustify,
                              double *vis_r, double *vis_J,
                              double *imu_r, double *imu_J,
                              double *wheel_r, double *wheel_J,
                              double *prior_r, double *cost);

/* ------------------------------------------------------------------------------------------
 * a6/a8  Pre-integration on the device (the producers of gfbe_imu_preint / gfbe_wheel_preint):
 *   IntegrationBase::push_back/propagate/midPointIntegration   integration_base.h:39-167
 *   WheelIntegrationBase::push_back/propagate/midPointIntegration wheel_integration_base.h:41-178
 * n_interval independent intervals; interval k owns samples [offset[k], offset[k+1]).
 * Sample rows: IMU [dt ax ay az gx gy gz], wheel [dt vx vy vz gx gy gz]. first_* is the
 * (acc_0,gyr_0)/(vel_0,gyr_0) the interval was constructed with. noise: ACC_N GYR_N ACC_W GYR_W
 * (integration_base.h:30-36) / VEL_N_wheel GYR_N_wheel (wheel_integration_base.h:32-36).
 * ------------------------------------------------------------------------------------------ */
gfbe_status gfbe_preintegrate_imu(gfbe_ctx *ctx, int32_t n_interval, const int32_t *offset,
                                  const double *samples, const double *first_acc_gyr /*[n][6]*/,
                                  const double *lin_ba_bg /*[n][6]*/, const double noise[4],
                                  gfbe_imu_preint *out);
gfbe_status gfbe_preintegrate_wheel(gfbe_ctx *ctx, int32_t n_interval, const int32_t *offset,
                                    const double *samples, const double *first_vel_gyr /*[n][6]*/,
                                    const double *lin_sx_sy_sw_td /*[n][4]*/, const double noise[2],
                                    gfbe_wheel_preint *out);

/* ------------------------------------------------------------------------------------------
 * a1/a2/a3/a11/a12  The whole optimization() call.
 *   solve:        ceres::Solve with DENSE_SCHUR + DOGLEG + HuberLoss   estimator.cpp:2956-3379
 *   write-back:   double2vector() yaw/position re-anchoring            estimator.cpp:2501-2630
 *   marginalise:  MARGIN_OLD / MARGIN_SECOND_NEW                       estimator.cpp:3394-3693,
 *                 MarginalizationInfo::{preMarginalize,marginalize,getParameterBlocks}
 *                                                                      marginalization_factor.cpp:119-330
 * out_state / out_feature receive the re-anchored parameter blocks (what a second vector2double()
 * would produce, estimator.cpp:3398).
 * FAILURE CONTRACT (one rule for gfbe_solve_window, gfbe_solve_batch and gfbe_batch_download; INTEGRATION.md and
 * examples/estimator_binding.cpp say the same; tests/test_gpu_branches.py asserts both halves through this entry point):
 *   status <= GFBE_NUMERICAL_FAILURE (GFBE_OK, GFBE_NO_CONVERGENCE, GFBE_NUMERICAL_FAILURE): the solve RAN and every output has been
 *     written — out_state / out_feature / the in-out prior / summary. For GFBE_NUMERICAL_FAILURE they hold what the failed solve left:
 *     the last state its iterations accepted (the uploaded state if none was), re-anchored, and the prior marginalised at that state —
 *     exactly what the reference goes on with, since Estimator::optimization() never looks at Ceres' termination_type
 *     (estimator.cpp:3377-3379). A caller that wants "keep the previous state on failure" tests the status and discards the outputs.
 *   status >= GFBE_BAD_INPUT: the CALL failed (bad argument, no device, device error) — no output has been touched.
 * (SURVEY.md section 8b asked for "outputs untouched on failure"; that holds for failed calls. A numerically failed solve is not a failed
 *  call: leaving its outputs unwritten would make a batch's result depend on which of its windows failed.)
 * prior_out (may be NULL when margin_flag == GFBE_MARGIN_NONE) is IN/OUT like last_marginalization_info: when a marginalisation
 * ran — margin_flag != GFBE_MARGIN_NONE and the window is full (frame_count == GFBE_WINDOW_SIZE, estimator.cpp:3391) — it
 * receives the new prior with block ids already shifted (slot i -> i-1 for MARGIN_OLD; slot 10 -> 9 for SECOND_NEW), `valid`
 * = 0 if the marginalisation produced none (marginalization_factor.cpp:205-210). When NO marginalisation ran (a window that is
 * still filling up) it is left exactly as the caller passed it: pass the previous prior, or a structure whose `valid` the
 * caller has set to 0 — an uninitialised gfbe_prior stays uninitialised.
 * ------------------------------------------------------------------------------------------ */
gfbe_status gfbe_solve_window(gfbe_ctx *ctx, const gfbe_window *win, int32_t margin_flag,
                              gfbe_state *out_state, double *out_feature,
                              gfbe_prior *prior_out, gfbe_summary *summary);

/* Batched form: n independent windows, one launch sequence. Arrays are per-window. The windows do not influence one another:
 * summary[w].status is window w's own outcome, the return value the worst of them (GFBE_NO_CONVERGENCE or
 * GFBE_NUMERICAL_FAILURE: every window's outputs have been written, a failed window's from the last state its iterations
 * accepted; anything larger: the call itself failed and no output is valid). The same holds for gfbe_batch_download. */
gfbe_status gfbe_solve_batch(gfbe_ctx *ctx, int32_t n_window, const gfbe_window *const *win,
                             int32_t margin_flag, gfbe_state *out_state, double *const *out_feature,
                             gfbe_prior *const *prior_out, gfbe_summary *summary);

/* Device-resident form used for throughput measurement: upload once, (re)solve many times.
 *   (Window order: outputs and gfbe_batch_feature_count always use the caller's indices; inside, a batch that is solved in parts
 *   holds its windows sorted by size — a window's result does not depend on its place.)
 *   gfbe_batch_upload   packs the windows into pinned staging memory and ENQUEUES one host-to-device copy plus the
 *                       preparation kernels on the context's private copy stream. It returns when the
 *                       caller's gfbe_window structures have been read — they may be reused at once — but the copy may still
 *                       be in flight: gfbe_batch_solve waits for it on the device (an event), never on the host
 *   gfbe_batch_solve    resets every window to its uploaded state, then runs the full
 *                       optimization() sequence on the ctx stream; asynchronous, no host sync
 *   gfbe_batch_download runs the gather kernel and ONE device-to-host copy on the context's private download stream
 *                       (after the solve's event), waits for that copy only, and unpacks. prior_out: in/out as described
 *                       above (untouched for windows whose marginalisation did not run)
 * ------------------------------------------------------------------------------------------ */
typedef struct gfbe_batch gfbe_batch;
gfbe_status gfbe_batch_upload(gfbe_ctx *ctx, int32_t n_window, const gfbe_window *const *win,
                              gfbe_batch **batch);
gfbe_status gfbe_batch_solve(gfbe_ctx *ctx, gfbe_batch *batch, int32_t margin_flag);
gfbe_status gfbe_batch_download(gfbe_ctx *ctx, gfbe_batch *batch, gfbe_state *out_state,
                                double *const *out_feature, gfbe_prior *const *prior_out,
                                gfbe_summary *summary);
void gfbe_batch_free(gfbe_ctx *ctx, gfbe_batch *batch);

/* Measurement hooks (bench.py): name/launch count/accumulated GPU milliseconds of each kernel
 * family since the last reset, measured with hipEvents on the ctx stream when profiling is on. */
gfbe_status gfbe_profile_enable(gfbe_ctx *ctx, int32_t on);
int32_t gfbe_profile_count(const gfbe_ctx *ctx);
gfbe_status gfbe_profile_get(const gfbe_ctx *ctx, int32_t i, const char **name, int64_t *launches,
                             double *total_ms, double *algorithmic_bytes);
void gfbe_profile_reset(gfbe_ctx *ctx);
/* Where the calling thread spent the LAST gfbe_batch_upload(_tables) and the LAST gfbe_batch_download of this context, milliseconds on the
 * host's clock: out4 = [ upload: scan + packing into pinned memory | upload: the rest of the call (allocation, enqueue of the copy and the
 * expansion kernels) | download: WAITING for the device (the solve, the gather kernel, the copy) | download: unpacking into the caller's
 * structures ]. A pipeline that keeps several batches in flight is host-bound when the first, second and fourth add up to its time per
 * batch, and device-bound when the third is what is left (bench.py's end_to_end block reports both). */
void gfbe_host_times(const gfbe_ctx *ctx, double *out4);

/* Diagnostics: phase time stamps (10 ns ticks) of the dense solve kernel for window w of a batch. */
gfbe_status gfbe_debug_timing(gfbe_ctx *ctx, gfbe_batch *batch, int32_t w, double *out32);
/* Diagnostics: a per-window vector of the LAST linearisation of a solved batch (GFBE_DENSE_DIM doubles each; waits for the solve):
 * which = 0 Gauss-Newton step y of the dense block (Jacobi-scaled), 1 Cauchy direction v, 2 Jacobi scaling s, 3 gradient g.
 * Lets a test compare the two factorisations of gfbe_options.solve_kernel entry by entry. which = 1000 + r: row r of the assembled
 * normal equations H (lower triangle valid); 2000 + r: row r of the Schur term E (73 entries; r = 73: its gradient share). */
gfbe_status gfbe_debug_vector(gfbe_ctx *ctx, gfbe_batch *batch, int32_t w, int32_t which, double *out);

/* Multi-GPU landmark sharding: when set, the library calls
 * fn(user, device_ptr, n_doubles, hip_stream) once per linearisation on the packed partial reduced
 * system [S | g | cost ...] (and on a few small exchange blocks per iteration); the callee performs an in-place sum
 * all-reduce (RCCL) on that stream and returns 0, or a non-zero code: the solve that enqueued the call then returns
 * GFBE_DEVICE_ERROR with the code in gfbe_last_error — un-reduced partial sums are never handed back as a result.
 * A hook installed with world_size 1 still runs the sharded launch sequence (every all-reduce is then the identity): that is
 * how a single-GPU box exercises the whole path. fn == NULL switches the sharding off.
 * Limits of the sharded mode: no solver time cap (max_solver_time_in_seconds is refused: the ranks' clocks would stop them at
 * different iterations), no device-resident feature tables, and ONE mu retry per linearisation — a factorisation that fails
 * again at the larger mu ends the window with GFBE_NUMERICAL_FAILURE, where the unsharded solve keeps raising mu up to max_mu. */
typedef int32_t (*gfbe_allreduce_fn)(void *user, void *device_ptr, int64_t n_doubles, void *hip_stream);
gfbe_status gfbe_set_allreduce(gfbe_ctx *ctx, gfbe_allreduce_fn fn, void *user, int32_t rank,
                               int32_t world_size);

#ifdef __cplusplus
}
#endif
#endif /* GFBE_H_ */

// gfbe_device.h — HBM data layout of a batch of sliding windows + kernel launch prototypes.
// Shared by the host side (gfbe_host.cpp) and the kernels (gfbe_kernels.hip, gfbe_marg.hip).
// See DESIGN.md §3 for the rationale; names follow the reference's domain (window, landmark,
// factor, prior), file:line citations are to Ground-Fusion++/vins_estimator/src/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <vector>

#include "../../include/gfbe.h"

#ifndef GFBE_LIN_SMALL_THREADS
#define GFBE_LIN_SMALL_THREADS 256     // (four waves, one per SIMD: each has the whole register file — arch + acc — for the scalar algebra of an inertial /
                                       //  wheel factor; eight waves = 256 registers each spilled 508 bytes per lane: 1.244 against 1.212 ms host to host)
#endif
#ifndef GFBE_LIN_SMALL_KS
#define GFBE_LIN_SMALL_KS 4     // shares of a visual tile's observation steps (k = share, share + KS, ...): the waves of the tile's workgroup in k_lin_small
                                // (round 5; until then five workgroups per tile: 4 -> 5 measured 1.244 -> 1.227 ms, 10: 1.316), workgroups in k_vis_split
#endif

// Small batches (< 32 windows, no landmark sharding): launches of one iteration merged on a single window's latency path. Bit 0:
// k_schur + k_visblock_small in one launch; bit 1: k_step and the dense half of k_candidate by the workgroup of k_lm_step that
// finishes last, the landmark half of k_candidate by k_lin_small<1>'s tile workgroups; bit 2: k_accept by the workgroup of
// k_lin_small<1> that finishes last; bit 3: the marginalisation's linearisation (k_vis_split<2> + k_dense) in one launch, its pair sums
// and its Schur partial in another. Same code, same order of every sum: the results do not change by a bit.
#ifndef GFBE_FUSE_SMALL
#define GFBE_FUSE_SMALL 15
#endif

// GFBE_DIAG = 1: the diagnostics build (libgfbe_diag.so; backend.build_native(diag=True)). Only that build reads the environment
// hooks of the measurement / test scripts (GFBE_POISON_UNCLEARED, GFBE_VIS_FULL, GFBE_DEBUG_UPLOAD) and accepts the kernel
// time-stamp / ablation macros (GFBE_ABLATE, GFBE_*_STAMP). The shipped library has none of them: no environment variable can
// change what it computes or writes.
#ifndef GFBE_DIAG
#define GFBE_DIAG 0
#endif
#if !GFBE_DIAG && (defined(GFBE_ABLATE) || defined(GFBE_KVIS_STAMP) || defined(GFBE_LIN_STAMP) || defined(GFBE_CHOL_STAMP) || \
                   defined(GFBE_CHAIN_STAMP) || defined(GFBE_BIG_STAMP) || defined(GFBE_LDLT_STAMP))
#error "kernel time stamps / ablations are diagnostics: build with -DGFBE_DIAG=1"
#endif

namespace gfd {

#if GFBE_DIAG
inline const char *diag_getenv(const char *name) { return getenv(name); }
#else
inline const char *diag_getenv(const char *) { return nullptr; }
#endif

// ---- dimensions -----------------------------------------------------------------------------
enum {
  NF = GFBE_NFRAMES,          // 11 frames
  ND = GFBE_DENSE_DIM,        // 246 tangent dims of the dense (pose / IMU / extrinsic / ground plane / GNSS) blocks: every row stride
  NC = GFBE_CORE_DIM,         // 187 of them without the GNSS blocks: BatchDev::nu is NC for a batch without GNSS windows, ND with
  NV = 73,                    // leading dims visual factors touch: 11 poses * 6 + ex_cam 6 + td 1
  NA = 259,                   // ambient doubles of gfbe_state
  MAXOBS = 10,                // factors per landmark (n_obs - 1)
  REC = 42,                   // doubles per visual block-CSR record: r(2) + J(2 x 20)   = 336 B
  NPAIR = NF * NF,            // (imu_i, imu_j) pair slots, index i * 11 + j
  VP_STRIDE = 336,            // fused visual partial of one (tile, observation step): T0 16x16 | T1 16x4 | T2 4x4
  XS_LD = 21,                 // LDS row stride of the [J | r] panel (20 + 1)
  NVP = 80,                   // NV padded to 5 tiles of 16; column 73 carries the landmark gradient
  SCHUR_TILES = 15,           // upper-triangular 16x16 tile pairs of the 80 x 80 block
  SCHUR_STRIDE = SCHUR_TILES * 256, // one partial per (window, group of start frames): 15 dense tiles
  SCHUR_GROUPS = 4,           // throughput batches: start frames {0, 1}, {2}, {3, 4, 5}, {6 .. 10}: one k_schur workgroup and one partial each
  LM_TILE = 64,               // landmarks per workgroup in the landmark kernels
  SCHUR_CHUNK = 64,           // landmarks per Schur work item
  IMU_PART = 30 * 30 + 30 + 2,      // J^T J, J^T r, cost
  WHEEL_PART = 22 * 22 + 22 + 2,
  MAX_IMU = 10, MAX_WHEEL = 10, MAX_PLANE = 10,
  PLANE_PART = 16 * 16 + 16 + 2,    // J^T J, J^T r, cost, candidate cost of one PlaneFactor (columns pose_i 6, ex_wheel 6, plane_R 3, plane_Z 1)
  ANCHOR_PART = 6 * 6 + 6 + 2,
  MAX_BATCH_PARTS = 16,       // upper bound of gfbe_options.split_batch (every part beyond the first owns a pair of streams)
  BATCH_SPLIT_MIN_B = 128,    // batches at least this big are uploaded as two halves solved side by side (gfbe_options.split_batch)
  LIN_SMALL_KS = GFBE_LIN_SMALL_KS,   // shares of a landmark tile's observation steps (dealt round-robin): waves of its workgroup in k_lin_small, workgroups in k_vis_split
  LIN_SMALL_THREADS = GFBE_LIN_SMALL_THREADS,    // k_lin_small: threads per workgroup (LIN_SMALL_KS waves per visual tile, all of them for an inertial / wheel / prior item)
  DENSE_SPLIT_MIN_B = 32,     // batches at least this big: k_dense_raw (lane = window) + aux-stream overlap of the dense factors
  VS_BLOCKS = 2 * (NF - 1),     // blocks of the split visual assembly of small batches (k_visblock_small)
  LIOW_WGS = 8, LIOW_PART = 32,   // LiDAR factors of a window: workgroups per window, doubles per partial (21 H | 6 g | cost | candidate cost)
  XCHG = 8,                   // doubles per (window, rank) row of a scalar exchange block
  HC = 13,                    // common part of a landmark's H_pl row: pose_i(6) ex(6) td(1)
  PAIR_CONST_DOUBLES = 75,    // sizeof(PairConst) / 8 (gfbe_factors.h; static_assert in gfbe_kernels.hip)
  VPY = 28,                   // fused visual partial of a window with constant extrinsic / td: upper triangle of the 7 x 7 [Y r]^T [Y r] (visual_lin_y)
  VPY_STRIDE = 32,            // doubles per (tile, observation step) slot of vis_part in that layout
  // GNSS inside the window (gfbe_gnss_solve.hip)
  GN_C = 124,                 // compact list of the tangent dims the GNSS factors reach in the solve: position of the 11 poses (33), velocity of
                              // the 11 speed-bias blocks (33), rcv_dt (44), rcv_ddt (11), anc_ecef (3); the yaw is constant there
  GN_M = 26,                  // local dims of the marginalisation set: P0 V0 P1 V1 (3 each) rcv_dt[0][4] rcv_ddt[0] rcv_dt[1][4] rcv_ddt[1] yaw anc(3)
  GN_MPART = GN_M * GN_M + GN_M + 2,   // J^T J, J^T r, cost of the frame-0 GNSS factors at the re-anchored state
  BIG_LD = 272                // k_solve_big: row stride of the factor in its global scratch (n + 1 <= 247 rows of <= 256 doubles; NOT a power of two:
                              // 16 tile rows 2 KB apart would sit in two L2 channels — measured 6.6 us per four-step operand load instead of ~1)
};

// tangent offsets (same convention as the ABI's block order)
__host__ __device__ inline int T_POSE(int k) { return 6 * k; }
enum { T_EX = 66, T_TD = 72, T_EXW = 172, T_SX = 178, T_SY = 179, T_SW = 180, T_TDW = 181,
       T_PLR = 182,          // para_plane_R: three tangent dims + the quaternion's 4th slot (185), which only ever exists in the prior
       T_PLZ = 186,
       T_ANC = 187, T_YAW = 190 };   // para_anc_ecef (3), para_yaw_enu_local; then rcv_dt[11][4] and rcv_ddt[11]
__host__ __device__ inline int T_DT(int i, int k) { return 191 + 4 * i + k; }
__host__ __device__ inline int T_DDT(int i) { return 235 + i; }
__host__ __device__ inline int T_SB(int k) { return 73 + 9 * k; }
// ambient offsets inside gfbe_state (195 doubles)
__host__ __device__ inline int A_POSE(int k) { return 7 * k; }
__host__ __device__ inline int A_SB(int k) { return 77 + 9 * k; }
enum { A_EX = 176, A_EXW = 183, A_IX = 190, A_TD = 193, A_TDW = 194, A_PLR = 195, A_PLZ = 199,
       A_DT = 200, A_DDT = 244, A_YAW = 255, A_ANC = 256 };   // gfbe_state::gnss: rcv_dt[11][4], rcv_ddt[11], yaw_enu_local, anc_ecef[3]

__host__ __device__ inline int blk_tan(int id) {
  if (id < GFBE_BLK_SB0) return T_POSE(id);
  if (id < GFBE_BLK_EX_CAM) return T_SB(id - GFBE_BLK_SB0);
  if (id >= GFBE_BLK_RCV_DDT0) return T_DDT(id - GFBE_BLK_RCV_DDT0);
  if (id >= GFBE_BLK_RCV_DT0) return T_DT(0, id - GFBE_BLK_RCV_DT0);
  switch (id) {
    case GFBE_BLK_ANC_ECEF: return T_ANC;
    case GFBE_BLK_YAW_ENU: return T_YAW;
    case GFBE_BLK_EX_CAM: return T_EX;
    case GFBE_BLK_EX_WHEEL: return T_EXW;
    case GFBE_BLK_SX: return T_SX;
    case GFBE_BLK_SY: return T_SY;
    case GFBE_BLK_SW: return T_SW;
    case GFBE_BLK_TD: return T_TD;
    case GFBE_BLK_PLANE_R: return T_PLR;
    case GFBE_BLK_PLANE_Z: return T_PLZ;
    default: return T_TDW;
  }
}
__host__ __device__ inline int blk_amb(int id) {
  if (id < GFBE_BLK_SB0) return A_POSE(id);
  if (id < GFBE_BLK_EX_CAM) return A_SB(id - GFBE_BLK_SB0);
  if (id >= GFBE_BLK_RCV_DDT0) return A_DDT + (id - GFBE_BLK_RCV_DDT0);
  if (id >= GFBE_BLK_RCV_DT0) return A_DT + (id - GFBE_BLK_RCV_DT0);
  switch (id) {
    case GFBE_BLK_ANC_ECEF: return A_ANC;
    case GFBE_BLK_YAW_ENU: return A_YAW;
    case GFBE_BLK_EX_CAM: return A_EX;
    case GFBE_BLK_EX_WHEEL: return A_EXW;
    case GFBE_BLK_SX: return A_IX;
    case GFBE_BLK_SY: return A_IX + 1;
    case GFBE_BLK_SW: return A_IX + 2;
    case GFBE_BLK_TD: return A_TD;
    case GFBE_BLK_PLANE_R: return A_PLR;
    case GFBE_BLK_PLANE_Z: return A_PLZ;
    default: return A_TDW;
  }
}
__host__ __device__ inline int blk_gsize(int id) {
  if (id < GFBE_BLK_SB0) return 7;
  if (id < GFBE_BLK_EX_CAM) return 9;
  if (id == GFBE_BLK_EX_CAM || id == GFBE_BLK_EX_WHEEL) return 7;
  if (id == GFBE_BLK_PLANE_R) return 4;
  if (id == GFBE_BLK_ANC_ECEF) return 3;
  return 1;
}
// (para_plane_R: local size 4 like MarginalizationInfo::localSize gives every block that is not 7 wide — the solve keeps its
//  4th tangent slot inactive, the prior carries a column for it)
__host__ __device__ inline int blk_lsize(int id) { const int g = blk_gsize(id); return g == 7 ? 6 : g; }

// ---- per-window descriptor (constant during a solve) ----------------------------------------
struct WinDesc {
  int L, K;                  // landmarks, visual factors
  int lm_off;                // first landmark slot of this window in the per-landmark arrays (64-aligned)
  int lm_slots;              // padded landmark slots (multiple of LM_TILE; start-frame groups are tile aligned)
  int rec_off;               // first visual record
  int vel_off;               // compact observation upload (BatchDev::obs_compact): first entry of fvel (the records of start frame 0 come first)
  int n_tiles;               // lm_slots / LM_TILE
  int tile_off;              // first entry of this window in tile_start[] (start frame of each tile)
  int n_imu, n_wheel;
  int imu_off, wheel_off;    // into the batch-wide preintegration arrays
  int imu_frame[MAX_IMU], wheel_frame[MAX_WHEEL];
  int imu_of_frame[NF], wheel_of_frame[NF];   // factor whose first frame is f, or -1
  int prior_n, prior_nblk;   // 0 => no prior
  int prior_blk_id[GFBE_MAX_PRIOR_BLOCKS], prior_blk_size[GFBE_MAX_PRIOR_BLOCKS], prior_blk_idx[GFBE_MAX_PRIOR_BLOCKS];
  int prior_x0_off[GFBE_MAX_PRIOR_BLOCKS];
  int prior_map[ND];         // tangent dim -> prior column (-1 if none)
  int frame_count;
  int pair_begin[NPAIR + 1]; // records of pair (i,j) are [pair_begin[i*11+j], pair_begin[i*11+j+1]) relative to rec_off
  int sf_tile_begin[NF + 1]; // tiles of start frame s are [sf_tile_begin[s], sf_tile_begin[s+1])
  unsigned char act[ND];     // tangent dim is in the reduced program (not constant, touched by a factor)
  unsigned char blk_free[GFBE_BLK_COUNT];
  unsigned char ex_cam_mask[6], ex_wheel_mask[6];
  unsigned char pad_[3];
  int lio_n, lio_off, lio_frame, lio_pad;   // LiDAR point-to-plane factors on pose lio_frame (gfbe_lio_block)
  double lio_sqrt_info, lio_huber;
  int n_plane, use_anchor;                  // PlaneFactors on poses 0 .. n_plane - 1 (0: none); PoseAnchorFactor on pose 0
  double plane_noise_inv[3], anchor_pose[7], anchor_sqrt_info;
  // GNSS (gfbe_window.gnss_*): gnss_factors = gnss_ready && !lowspeed (estimator.cpp:2969-2984, 3239)
  int gnss_ready, gnss_factors, n_gnss, gnss_off;   // observations [gnss_off, gnss_off + n_gnss) of BatchDev::gnss_obs, sorted by frame
  int gnss_frame_begin[NF + 1];                     // observations of frame i: [gnss_frame_begin[i], gnss_frame_begin[i + 1]) relative to gnss_off
  int gnss_has_iono, gnss_pad;
  double gnss_iono[8], gnss_frame_dt[GFBE_WINDOW_SIZE], gnss_ddt_weight;
};

// ---- per-window solver state (mutated by kernels; mirrors TrustRegionMinimizer + DoglegStrategy)
struct WinCtl {
  int cur;                   // index (0/1) of the current parameter buffer; 1-cur is the candidate
  int iter;                  // trust-region iterations started
  int done;                  // solve finished
  int reuse;                 // DoglegStrategy::reuse_ : GN/Cauchy vectors valid, only the radius changed
  int have_step;             // k_step produced a valid candidate this iteration
  int num_successful, termination, status, invalid_steps, lin_fail;
  int n_clamped;
  int lin_retry;             // landmark sharding: the factorisation failed at the previous mu; the retry pass factorises with c.mu and the rebuilt, all-reduced E
  int lb;                    // speculative linearisation (BatchDev::spec): which set of the linearisation's outputs is current (0: the first) — next to
                             // done / reuse: the kernels between two linearisations read it with them, before their first load of the set
  int marg_ran;              // the marginalisation kernels handled this window in the last solve
  double radius, mu, cost, cand_cost, x_norm, cand_norm2, step_amb2;
  double G2, N2, gy, vHv, vHy, yHy, alpha, grad_max;
  double c1, c2, step_norm, model_change;
  double initial_cost;
  double cost_history[16];
  unsigned char accepted[16];
  long long t_start, t_solved, t_marg;   // device wall clock (100 MHz ticks): k_reset, k_reanchor, end of the marginalisation (0: none)
  double sw_mu[2];                       // BatchDev::spec: the mu the weights lm_sw of set 0 / 1 were formed with (k_schur uses them when it is the
                                         // window's mu — an accepted step's new mu is known to the pass that linearises its candidate — and forms them itself otherwise)
};

// ---- batch: all device pointers ----------------------------------------------------------------
struct BatchDev {
  int B;
  int tot_lm;                 // total padded landmark slots
  int max_tiles;              // max n_tiles over windows
  int max_sf_tiles;           // most landmark tiles one start frame of one window has (k_vis_chunk's grid)
  WinDesc *desc;              // [B]
  WinCtl *ctl;                // [B]
  gfbe_options opt;
  // dense parameters: x0 = uploaded state, x[2] = current/candidate, xout = re-anchored result
  double *x0, *x, *xout;      // [B][NA], [B][2][NA], [B][NA]
  // pose-pair constants of the visual factors (PairConst, gfbe_factors.h) of the three states a window can be evaluated at:
  // [B][3][NPAIR][PC_DOUBLES] — slot 0 / 1 follow x[0] / x[1], slot 2 the re-anchored state xout. Written by the single-wave
  // kernels that produce those states (k_reset, k_candidate, k_reanchor); every visual tile then just loads its <= 10 records.
  double *pc;
  // landmarks (internal order: sorted by start frame, tile aligned). SoA over tot_lm slots.
  int *lm_info;               // start | m << 8 | const << 16 | valid << 24
  int *lm_abi;                // ABI feature_index of the slot (-1 for padding)
  double *lm_pts;             // [6][tot_lm]: pix piy piz vix viy td_i
  double *lm_obs;             // [MAXOBS][5][tot_lm]: pjx pjy vjx vjy td_j
  int *lm_rec;                // [MAXOBS][tot_lm]: record position (relative to rec_off) of factor k
  double *fvel;               // [.][3] compact observation upload: vjx vjy td_j of the factors whose landmark starts in frame 0 (WinDesc::vel_off)
  int obs_compact;            // 1: fobs is [tot_rec][2] — the observations already shifted to the window's td by the host (upload_one)
  double *fobs;               // [tot_rec][5] host upload only: pjx pjy vjx vjy td_j of every factor in record (pair-major) order; k_expand
                              // scatters them into lm_obs / lm_rec (the ELL rows never cross PCIe: 40 B per factor instead of ~100)
  double *lam0, *lam;         // [tot_lm], [2][tot_lm]
  double *lm_Hll, *lm_gl;     // [tot_lm]
  double *lm_hC;              // [HC][tot_lm]
  double *lm_hP;              // [MAXOBS][6][tot_lm]
  double *lm_sl, *lm_yl, *lm_vl;   // Jacobi scale, GN component, Cauchy direction component
  double *lm_sw;              // [tot_lm] sqrt of the landmark's weight in the Schur term at the current linearisation (k_vis<0> -> k_schur)
  // visual block-CSR records, pair-major: [tot_rec][REC]
  double *rec;
  int tot_rec;
  int *tile_start;            // start frame per tile (batch-wide list)
  // factor inputs
  gfbe_imu_preint *imu; gfbe_wheel_preint *wheel;
  double *imu_sqrt, *wheel_sqrt;     // [n][225], [n][36]
  double *prior_J0, *prior_r0, *prior_x0, *prior_H;   // [B][ND*ND], [B][ND], [B][PRIOR_X0], [B][ND*ND]
  // partial results
  double *pair_part;          // [B][NF][VP_STRIDE]   marginalisation: X^T X of the pose pairs (0, j), slot j (sum of vis_part over the tiles of start frame 0)
  double *vis_part;           // [B][max_tiles][MAXOBS][VP_STRIDE]  X^T X of the 64 factors of one tile at one step, X = [J | r]
  double *schur_part;         // [B][schur_groups][SCHUR_STRIDE]  sum over the landmarks of one group of start frames
  double *schur_part2;        // the second set's (k_linschur<SPEC>: the landmark elimination of the candidate's linearisation; nullptr without it)
  int linschur;               // throughput batch with constant extrinsic / td, not sharded: k_linschur (evaluation + landmark elimination in one
                              // launch) in the place of k_vis<0, false> + k_schur (gfbe_options.merge_lin_schur)
  int schur_groups;           // SCHUR_GROUPS for throughput batches; small batches: 2 NF (two workgroups per start frame), NF when sharded
  double *imu_part, *wheel_part;     // [B][MAX_IMU][IMU_PART], [B][MAX_WHEEL][WHEEL_PART]
  double *plane_part, *anchor_part;  // [B][MAX_PLANE][PLANE_PART], [B][ANCHOR_PART]   (only read for windows with n_plane / use_anchor)
  int any_plane;                     // some window of the batch has plane or anchor factors (else their workgroups are not launched)
  int prior_n_max;                   // largest prior dimension of the batch (k_prior_tp stages J0 in LDS when it fits)
  double *prior_g;            // [B][ND + 2]  J0^T r, cost
  // ---- speculative linearisation (small batches, gfbe_options.speculative_linearization): the pass that evaluates the candidate of an
  // iteration LINEARISES there — into the second set of the linearisation's outputs below; k_accept's tail flips WinCtl::lb when the
  // step is accepted, and the next iteration starts at the Schur elimination (lin_view: the set WinCtl::lb names). A rejected step
  // leaves the current set alone, exactly what DoglegStrategy's reuse needs.
  int spec;
  double *lm_Hll2, *lm_gl2, *lm_hC2, *lm_hP2, *lm_sw2, *vis_part2, *imu_part2, *wheel_part2, *plane_part2, *anchor_part2, *prior_g2, *lio_part2, *gnss_J2, *gnss_r2, *gnss_cost2;
  // ---- landmark sharding over ranks (gfbe_set_allreduce): tile t of a window belongs to rank t % world.
  int rank, world;
  int sharded;                      // an all-reduce hook is installed (gfbe_set_allreduce): the launch sequence with the exchange blocks, also for world == 1
  int test_fail_chol_iter;          // gfbe_options.test_fail_chol_iter (test hook): the first factorisation of that iteration "fails"
  int vis_full;                     // some window of the batch has a free camera extrinsic or td: the visual kernels form those Jacobian blocks
  double *xa, *xb, *xc;       // [B][world][XCHG] scalar exchange rows (own row written, the others zeroed, then sum all-reduce):
                              //   xa: visual cost of the linearisation point; xb: landmark shares of the dogleg scalars;
                              //   xc: candidate cost / step norms
  double *vis_H;              // [B][73][74]  visual block of the normal equations + gradient column (k_visblock)
  double *vis_Hs;             // small batches only (else nullptr): [B][vs_blocks][73][74], one block per (start frame, thread group), summed by k_assemble
  int vs_blocks;              // VS_BLOCKS when some window has a free extrinsic / td (13/20-column partials); 1 otherwise: block 0 is the whole visual block (visblock_y)
  double *raw_imu, *raw_wheel; // [MAX_IMU][15 + 450][B], [MAX_WHEEL][6 + 132][B]  un-whitened residuals / Jacobians (k_dense_raw), window-minor
  int *asm_tab;               // [asm_n][4]  window-independent assembly table, owned by the context (asm_tables_build): every entry of the lower
                              // triangle over the dims in use, or — a batch without GNSS blocks whose priors hold no speed-bias block
                              // but SpeedBias[0] — only the entries some factor of such a window can reach (~43 %: the others of H stay
                              // the zeros of the upload)
  int asm_n;
  double *zero;               // a few zeros: target of the "absent contribution" loads of k_assemble
  double *tile_cost;          // [B][max_tiles]   visual cost partials (current linearisation)
  double *vis_contrib;        // [B][max_tiles][MAXOBS][16][64] small batches: per-step contributions to Hll, gl, hC, cost (k_lin_small)
  int *tile_cnt;              // [B][max_tiles]   arrival counter of the tile's LIN_SMALL_KS workgroups in k_vis_split (zero between launches)
  int *win_cnt;               // [B][2]           arrival counters of a window's k_lm_step / k_lin_small<1> workgroups (GFBE_FUSE_SMALL; zero between launches)
  double *tile_cand;          // [B][max_tiles][4] candidate: cost, |x-xc|^2, |xc|^2, pad
  double *tile_gram;          // [B][max_tiles][8] landmark parts of G2 N2 gy vHv vHy yHy gradmax
  double *dense_cand;         // [B][4] dense-factor candidate cost, |x-xc|^2, |xc|^2
  int tot_lio;                // LiDAR factors of the whole batch (0: the kernels are not launched)
  double *lio;                // [tot_lio][8]  p(3) n(3) offset weight
  double *lio_part;           // [B][LIOW_WGS][LIOW_PART]
  // assembled system
  // GNSS inside the solve (gfbe_gnss_solve.hip): observations, their residuals / Jacobians at the current linearisation, costs
  int any_gnss, tot_gnss, gnss_max_obs;   // some window has gnss_ready; observations of the whole batch / of its largest window
  gfbe_gnss_obs *gnss_obs;          // [tot_gnss]
  double *gnss_J, *gnss_r;          // [tot_gnss][36], [tot_gnss][2]
  double *gnss_cost;                // [B][2]  cost of the GNSS factors at the linearisation point / at the candidate
  double *gnss_marg;                // [B][GN_MPART]  the frame-0 GNSS factors of MARGIN_OLD at the re-anchored state (cost < 0: none)
  int solve_big;                    // some window has active GNSS dims: the batch takes k_solve_big (factor in global memory, n <= 246)
  double *solveS;                   // k_solve_big: [B][BIG_LD * BIG_LD] scaled system / its factor
  int nu;                     // tangent dims in use: NC (no window of the batch has GNSS blocks) or ND
  double *H, *g;              // [B][ND*ND], [B][ND]  unscaled J^T J, J^T r of the dense block
  double *E, *eg;             // [B][NV*NV], [B][NV]  sum_l w_l h_l h_l^T, sum_l w_l h_l gl  (unscaled h)
  double *sys_pack;           // landmark sharding only: [B][sys_pack_doubles] the packed partial system the all-reduce sums (k_sys_pack)
  double *Er;                 // landmark sharding only: [B][NV*NV + NV] E | eg rebuilt for a larger mu by every rank from its own tiles
                              // (zeros for the windows that do not retry), summed by one all-reduce before the retry pass of k_solve
  double *sp, *Dp, *gts, *vp, *yp, *step;   // [B][ND] each
  // k_solve_chain (gfbe_solve.hip): the speed-bias blocks are eliminated before the dense factorisation
  size_t solve_scratch_stride;   // doubles per window of solveY / solveS (dead after the solve: scratch of the marginalisation's eigen-decomposition)
  double *solveY;             // [B][96][104]  Yr rows of the chain, transposed (written by the elimination, read back by the back-substitution)
  int solve_ntile;            // dense tiles (16 x 16, lower triangle incl. the right-hand side row) of the largest window: sizes the dynamic LDS
  int solve_mono;             // some window's prior couples a speed-bias block other than SpeedBias[0]: the whole batch takes the monolithic k_solve
  int solve_tw;               // the chain eliminated from both ends (k_solve_chain_tw, one workgroup per CU): small batches, where a window's latency counts
  int solve_wide;             // a batch with GNSS dims (solve_big) whose windows all fit k_solve_chain_wide: the chain kernel with nine tile columns of dense dims
  int asm_legacy;             // diagnostics build only (GFBE_ASM_LEGACY=1 at upload): the assembly's entry loops in their form of rounds 4-6 (gfbe_kernels.hip: ASM_TP_ON)
  // debug / inspection outputs (gfbe_eval_factors)
  double *dbg_imu, *dbg_wheel, *dbg_prior;  // [B][MAX_IMU][15*31], [B][MAX_WHEEL][6*23], [B][ND]
  // marginalisation
  double *mA, *mb;            // [B][ND*ND], [B][ND]
  double *mJ0, *mr0;          // [B][ND*ND], [B][ND]
  double *mV;                 // [B][ND*ND] eigenvectors scratch
  int *mmeta;                 // [B][4 + 3*GFBE_MAX_PRIOR_BLOCKS]: valid, n, n_blocks, pad, ids, sizes, idx
  int marg_nmax;              // host-known upper bound of the new priors' size over the batch
  double *mx0;                // [B][PRIOR_X0]
  double *timing;             // [B][32] phase time stamps of k_solve (wall_clock64, 10 ns ticks; diagnostics)
  // ---- result hand-over (k_gather): everything gfbe_batch_download returns, packed for ONE device-to-host copy
  //   dl_fix [B][DL_FIX doubles]: WinCtl | xout | prior meta (ints) | prior x0 | prior r0     dl_feat [sum L]: para_Feature in ABI order
  //   dl_J0: the new priors' J0, n x n each, at the host-known offsets dl_j0_off[w] (upper bounds of n from the block tables)
  double *dl_fix, *dl_feat, *dl_J0;
  int *dl_feat_off;           // [B + 1]
  long long *dl_j0_off;       // [B + 1]
};
enum { DL_CTL = (sizeof(WinCtl) + 7) / 8, DL_META = (4 + 3 * GFBE_MAX_PRIOR_BLOCKS + 1) / 2,
       DL_OFF_X = DL_CTL, DL_OFF_META = DL_OFF_X + NA, DL_OFF_X0 = DL_OFF_META + DL_META, DL_OFF_R0 = DL_OFF_X0 + GFBE_NFRAMES * 16 + 32,
       DL_FIX = DL_OFF_R0 + ND };

enum { PRIOR_X0 = GFBE_NFRAMES * 16 + 32 };

// ---- kernel launchers (gfbe_kernels.hip / gfbe_marg.hip) -------------------------------------------
hipError_t kernels_init_device();   // per-device kernel attributes (k_solve's dynamic LDS); called by gfbe_create
hipError_t marg_init_device();      // same for k_marg
hipError_t gnss_init_device();      // same for k_gnss
hipError_t dense_init_device();     // same for k_prior_tp
hipError_t lin_small_init_device(); // same for k_lin_small
void launch_ingest_small(const void *src_host, void *dst, size_t bytes, void *z0, size_t z0_bytes, void *z1, size_t z1_bytes,
                         const double *jsrc_host, double *jdst, int nrow, size_t row, size_t jstride, hipStream_t s);
void launch_prep(const BatchDev &d, hipStream_t s);
void launch_upload_small(const BatchDev &d, int with_expand, hipStream_t s);   // small batches: k_expand + k_prep + k_prep_prior + k_asm_table in one launch
void launch_expand(const BatchDev &d, hipStream_t s);                 // host upload: fobs -> lm_obs / lm_rec
void launch_gather(const BatchDev &d, int margin_flag, hipStream_t s);   // results -> dl_fix / dl_feat / dl_J0
void launch_reset(const BatchDev &d, hipStream_t s);
// mode 0: linearise at the current parameters (writes records, landmark sums, cost partials)
// mode 1: candidate cost only   mode 2: linearise the marginalisation set at xout (start frame 0 only)
void launch_vis(const BatchDev &d, int mode, hipStream_t s, int write_records = 0, int spec = 0);
void launch_pair(const BatchDev &d, int marg, hipStream_t s);
void launch_pair_schur_marg(const BatchDev &d, hipStream_t s);   // small batches: k_pairsum (marg) + k_schur (marg) in one launch
void launch_lin_small(const BatchDev &d, int mode, hipStream_t s, int fuse = 0);   // fuse (mode 1): bit 1 candidate tiles first, bit 2 k_accept last
void launch_dense_factors(const BatchDev &d, int mode, int debug_out, hipStream_t s, int spec = 0);
void launch_schur(const BatchDev &d, int marg, hipStream_t s, int with_visblock = 0);   // with_visblock: k_schur_visblock_small
void launch_linschur(const BatchDev &d, int spec, int gate_mu, hipStream_t s);            // BatchDev::linschur: k_vis<0, false> + k_schur in one launch
void launch_visblock(const BatchDev &d, hipStream_t s);
void launch_lio_window(const BatchDev &d, int mode, hipStream_t s, int spec = 0);
void launch_assemble(const BatchDev &d, hipStream_t s);
// the context's assembly tables (built once per context): full[ND (ND + 1) / 2] and compact[*n_compact], both ordered by the larger dim
hipError_t asm_tables_build(int **full, int **compact, int *n_compact, hipStream_t s);
// context accessors for the translation units that do not see the gfbe_ctx definition (gfbe_host.cpp)
hipStream_t ctx_stream(gfbe_ctx *c);
int ctx_device(const gfbe_ctx *c);
void ctx_set_error(gfbe_ctx *c, const char *msg);
void *ctx_scratch(gfbe_ctx *c, size_t bytes);   // grow-only device scratch of the context (one caller at a time), nullptr on failure
void *ctx_scratch_pinned(gfbe_ctx *c, size_t bytes);   // its pinned host mirror (grow-only, same rules)
void launch_xchg_gram(const BatchDev &d, hipStream_t s);
void launch_xchg_cand(const BatchDev &d, hipStream_t s);
void launch_lam_mask(const BatchDev &d, hipStream_t s);
void launch_marginalize_partials(const BatchDev &d, hipStream_t s, bool dense_elsewhere = false);
void launch_marginalize_finish(const BatchDev &d, int flag, hipStream_t s);
void launch_solve(const BatchDev &d, hipStream_t s, int retry_pass = 0);
int solve_chain_tiles(const unsigned char *act);
bool solve_chain_tw_fits(int ntile);                // k_solve_chain_tw holds every row of Yr in LDS: up to five tile columns of the dense part
bool solve_chain_wide_fits(int n_dense_max);   // a batch with GNSS dims on k_solve_chain_wide (nine tile columns of dense dims)
size_t solve_chain_scratch_doubles();              // doubles of BatchDev::solveY per window   // dense 16 x 16 tiles k_solve_chain needs for a window with these active dims
void launch_rebuild_E_shard(const BatchDev &d, hipStream_t s);
void launch_sys_pack(const BatchDev &d, int dir, hipStream_t s);   // landmark sharding: pack (0) / unpack (1) the partial system around its all-reduce
size_t sys_pack_doubles_host(int nu, int world);
void launch_lm_step(const BatchDev &d, hipStream_t s, int fuse = 0);   // fuse: k_step + the dense blocks of k_candidate by the last workgroup of a window
void launch_step(const BatchDev &d, hipStream_t s);
void launch_candidate(const BatchDev &d, hipStream_t s);
void launch_accept(const BatchDev &d, hipStream_t s, int spec = 0);
void launch_reanchor(const BatchDev &d, hipStream_t s);
void launch_marginalize(const BatchDev &d, int flag, hipStream_t s);
// mode 0: GNSS factors at the current parameters, added to H / g (after launch_assemble); 1: candidate cost; 2: the frame-0 factors at
// the re-anchored state for MARGIN_OLD
void launch_gnss(const BatchDev &d, int mode, hipStream_t s, int sub = 0);
bool lin_small_takes_gnss(const BatchDev &d, int mode);   // k_lin_small's candidate passes evaluate the batch's GNSS factors themselves (gfbe_gnss_item.h)

// ---- device-resident feature tables (gfbe_ftab.hip; shared with the batch upload that reads them)
enum { FT_NOBS = GFBE_WINDOW_SIZE + 1, FT_OW = 8, FT_BINS = NF * 8,
       FT_LAY_BIN = 1, FT_LAY_GRP = 1 + FT_BINS, FT_LAY_PAIR = 1 + FT_BINS + NF, FT_LAY_STRIDE = 1 + FT_BINS + NF + NPAIR + 1 };   // bins: (start frame, factors 3..10) of a landmark
struct FtabDev {
  int W, F;             // tables, capacity (features per table)
  int *count;           // [W]
  int *id[2], *start[2], *nobs[2], *eflag[2], *sflag[2];
  double *depth[2], *obs[2], *td[2];   // obs [W][F][FT_NOBS][FT_OW], td [W][F][FT_NOBS]
  int *keep;            // [W][F] scratch: survivor flag / erased observation (+2) / match index
  double *ndepth;       // [W][F] scratch: edited depth
  int *ids_scratch;     // [W][F] flagged ids of check_outliers (ascending)
  int *cnt_scratch;     // [W]
  int *hist, *layout;   // [W][FT_BINS + 2], [W][FT_LAY_STRIDE]: landmark histogram / slot layout of gfbe_batch_upload_tables
  int *err;             // [W] sticky error flags (capacity / more than FT_NOBS observations)
  gfbe_ftab_options opt;
};
// landmarks (features with >= 4 observations) of tables [w0, w0 + n): counts[w][0] = L, [1] = K, [2 + bin] per (start, factors)
void launch_ftab_count(const FtabDev &T, int cur, int w0, int n, int *counts, hipStream_t s);
// fills the landmark arrays of a batch from the tables (layout tables from the host: per window FT_BINS slot bases, NF group
// bases, NPAIR + 1 pair_begin, lm_off); slot_of [n][F] receives the slot of landmark k (list order) for the download
void launch_ftab_pack(const FtabDev &T, int cur, int w0, int n, const BatchDev &d, const int *layout, int *slot_of, hipStream_t s);

}  // namespace gfd

struct gfbe_ftab {
  gfd::FtabDev d;
  int cur = 0;
  std::vector<void *> allocs;
  // argument staging of the table operations: one device chunk + its pinned host mirror, grown on demand and kept (a
  // hipMalloc / hipFree pair per argument cost more than the kernels: 0.4 ms per call measured)
  char *stage_d = nullptr, *stage_h = nullptr;
  size_t stage_cap = 0;
  // the operations that return nothing (triangulate, setDepth, the erasing ones) do not wait for the device: their arguments go
  // through a ring of small staging slots, a slot is reused when the event recorded behind its operation has passed
  enum { RING = 8, RING_SLOT = 64 << 10 };
  char *ring_d = nullptr, *ring_h = nullptr;
  hipEvent_t ring_ev[RING] = {};
  bool ring_used[RING] = {};
  int ring_next = 0;
  // gfbe_batch_upload_tables reads the tables on the context's COPY stream, beside the solves queued on the main one: it waits for
  // ev_ops (recorded on the main stream behind every table operation) and leaves ev_read behind its last reader; the next table
  // operation waits for that one (read_pending)
  hipEvent_t ev_ops = nullptr, ev_read = nullptr;
  bool read_pending = false;
};

// gfbe_kernels.hip — hand-written HIP kernels (gfx950 / MI355X, FP64) for the sliding-window solve.
//
// One launch sequence runs a whole batch of windows; grid.y = window. Every kernel early-exits for
// windows whose trust-region state says it has nothing to do, so the host enqueues a fixed
// sequence and never synchronises inside a solve.
//
// What each kernel stands in for (reference = Ground-Fusion++/vins_estimator/src):
//   k_vis          ProjectionTwoFrameOneCamFactor::Evaluate (factor/projectionTwoFrameOneCamFactor.cpp:43-151)
//                  + loss Corrector (factor/marginalization_factor.cpp:46-77) + the landmark row of J^T J
//   k_pair         J^T J / J^T r of the visual factors of one (imu_i, imu_j) pose pair, from the
//                  block-CSR records (what Ceres' BlockSparseMatrix + SchurEliminator hold)
//   k_dense        IMUFactor / WheelFactor / MarginalizationFactor ::Evaluate
//                  (factor/imu_factor.h:28-191, factor/wheel_factor.h:28-247, factor/marginalization_factor.cpp:344-392)
//   k_schur        SchurEliminator: sum_l H_pl H_ll^-1 H_lp over the 1-D inverse-depth blocks
//   k_assemble     reduced camera system assembly (fixed-order, owner-computes => deterministic)
//   k_solve        DoglegStrategy::ComputeStep dense part: Jacobi scaling, mu-regularised Cholesky, GN step
//   k_lm_step      back-substitution of the eliminated landmarks + their share of the dogleg scalars
//   k_step         DoglegStrategy::ComputeTraditionalDoglegStep + TrustRegionMinimizer bookkeeping
//   k_candidate    x (+) delta  (factor/pose_local_parameterization.cpp:12-27)
//   k_accept       TrustRegionMinimizer: step acceptance, radius update, termination tests
//   k_reanchor     Estimator::double2vector gauge fix (estimator/estimator.cpp:2501-2555)
#include "gfbe_devutil.h"
#include "gfbe_factors.h"
#include "gfbe_gnss_item.h"
#include <type_traits>

namespace gfd {

#ifndef GFBE_ABLATE
#define GFBE_ABLATE 0   // timing ablations of k_vis (tools/diag_ablate.sh); 0 in every shipped build
#endif
#ifndef GFBE_KVIS_EARLY
#define GFBE_KVIS_EARLY 1   // k_vis: the prefetched observation of the next step is waited for BEFORE this step's stores are issued
#endif
#ifndef GFBE_FUSE_CAND
#define GFBE_FUSE_CAND 1   // throughput batches: the landmark half of k_candidate at the head of the cost pass (k_vis<1>)
#endif
#ifndef GFBE_PCS_ONE_ROUND
#define GFBE_PCS_ONE_ROUND 1      // k_vis_chunk: the pair records of a wave's start frame in one round of loads (0: one dependent round trip per pair, rounds 4-6)
#endif
#ifndef GFBE_ASM_U
#define GFBE_ASM_U 4       // k_visasm: entries of H a thread has in flight
#endif
#ifndef GFBE_ASM_TP
#define GFBE_ASM_TP 29     // k_visasm (end of round 6; every output keeps its bits): 1 = the entries of H by asm_H_tp,
                           // 4 = the descriptor tables requested before the visual block is gathered, 8 = the gradient's dense terms by gather_g_dense_tp
                           // (k_assemble's as well), 16 = the gather of the visual block's partials (visblock_y); 0 = rounds 4-6
#endif
// which form runs: the build's choice — and, in the diagnostics build, the older form on request (BatchDev::asm_legacy, GFBE_ASM_LEGACY=1 at upload:
// tests/test_gpu_uncleared.py compares H entry by entry and every output's bits between the two)
#if GFBE_DIAG
#define ASM_TP_ON(d, bit) (((GFBE_ASM_TP) & (bit)) && !(d).asm_legacy)
#else
#define ASM_TP_ON(d, bit) (((GFBE_ASM_TP) & (bit)) != 0)
#endif
#ifndef GFBE_DENSE_TP
#define GFBE_DENSE_TP 1    // throughput batches: k_dense_tp (matrix-core whitening / J^T J, four windows per workgroup) instead of k_dense<false>
#endif
#ifndef GFBE_KVIS_WAVES
#define GFBE_KVIS_WAVES 3   // k_vis<0, false>: waves per SIMD the register allocation aims at (measured: 4 — 128 VGPRs, a 12-byte spill — 152 us per
                            // 512 windows against 149 at 3: the kernel waits for memory, not for a free wave slot; -ffp-contract=fast: no difference either)
#endif
#ifndef GFBE_KVIS_STAMP
#define GFBE_KVIS_STAMP 0   // diagnostics build (tools/diag_variants.py): phase time stamps of one wave of k_vis<0> into d.timing
#endif

// =============================================================================================
// k_prep: once per upload. sqrt_info of every IMU / wheel factor (imu_factor.h:73, wheel_factor.h:85,
// hoisted out of Evaluate: SURVEY App. A.5) and H_prior = J0^T J0.
// =============================================================================================
// k_prep, grid (B, PREP_FACT_WGS): the workgroups factorise the covariances (one WAVE per factor: the 20 waves take the <= 10
// inertial and <= 10 wheel factors of a window side by side); k_prep_prior, grid (B, PREP_PRIOR_WGS), shares H_prior.
//
// sqrt_info_from_cov (gfbe_factors.h: inverse by partially pivoted LU, then a lower Cholesky of the inverse, written transposed)
// by the 64 lanes of one wave. Every entry goes through exactly the operations, in exactly the order, of the one-thread form
// (row updates of an elimination step are independent of each other; a column of the inverse is one lane's own substitution;
// a column of the Cholesky factor is a dot product per row): the results are bit-identical to it (tests/test_gpu_parity.py
// compares them with the host build of the one-thread form through the oracle), at ~10 us instead of ~150 us per factor — the
// one-lane version was the longest kernel on the upload path of a single window.
// lu, inv, U: n * n doubles of LDS each, private to the wave. Returns 0 on success (wave-uniform).
#define PREP_WSYNC() do { __threadfence_block(); __builtin_amdgcn_wave_barrier(); } while (0)
template <int n>
__device__ __forceinline__ int sqrt_info_from_cov_wave(const double *cov, double *lu, double *inv, double *U, const int lane) {
  for (int e = lane; e < n * n; e += 64) { lu[e] = cov[e]; U[e] = 0.0; }
  int perm = lane;                       // lane i < n holds perm[i]
  PREP_WSYNC();
  for (int k = 0; k < n; k++) {
    // the FIRST row of maximal |lu[i][k]|, i >= k (the scan of the one-thread form replaces the pivot on a strictly larger value)
    const bool cand = lane >= k && lane < n;
    const double a = cand ? fabs(lu[lane * n + k]) : -1.0;
    double best = a;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = fmax(best, __shfl_xor(best, o, 64));
    if (best == 0.0) return 1;
    const int piv = __ffsll((long long)__ballot(cand && a == best)) - 1;
    if (piv != k) {
      if (lane < n) { const double t = lu[k * n + lane]; lu[k * n + lane] = lu[piv * n + lane]; lu[piv * n + lane] = t; }
      const int pk = __shfl(perm, k, 64), pp = __shfl(perm, piv, 64);
      if (lane == k) perm = pp;
      if (lane == piv) perm = pk;
      PREP_WSYNC();
    }
    if (lane > k && lane < n) lu[lane * n + k] = lu[lane * n + k] / lu[k * n + k];
    PREP_WSYNC();
    const int m = n - k - 1;
    for (int e = lane; e < m * m; e += 64) {
      const int i = k + 1 + e / m, j = k + 1 + e % m;
      const double f = lu[i * n + k];
      lu[i * n + j] -= f * lu[k * n + j];
    }
    PREP_WSYNC();
  }
  if (lane < n) {                        // column `lane` of the inverse
    const int c = lane;
    double y[n], x[n];                   // (n is a compile-time constant and the loops are unrolled: registers, not scratch)
#pragma unroll
    for (int i = 0; i < n; i++) {
      double sum = (__shfl(perm, i, 64) == c) ? 1.0 : 0.0;
#pragma unroll
      for (int j = 0; j < i; j++) sum -= lu[i * n + j] * y[j];
      y[i] = sum;
    }
#pragma unroll
    for (int i = n - 1; i >= 0; i--) {
      double sum = y[i];
#pragma unroll
      for (int j = i + 1; j < n; j++) sum -= lu[i * n + j] * x[j];
      x[i] = sum / lu[i * n + i];
      inv[i * n + c] = x[i];
    }
  }
  PREP_WSYNC();
  for (int j = 0; j < n; j++) {          // lower Cholesky of inv, written transposed (upper) into U
    double dd = inv[j * n + j];
    for (int k = 0; k < j; k++) dd -= U[k * n + j] * U[k * n + j];
    if (!(dd > 0.0)) return 2;
    dd = sqrt(dd);
    if (lane == j) U[j * n + j] = dd;
    if (lane > j && lane < n) {
      double sum = inv[lane * n + j];
      for (int k = 0; k < j; k++) sum -= U[k * n + lane] * U[k * n + j];
      U[j * n + lane] = sum / dd;        // L(i,j) stored at U(j,i)
    }
    PREP_WSYNC();
  }
  return 0;
}
enum { PREP_FACT_WGS = 5, PREP_PRIOR_WGS = 8, PREP_ROWS = 16 };
__device__ __forceinline__ void prep_body(const BatchDev &d, const int w, const int by) {
  const WinDesc &ds = d.desc[w];
  __shared__ double work[4][3 * 225];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  double *lu = work[wave], *inv = lu + 225, *U = inv + 225;
  for (int f = by * 4 + wave; f < ds.n_imu + ds.n_wheel; f += 4 * PREP_FACT_WGS) {      // wave-uniform
    const bool imu = f < ds.n_imu;
    const int k = imu ? f : f - ds.n_imu, n = imu ? 15 : 6;
    const double *cov = imu ? d.imu[ds.imu_off + k].covariance : d.wheel[ds.wheel_off + k].covariance;
    double *out = imu ? d.imu_sqrt + (size_t)(ds.imu_off + k) * 225 : d.wheel_sqrt + (size_t)(ds.wheel_off + k) * 36;
    const int rc = imu ? sqrt_info_from_cov_wave<15>(cov, lu, inv, U, lane) : sqrt_info_from_cov_wave<6>(cov, lu, inv, U, lane);
    for (int e = lane; e < n * n; e += 64) out[e] = rc ? nan("") : U[e];
    PREP_WSYNC();
  }
}
__global__ __launch_bounds__(256) void k_prep(BatchDev d) { prep_body(d, blockIdx.x, blockIdx.y); }
// H_prior = J0^T J0, lower triangle mirrored (PREP_PRIOR_WGS workgroups per window; its own kernel: the factorisations above take
// 256 VGPRs, this one a fraction — eight workgroups per CU instead of one). J0 goes through LDS sixteen rows at a time (coalesced
// loads, every row read once per pass); a thread keeps PREP_MAXE entries in registers across the row panels and adds the products
// in row order — the same sums as an entry-by-entry loop over global memory. A prior larger than 8 x 256 x PREP_MAXE entries of
// the triangle (n > 127) takes more than one pass over J0.
enum { PREP_MAXE = 4 };
__device__ __forceinline__ void prep_prior_body(const BatchDev &d, const int w, const int by) {
  const WinDesc &ds = d.desc[w];
  const int t = threadIdx.x;
  const int n = ds.prior_n;
  if (n == 0) return;
  const double *J0 = d.prior_J0 + (size_t)w * ND * ND;
  double *Hp = d.prior_H + (size_t)w * ND * ND;
  __shared__ double panel[PREP_ROWS * ND];
  const int ntri = n * (n + 1) / 2, stride = PREP_PRIOR_WGS * 256;
  for (int pass0 = 0; pass0 < ntri; pass0 += PREP_MAXE * stride) {     // (uniform trip count: barriers inside)
    const int base = pass0 + by * 256 + t;
    double acc[PREP_MAXE];
    int ei[PREP_MAXE], ej[PREP_MAXE];
#pragma unroll
    for (int q = 0; q < PREP_MAXE; q++) {
      acc[q] = 0.0; ei[q] = -1; ej[q] = 0;
      const int e = base + q * stride;
      if (e < ntri) tri_decode(e, ei[q], ej[q]);       // ej <= ei
    }
    for (int r0 = 0; r0 < n; r0 += PREP_ROWS) {
      const int nr = min((int)PREP_ROWS, n - r0);
      __syncthreads();
      for (int e = t; e < nr * n; e += 256) panel[e] = J0[(size_t)r0 * n + e];
      __syncthreads();
#pragma unroll
      for (int q = 0; q < PREP_MAXE; q++) {
        if (ei[q] < 0) continue;
        double sacc = acc[q];
        for (int rr = 0; rr < nr; rr++) sacc += panel[rr * n + ei[q]] * panel[rr * n + ej[q]];
        acc[q] = sacc;
      }
    }
#pragma unroll
    for (int q = 0; q < PREP_MAXE; q++) {
      if (ei[q] < 0) continue;
      Hp[(size_t)ei[q] * n + ej[q]] = acc[q];
      Hp[(size_t)ej[q] * n + ei[q]] = acc[q];
    }
  }
}
__global__ __launch_bounds__(256) void k_prep_prior(BatchDev d) { prep_prior_body(d, blockIdx.x, blockIdx.y); }

// Pose-pair constants of the visual factors at state X (gfbe_state layout) for all 55 pairs i < j: ONE wave (lanes 0..63 of the
// calling workgroup; `sp` = 12 PoseRT of LDS), one pair per lane. Called by the kernels that produce a state; the visual
// tiles (36 workgroups per window and pass) load the records instead of rebuilding them from the poses each.
static_assert(PC_DOUBLES == PAIR_CONST_DOUBLES, "PairConst layout");
// (second half of pair_consts_of_state: the NF + 1 poses are staged in sp, the caller's barrier has passed)
__device__ __forceinline__ void pair_consts_from_staged(double *pc_out, const PoseRT *sp, int lane) {
  if (lane < NF) {     // the per-frame constants of visual_lin_y in the (unused) diagonal slot (i, i)
    const FrameConst f = make_frame_const(sp[lane], sp[NF], sp[0]);
    double *o = pc_out + (size_t)(lane * NF + lane) * PC_DOUBLES;
    const double *src = (const double *)&f;
#pragma unroll
    for (int q = 0; q < FC_DOUBLES; q++) o[q] = src[q];
  }
  if (lane < NF * (NF - 1) / 2) {
    int i = 0, rem = lane;
    while (rem >= NF - 1 - i) { rem -= NF - 1 - i; i++; }
    const int j = i + 1 + rem;
    const PairConst p = make_pair_const(sp[i], sp[j], sp[NF]);
    double *o = pc_out + (size_t)(i * NF + j) * PC_DOUBLES;
    const double *src = (const double *)&p;
#pragma unroll
    for (int q = 0; q < PC_DOUBLES; q++) o[q] = src[q];
  }
}
__device__ __forceinline__ void pair_consts_of_state(const double *X, double *pc_out, PoseRT *sp, int lane) {
  if (lane < NF) sp[lane] = make_pose(X + A_POSE(lane));
  if (lane == NF) sp[NF] = make_pose(X + A_EX);
  __syncthreads();
  pair_consts_from_staged(pc_out, sp, lane);
}

// =============================================================================================
// k_reset: start of every solve — restore the uploaded state, reset the trust-region bookkeeping.
// =============================================================================================
__global__ __launch_bounds__(256) void k_reset(BatchDev d) {
  const int w = blockIdx.y;
  const WinDesc &ds = d.desc[w];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  __shared__ PoseRT sp_reset[NF + 1];
  if (blockIdx.x == 0) {
    pair_consts_of_state(d.x0 + (size_t)w * NA, d.pc + (size_t)w * 3 * NPAIR * PC_DOUBLES, sp_reset, threadIdx.x);   // (block-uniform branch: the barrier inside is safe)
    for (int q = threadIdx.x; q < NA; q += blockDim.x) {
      const double v = d.x0[(size_t)w * NA + q];
      d.x[((size_t)w * 2) * NA + q] = v;
      d.x[((size_t)w * 2 + 1) * NA + q] = v;
    }
    if (threadIdx.x == 0) {
      WinCtl c;
      c.cur = 0; c.iter = 0; c.done = 0; c.reuse = 0; c.have_step = 0;
      c.num_successful = 0; c.termination = 0; c.status = GFBE_NO_CONVERGENCE; c.invalid_steps = 0; c.lin_fail = 0;
      c.n_clamped = 0; c.lin_retry = 0;
      c.radius = d.opt.initial_trust_region_radius; c.mu = GF_MIN_MU; c.cost = 0; c.cand_cost = 0; c.x_norm = 0;
      c.cand_norm2 = 0; c.step_amb2 = 0;
      c.G2 = c.N2 = c.gy = c.vHv = c.vHy = c.yHy = c.alpha = c.grad_max = 0;
      c.c1 = c.c2 = c.step_norm = c.model_change = 0; c.initial_cost = 0;
      for (int i = 0; i < 16; i++) { c.cost_history[i] = 0; c.accepted[i] = 0; }
      c.t_start = (long long)wall_clock64(); c.t_solved = 0; c.t_marg = 0; c.marg_ran = 0; c.lb = 0; c.sw_mu[0] = -1.0; c.sw_mu[1] = -1.0;
      d.ctl[w] = c;
    }
  }
  if (t < ds.lm_slots) {
    const double v = d.lam0[ds.lm_off + t];
    d.lam[ds.lm_off + t] = v;
    d.lam[(size_t)d.tot_lm + ds.lm_off + t] = v;
  }
}

// =============================================================================================
// k_vis: one lane = one landmark, looping over its observations (all lanes of a tile share the
// start frame, so at step k the whole wave evaluates the same pose pair). SE(3) state tiles are
// staged once per workgroup in LDS as (t, R).
// MODE 0: linearise at the current point; MODE 1: candidate cost; MODE 2: marginalisation set
// (landmarks with start_frame 0, estimator.cpp:3498-3531) at the re-anchored state.
// =============================================================================================
// KS > 1 (small batches): the observation steps of a tile are independent pose pairs — their partials, their
// H_pl blocks — except for three per-landmark running sums (Hll, gl, hC) and the cost. KS shares split a tile — waves of ONE workgroup
// in k_lin_small (WS below, round 5), workgroups in k_vis_split (the launch sequence of the profiling mode) —, share
// kq takes the steps k = kq, kq + KS, ..., every step's CONTRIBUTION to the running sums goes to a scratch array, and the first wave
// behind the workgroup's barrier (the workgroup that arrives last at an atomic counter per tile in the other form; nobody waits for
// anybody there) adds them up in step order: the same additions in the same order as the one-wave loop of the throughput path, so
// the results stay bit-identical to it.
#ifndef GFBE_LIN_STAMP
#define GFBE_LIN_STAMP 0
#endif
#ifndef GFBE_LIN_STAMP_MODE
#define GFBE_LIN_STAMP_MODE 0      // which launch mode of k_lin_small the diagnostics build stamps
#endif
#define VC_STRIDE 16      // doubles per (step, lane) in vis_contrib: Hll, gl, hC[<= 13], cost
// a landmark's candidate inverse depth x + s_l (c1 v_l + c2 y_l) and its shares of |x - x_cand|^2, |x_cand|^2 (k_candidate's landmark half)
__device__ __forceinline__ double candidate_lm(const double lam, const double sl, const double vl, const double yl, const double c1, const double c2,
                                               const bool free_lm, double &d2, double &n2) {
  double lc = lam;
  d2 = 0.0; n2 = 0.0;
  if (free_lm) {
    lc = lam + sl * (c1 * vl + c2 * yl);
    d2 = (lam - lc) * (lam - lc);
    n2 = lc * lc;
  }
  return lc;
}
// SPEC (MODE 0, speculative linearisation of a small batch: BatchDev::spec): the linearisation AT THE CANDIDATE of the iteration, in
// the place of its cost pass — d is the view of the set the candidate's linearisation goes to (lin_view), the tile's cost goes
// where the cost pass leaves it.
// head (passes at the candidate): the tile first forms the candidate inverse depths of its landmarks — candidate_tile's work (the landmark
// half of k_candidate), its loads requested with the evaluation's own and the lane's result kept in its register.
// WS (KS > 1; k_lin_small since the end of round 5): the KS shares of a tile are WAVES of one workgroup — kq = the wave, `wsm` = KS panels
// of dynamic LDS — that meet at a workgroup barrier, where the first wave adds the contributions up: the workgroups of the other form met
// at an atomic counter behind two device-scope fences, and the last one read the others' contributions across the XCDs (~8 us of a
// ~17 us linearisation launch of a single window).
template <int MODE, bool FULL, int KS = 1, bool SPEC = false, bool WS = false>
__device__ __forceinline__ void vis_body(const BatchDev &d, int write_records, const int w, const int tile, const int kq = 0, const bool head = false,
                                         double *wsm = nullptr) {
  static_assert(!WS || KS > 1, "wave shares are shares");
  static_assert(!SPEC || MODE == 0, "the speculative pass is a linearisation");
  constexpr bool CAND = (MODE == 1) || SPEC;      // evaluated at the candidate state
  const WinDesc &ds = d.desc[w];
  if (tile >= ds.n_tiles || !TILE_OWNED(d, tile)) return;
  const WinCtl &c = d.ctl[w];
  if (MODE == 0 && !SPEC && (c.done || c.reuse)) return;
  if (CAND && (c.done || !c.have_step)) return;
  // start frame of the tile from the descriptor's own table (scalar loads next to n_tiles) instead of tile_start[]: one dependent
  // memory round trip less before the pair constants can be fetched
  int sframe = 0;
#pragma unroll
  for (int q = 1; q < NF; q++) sframe += (tile >= ds.sf_tile_begin[q]) ? 1 : 0;
  if (MODE == 2 && sframe != 0) return;
  const int buf = CAND ? 1 - c.cur : c.cur;
  const double *X = (MODE == 2) ? d.xout + (size_t)w * NA : d.x + ((size_t)w * 2 + buf) * NA;
  const double *lamv = d.lam + (size_t)buf * d.tot_lm;

#if GFBE_KVIS_STAMP
  double *kstamp = d.timing + (size_t)d.B * 32;      // (the extra block behind the windows' own slots)
  const bool kst = MODE == 0 && w == d.B / 2 && tile == 0 && threadIdx.x == 0 && c.iter == 0;
#define KSTAMP(i) do { if (kst) kstamp[i] = (double)wall_clock64(); } while (0)
#else
#define KSTAMP(i) do { } while (0)
#endif
#if GFBE_LIN_STAMP
  unsigned long long *lst = (unsigned long long *)(d.timing + (size_t)d.B * 32);
  const bool lstamp = KS > 1 && MODE == 0 && tile == 0 && (threadIdx.x & 63) == 0 && (WS || threadIdx.x == 0);
#define LSTAMP(i) do { if (lstamp && kq == 0) lst[i] = wall_clock64(); } while (0)
#define LSTAMP_ANY(i) do { if (lstamp) lst[i] = wall_clock64(); } while (0)
#else
#define LSTAMP(i) do { } while (0)
#define LSTAMP_ANY(i) do { } while (0)
#endif
  LSTAMP(14);
  KSTAMP(0);
  // pair (sframe, j) constants, j = sframe+1 .. 10, from the records the state's producer left (k_reset / k_candidate /
  // k_reanchor): one coalesced load instead of 12 quaternion -> matrix conversions, 10 triple products and two barriers per
  // tile. A window with constant extrinsic and td stages only the members its Jacobian blocks need (PairConstR).
  // YM (round 4): the linearisation of a window with constant extrinsic and td sums [Y r]^T [Y r] (7 x 7; visual_lin_y in
  // gfbe_factors.h) instead of [J_i J_j r]^T [J_i J_j r] (13 x 13): both rows of a factor fit ONE 16-wide matrix-core tile (16
  // instructions per step instead of 32), the per-factor work is G and two cross products instead of four 3 x 3 products, and a
  // (tile, step) partial is 28 doubles instead of 157; k_visasm applies the pair's 6 x 12 transform T once per pair. The
  // marginalisation set (MODE 2) keeps the 13-column panel: k_marg reads its pair sums in the full layout.
  constexpr bool YM = (MODE == 0 && !FULL);
  typedef typename std::conditional<FULL, PairConst, typename std::conditional<MODE == 2, PairConstR, PairConstY>::type>::type PCT;
  constexpr int PCW = (MODE == 1) ? 12 : sizeof(PCT) / sizeof(double), XLD = FULL ? XS_LD : 17;   // (the cost pass uses Tm and u: the first twelve doubles)
  __shared__ PCT pcs[NF];
  __shared__ FrameConst fcs;
  __shared__ double xs_own[(MODE == 1 || WS) ? 1 : LM_TILE * XLD];   // one [J | r] row of each of the wave's 64 factors (YM: both [Y r] rows)
  double *const xs = WS ? wsm + (size_t)kq * (LM_TILE * XLD) : xs_own;
  // ---- the wave's second level of loads, ALL requested before the pair constants are staged (round 5): the landmark's slot data, its
  // inverse depth and the first observation row need nothing but the descriptor — a wave used to fetch them one dependent round trip
  // after the other BEHIND the staging barrier (prologue 6.3 us of a ~22 us wave under load)
  const int lane = WS ? (threadIdx.x & 63) : threadIdx.x;
  const int slot = ds.lm_off + tile * LM_TILE + lane;
  const size_t TL = d.tot_lm;
  // (tdc: the observations of this batch are stored shifted to the windows' constant td — expand_body — and need neither their
  //  velocities nor their own td here; compile-time for the 7 x 7 linearisation, a launch-uniform flag for the cost pass)
  const bool tdc = YM || (MODE == 1 && !d.vis_full);
  const int nobq = tdc ? 2 : 5;
  const int info = d.lm_info[slot];
  const double pix = d.lm_pts[0 * TL + slot], piy = d.lm_pts[1 * TL + slot], piz = d.lm_pts[2 * TL + slot];
  double vix = 0.0, viy = 0.0, tdi = 0.0;
  if (!tdc) { vix = d.lm_pts[3 * TL + slot]; viy = d.lm_pts[4 * TL + slot]; tdi = d.lm_pts[5 * TL + slot]; }
  const bool chead = CAND && head;
  double lam = chead ? d.lam[(size_t)c.cur * TL + slot] : lamv[slot];
  double c_sl = 0.0, c_vl = 0.0, c_yl = 0.0;
  if (chead) { c_sl = d.lm_sl[slot]; c_vl = d.lm_vl[slot]; c_yl = d.lm_yl[slot]; }
  const double td = X[A_TD];
  // the observation of step k + 1 is fetched while step k is evaluated (rows beyond a track's length hold whatever the memory held —
  // lm_obs is not cleared at upload — and are used below the track's length only); the first one here, unconditionally
  double nob[5] = {0.0, 0.0, 0.0, 0.0, 0.0};      // (tdc: rows 2..4 are not loaded — zero velocity, the window's td: the shift is an exact zero)
  {
    const double *ob = d.lm_obs + (size_t)kq * 5 * TL + slot;
#pragma unroll
    for (int q = 0; q < 5; q++) if (q < nobq) nob[q] = ob[q * TL];
  }
  // KS > 1 (round 5): the observations of the share's LATER steps too (k = kq + KS, kq + 2 KS: a share has two steps with five shares per
  // tile, three with four) — a lone wave per SIMD waited a memory round trip per step for the one requested a step ahead (and, vmcnt
  // counting loads and stores in one queue, for the contribution stores issued before it; a load inside the loop makes its back edge
  // wait for the step's stores as well). No load is left inside the loop.
  constexpr int NLATER = KS > 1 ? (MAXOBS + KS - 1) / KS - 1 : 0;
  double nobl[NLATER > 0 ? NLATER : 1][5];
#pragma unroll
  for (int u = 0; u < NLATER; u++) {
#pragma unroll
    for (int q = 0; q < 5; q++) nobl[u][q] = 0.0;
    if (kq + (u + 1) * KS < MAXOBS) {
      const double *ob = d.lm_obs + (size_t)(kq + (u + 1) * KS) * 5 * TL + slot;
#pragma unroll
      for (int q = 0; q < 5; q++) if (q < nobq) nobl[u][q] = ob[q * TL];
    }
  }
  {
    const double *src = d.pc + (((size_t)w * 3 + (MODE == 2 ? 2 : buf)) * NPAIR + sframe * NF) * PC_DOUBLES;
    // (WS: the records dealt over the workgroup's waves)
#if GFBE_PCS_ONE_ROUND
    // (end of round 6, as in k_vis_chunk: every record this wave stages is requested before the first is written — the loop below is a
    //  dependent round trip per pair; a slot beyond the window's last frame re-reads a valid entry and writes nothing)
    {
      constexpr int JS = WS ? KS : 1, NJ = (NF - 1 + JS - 1) / JS, NR = (PCW + LM_TILE - 1) / LM_TILE;
      double rec[NJ][NR];
#pragma unroll
      for (int jj = 0; jj < NJ; jj++)
#pragma unroll
        for (int r = 0; r < NR; r++)
          rec[jj][r] = src[(size_t)min(sframe + 1 + (WS ? kq : 0) + jj * JS, NF - 1) * PC_DOUBLES + min(lane + r * LM_TILE, PCW - 1)];
      const double fcv = src[(size_t)sframe * PC_DOUBLES + min(lane, (int)FC_DOUBLES - 1)];
#pragma unroll
      for (int jj = 0; jj < NJ; jj++) {
        const int j = sframe + 1 + (WS ? kq : 0) + jj * JS;
#pragma unroll
        for (int r = 0; r < NR; r++) if (j < NF && lane + r * LM_TILE < PCW) ((double *)&pcs[min(j, NF - 1)])[lane + r * LM_TILE] = rec[jj][r];
      }
      if (YM && (!WS || kq == 0) && lane < FC_DOUBLES) ((double *)&fcs)[lane] = fcv;
    }
#else
    for (int j = sframe + 1 + (WS ? kq : 0); j < NF; j += (WS ? KS : 1))
      for (int q = lane; q < PCW; q += LM_TILE) ((double *)&pcs[j])[q] = src[(size_t)j * PC_DOUBLES + q];     // (the whole record is 75 doubles: two rounds)
    if (YM && (!WS || kq == 0) && lane < FC_DOUBLES) ((double *)&fcs)[lane] = src[(size_t)sframe * PC_DOUBLES + lane];
#endif
  }
  __syncthreads();
  const double sq = d.opt.vis_sqrt_info, delta = d.opt.huber_delta;
  KSTAMP(1);
  LSTAMP(15);
  const bool valid = (info >> 24) & 1;
  const int m = valid ? ((info >> 8) & 0xff) : 0;
  const bool is_const = (info >> 16) & 1;
  // wave-uniform trip count (tracks are sorted longest first inside a start-frame group)
  int mmax = m;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mmax = max(mmax, __shfl_xor(mmax, o, 64));
  double cost = 0.0;
  if (tdc) {
    nob[4] = td;
#pragma unroll
    for (int u = 0; u < NLATER; u++) nobl[u][4] = td;
  }
  if (chead) {      // x_cand = x + s_l (c1 v_l + c2 y_l): candidate_lm is candidate_tile's arithmetic
    double d2, n2;
    lam = candidate_lm(lam, c_sl, c_vl, c_yl, c.c1, c.c2, valid && !is_const && m > 0, d2, n2);
    d.lam[(size_t)(1 - c.cur) * TL + slot] = lam;
    d2 = wave_sum(d2); n2 = wave_sum(n2);
    if (lane == 0) {
      double *o = d.tile_cand + ((size_t)w * d.max_tiles + tile) * 4;
      o[1] = d2; o[2] = n2;
    }
  }

  // landmark row of the normal equations: pose_i (6) [| extrinsic (6) | td] — the latter only when they are free somewhere
  constexpr int NHC = FULL ? HC : 6;
  double hC[NHC], Hll = 0.0, gl = 0.0;
#pragma unroll
  for (int q = 0; q < NHC; q++) hC[q] = 0.0;
  if (MODE != 1 && !FULL && !YM) {   // reduced panel [Ji Jj r 0 0 0]: the three padding columns are written once
    double *xr = xs + lane * XLD;
    xr[13] = 0.0; xr[14] = 0.0; xr[15] = 0.0;
  }
  // YM: the landmark's camera-frame point P_ci and its world-frame vectors (f: from the camera centre of frame i, e: from its
  // body origin, x: from the window's origin P_0) are the same for all of its factors
  double ycx = 0.0, ycy = 0.0, ycz = 0.0, yinv_l = 0.0, Dsum[3] = {0.0, 0.0, 0.0};
  vec3 yf = mk3(0.0, 0.0, 0.0), ye = yf, yx = yf;
  if (YM || MODE == 1) {
    const double dti = tdc ? 0.0 : td - tdi;
    yinv_l = 1.0 / lam;
    ycx = (tdc ? pix : __builtin_fma(-dti, vix, pix)) * yinv_l; ycy = (tdc ? piy : __builtin_fma(-dti, viy, piy)) * yinv_l; ycz = piz * yinv_l;
    if (YM) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        yf[a] = __builtin_fma(fcs.W(a, 0), ycx, __builtin_fma(fcs.W(a, 1), ycy, fcs.W(a, 2) * ycz));
        ye[a] = yf[a] + fcs.wt[a];
        yx[a] = ye[a] + fcs.dPc[a];      // from the window's origin P_0
      }
      double *xr = xs + lane * XLD;      // panel row [g0 y0 r0 0 | g1 y1 r1 0]: the two padding columns are written once
      xr[7] = 0.0; xr[15] = 0.0;
    }
  }
  KSTAMP(2);
  double *contrib = KS > 1 ? d.vis_contrib + (((size_t)w * d.max_tiles + tile) * MAXOBS) * (VC_STRIDE * LM_TILE) + lane : nullptr;
  for (int k = kq; k < mmax; k += KS) {
    if (k < 5) KSTAMP(3 + 5 * k);
    double r[2], Ji[12], Jj[12], Je[FULL ? 12 : 1], Jl[2], Jt[2], hp[6];
    const double pjx = nob[0], pjy = nob[1], vjx = nob[2], vjy = nob[3], tdj = nob[4];
    if (KS > 1) {      // the next step's observation: fetched before the loop
#pragma unroll
      for (int q = 0; q < 5; q++) nob[q] = nobl[0][q];
#pragma unroll
      for (int u = 0; u + 1 < NLATER; u++) {
#pragma unroll
        for (int q = 0; q < 5; q++) nobl[u][q] = nobl[u + 1][q];
      }
    } else if (k + KS < mmax) {
      const double *ob = d.lm_obs + (size_t)(k + KS) * 5 * TL + slot;
#pragma unroll
      for (int q = 0; q < 5; q++) if (q < nobq) nob[q] = ob[q * TL];
    }
    if constexpr (YM) {
      double *xr = xs + lane * XLD;
      if (k < m) {
        const PCT &pc = pcs[sframe + 1 + k];
        double g0[3], g1[3];
        const double ck = visual_lin_y(pc, ycx, ycy, ycz, yf, yinv_l, td, pjx, pjy, vjx, vjy, tdj, sq, delta, r, g0, g1, Jl);
        if (KS > 1) contrib[((size_t)k * VC_STRIDE + 15) * LM_TILE] = ck; else cost += ck;
        const vec3 y0 = cross3(mk3(g0[0], g0[1], g0[2]), yx), y1 = cross3(mk3(g1[0], g1[1], g1[2]), yx);   // rows of G [x]x
        // landmark row of the normal equations (w = d/dlambda; a constant landmark has none)
        const double w0 = is_const ? 0.0 : Jl[0], w1 = is_const ? 0.0 : Jl[1];
        vec3 dv;
#pragma unroll
        for (int q = 0; q < 3; q++) dv[q] = __builtin_fma(g0[q], w0, g1[q] * w1);                         // d = G^T w
        if (KS > 1) {
          double *cb = contrib + (size_t)k * VC_STRIDE * LM_TILE;
          cb[0] = __builtin_fma(w0, w0, w1 * w1);
          cb[LM_TILE] = __builtin_fma(w0, r[0], w1 * r[1]);
#pragma unroll
          for (int q = 0; q < 3; q++) cb[(2 + q) * LM_TILE] = dv[q];
        } else {
          Hll += __builtin_fma(w0, w0, w1 * w1);
          gl += __builtin_fma(w0, r[0], w1 * r[1]);
#pragma unroll
          for (int q = 0; q < 3; q++) Dsum[q] += dv[q];
        }
        // what k_schur / k_lm_step need of this factor is d alone (24 instead of 48 bytes): the H_pl block of the observing pose
        // j = s + 1 + k is [ -d ; Rj^T (d x (x - t_j)) ] with the landmark's x and the frame's constants (lm_row_* in gfbe_devutil.h)
#pragma unroll
        for (int q = 0; q < 3; q++) hp[q] = dv[q];
#pragma unroll
        for (int q = 0; q < 3; q++) { xr[q] = g0[q]; xr[3 + q] = y0[q]; xr[8 + q] = g1[q]; xr[11 + q] = y1[q]; }
        xr[6] = r[0]; xr[14] = r[1];
      } else {
#pragma unroll
        for (int q = 0; q < 7; q++) { xr[q] = 0.0; xr[8 + q] = 0.0; }
      }
#if GFBE_KVIS_EARLY
#pragma unroll
      for (int q = 0; q < 2; q++) asm volatile("" : "+v"(nob[q]));      // (see the 13-column path below)
#endif
      if (k < m) {
#pragma unroll
        for (int q = 0; q < 3; q++) d.lm_hP[((size_t)k * 6 + q) * TL + slot] = hp[q];
      }
      // [Y r]^T [Y r] of the step's 64 factors: lane's LDS row holds BOTH residual rows of its factor, eight columns each, so the
      // 16 x 16 product has the two 8 x 8 blocks wanted on its diagonal (the off-diagonal blocks mix the rows: dropped)
      typedef double dbl4_y __attribute__((ext_vector_type(4)));
      dbl4_y acc0 = {0, 0, 0, 0};
      const int lr = lane & 15, lk = lane >> 4;
      __threadfence_block();
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int blk = 0; blk < LM_TILE / 4 / 8; blk++) {
        double va[8];
#pragma unroll
        for (int u = 0; u < 8; u++) va[u] = xs[(4 * (8 * blk + u) + lk) * XLD + lr];
#pragma unroll
        for (int u = 0; u < 8; u++) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u], va[u], acc0, 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
      // lane (lk, lr): acc0[q] = entry (lk + 4 q, lr). Rows 0..7 x columns 0..7 (first residual row) live in q = 0, 1 of the lanes
      // lr < 8; rows 8..15 x columns 8..15 (second row) in q = 2, 3 of the lanes lr >= 8: folded onto the lanes lr < 8
      const double s0 = acc0[0] + __shfl_down(acc0[2], 8, 64), s1 = acc0[1] + __shfl_down(acc0[3], 8, 64);
      double *vo = d.vis_part + (((size_t)w * d.max_tiles + tile) * MAXOBS + k) * VPY_STRIDE;
      if (lr < 7) {      // upper triangle of the 7 x 7 sum, row-major packed: (a, b), a <= b, at 7 a - a (a - 1) / 2 + b - a
        { const int a = lk; if (a <= lr) vo[7 * a - a * (a - 1) / 2 + lr - a] = s0; }
        { const int a = lk + 4; if (a <= lr) vo[7 * a - a * (a - 1) / 2 + lr - a] = s1; }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if constexpr (MODE == 1) {
      if (k < m) cost += visual_cost_y(pcs[sframe + 1 + k], ycx, ycy, ycz, td, pjx, pjy, vjx, vjy, tdj, sq, delta);
    }
    if constexpr (!YM && MODE != 1) {
    if (k < m) {
      if (GFBE_ABLATE == 4 && MODE == 0) {
#pragma unroll
        for (int q = 0; q < 12; q++) { Ji[q] = pjx + q; Jj[q] = pjy * q; if (FULL) Je[q] = vjx - q; }
        Jl[0] = lam; Jl[1] = tdj; Jt[0] = vjy; Jt[1] = pix; r[0] = piy * 1e-3; r[1] = piz * 1e-3;
      } else
      {
        const double ck = visual_lin<MODE != 1, FULL>(pcs[sframe + 1 + k], lam, td, pix, piy, piz, pjx, pjy, vix, viy, vjx, vjy, tdi, tdj,
                                                      sq, delta, r, Ji, Jj, Je, Jl, Jt);
        if (KS > 1) contrib[((size_t)k * VC_STRIDE + 15) * LM_TILE] = ck; else cost += ck;
      }
      if (MODE != 1) {
        if (FULL && write_records) {   // inspection path (gfbe_eval_factors): block-CSR record r(2) | row0: Ji Jj Je Jl Jt | row1
          double *rb = d.rec + ((size_t)ds.rec_off + d.lm_rec[(size_t)k * TL + slot]) * REC;
          rb[0] = r[0]; rb[1] = r[1];
#pragma unroll
          for (int q = 0; q < 6; q++) {
            rb[2 + q] = Ji[q]; rb[8 + q] = Jj[q]; rb[14 + q] = Je[FULL ? q : 0];
            rb[22 + q] = Ji[6 + q]; rb[28 + q] = Jj[6 + q]; rb[34 + q] = Je[FULL ? 6 + q : 0];
          }
          rb[20] = Jl[0]; rb[21] = Jt[0]; rb[40] = Jl[1]; rb[41] = Jt[1];
        }
        // landmark row of the normal equations (w = Jl)
        const double w0 = (is_const && MODE == 0) ? 0.0 : Jl[0], w1 = (is_const && MODE == 0) ? 0.0 : Jl[1];
        if (KS > 1) {
          double *cb = contrib + (size_t)k * VC_STRIDE * LM_TILE;
          cb[0] = __builtin_fma(w0, w0, w1 * w1);
          cb[LM_TILE] = __builtin_fma(w0, r[0], w1 * r[1]);
#pragma unroll
          for (int q = 0; q < 6; q++) {
            cb[(2 + q) * LM_TILE] = __builtin_fma(Ji[q], w0, Ji[6 + q] * w1);
            if (FULL) cb[(8 + q) * LM_TILE] = __builtin_fma(Je[q], w0, Je[6 + q] * w1);
            hp[q] = __builtin_fma(Jj[q], w0, Jj[6 + q] * w1);
          }
          if (FULL) cb[14 * LM_TILE] = __builtin_fma(Jt[0], w0, Jt[1] * w1);
        } else {
        Hll += __builtin_fma(w0, w0, w1 * w1);
        gl += __builtin_fma(w0, r[0], w1 * r[1]);
#pragma unroll
        for (int q = 0; q < 6; q++) {
          hC[q] += __builtin_fma(Ji[q], w0, Ji[6 + q] * w1);
          if (FULL) hC[6 + q] += __builtin_fma(Je[q], w0, Je[6 + q] * w1);
          hp[q] = __builtin_fma(Jj[q], w0, Jj[6 + q] * w1);
        }
        if (FULL) hC[12] += __builtin_fma(Jt[0], w0, Jt[1] * w1);
        }
      }
    } else if (MODE != 1) {
#pragma unroll
      for (int q = 0; q < 12; q++) { Ji[q] = 0.0; Jj[q] = 0.0; if (FULL) Je[q] = 0.0; }
      Jt[0] = Jt[1] = 0.0; r[0] = r[1] = 0.0;
    }
    if (k < 5) KSTAMP(4 + 5 * k);
    if (MODE != 1) {
#if GFBE_KVIS_EARLY
      // vmcnt counts loads and stores in one queue: if the next step's observation (loaded at the top of this step) were first
      // touched at the loop's back edge, the wave would sit there until this step's partial-sum stores have been acknowledged.
      // Touching it here — after the evaluation, before any store of this step, on a path every lane takes — costs nothing.
#pragma unroll
      for (int q = 0; q < 5; q++) asm volatile("" : "+v"(nob[q]));
#endif
      if (k < m && (GFBE_ABLATE != 3 || hp[0] == 1.2345)) {   // the landmark's H_pl block of observing pose s + 1 + k
        // (lm_hP is not cleared at upload: rows from a track's length on hold whatever the memory held; k_schur masks them per
        //  landmark, nobody else reads past a track's length)
#pragma unroll
        for (int q = 0; q < 6; q++) d.lm_hP[((size_t)k * 6 + q) * TL + slot] = hp[q];
      }
    }
    if (MODE != 1 && GFBE_ABLATE != 1) {
      // X^T X of the step's 128-row panel on the FP64 matrix cores (the J^T J / J^T r of this tile's factors of pose pair
      // (sframe, sframe+1+k)); J never leaves the CU. The panel goes through LDS 64 rows at a time (row h of every lane's
      // factor, h = 0, 1).
      //   FULL: X = [Ji Jj Je Jt | r], 20 columns. T0 = X(:,0:16)^T X(:,0:16) with v_mfma_f64_16x16x4_f64 (64 clk / 4 rows);
      //     the thin blocks use v_mfma_f64_4x4x4_4b_f64 (18 clk, lane layout measured in profiles/ubench/
      //     mfma_f64_4x4x4_layout.hip: A_blk[i][k] at lane 16k+4blk+i, B_blk[k][j] at 16k+4blk+j, D_blk[i][j] at 16i+4blk+j):
      //     T1 = X(:,0:16)^T X(:,16:20) (block blk = rows 4blk..4blk+3, same A operand as T0), T2 = X(:,16:20)^T X(:,16:20)
      //     (the four blocks take four different row quads, summed at the end).
      //   reduced (extrinsic and td constant in every window of the batch): X = [Ji Jj r 0 0 0] is one 16-column tile —
      //     16 matrix-core instructions per half instead of 36 — and the partial is written in the FULL layout's positions
      //     (gradient column into T1(:,3), r^T r into T2(3,3)); the extrinsic / td positions keep whatever they hold
      //     (zero, or the marginalisation pass's values): those dims are inactive and k_assemble never reads them.
      typedef double dbl4_v __attribute__((ext_vector_type(4)));
      dbl4_v acc0 = {0, 0, 0, 0};
      double acc1 = 0.0, acc2 = 0.0;
      const int lr = lane & 15, lk = lane >> 4, lj = lane & 3, lb = (lane & 15) >> 2;
      if (k < 5) KSTAMP(5 + 5 * k);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        if (k < 5 && h == 1) KSTAMP(6 + 5 * k);
        {
          double *xr = xs + lane * XLD;
#pragma unroll
          for (int q = 0; q < 6; q++) { xr[q] = Ji[6 * h + q]; xr[6 + q] = Jj[6 * h + q]; if (FULL) xr[12 + q] = Je[6 * h + q]; }
          if (FULL) { xr[18] = Jt[h]; xr[19] = r[h]; }
          else xr[12] = r[h];
        }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int blk = 0; blk < LM_TILE / 4 / 8; blk++) {   // 8 row-quads at a time: operands first, then the MFMAs
          double va[8], vb[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const double *rowp = xs + (4 * (8 * blk + u) + lk) * XLD;
            va[u] = rowp[lr];
            if (FULL) vb[u] = rowp[16 + lj];
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u], va[u], acc0, 0, 0, 0);
            if (FULL) acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(va[u], vb[u], acc1, 0, 0, 0);
          }
        }
        if (FULL) {
          double vc[LM_TILE / 16];
#pragma unroll
          for (int qd = 0; qd < LM_TILE / 16; qd++) vc[qd] = xs[(16 * qd + 4 * lb + lk) * XLD + 16 + lj];
#pragma unroll
          for (int qd = 0; qd < LM_TILE / 16; qd++) acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(vc[qd], vc[qd], acc2, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (FULL) {
        acc2 += __shfl_xor(acc2, 4, 64);
        acc2 += __shfl_xor(acc2, 8, 64);
      }
      if (k < 5) KSTAMP(7 + 5 * k);
      double *vo = d.vis_part + (((size_t)w * d.max_tiles + tile) * MAXOBS + k) * VP_STRIDE;
      if (GFBE_ABLATE == 2 && acc0[0] != 1.2345) continue;
      if (FULL) {
#pragma unroll
        for (int q = 0; q < 4; q++) vo[(lk + 4 * q) * 16 + lr] = acc0[q];
        vo[256 + (4 * lb + lk) * 4 + lj] = acc1;        // T1[row 4 blk + i][col j], i = lane >> 4
        if (lb == 0) vo[320 + lk * 4 + lj] = acc2;       // T2[i][j]
      } else {
        // lane holds (row lk + 4 q, column lr) of the 16 x 16 tile: rows / columns 0..11 are pose_i, pose_j; 12 is r
#pragma unroll
        for (int q = 0; q < 3; q++) {                    // rows 0..11
          const int a = lk + 4 * q;
          if (lr < 12) vo[a * 16 + lr] = acc0[q];
          else if (lr == 12) vo[256 + a * 4 + 3] = acc0[q];     // J^T r -> column 19 of the full layout
        }
        if (lk == 0 && lr == 12) vo[320 + 15] = acc0[3];        // r^T r (row 12, column 12)
      }
      __builtin_amdgcn_wave_barrier();
    }
    }   // (!YM)
  }
  LSTAMP(16);
  if (KS > 1) {
    if (WS) {
      // the tile's waves meet here (a workgroup barrier orders their contribution stores before the loads below: one CU, one vector
      // cache); the first one adds the steps' contributions up, in step order
      __syncthreads();
      if (kq != 0) return;
      LSTAMP_ANY(17);
    } else {
    // the last of the tile's KS workgroups to get here adds the steps' contributions up, in step order
    __threadfence();
    int last = 0;
    int *cnt = d.tile_cnt + (size_t)w * d.max_tiles + tile;
    if (lane == 0) last = atomicAdd(cnt, 1) == KS - 1;
    last = __shfl(last, 0, 64);
    if (!last) return;
    __threadfence();
    LSTAMP_ANY(17);
    if (lane == 0) *cnt = 0;     // (ready for the next launch)
    }
    const double *cr = d.vis_contrib + (((size_t)w * d.max_tiles + tile) * MAXOBS) * (VC_STRIDE * LM_TILE) + lane;
    // (round 5: ALL the steps' values are requested before the first sum — the loop used to take one memory round trip per step, ten
    //  in a row for the tiles of start frame 0: ~6 of the ~17 us of a single window's linearisation launch. Same sums, same order.)
    constexpr int NVC = (MODE == 1) ? 1 : (YM ? 6 : (FULL ? 16 : 9));      // cost [| Hll gl | D (3) or hC (6 [+ 7])]
    constexpr int CH = (NVC > 9) ? 4 : MAXOBS;      // steps in flight (the 20-column panel's 16 values per step: four at a time)
#define VC_LD(k, i) (WS ? __hip_atomic_load(cr + ((size_t)(k) * VC_STRIDE + (i)) * LM_TILE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) \
                        : __hip_atomic_load(cr + ((size_t)(k) * VC_STRIDE + (i)) * LM_TILE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
#pragma unroll
    for (int k0 = 0; k0 < MAXOBS; k0 += CH) {
      double vc[CH][NVC];
#pragma unroll
      for (int u = 0; u < CH; u++) {
        const int k = k0 + u;
        if (k < MAXOBS && k < mmax) {      // (wave-uniform)
          vc[u][0] = VC_LD(k, 15);
          if (MODE != 1) {
            vc[u][1] = VC_LD(k, 0); vc[u][2] = VC_LD(k, 1);
#pragma unroll
            for (int q = 0; q < NVC - 3; q++) vc[u][3 + q] = VC_LD(k, 2 + q);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < CH; u++) {
        const int k = k0 + u;
        if (k < MAXOBS && k < m) {
          cost += vc[u][0];
          if (MODE != 1) {
            Hll += vc[u][1];
            gl += vc[u][2];
            if (YM) {
#pragma unroll
              for (int q = 0; q < 3; q++) Dsum[q] += vc[u][3 + q];
            } else {
#pragma unroll
              for (int q = 0; q < 6; q++) { hC[q] += vc[u][3 + q]; if (FULL) hC[6 + q] += vc[u][9 + q]; }
              if (FULL) hC[12] += vc[u][15];
            }
          }
        }
      }
    }
#undef VC_LD
    asm volatile("" :: "v"(cost));
    LSTAMP_ANY(18);
  }
  if (YM) {      // the landmark's own part: D = the sum of its factors' d, and x, its position from the window's origin — the H_pl block
                 // of the start pose is [ D ; Ri^T ((x - t_i) x D) ] (lm_row_* in gfbe_devutil.h)
#pragma unroll
    for (int q = 0; q < 3; q++) { hC[q] = Dsum[q]; hC[3 + q] = yx[q]; }
  }
  // (one wave gets here for tile 0: the only one of a k_vis launch, the first of a k_lin_small tile workgroup)
  if (MODE == 0 && d.spec && tile == 0 && lane == 0) d.ctl[w].sw_mu[SPEC ? 1 - c.lb : c.lb] = SPEC ? mu_after_accept(c.mu) : c.mu;
  if (MODE == 0 && valid) {
    // the landmark's weight in the Schur term, w_l = s_l^2 / (s_l^2 Hll + mu clamp(s_l^2 Hll)) (Jacobi-scaled, mu-regularised), once
    // per landmark here instead of once per wave that stages it in k_schur; the Jacobi scale s_l is fixed at iteration 0
    // (TrustRegionMinimizer::IterationZero). A constant landmark has none.
    // (SPEC: with the mu an accepted step leaves — this set is current only then; c.sw_mu records what the weights of a set were formed
    //  with, and k_schur forms them itself when that is not the window's mu: after TrustRegionMinimizer::HandleInvalidStep)
    const bool first = !SPEC && c.iter == 0;
    const double mu_w = SPEC ? mu_after_accept(c.mu) : c.mu;
    double sl = 1.0, sw = 0.0;
    if (m > 0 && !is_const) {
      sl = first ? (d.opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(Hll)) : 1.0) : d.lm_sl[slot];      // (loaded here, not ahead of the loop: two registers over
                                                                                                 //  the 128 that four waves per SIMD allow cost the kernel 6 %)
      const double hs2 = sl * sl * Hll;
      sw = sqrt(sl * sl / (hs2 + mu_w * clamp_diag(hs2)));
    }
    if (first) d.lm_sl[slot] = sl;
    d.lm_sw[slot] = sw;
  }
  if (MODE != 1 && valid) {
    d.lm_Hll[slot] = Hll;
    d.lm_gl[slot] = gl;
#pragma unroll
    for (int q = 0; q < NHC; q++) d.lm_hC[(size_t)q * TL + slot] = hC[q];
  }
  KSTAMP(30);
  LSTAMP_ANY(19);
  cost = wave_sum(cost);
  if (lane == 0) {
    if (CAND) d.tile_cand[((size_t)w * d.max_tiles + tile) * 4] = cost;
    else d.tile_cost[(size_t)w * d.max_tiles + tile] = cost;
  }
}

__device__ __forceinline__ double candidate_tile(const BatchDev &d, const WinDesc &ds, const WinCtl &c, const int w, const int tile, const int t);
// SPEC (MODE 0; BatchDev::spec): the candidate's linearisation in the place of its cost pass, into the set of outputs that is not the current one
template <int MODE, bool FULL, bool SPEC = false>
__global__ __launch_bounds__(LM_TILE, (MODE == 0 && !FULL) ? GFBE_KVIS_WAVES : 2) void k_vis(BatchDev d0, int write_records) {
  const BatchDev d = MODE == 0 ? lin_view(d0, SPEC ? 1 - d0.ctl[blockIdx.x].lb : d0.ctl[blockIdx.x].lb) : d0;
  // tile-major dispatch order (x = window): all windows' tile 0 (start frame 0, the longest tracks) first, the
  // short start-frame-7 tiles last — a longest-first schedule that shortens the tail of the launch
  // (the cost pass of a throughput batch, write_records = 1: the tile first forms the candidate inverse depths of its landmarks —
  //  the landmark half of k_candidate, the same 64 lanes and the same sums (vis_body's head); k_candidate_dense has formed the candidate's
  //  dense blocks and pair constants before. A launch of its own walked a window's tiles four at a time: 115 us per 2048 windows.)
  vis_body<MODE, FULL, 1, SPEC>(d, (MODE == 1 || SPEC) ? 0 : write_records, blockIdx.x, blockIdx.y, 0, (MODE == 1 || SPEC) && write_records);
}

// =============================================================================================
// k_vis_chunk (round 6): k_vis<0, false> for throughput batches with a wave that STAYS — one wave takes up to VIS_CHUNK consecutive
// tiles of ONE start frame of its window instead of one tile. Phase stamps of a k_vis wave under load (profiles/r6_kvis_ablation.txt):
// 6.6 us of prologue — descriptor, control block, then the landmark's data and the pair records, each a dependent round trip of 2 - 3 us
// under load, then the LDS staging and its barrier — in front of 5.3 observation steps of ~2.4 us: a third of a wave's life. Here the pair
// records of the start frame are staged once per wave, and a tile's landmark data (slot info, first observation, inverse depth, and what
// the candidate's head needs) is requested while the tile before it is evaluated. Per tile the arithmetic is vis_body<0, false, 1, SPEC>'s,
// operation for operation, in its order: every output bit for bit (the whole suite runs through it).
// Grid: (window, start frame x sub-chunk): blockIdx.y = s * nsub + sub takes the tiles [begin(s) + VIS_CHUNK sub, + VIS_CHUNK) of start
// frame s (start frame 0 — the longest tracks — is dispatched first).
// =============================================================================================
#ifndef GFBE_VIS_CHUNK
#define GFBE_VIS_CHUNK 1
#endif
#ifndef VIS_CHUNK
#define VIS_CHUNK 4
#endif
template <bool SPEC>
__global__ __launch_bounds__(LM_TILE, GFBE_KVIS_WAVES) void k_vis_chunk(BatchDev d0, int head_in, int nsub) {
  const int w = blockIdx.x;
  const WinCtl &c = d0.ctl[w];
  if (!SPEC && (c.done || c.reuse)) return;
  if (SPEC && (c.done || !c.have_step)) return;
  const int lbw = SPEC ? 1 - c.lb : c.lb;
  const BatchDev d = lin_view(d0, lbw);
  const WinDesc &ds = d.desc[w];
  const int sframe = blockIdx.y / nsub, sub = blockIdx.y - sframe * nsub;
  const int t0 = ds.sf_tile_begin[sframe] + VIS_CHUNK * sub, t1 = min(t0 + VIS_CHUNK, ds.sf_tile_begin[sframe + 1]);
  if (t0 >= t1) return;
  const int cur = c.cur, buf = SPEC ? 1 - cur : cur;
  const double *X = d.x + ((size_t)w * 2 + buf) * NA;
  const bool chead = SPEC && head_in;
  const double c1 = c.c1, c2 = c.c2;
  const bool first = !SPEC && c.iter == 0;
  const double mu_w = SPEC ? mu_after_accept(c.mu) : c.mu;
  const double sq = d.opt.vis_sqrt_info, delta = d.opt.huber_delta;
  const int jac_scale = d.opt.jacobi_scaling;
  const int lm_off_c = ds.lm_off;
  constexpr int XLD = 17;
  __shared__ PairConstY pcs[NF];
  __shared__ FrameConst fcs;
  __shared__ double xs[LM_TILE * XLD];
  const int lane = threadIdx.x;
  const size_t TL = d.tot_lm;
  // ---- what a lane holds of a tile before its first step; the NEXT tile's is requested while the current one is evaluated
  int p_info;
  double p_pt[3], p_lam, p_sl = 0.0, p_vl = 0.0, p_yl = 0.0, p_ob[2];
#define VC_PREFETCH(TILE)                                                                                           \
  {                                                                                                                  \
    const int slot_ = lm_off_c + (TILE) * LM_TILE + lane;                                                            \
    p_info = d.lm_info[slot_];                                                                                       \
    _Pragma("unroll") for (int q = 0; q < 3; q++) p_pt[q] = d.lm_pts[(size_t)q * TL + slot_];                        \
    p_lam = d.lam[(size_t)(chead ? cur : buf) * TL + slot_];                                                         \
    if (chead) { p_sl = d.lm_sl[slot_]; p_vl = d.lm_vl[slot_]; p_yl = d.lm_yl[slot_]; }                              \
    p_ob[0] = d.lm_obs[slot_]; p_ob[1] = d.lm_obs[TL + slot_];                                                       \
  }
  VC_PREFETCH(t0)
  const double td = X[A_TD];
  {
    const double *src = d.pc + (((size_t)w * 3 + buf) * NPAIR + sframe * NF) * PC_DOUBLES;
#if GFBE_PCS_ONE_ROUND
    // (end of round 6) every record of the start frame's pairs requested before the first is written to LDS: the loop below compiles to
    // load, s_waitcnt vmcnt(0), ds_write per pair — up to ten dependent round trips in front of a wave's first tile (its ISA; the prologue
    // was a third of a wave's life, profiles/r6_kvis_ablation.txt). A pair beyond the window's last frame re-reads the last record and writes nothing.
    static_assert(PCY_DOUBLES <= LM_TILE && FC_DOUBLES <= LM_TILE, "one lane per entry of a pair record");
    {
      const int ql = min(lane, (int)PCY_DOUBLES - 1);
      double rec[NF - 1];
#pragma unroll
      for (int jj = 0; jj < NF - 1; jj++) rec[jj] = src[(size_t)min(sframe + 1 + jj, NF - 1) * PC_DOUBLES + ql];
      const double fcv = src[(size_t)sframe * PC_DOUBLES + min(lane, (int)FC_DOUBLES - 1)];
#pragma unroll
      for (int jj = 0; jj < NF - 1; jj++) if (sframe + 1 + jj < NF && lane < (int)PCY_DOUBLES) ((double *)&pcs[min(sframe + 1 + jj, NF - 1)])[lane] = rec[jj];
      if (lane < (int)FC_DOUBLES) ((double *)&fcs)[lane] = fcv;
    }
#else
    for (int j = sframe + 1; j < NF; j++)
      for (int q = lane; q < (int)PCY_DOUBLES; q += LM_TILE) ((double *)&pcs[j])[q] = src[(size_t)j * PC_DOUBLES + q];
    if (lane < (int)FC_DOUBLES) ((double *)&fcs)[lane] = src[(size_t)sframe * PC_DOUBLES + lane];
#endif
  }
  __syncthreads();
  for (int tile = t0; tile < t1; tile++) {
    const int slot = lm_off_c + tile * LM_TILE + lane;
    const int info = p_info;
    const double pix = p_pt[0], piy = p_pt[1], piz = p_pt[2];
    double lam = p_lam;
    const double c_sl = p_sl, c_vl = p_vl, c_yl = p_yl;
    double nob[2] = {p_ob[0], p_ob[1]};
    if (tile + 1 < t1) VC_PREFETCH(tile + 1)
    const bool valid = (info >> 24) & 1;
    const int m = valid ? ((info >> 8) & 0xff) : 0;
    const bool is_const = (info >> 16) & 1;
    int mmax = m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mmax = max(mmax, __shfl_xor(mmax, o, 64));
    double cost = 0.0;
    if (chead) {      // x_cand = x + s_l (c1 v_l + c2 y_l): candidate_lm is candidate_tile's arithmetic
      double d2, n2;
      lam = candidate_lm(lam, c_sl, c_vl, c_yl, c1, c2, valid && !is_const && m > 0, d2, n2);
      d.lam[(size_t)(1 - cur) * TL + slot] = lam;
      d2 = wave_sum(d2); n2 = wave_sum(n2);
      if (lane == 0) {
        double *o = d.tile_cand + ((size_t)w * d.max_tiles + tile) * 4;
        o[1] = d2; o[2] = n2;
      }
    }
    double Hll = 0.0, gl = 0.0, Dsum[3] = {0.0, 0.0, 0.0};
    const double yinv_l = 1.0 / lam;
    const double ycx = pix * yinv_l, ycy = piy * yinv_l, ycz = piz * yinv_l;
    vec3 yf, ye, yx;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      yf[a] = __builtin_fma(fcs.W(a, 0), ycx, __builtin_fma(fcs.W(a, 1), ycy, fcs.W(a, 2) * ycz));
      ye[a] = yf[a] + fcs.wt[a];
      yx[a] = ye[a] + fcs.dPc[a];      // from the window's origin P_0
    }
    double *xr = xs + lane * XLD;      // panel row [g0 y0 r0 0 | g1 y1 r1 0]: the two padding columns are written once
    xr[7] = 0.0; xr[15] = 0.0;
    for (int k = 0; k < mmax; k++) {
      double r[2], Jl[2], hp[3];
      const double pjx = nob[0], pjy = nob[1];
      if (k + 1 < mmax) {
        const double *ob = d.lm_obs + (size_t)(k + 1) * 5 * TL + slot;
        nob[0] = ob[0]; nob[1] = ob[TL];
      }
      if (k < m) {
        double g0[3], g1[3];
        const double ck = visual_lin_y(pcs[sframe + 1 + k], ycx, ycy, ycz, yf, yinv_l, td, pjx, pjy, 0.0, 0.0, td, sq, delta, r, g0, g1, Jl);
        cost += ck;
        const vec3 y0 = cross3(mk3(g0[0], g0[1], g0[2]), yx), y1 = cross3(mk3(g1[0], g1[1], g1[2]), yx);   // rows of G [x]x
        const double w0 = is_const ? 0.0 : Jl[0], w1 = is_const ? 0.0 : Jl[1];
        vec3 dv;
#pragma unroll
        for (int q = 0; q < 3; q++) dv[q] = __builtin_fma(g0[q], w0, g1[q] * w1);                         // d = G^T w
        Hll += __builtin_fma(w0, w0, w1 * w1);
        gl += __builtin_fma(w0, r[0], w1 * r[1]);
#pragma unroll
        for (int q = 0; q < 3; q++) Dsum[q] += dv[q];
#pragma unroll
        for (int q = 0; q < 3; q++) hp[q] = dv[q];
#pragma unroll
        for (int q = 0; q < 3; q++) { xr[q] = g0[q]; xr[3 + q] = y0[q]; xr[8 + q] = g1[q]; xr[11 + q] = y1[q]; }
        xr[6] = r[0]; xr[14] = r[1];
      } else {
#pragma unroll
        for (int q = 0; q < 7; q++) { xr[q] = 0.0; xr[8 + q] = 0.0; }
      }
#if GFBE_KVIS_EARLY
#pragma unroll
      for (int q = 0; q < 2; q++) asm volatile("" : "+v"(nob[q]));      // (the next step's observation is waited for BEFORE this step's stores are issued: vis_body)
#endif
      if (k < m) {
#pragma unroll
        for (int q = 0; q < 3; q++) d.lm_hP[((size_t)k * 6 + q) * TL + slot] = hp[q];
      }
      typedef double dbl4_y __attribute__((ext_vector_type(4)));
      dbl4_y acc0 = {0, 0, 0, 0};
      const int lr = lane & 15, lk = lane >> 4;
      __threadfence_block();
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int blk = 0; blk < LM_TILE / 4 / 8; blk++) {
        double va[8];
#pragma unroll
        for (int u = 0; u < 8; u++) va[u] = xs[(4 * (8 * blk + u) + lk) * XLD + lr];
#pragma unroll
        for (int u = 0; u < 8; u++) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u], va[u], acc0, 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
      const double s0 = acc0[0] + __shfl_down(acc0[2], 8, 64), s1 = acc0[1] + __shfl_down(acc0[3], 8, 64);
      double *vo = d.vis_part + (((size_t)w * d.max_tiles + tile) * MAXOBS + k) * VPY_STRIDE;
      if (lr < 7) {      // upper triangle of the 7 x 7 sum, row-major packed: (a, b), a <= b, at 7 a - a (a - 1) / 2 + b - a
        { const int a = lk; if (a <= lr) vo[7 * a - a * (a - 1) / 2 + lr - a] = s0; }
        { const int a = lk + 4; if (a <= lr) vo[7 * a - a * (a - 1) / 2 + lr - a] = s1; }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (d.spec && tile == 0 && lane == 0) d.ctl[w].sw_mu[lbw] = mu_w;
    if (valid) {
      // the landmark's weight in the Schur term (vis_body)
      double sl = 1.0, sw = 0.0;
      if (m > 0 && !is_const) {
        sl = first ? (jac_scale ? 1.0 / (1.0 + sqrt(Hll)) : 1.0) : (chead ? c_sl : d.lm_sl[slot]);
        const double hs2 = sl * sl * Hll;
        sw = sqrt(sl * sl / (hs2 + mu_w * clamp_diag(hs2)));
      }
      if (first) d.lm_sl[slot] = sl;
      d.lm_sw[slot] = sw;
      d.lm_Hll[slot] = Hll;
      d.lm_gl[slot] = gl;
#pragma unroll
      for (int q = 0; q < 3; q++) { d.lm_hC[(size_t)q * TL + slot] = Dsum[q]; d.lm_hC[(size_t)(3 + q) * TL + slot] = yx[q]; }
    }
    cost = wave_sum(cost);
    if (lane == 0) {
      if (SPEC) d.tile_cand[((size_t)w * d.max_tiles + tile) * 4] = cost;
      else d.tile_cost[(size_t)w * d.max_tiles + tile] = cost;
    }
  }
#undef VC_PREFETCH
}

// =============================================================================================
// k_lio_window: LiDAR point-to-plane factors attached to one window pose (gfbe_lio_block; the joint LIO + VIO solve,
// LidarPlaneNormFactor lidarFactor.cpp:18-51 + HuberLoss lidarodom.cpp:539). LIOW_WGS workgroups stride over the
// window's factors; every thread keeps its share of the 6 x 6 J^T J (upper triangle), J^T r and the cost in registers,
// the workgroup reduces in a fixed order and writes one partial. MODE 0: linearise; MODE 1: candidate cost.
// k_visblock adds the partials into the pose block / gradient / cost of the visual part; k_accept adds the candidate cost.
// =============================================================================================
// small batches: LIN_SMALL_KS workgroups per tile (see vis_body)
template <int MODE, bool FULL>
__global__ __launch_bounds__(LM_TILE, 1) void k_vis_split(BatchDev d, int write_records) {
  vis_body<MODE, FULL, LIN_SMALL_KS>(d, write_records, blockIdx.x, blockIdx.y / LIN_SMALL_KS, blockIdx.y % LIN_SMALL_KS);
}

// spec (MODE 0; BatchDev::spec): the linearisation at the candidate in the place of the cost pass, into the set of outputs that is not current
template <int MODE>
__global__ __launch_bounds__(256) void k_lio_window(BatchDev d0, int spec) {
  const int w = blockIdx.y, wg = blockIdx.x, t = threadIdx.x;
  const WinDesc &ds = d0.desc[w];
  if (ds.lio_n == 0) return;
  const WinCtl &c = d0.ctl[w];
  if (MODE == 0 && !spec && (c.done || c.reuse)) return;
  if ((MODE == 1 || spec) && (c.done || !c.have_step)) return;
  const BatchDev d = MODE == 0 ? lin_view(d0, spec ? 1 - c.lb : c.lb) : d0;
  const int buf = (MODE == 1 || spec) ? 1 - c.cur : c.cur;
  const double *X = d.x + ((size_t)w * 2 + buf) * NA + A_POSE(ds.lio_frame);
  const vec3 tw = ld3(X);
  const mat3 R = qrot(ldq(X + 3));
  constexpr int NACC = (MODE == 0) ? 28 : 1;
  double acc[NACC];
#pragma unroll
  for (int q = 0; q < NACC; q++) acc[q] = 0.0;
  const double sq = ds.lio_sqrt_info, delta = ds.lio_huber;
  for (int k = wg * 256 + t; k < ds.lio_n; k += LIOW_WGS * 256) {
    const double *f = d.lio + (size_t)(ds.lio_off + k) * 8;
    const vec3 p = ld3(f), nv = ld3(f + 3);
    const double sw = sq * f[7];
    double r = sw * (dot3(nv, add(mv(R, p), tw)) + f[6]);
    double scale = 1.0, cost = 0.5 * r * r;
    if (delta > 0.0) {
      double s1, rs, asn;
      cost = corrector(r * r, delta, &s1, &rs, &asn);     // 1-D residual under HuberLoss: rho'' <= 0, so J <- sqrt(rho') J, r <- sqrt(rho') r
      scale = s1;
    }
    if (MODE == 1) { acc[0] += cost; continue; }
    const vec3 nR = tmv(R, nv);
    double J[6];
    J[0] = sw * nv[0]; J[1] = sw * nv[1]; J[2] = sw * nv[2];
    J[3] = -sw * (nR[1] * p[2] - nR[2] * p[1]); J[4] = -sw * (nR[2] * p[0] - nR[0] * p[2]); J[5] = -sw * (nR[0] * p[1] - nR[1] * p[0]);
#pragma unroll
    for (int a = 0; a < 6; a++) J[a] *= scale;
    r *= scale;
    int e = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int b = a; b < 6; b++) acc[e++] += J[a] * J[b];
#pragma unroll
    for (int a = 0; a < 6; a++) acc[21 + a] += J[a] * r;
    acc[27] += cost;
  }
  __shared__ double red[4][NACC];
  const int lane = t & 63, wave = t >> 6;
#pragma unroll
  for (int q = 0; q < NACC; q++) {
    double v = acc[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) red[wave][q] = v;
  }
  __syncthreads();
  double *out = d.lio_part + ((size_t)w * LIOW_WGS + wg) * LIOW_PART;
  if (t < NACC) out[MODE == 1 ? 28 : t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
}

// =============================================================================================
// k_pairsum: sums the fused visual partials over the landmark tiles of a start frame (fixed order):
//   pair_part[w][(i,j)][:] = sum_{tiles t of start frame i} vis_part[w][t][j-i-1][:]
// Used by the marginalisation (pairs (0, j)); the solve loop sums the tiles inside k_assemble.
// =============================================================================================
__device__ __forceinline__ void pairsum_body(const BatchDev &d, const int marg, const int w, const int pair) {
  const WinDesc &ds = d.desc[w];
  const WinCtl &c = d.ctl[w];
  if (!marg && (c.done || c.reuse)) return;
  int i = 0, rem = pair;
  while (rem >= NF - 1 - i) { rem -= NF - 1 - i; i++; }
  const int j = i + 1 + rem;
  if (marg && i != 0) return;
  const int p = i * NF + j;
  if (ds.pair_begin[p + 1] == ds.pair_begin[p]) return;   // no factor on this pose pair: block stays zero
  const double *vp = d.vis_part + (size_t)w * d.max_tiles * MAXOBS * VP_STRIDE + (size_t)(j - i - 1) * VP_STRIDE + threadIdx.x;
  double s = 0.0;
  for (int t = ds.sf_tile_begin[i]; t < ds.sf_tile_begin[i + 1]; t++) {
    if (((d.lm_info[ds.lm_off + t * LM_TILE] >> 8) & 0xff) <= j - i - 1) break;   // tiles are sorted longest first: the others never ran this step (vis_part is not cleared)
    if (TILE_OWNED(d, t)) s += vp[(size_t)t * MAXOBS * VP_STRIDE];                // (landmark sharding: this rank's tiles)
  }
  if (marg) d.pair_part[((size_t)w * NF + j) * VP_STRIDE + threadIdx.x] = s;     // (only the marginalisation keeps per-pair sums: pairs (0, j), slot j)
}
__global__ __launch_bounds__(VP_STRIDE) void k_pairsum(BatchDev d, int marg) { pairsum_body(d, marg, blockIdx.y, blockIdx.x); }

// =============================================================================================
// k_dense: blocks 0..9 IMU factors, 10..19 wheel factors, 20 the prior. 64 threads each.
// mode 0 linearise at current; 1 candidate cost; 2 MARGIN_OLD set at xout (frame-0 IMU/wheel + prior);
// 3 MARGIN_SECOND_NEW set at xout (prior only).
// =============================================================================================
// Structural non-zeros of the un-whitened Jacobians written by imu_raw (15 x 30) / wheel_raw (6 x 22).
__device__ __forceinline__ bool imu_nz(int r, int c) {
  const int mask[5] = {0x03F, 0x052, 0x09E, 0x108, 0x210};   // 3 x 3 column blocks present in row block r / 3
  return (mask[r / 3] >> (c / 3)) & 1;
}
__device__ __forceinline__ bool wheel_nz(int r, int c) {
  if (c >= 20) return true;
  if (c >= 18) return r < 3;
  return r < 3 || ((c / 3) & 1);
}
// Which inertial / wheel factors a pass evaluates (shared by k_dense_raw and k_dense):
//   mode 0 linearise, 1 candidate cost, 2 MARGIN_OLD set at the re-anchored state (factor of frame 0), 3 nothing.
// spec (mode 0, BatchDev::spec): the linearisation AT THE CANDIDATE in the place of the cost pass — the candidate's state, the windows
// that have a step, the set of outputs that is not the current one.
__device__ __forceinline__ bool dense_pass_active(const WinCtl &c, int mode, int spec = 0) {
  if (mode == 0 && !spec && (c.done || c.reuse)) return false;
  if ((mode == 1 || spec) && (c.done || !c.have_step)) return false;
  return true;
}
#define RAW_IMU (15 + 15 * 30)
#define RAW_WHEEL (6 + 6 * 22)
// raw values of factor slot f: [f][window / 4][value q][window % 4] — k_dense_raw (lane = window) stores 32-byte pieces, a workgroup
// of k_dense_tp reads the RAW x 4 doubles of its four windows as one contiguous block; value q of window w is raw_of(...)[4 q]
__device__ __forceinline__ size_t raw_of(int f, int w, int B, int raw_len) { return ((size_t)f * ((B + 3) >> 2) + (w >> 2)) * raw_len * 4 + (w & 3); }
// k_dense_raw: the SE(3) / quaternion algebra of the inertial and wheel factors (imu_factor.h:69-190,
// wheel_factor.h:80-243) is scalar code; here one LANE = one window (factor f of 64 windows per wave), so the 64
// lanes of the wave are all busy. Output: raw residual + raw Jacobian non-zeros, window-minor
// (raw_imu[f][q][B]: coalesced stores here, one 32-byte sector per value for the per-factor workgroup of k_dense).
__global__ __launch_bounds__(64) void k_dense_raw(BatchDev d, int mode, int spec) {
  GFBE_SMALL_KERNEL_PRIO();
  const int f = blockIdx.x, w = blockIdx.y * 64 + threadIdx.x;
  if (w >= d.B) return;
  const WinDesc &ds = d.desc[w];
  const WinCtl &c = d.ctl[w];
  if (!dense_pass_active(c, mode, spec)) return;
  const int buf = (mode == 1 || spec) ? 1 - c.cur : c.cur;
  const double *X = (mode >= 2) ? d.xout + (size_t)w * NA : d.x + ((size_t)w * 2 + buf) * NA;
  const size_t B = d.B;
  if (f < MAX_IMU) {
    if (f >= ds.n_imu) return;
    const int fi = ds.imu_frame[f];
    if (mode >= 2 && !(mode == 2 && fi == 0 && d.imu[ds.imu_off + f].sum_dt < 10.0)) return;
    double *out = d.raw_imu + raw_of(f, w, (int)B, RAW_IMU);
    imu_raw(&d.imu[ds.imu_off + f], d.opt.g_norm, X + A_POSE(fi), X + A_SB(fi), X + A_POSE(fi + 1), X + A_SB(fi + 1), out,
            mode == 1 ? nullptr : out + 15 * 4, 4);
  } else {
    const int k = f - MAX_IMU;
    if (k >= ds.n_wheel) return;
    const int fi = ds.wheel_frame[k];
    if (mode >= 2 && !(mode == 2 && fi == 0 && d.wheel[ds.wheel_off + k].sum_dt < 10.0)) return;
    double *out = d.raw_wheel + raw_of(k, w, (int)B, RAW_WHEEL);
    wheel_raw(&d.wheel[ds.wheel_off + k], X + A_POSE(fi), X + A_POSE(fi + 1), X + A_EXW, X[A_IX], X[A_IX + 1], X[A_IX + 2],
              X[A_TDW], out, mode == 1 ? nullptr : out + 6 * 4, 4);
  }
}

// FUSED: small batches evaluate the raw factor inline (lane 0) — one launch less on the latency path of a single
// window; otherwise the raw residuals / Jacobians come from k_dense_raw.
enum { PRIOR_CHUNK = 4000 };      // doubles of LDS the small-batch prior item stages J0 through (dense_body; k_lin_small's dynamic LDS)
// spec (mode 0 only; BatchDev::spec): the linearisation AT THE CANDIDATE in the place of its cost pass — d is the view of the set it goes to.
template <bool FUSED>
__device__ __forceinline__ void dense_body(const BatchDev &d, int mode, int debug_out, const int w, const int f, const bool spec = false,
                                           double *pbuf = nullptr) {      // pbuf: PRIOR_CHUNK doubles of LDS for the prior's J0 (k_lin_small), or none
  const WinDesc &ds = d.desc[w];
  const WinCtl &c = d.ctl[w];
  const bool cand = mode == 1 || spec;
  if (mode == 0 && !spec && (c.done || c.reuse)) return;
  if (cand && (c.done || !c.have_step)) return;
  const int buf = cand ? 1 - c.cur : c.cur;
  const double *X = (mode >= 2) ? d.xout + (size_t)w * NA : d.x + ((size_t)w * 2 + buf) * NA;
  const int t = threadIdx.x, nthr = blockDim.x;      // (64 threads; 256 in k_lin_small: four waves share an inertial factor)
  __shared__ double raw[16], rw[16], Jraw[15 * 30], Jw[15 * 30], red[16];
  __shared__ double dx[ND], rp[ND];
#if GFBE_LIN_STAMP
  unsigned long long *dst_ = (unsigned long long *)(d.timing + (size_t)d.B * 32);
  const int dsb_ = (mode == (GFBE_LIN_STAMP_MODE == 1 ? 1 : 0) && t == 0) ? (f == MAX_IMU ? 24 : (f == MAX_IMU + MAX_WHEEL ? 28 : -1)) : -1;
#define DSTAMP(i) do { if (dsb_ >= 0) dst_[dsb_ + (i)] = wall_clock64(); } while (0)
#else
#define DSTAMP(i) do { } while (0)
#endif
  DSTAMP(0);

  if (f < MAX_IMU) {
    double *part = d.imu_part + ((size_t)w * MAX_IMU + f) * IMU_PART;
    if (f >= ds.n_imu) { if (mode == 1 && t == 0) part[IMU_PART - 1] = 0.0; return; }
    const int fi = ds.imu_frame[f];
    if (mode >= 2 && !(mode == 2 && fi == 0 && d.imu[ds.imu_off + f].sum_dt < 10.0)) { if (t == 0) part[IMU_PART - 2] = -1.0; return; }
    if (FUSED) {
      for (int q = t; q < 450; q += nthr) Jraw[q] = 0.0;
      __syncthreads();
      // the scalar quaternion algebra of the factor: lane 0 of EVERY wave takes its share of the 18 Jacobian blocks
      if ((t & 63) == 0 && (mode != 1 || t == 0))
        imu_raw(&d.imu[ds.imu_off + f], d.opt.g_norm, X + A_POSE(fi), X + A_SB(fi), X + A_POSE(fi + 1), X + A_SB(fi + 1), raw,
                mode == 1 ? nullptr : Jraw, 1, t >> 6, nthr >> 6);
    } else {   // raw residual / Jacobian of this factor from k_dense_raw (window-minor layout)
      const double *in = d.raw_imu + raw_of(f, w, d.B, RAW_IMU);
      if (t < 15) raw[t] = in[4 * t];
      if (mode != 1)
        for (int q = t; q < 450; q += nthr) Jraw[q] = imu_nz(q / 30, q % 30) ? in[4 * (15 + q)] : 0.0;
    }
    __syncthreads();
    DSTAMP(1);
    const double *S = d.imu_sqrt + (size_t)(ds.imu_off + f) * 225;   // upper triangular
    if (t < 15) { double s = 0.0; for (int b = t; b < 15; b++) s += S[t * 15 + b] * raw[b]; rw[t] = s; }
    if (mode != 1 && t >= 16 && t < 46) {
      const int col = t - 16;
      for (int a = 0; a < 15; a++) { double s = 0.0; for (int b = a; b < 15; b++) s += S[a * 15 + b] * Jraw[b * 30 + col]; Jw[a * 30 + col] = s; }
    }
    __syncthreads();
    double cst = 0.0;
    if (t == 0) for (int a = 0; a < 15; a++) cst += 0.5 * rw[a] * rw[a];
    DSTAMP(2);
    if (mode == 1) { if (t == 0) part[IMU_PART - 1] = cst; return; }
    for (int e = t; e < 930; e += nthr) {
      double s = 0.0;
      if (e < 900) { const int a = e / 30, b = e % 30; for (int r = 0; r < 15; r++) s += Jw[r * 30 + a] * Jw[r * 30 + b]; }
      else { const int a = e - 900; for (int r = 0; r < 15; r++) s += Jw[r * 30 + a] * rw[r]; }
      part[e] = s;
    }
    DSTAMP(3);
    if (t == 0) part[IMU_PART - 2] = cst;
    if (debug_out) {
      double *dbg = d.dbg_imu + ((size_t)w * MAX_IMU + f) * (15 * 31);
      for (int q = t; q < 450; q += nthr) dbg[15 + q] = Jw[q];
      if (t < 15) dbg[t] = rw[t];
    }
  } else if (f < MAX_IMU + MAX_WHEEL) {
    const int k = f - MAX_IMU;
    double *part = d.wheel_part + ((size_t)w * MAX_WHEEL + k) * WHEEL_PART;
    if (k >= ds.n_wheel) { if (mode == 1 && t == 0) part[WHEEL_PART - 1] = 0.0; return; }
    const int fi = ds.wheel_frame[k];
    if (mode >= 2 && !(mode == 2 && fi == 0 && d.wheel[ds.wheel_off + k].sum_dt < 10.0)) { if (t == 0) part[WHEEL_PART - 2] = -1.0; return; }
    if (FUSED) {
      for (int q = t; q < 132; q += nthr) Jraw[q] = 0.0;
      __syncthreads();
      // (lane 0 of EVERY wave takes its share of the Jacobian's four items, as for the inertial factor: one lane took ~16 us, the longest
      //  item of a single window's linearisation launch)
      if ((t & 63) == 0 && (mode != 1 || t == 0))
        wheel_raw(&d.wheel[ds.wheel_off + k], X + A_POSE(fi), X + A_POSE(fi + 1), X + A_EXW, X[A_IX], X[A_IX + 1], X[A_IX + 2],
                  X[A_TDW], raw, mode == 1 ? nullptr : Jraw, 1, t >> 6, nthr >> 6);
    } else {
      const double *in = d.raw_wheel + raw_of(k, w, d.B, RAW_WHEEL);
      if (t < 6) raw[t] = in[4 * t];
      if (mode != 1)
        for (int q = t; q < 132; q += nthr) Jraw[q] = wheel_nz(q / 22, q % 22) ? in[4 * (6 + q)] : 0.0;
    }
    __syncthreads();
    DSTAMP(1);
    const double *S = d.wheel_sqrt + (size_t)(ds.wheel_off + k) * 36;
    if (t < 6) { double s = 0.0; for (int b = t; b < 6; b++) s += S[t * 6 + b] * raw[b]; rw[t] = s; }
    if (mode != 1 && t >= 16 && t < 38) {
      const int col = t - 16;
      for (int a = 0; a < 6; a++) { double s = 0.0; for (int b = a; b < 6; b++) s += S[a * 6 + b] * Jraw[b * 22 + col]; Jw[a * 22 + col] = s; }
    }
    __syncthreads();
    double cst = 0.0;
    if (t == 0) for (int a = 0; a < 6; a++) cst += 0.5 * rw[a] * rw[a];
    DSTAMP(2);
    if (mode == 1) { if (t == 0) part[WHEEL_PART - 1] = cst; return; }
    for (int e = t; e < 506; e += nthr) {
      double s = 0.0;
      if (e < 484) { const int a = e / 22, b = e % 22; for (int r = 0; r < 6; r++) s += Jw[r * 22 + a] * Jw[r * 22 + b]; }
      else { const int a = e - 484; for (int r = 0; r < 6; r++) s += Jw[r * 22 + a] * rw[r]; }
      part[e] = s;
    }
    DSTAMP(3);
    if (t == 0) part[WHEEL_PART - 2] = cst;
    if (debug_out) {
      double *dbg = d.dbg_wheel + ((size_t)w * MAX_WHEEL + k) * (6 * 23);
      for (int q = t; q < 132; q += nthr) dbg[6 + q] = Jw[q];
      if (t < 6) dbg[t] = rw[t];
    }
  } else if (f > MAX_IMU + MAX_WHEEL && f <= MAX_IMU + MAX_WHEEL + MAX_PLANE) {
    // PlaneFactor of pose i (estimator.cpp:3214-3220; plane_factor.h:25-122): J^T J (16 x 16: pose_i, ex_wheel, plane_R, plane_Z),
    // J^T r, cost. mode 2: the one of frame 0 joins the MARGIN_OLD set (estimator.cpp:3441-3448).
    const int i = f - (MAX_IMU + MAX_WHEEL + 1);
    double *part = d.plane_part + ((size_t)w * MAX_PLANE + i) * PLANE_PART;
    if (i >= ds.n_plane) return;
    if (mode >= 2 && !(mode == 2 && i == 0)) { if (t == 0) part[PLANE_PART - 2] = -1.0; return; }
    if (t == 0) plane_eval(X + A_POSE(i), X + A_EXW, X + A_PLR, X[A_PLZ], ds.plane_noise_inv, raw, mode == 1 ? nullptr : Jraw);
    __syncthreads();
    double cst = 0.0;
    if (t == 0) for (int a = 0; a < 3; a++) cst += 0.5 * raw[a] * raw[a];
    if (mode == 1) { if (t == 0) part[PLANE_PART - 1] = cst; return; }
    for (int e = t; e < 272; e += nthr) {
      double s = 0.0;
      if (e < 256) { const int a = e >> 4, b = e & 15; for (int r = 0; r < 3; r++) s += Jraw[r * 16 + a] * Jraw[r * 16 + b]; }
      else { const int a = e - 256; for (int r = 0; r < 3; r++) s += Jraw[r * 16 + a] * raw[r]; }
      part[e] = s;
    }
    if (t == 0) part[PLANE_PART - 2] = cst;
  } else if (f == MAX_IMU + MAX_WHEEL + MAX_PLANE + 1) {
    // PoseAnchorFactor on Pose[0] (estimator.cpp:3004-3012; pose_anchor_factor.cpp:8-32)
    double *part = d.anchor_part + (size_t)w * ANCHOR_PART;
    if (!ds.use_anchor || mode >= 2) return;
    if (t == 0) anchor_eval(X + A_POSE(0), ds.anchor_pose, ds.anchor_sqrt_info, raw, mode == 1 ? nullptr : Jraw);
    __syncthreads();
    double cst = 0.0;
    if (t == 0) for (int a = 0; a < 6; a++) cst += 0.5 * raw[a] * raw[a];
    if (mode == 1) { if (t == 0) part[ANCHOR_PART - 1] = cst; return; }
    if (t < 42) {
      double s = 0.0;
      if (t < 36) { const int a = t / 6, b = t % 6; for (int r = 0; r < 6; r++) s += Jraw[r * 6 + a] * Jraw[r * 6 + b]; }
      else { const int a = t - 36; for (int r = 0; r < 6; r++) s += Jraw[r * 6 + a] * raw[r]; }
      part[t] = s;
    }
    if (t == 0) part[ANCHOR_PART - 2] = cst;
  } else {
    // prior: r = r0 + J0 dx ; g = J0^T r   (marginalization_factor.cpp:375-389)
    double *pg = d.prior_g + (size_t)w * (ND + 2);
    const int n = ds.prior_n;
    if (n == 0) { if (t == 0) { pg[ND] = 0.0; pg[ND + 1] = 0.0; } return; }
    if (t < ds.prior_nblk)
      prior_block_dx(X + blk_amb(ds.prior_blk_id[t]), d.prior_x0 + (size_t)w * PRIOR_X0 + ds.prior_x0_off[t], ds.prior_blk_size[t],
                     dx + ds.prior_blk_idx[t]);
    __syncthreads();
    const double *J0 = d.prior_J0 + (size_t)w * ND * ND;
    const double *r0 = d.prior_r0 + (size_t)w * ND;
    double cst = 0.0;
    DSTAMP(1);
    if (nthr >= n && pbuf) {
      // round 5: J0 goes through LDS in chunks of whole rows, fetched as contiguous memory by all the threads; thread i forms r_i from
      // its row (the products added in the old order), then thread k adds the chunk's rows to g_k (rows ascending: the old order
      // too). One sweep over J0 instead of two, no thread walking a row of its own in memory (64 cache lines per load of a wave, a
      // memory round trip every few entries: the prior took 13-17 us of a single window's ~17 us linearisation launch).
      constexpr int PBUF = PRIOR_CHUNK;
      const int ld = n | 1, R = min(PBUF / ld, n);      // rows per chunk (n <= 246: at least 16)
      double gk = 0.0;
      // (the NEXT chunk's entries are requested — sixteen per thread: PBUF <= 16 x 256 — before the current one is used: its load round
      //  trip hides behind the two passes)
      constexpr int LU = 16;
      static_assert(PRIOR_CHUNK <= LU * 256, "a thread's share of a chunk");
      double nx[LU];
      {
        const int tot0 = min(R, n) * n;
#pragma unroll
        for (int u = 0; u < LU; u++) nx[u] = (t + u * nthr < tot0) ? J0[t + u * nthr] : 0.0;
      }
      for (int i0 = 0; i0 < n; i0 += R) {
        const int rows = min(R, n - i0);
        {   // the chunk the registers hold goes to LDS: the row / column of an entry carried along instead of divided out
          const int tot = rows * n, dr = nthr / n, dc = nthr - dr * n;      // e += nthr: (row, column) += (dr, dc), with carry
          int r = t / n, c = t - r * n;
#pragma unroll
          for (int u = 0; u < LU; u++) {
            if (t + u * nthr < tot) pbuf[r * ld + c] = nx[u];
            r += dr; c += dc;
            if (c >= n) { c -= n; r++; }
          }
        }
        __syncthreads();
#if GFBE_LIN_STAMP
        if (dsb_ >= 0) dst_[i0 == 0 ? 20 : 23] = wall_clock64();
#endif
        if (i0 + rows < n) {
          const int tot1 = min(R, n - i0 - rows) * n;
          const double *src1 = J0 + (size_t)(i0 + rows) * n;
#pragma unroll
          for (int u = 0; u < LU; u++) nx[u] = (t + u * nthr < tot1) ? src1[t + u * nthr] : 0.0;
        }
        if (t >= i0 && t < i0 + rows) {
          double s = r0[t];
          const double *row = pbuf + (t - i0) * ld;
          constexpr int PU = 8;      // (reads of eight entries in flight, their products added in order)
          int k = 0;
          for (; k + PU <= n; k += PU) {
            double a[PU], b[PU];
#pragma unroll
            for (int u = 0; u < PU; u++) { a[u] = row[k + u]; b[u] = dx[k + u]; }
#pragma unroll
            for (int u = 0; u < PU; u++) s += a[u] * b[u];
          }
          for (; k < n; k++) s += row[k] * dx[k];
          rp[t] = s;
          cst += 0.5 * s * s;
        }
#if GFBE_LIN_STAMP
        if (dsb_ >= 0 && i0 == 0) dst_[21] = wall_clock64();
#endif
        if (mode != 1) {
          __syncthreads();
#if GFBE_LIN_STAMP
          if (dsb_ >= 0 && i0 == 0) dst_[22] = wall_clock64();
#endif
          if (t < n) {
            constexpr int PU = 8;
            int r = 0;
            for (; r + PU <= rows; r += PU) {
              double a[PU], b[PU];
#pragma unroll
              for (int u = 0; u < PU; u++) { a[u] = pbuf[(r + u) * ld + t]; b[u] = rp[i0 + r + u]; }
#pragma unroll
              for (int u = 0; u < PU; u++) gk += a[u] * b[u];
            }
            for (; r < rows; r++) gk += pbuf[r * ld + t] * rp[i0 + r];
          }
        }
        __syncthreads();
      }
      cst = block_sum(cst, red);
      DSTAMP(2);
      if (mode == 1) { if (t == 0) pg[ND + 1] = cst; return; }
      if (t < n) pg[t] = gk;
    } else {
    for (int i = t; i < n; i += nthr) {
      double s = r0[i];
      for (int k = 0; k < n; k++) s += J0[(size_t)i * n + k] * dx[k];
      rp[i] = s;
      cst += 0.5 * s * s;
    }
    cst = block_sum(cst, red);
    if (mode == 1) { if (t == 0) pg[ND + 1] = cst; return; }
    __syncthreads();
    for (int k = t; k < n; k += nthr) {
      double s = 0.0;
      for (int i = 0; i < n; i++) s += J0[(size_t)i * n + k] * rp[i];
      pg[k] = s;
    }
    }
    DSTAMP(3);
    if (t == 0) pg[ND] = cst;
    if (debug_out) for (int i = t; i < n; i += nthr) d.dbg_prior[(size_t)w * ND + i] = rp[i];
  }
}

template <bool FUSED>
__global__ __launch_bounds__(64, FUSED ? 1 : 4) void k_dense(BatchDev d0, int mode, int debug_out, int f0, int spec) {
  // (the set of the linearisation's outputs this pass writes: lin_view; the marginalisation's passes and the cost pass use the first)
  const BatchDev d = mode == 0 ? lin_view(d0, spec ? 1 - d0.ctl[blockIdx.y].lb : d0.ctl[blockIdx.y].lb) : d0;
  dense_body<FUSED>(d, mode, debug_out, blockIdx.y, blockIdx.x + f0, spec != 0);
}

// ---- k_dense_tp: the inertial / wheel factors and the prior of throughput batches (B >= DENSE_SPLIT_MIN_B), modes 0 and 1.
// k_dense gives every factor a 64-thread workgroup whose lanes walk short serial loops (43 000 workgroups per 2048 windows, the raw
// values fetched one 32-byte sector per double from the window-minor arrays of k_dense_raw). Here
//   slots 0 .. 19   a workgroup = factor slot f of FOUR consecutive windows, one wave per window. The raw residual / Jacobian
//                   values of the four windows share their sectors (thread -> (value q, window)); the whitening S [J | r] and
//                   [Jw | rw]^T [Jw | rw] run on the FP64 matrix cores: with X = [J | r] as two 16-column tiles, Y_t = S X_t comes
//                   out of v_mfma_f64_16x16x4_f64 in exactly the operand layout of the next product (lane (lr, lk), entry q =
//                   Y[4 q + lk][16 t + lr]), so G_00 = Y_0^T Y_0, G_10 = Y_1^T Y_0, G_11 = Y_1^T Y_1 follow without a trip through
//                   LDS: 8 + 12 instructions per inertial factor (15 x 30), 4 + 6 per wheel factor (6 x 22); J^T r is row C of G;
//   (the priors: k_prior_tp below, a launch of its own because of its LDS.)
// The cost of a factor is summed from the whitened residual in the same order in both modes. Sums differ from k_dense's in their
// association only (the small-batch kernel set keeps k_dense; the tests compare the two sets on a tolerance).
template <bool IMU> struct DenseKind;
template <> struct DenseKind<true> { enum { R = 15, C = 30, NK = 4, PART = IMU_PART, RAW = RAW_IMU }; };
template <> struct DenseKind<false> { enum { R = 6, C = 22, NK = 2, PART = WHEEL_PART, RAW = RAW_WHEEL }; };
enum { DTP_X = 4 * 512, DTP_S = 4 * 256, DTP_LDS = DTP_X + DTP_S, DTP_PRIOR0 = MAX_IMU + MAX_WHEEL };

template <bool IMU>
__device__ __forceinline__ void dense_tp_factor(const BatchDev &d, int mode, int spec, int f, int w0, double *X, double *St, int *s_act) {
  typedef DenseKind<IMU> K;
  constexpr int R = K::R, C = K::C, NK = K::NK;
  typedef double dbl4_d __attribute__((ext_vector_type(4)));
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, lr = lane & 15, lk = lane >> 4;
  const int B = d.B;
  if (t < 4) {
    const int w = w0 + t;
    int act = 0;
    if (w < B && dense_pass_active(d.ctl[w], mode, spec)) act = f < (IMU ? d.desc[w].n_imu : d.desc[w].n_wheel) ? 1 : 2;   // 2: no such factor
    s_act[t] = act;
  }
  for (int e = t; e < DTP_LDS; e += 256) X[e] = 0.0;      // (X and St are one array: pads and structural zeros)
  __syncthreads();
  {   // raw values of the four windows: one contiguous block [value q][window % 4] (raw_of), thread -> (q, window)
    const int wi = t & 3, qq = t >> 2;
    if (s_act[wi] == 1) {
      const double *in = (IMU ? d.raw_imu : d.raw_wheel) + raw_of(f, w0, B, K::RAW) + wi;
      const int nq = mode == 1 ? R : R + R * C;
      for (int q = qq; q < nq; q += 64) {
        int r = q, c = C;
        bool nz = true;
        if (q >= R) { const int e = q - R; r = e / C; c = e - r * C; nz = IMU ? imu_nz(r, c) : wheel_nz(r, c); }
        if (nz) X[((wi * 2 + (c >> 4)) * 16 + r) * 16 + (c & 15)] = in[4 * q];
      }
    }
  }
  if (s_act[wv] == 1) {   // S of the wave's window, transposed (St[k][i] = S[i][k]; S is upper triangular)
    const WinDesc &ds = d.desc[w0 + wv];
    const double *S = IMU ? d.imu_sqrt + (size_t)(ds.imu_off + f) * (R * R) : d.wheel_sqrt + (size_t)(ds.wheel_off + f) * (R * R);
    for (int e = lane; e < R * R; e += 64) { const int a = e / R, b = e - a * R; if (b >= a) St[(wv * 16 + b) * 16 + a] = S[e]; }
  }
  __syncthreads();
  const int act = s_act[wv], w = w0 + wv;       // (wave-uniform from here on; no block barrier below)
  if (act == 0) return;
  // (BatchDev::spec: the set of outputs of the window this linearisation goes to — lin_view's selection for the two arrays written here)
  const int lset = (d.spec && mode == 0) ? (spec ? 1 - d.ctl[w].lb : d.ctl[w].lb) : 0;
  double *part = IMU ? (lset ? d.imu_part2 : d.imu_part) + ((size_t)w * MAX_IMU + f) * IMU_PART
                     : (lset ? d.wheel_part2 : d.wheel_part) + ((size_t)w * MAX_WHEEL + f) * WHEEL_PART;
  if (act == 2) { if (mode == 1 && lane == 0) part[K::PART - 1] = 0.0; return; }
  const double *Xw = X + wv * 512, *Sw = St + wv * 256;
  double sa[NK], xb[NK];
#pragma unroll
  for (int kk = 0; kk < NK; kk++) { sa[kk] = Sw[(4 * kk + lk) * 16 + lr]; xb[kk] = Xw[256 + (4 * kk + lk) * 16 + lr]; }
  dbl4_d Y1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < NK; kk++) Y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[kk], xb[kk], Y1, 0, 0, 0);
  // cost: column C - 16 of tile 1 is the whitened residual — lane (lr = C - 16, lk), entry q = rw[4 q + lk] (rows >= R are zero)
  double sq = 0.0;
#pragma unroll
  for (int q = 0; q < 4; q++) sq += Y1[q] * Y1[q];
  const double s0 = __shfl(sq, C - 16, 64), s1 = __shfl(sq, C, 64), s2 = __shfl(sq, C + 16, 64), s3 = __shfl(sq, C + 32, 64);
  const double cst = 0.5 * (((s0 + s1) + s2) + s3);
  if (mode == 1) { if (lane == 0) part[K::PART - 1] = cst; return; }
#pragma unroll
  for (int kk = 0; kk < NK; kk++) xb[kk] = Xw[(4 * kk + lk) * 16 + lr];
  dbl4_d Y0 = {0.0, 0.0, 0.0, 0.0}, G00 = {0.0, 0.0, 0.0, 0.0}, G10 = {0.0, 0.0, 0.0, 0.0}, G11 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < NK; kk++) Y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[kk], xb[kk], Y0, 0, 0, 0);
#pragma unroll
  for (int kk = 0; kk < NK; kk++) {      // (rows >= 4 NK of Y are zero)
    G00 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y0[kk], Y0[kk], G00, 0, 0, 0);
    G10 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y1[kk], Y0[kk], G10, 0, 0, 0);
    G11 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y1[kk], Y1[kk], G11, 0, 0, 0);
  }
  // entry q of a tile's result: G[4 q + lk][lr]; J^T J (C x C, both triangles), J^T r = row C
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = 4 * q + lk, gi = 16 + i, gj = 16 + lr;
    if (i >= lr) part[i * C + lr] = G00[q];                 // (the lower triangle: what the assembly tables read, part_lower())
    if (gi < C) { part[gi * C + lr] = G10[q]; if (gj <= gi) part[gi * C + gj] = G11[q]; }
    else if (gi == C) { part[C * C + lr] = G10[q]; if (gj < C) part[C * C + gj] = G11[q]; }
  }
  if (lane == 0) part[K::PART - 2] = cst;
}

// The prior of one window on 256 threads: r = r0 + J0 dx, g = J0^T r, cost (marginalization_factor.cpp:359-389). J0 (n x n,
// 59 KB for the usual n = 86) is read ONCE: every thread issues its share of the loads at the start (up to PRIOR_REGS in flight per
// thread; the dependent descriptor / state loads of the dx computation run behind them), the copy lands in LDS and both products
// read it there. Measured before (a wave per row, four rows = eight loads in flight, then J0^T r from global memory again): 27 +
// 23 us of load latency per launch over 512 windows. A batch whose largest prior does not fit (n > PRIOR_LDS_N) takes the
// global-memory path (lds_n = 0).
enum { PRIOR_REGS = 32, PRIOR_LDS_N = 90 };      // 256 threads x 32 values >= 90 x 90; 90 x 90 x 8 B = 63 KB of LDS
static_assert(256 * PRIOR_REGS >= PRIOR_LDS_N * PRIOR_LDS_N, "k_prior_tp: a thread's share of J0");
static_assert(ND <= 256, "k_prior_tp: one thread per row / column of the prior (n <= GFBE_DENSE_DIM) in a 256-thread workgroup");
__global__ __launch_bounds__(256) void k_prior_tp(BatchDev d, int mode, int lds_n, int spec) {
  GFBE_SMALL_KERNEL_PRIO();
  extern __shared__ __attribute__((aligned(16))) double psm[];      // dx[ND] | r[ND] | partial sums [4][ND] | J0 [lds_n x lds_n]
  __shared__ double red[16];
  const int w = blockIdx.x, t = threadIdx.x, wv = t >> 6, lane = t & 63;
  const WinDesc &ds = d.desc[w];
  const WinCtl &c = d.ctl[w];
  if (!dense_pass_active(c, mode, spec)) return;
  // (BatchDev::spec: the set of outputs this linearisation goes to, as lin_view selects it)
  double *pg = ((d.spec && mode == 0 && (spec ? 1 - c.lb : c.lb)) ? d.prior_g2 : d.prior_g) + (size_t)w * (ND + 2);
  const int n = ds.prior_n;
  if (n == 0) { if (t == 0) { pg[ND] = 0.0; pg[ND + 1] = 0.0; } return; }
  double *dx = psm, *rp = psm + ND, *pp = psm + 2 * ND, *sJ = psm + 6 * ND;
  const double *J0 = d.prior_J0 + (size_t)w * ND * ND, *r0 = d.prior_r0 + (size_t)w * ND;
  const bool staged = n <= lds_n;
  double v[PRIOR_REGS];
  if (staged) {
#pragma unroll
    for (int j = 0; j < PRIOR_REGS; j++) { const int e = t + 256 * j; v[j] = e < n * n ? J0[e] : 0.0; }
  }
  const double r0t = t < n ? r0[t] : 0.0;
  const double *X = d.x + ((size_t)w * 2 + ((mode == 1 || spec) ? 1 - c.cur : c.cur)) * NA;
  if (t < ds.prior_nblk)
    prior_block_dx(X + blk_amb(ds.prior_blk_id[t]), d.prior_x0 + (size_t)w * PRIOR_X0 + ds.prior_x0_off[t], ds.prior_blk_size[t],
                   dx + ds.prior_blk_idx[t]);
  if (staged) {
#pragma unroll
    for (int j = 0; j < PRIOR_REGS; j++) { const int e = t + 256 * j; if (e < n * n) sJ[e] = v[j]; }
  }
  for (int e = t; e < 4 * ND; e += 256) pp[e] = 0.0;
  __syncthreads();
  // r = r0 + J0 dx: thread (h, i) sums its quarter (h of ng) of row i; the partial sums are added in the order of h
  const int ng = min(4, 256 / n), hh = t / n, ii = t - hh * n;      // (n <= 256: GFBE_DENSE_DIM = 246)
  if (staged) {
    if (hh < ng) {
      const int k0 = hh * n / ng, k1 = (hh + 1) * n / ng;
      double sacc = 0.0;
      for (int k = k0; k < k1; k++) sacc += sJ[ii * n + k] * dx[k];
      pp[hh * ND + ii] = sacc;
    }
    __syncthreads();
    if (t < n) rp[t] = r0t + (((pp[t] + pp[ND + t]) + pp[2 * ND + t]) + pp[3 * ND + t]);
  } else {
    for (int i0 = 4 * wv; i0 < n; i0 += 16) {      // a wave per row, lanes along the row, four rows in flight
      double sr[4] = {0.0, 0.0, 0.0, 0.0};
      for (int k = lane; k < n; k += 64) {
        const double x = dx[k];
#pragma unroll
        for (int u = 0; u < 4; u++) sr[u] += (i0 + u < n ? J0[(size_t)(i0 + u) * n + k] : 0.0) * x;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) sr[u] = wave_sum(sr[u]);
      if (lane == 0)
        for (int u = 0; u < 4; u++) if (i0 + u < n) rp[i0 + u] = r0[i0 + u] + sr[u];
    }
  }
  __syncthreads();
  double cst = 0.0;
  for (int i = t; i < n; i += 256) cst += 0.5 * rp[i] * rp[i];
  cst = block_sum(cst, red);
  if (mode == 1) { if (t == 0) pg[ND + 1] = cst; return; }
  // g = J0^T r: thread (h, k) sums its quarter of column k
  if (staged) {
    if (hh < ng) {
      const int i0 = hh * n / ng, i1 = (hh + 1) * n / ng;
      double sacc = 0.0;
      for (int i = i0; i < i1; i++) sacc += sJ[i * n + ii] * rp[i];
      pp[hh * ND + ii] = sacc;
    }
  } else {
    constexpr int NG = (ND + 63) / 64;
    double ag[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) ag[g] = 0.0;
#pragma unroll 4
    for (int i = wv; i < n; i += 4) {      // wave wv: rows wv, wv + 4, ...; lanes along the columns
      const double r = rp[i];
      const double *row = J0 + (size_t)i * n;
#pragma unroll
      for (int g = 0; g < NG; g++) if (lane + 64 * g < n) ag[g] += row[lane + 64 * g] * r;
    }
#pragma unroll
    for (int g = 0; g < NG; g++) if (lane + 64 * g < ND) pp[wv * ND + lane + 64 * g] = ag[g];
  }
  __syncthreads();
  for (int k = t; k < n; k += 256) pg[k] = ((pp[k] + pp[ND + k]) + pp[2 * ND + k]) + pp[3 * ND + k];
  if (t == 0) pg[ND] = cst;
}

__global__ __launch_bounds__(256) void k_dense_tp(BatchDev d, int mode, int spec) {
  GFBE_SMALL_KERNEL_PRIO();
  __shared__ double sm[DTP_LDS];
  __shared__ int s_act[4];
  const int slot = blockIdx.x, w0 = blockIdx.y * 4;
  if (slot < MAX_IMU) dense_tp_factor<true>(d, mode, spec, slot, w0, sm, sm + DTP_X, s_act);
  else dense_tp_factor<false>(d, mode, spec, slot - MAX_IMU, w0, sm, sm + DTP_X, s_act);
}
static size_t prior_tp_lds(int lds_n) { return sizeof(double) * (6 * (size_t)ND + (size_t)lds_n * lds_n); }
hipError_t dense_init_device() {   // per device, from gfbe_create (see kernels_init_device)
  return hipFuncSetAttribute((const void *)k_prior_tp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)prior_tp_lds(PRIOR_LDS_N));
}

// Small batches (one window per camera frame is the reference's call pattern): the visual tiles and the inertial / wheel /
// prior factors of a linearisation (MODE 0) or of a candidate evaluation (MODE 1) in ONE launch — the ~35 us of a single
// lane evaluating an IMU factor hide behind the visual tiles instead of following them on the stream.
__device__ __forceinline__ double candidate_tile(const BatchDev &d, const WinDesc &ds, const WinCtl &c, const int w, const int tile, const int t);
struct AcceptLocal {
  int done, have_step, iter, cur, num_successful, termination, status, reuse, lb;
  double cost, x_norm, model_change, radius, step_norm, mu, cand_cost;
};
template <class CT>
__device__ __forceinline__ void accept_body(const BatchDev &d, const int w, const int lane, CT &c, WinCtl &cg, const int cslot = 1);
__device__ __forceinline__ bool arrive_last(int *cnt, const int expected, const int lane);
// fuse (MODE 1, GFBE_FUSE_SMALL): bit 1 — a tile workgroup first forms the candidate inverse depths of its tile (the landmark half of
// k_candidate; the dense half ran at the tail of k_lm_step_fused); bit 2 — the workgroup of a window that finishes last goes on with
// k_accept.
// MODE 3 (BatchDev::spec, every iteration of the batch but the last): the candidate is LINEARISED — MODE 0's work at the candidate state,
// into the set of outputs that is not the current one — and the accepting tail makes that set the current one: the next iteration
// starts at the Schur elimination. Costs more than MODE 1 (~17 against ~14 us for a 2k-landmark window) and saves the next MODE 0 launch.
template <int MODE, bool FULL>
__global__ __launch_bounds__(LIN_SMALL_THREADS, 1) void k_lin_small(BatchDev d0, int fuse) {
  const int w = blockIdx.x, y = blockIdx.y;
  constexpr bool SPEC = MODE == 3;
  constexpr int VM = SPEC ? 0 : MODE;                   // what vis_body / dense_body do
  constexpr int KS = VM != 1 ? LIN_SMALL_KS : 1;      // (the cost-only pass is too short to gain: 9 -> 12 us when split)
  // the set of the linearisation's outputs this pass writes: the current one (MODE 0: lb = 0 unless the batch is speculative), the other
  // one (MODE 3); the cost pass and the marginalisation's set use the first whatever lb
  const BatchDev d = (MODE == 0 || SPEC) ? lin_view(d0, SPEC ? 1 - d0.ctl[w].lb : d0.ctl[w].lb) : d0;
  // LIN_SMALL_THREADS / 64 waves per workgroup. A visual tile is ONE workgroup whose first KS waves take the observation steps
  // k = wave, wave + KS, ... (vis_body's WS form; the others leave at once); the inertial / wheel / prior items use all the waves — the
  // single lane that evaluated an inertial or a wheel factor was the longest chain of the launch. lsm: the dynamic LDS of the
  // workgroup — the KS panels of a tile, or the prior's chunk of J0 (lin_small_lds_doubles).
  extern __shared__ __attribute__((aligned(16))) double lsm[];
  constexpr bool WS = KS > 1;
  const int ny_tiles = d.max_tiles;       // (one workgroup per tile in every mode)
  const bool tile_wg = y < ny_tiles;
#if GFBE_LIN_STAMP
  // diagnostics build (tools/diag_scripts/lin_stamps.py): start of workgroup 0 and the latest end per kind of item, of the LAST launch of MODE 0
  unsigned long long *ls = (unsigned long long *)(d.timing + (size_t)d.B * 32);
  if (MODE == GFBE_LIN_STAMP_MODE && y == 0 && threadIdx.x == 0) ls[0] = wall_clock64();
  if (MODE == GFBE_LIN_STAMP_MODE && y == ny_tiles && threadIdx.x == 0) ls[8] = wall_clock64();                 // first inertial item starts
  if (MODE == GFBE_LIN_STAMP_MODE && y == ny_tiles + MAX_IMU && threadIdx.x == 0) ls[9] = wall_clock64();       // first wheel item starts
  if (MODE == GFBE_LIN_STAMP_MODE && y == ny_tiles + MAX_IMU + MAX_WHEEL && threadIdx.x == 0) ls[10] = wall_clock64();   // the prior starts
#endif
  if (tile_wg) {
    if (threadIdx.x >= KS * LM_TILE) return;
    // (fuse bit 1: the tile forms its candidate inverse depths first; MODE 3: each of the tile's KS waves does, for its own lanes — the same values)
    if constexpr (WS) vis_body<VM, FULL, KS, SPEC, true>(d, 0, w, y, threadIdx.x >> 6, SPEC && (fuse & 2), lsm);
    else vis_body<VM, FULL, 1, SPEC, false>(d, 0, w, y, 0, MODE == 1 && (fuse & 2));
    if (threadIdx.x >= LM_TILE) return;      // (the tile's first wave goes on: it has summed the waves' shares)
  } else if ((MODE == 1 || SPEC) && y - ny_tiles == MAX_IMU + MAX_WHEEL + 1 + (d.any_plane ? MAX_PLANE + 1 : 0)) {
    // (round 6) the GNSS factors at the candidate: one more workgroup of the launch (launch_lin_small adds it for batches with GNSS
    // windows of up to GN_ITEM_MAX_OBS observations) instead of a launch of k_gnss behind this one
    __shared__ double gn_red[16];
    gnss_candidate_item(d, w, SPEC, gn_red);
  } else {
    dense_body<true>(d, VM, 0, w, y - ny_tiles, SPEC, lsm);
  }
#if GFBE_LIN_STAMP
  if (MODE == GFBE_LIN_STAMP_MODE && (threadIdx.x & 63) == 0) {
    const int f = y - ny_tiles;
    const int kind = tile_wg ? 1 : (f < MAX_IMU ? 2 : (f < MAX_IMU + MAX_WHEEL ? 3 : (f == MAX_IMU + MAX_WHEEL ? 4 : 5)));
    atomicMax(ls + kind, (unsigned long long)wall_clock64());
    if (tile_wg && y == 0) ls[6] = wall_clock64();
    if (tile_wg && y == ny_tiles - 1) ls[7] = wall_clock64();
    if (threadIdx.x == 0 && f == MAX_IMU) ls[11] = wall_clock64();                                      // first wheel item ends
  }
#endif
  if ((MODE == 1 || SPEC) && (fuse & 4)) {
    // (a factor's four waves are done with their stores before the workgroup arrives — dense_body leaves workgroup-uniformly —; its first wave goes on)
    if (!tile_wg) { __syncthreads(); if (threadIdx.x >= 64) return; }
    // (the scalars k_accept reads are this launch's inputs: loaded before the arrival, by every workgroup — any may be the last)
    WinCtl &c = d.ctl[w];
    AcceptLocal la;
    la.done = c.done; la.have_step = c.have_step; la.iter = c.iter; la.cur = c.cur; la.num_successful = c.num_successful;
    la.termination = c.termination; la.status = c.status; la.reuse = c.reuse; la.lb = c.lb;
    la.cost = c.cost; la.x_norm = c.x_norm; la.model_change = c.model_change; la.radius = c.radius; la.step_norm = c.step_norm; la.mu = c.mu;
    la.cand_cost = c.cand_cost;
    if (!arrive_last(d.win_cnt + 2 * w + 1, gridDim.y, threadIdx.x)) return;
#if GFBE_LIN_STAMP
    if (MODE == GFBE_LIN_STAMP_MODE && threadIdx.x == 0) ls[12] = wall_clock64();      // the last workgroup has arrived
#endif
    accept_body(d, w, threadIdx.x, la, c, SPEC ? 2 : 1);      // (MODE 3: the candidate's costs are those of its linearisation)
    if (threadIdx.x == 0) {
      if (SPEC) c.lb = la.lb;
      c.done = la.done; c.have_step = la.have_step; c.cur = la.cur; c.num_successful = la.num_successful; c.termination = la.termination;
      c.status = la.status; c.reuse = la.reuse; c.cost = la.cost; c.x_norm = la.x_norm; c.radius = la.radius; c.mu = la.mu; c.cand_cost = la.cand_cost;
#if GFBE_LIN_STAMP
      if (MODE == GFBE_LIN_STAMP_MODE) ls[13] = wall_clock64();      // the accepting tail is done
#endif
    }
  }
}

// =============================================================================================
// k_schur: Schur elimination of the 1-D inverse-depth blocks, E = sum_l w_l h_l h_l^T, on the FP64
// matrix cores. One workgroup per (window, start frame s): all its landmarks have their H_pl rows on
// the contiguous dims [6s, 73). Landmark tiles (64 rows, pre-multiplied by sqrt(w_l)) are staged in
// LDS as a 64 x 80 panel whose column 73 carries sqrt(w_l) g_l, so E[:,73] is the reduced gradient
// share. Each wave owns a few 16x16 output tiles and keeps them in registers across all landmark
// tiles of the start frame: v_mfma_f64_16x16x4_f64 with A = panel^T, B = panel.
//   solve:  w_l = s_l^2 / (s_l^2 Hll + mu clamp(s_l^2 Hll))      (Jacobi-scaled, mu-regularised)
//   marg :  w_l = 1 / Hll                                         (marginalization_factor.cpp:286-292)
// =============================================================================================
typedef double dbl4_t __attribute__((ext_vector_type(4)));
#ifndef GFBE_SCHUR_COMPACT
#define GFBE_SCHUR_COMPACT 1      // 0 (diagnostics build): the absolute column layout of rounds 1-3 in the solve's Schur panels as well
#endif
#ifndef HS_LD
#define HS_LD 75   // LDS row stride of the landmark panel: odd (the 64 lanes that stage one column spread over 32 bank pairs, two-way instead of
                   // four-way) and just wide enough for the 74 columns in use — the matrix-core operand loads of the last 16-column block run
                   // into the next row's first columns: products that only reach output columns >= 75, which nobody reads (38.4 KB: four
                   // workgroups per CU)
#endif

__device__ __forceinline__ int schur_pair(int I, int J) { return I * 5 - I * (I - 1) / 2 + (J - I); }   // I <= J < 5

// One workgroup per (window, GROUP of start frames). Throughput batches (>= 32 windows): the landmark tiles of start frames {0, 1},
// {2}, {3, 4, 5}, {6 .. 10} go through the same accumulators one after the other (a window's tiles are sorted by start frame: a
// group is a contiguous range of tiles), so that a window leaves FOUR partials instead of eleven — k_assemble's gather of E reads 4
// instead of 11 values per entry and two thirds of the partial stores are gone. Small batches keep one group per start frame and deal
// its tiles over TWO workgroups (22 partials): a start frame of a 2k-landmark window has 4-5 tiles of ~3.5 us each, and a single
// window's latency wants them side by side. The marginalisation pass (start frame 0 alone, one workgroup) uses slot 0.
// (the groups follow the tile rows a start frame's landmarks can reach — I0 = 6 s / 16 steps at s = 3 and s = 6 — so that no group
//  multiplies tile pairs its later start frames do not touch; work ~ tiles x pairs: {0,1} 30, {2} 15, {3,4,5} 30, {6..10} 12)
__device__ __forceinline__ int schur_group_first(int g, int ng) { return ng >= NF ? g : (g == 0 ? 0 : (g == 1 ? 2 : (g == 2 ? 3 : 6))); }
__device__ __forceinline__ void schur_body(const BatchDev &d, const int marg, const int w, const int grp) {
  const WinDesc &ds = d.desc[w];
  WinCtl &c = d.ctl[w];
  if (!marg && (c.done || c.reuse)) return;
  if (marg && grp != 0) return;
  const int sub = (!marg && d.schur_groups == 2 * NF) ? 2 : 1;        // workgroups per start-frame group (small batches: two, tiles dealt alternately)
  const int ng = marg ? NF : d.schur_groups / sub, half = grp % sub;
  const int gi = grp / sub;
  const int s_first = schur_group_first(gi, ng), s_end = marg ? 1 : (gi == ng - 1 ? NF : schur_group_first(gi + 1, ng));
  const int tb = ds.sf_tile_begin[s_first], te = ds.sf_tile_begin[s_end];
  const int tfirst = sub == 2 ? tb + half : tb + ((d.rank - tb) % d.world + d.world) % d.world;   // first tile of this workgroup / rank (world 1: tb)
  if (tfirst >= te) {
    // solve: the partial stays zero (zeroed at upload; the structure never changes). Marginalisation: slot 0 may hold the solve's
    // partial of the group {0, 1} — no landmark (of this rank) starts in frame 0, so its Schur term is zero
    if (marg) for (int q = threadIdx.x; q < SCHUR_STRIDE; q += 256) d.schur_part[((size_t)w * d.schur_groups) * SCHUR_STRIDE + q] = 0.0;
    return;
  }
  __shared__ double hs[LM_TILE * HS_LD + 8];      // (+ the overrun of the last row's last block)
  const size_t TL = d.tot_lm;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // output tiles of this group: tile rows/cols >= I0 (of its first start frame), upper pairs, round-robin over the 4 waves
  // COMPACT panel (the solve; round 4): the columns of a group's panel are taken relative to its first start frame, the gradient
  // column first — column 0 = sqrt(w) g_l, dim a at 1 + a - 6 s_first — so that a tile's non-zero columns are [0, 6 (s - s_first + m0
  // + 1)] whatever the group: the 16-column blocks it multiplies are 0 .. jl, (jl + 1)(jl + 2) / 2 pairs, against the absolute columns'
  // blocks of [6 s, 6 (s + m0 + 1)) plus the block of column 73 (counted on the 2k-landmark windows of the bench: 219 instead of 326
  // tile pairs per window and linearisation). Pairs are numbered column-major — (I, J) at J (J + 1) / 2 + I — so the active ones of
  // any tile are a prefix dealt evenly over the four waves. The marginalisation pass keeps the absolute layout k_marg reads.
  const bool compact = GFBE_SCHUR_COMPACT && !marg;
  const int coff = compact ? 1 - 6 * s_first : 0;                     // column of dim a: a + coff
  const int I0 = compact ? 0 : (6 * s_first) / 16;
  // slot q of this wave owns the (4 q + wave)-th upper tile pair (I, J), I <= J — all wave-uniform
  // scalars, and the four accumulators are separate named registers (no dynamic indexing of AGPRs).
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int nside = compact ? (1 + 6 * (NF - s_first) + (d.vis_full ? 7 : 0) + 15) / 16 : 5 - I0, npairs = nside * (nside + 1) / 2;
  int pI[4], pJ[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    int idx = 4 * q + wv, I = I0;
    pI[q] = -1; pJ[q] = -1;
    if (idx < npairs) {
      if (compact) { int J = 0; while ((J + 1) * (J + 2) / 2 <= idx) J++; pJ[q] = J; pI[q] = idx - J * (J + 1) / 2; }
      else { while (idx >= 5 - I) { idx -= 5 - I; I++; } pI[q] = I; pJ[q] = I + idx; }
    }
  }
  dbl4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  const int lr = lane & 15, lk = lane >> 4;
  // thread (l, part): landmark l of the tile; part 0 stages the common columns (start pose, extrinsic, td,
  // gradient), parts 1..3 the observing poses s+1+k with k = part-1, part+2, ... All global loads of a tile
  // are issued together (rows beyond a track's length are zero in memory) and the NEXT tile's loads are
  // in flight while the matrix cores work on the current one.
  const int l = t & 63, part = t >> 6;
  // sqrt(w_l) from Hll, s_l and the window's mu instead of lm_sw: a batch with the second set (BatchDev::spec) whose current set's weights
  // were formed with another mu (WinCtl::sw_mu)
  const bool spec_sw = !marg && d.spec && c.sw_mu[c.lb] != c.mu;
  const double spec_mu = spec_sw ? c.mu : 0.0;
  const bool hc_full = marg || d.vis_full;      // the linearisation of a batch with constant extrinsic / td everywhere leaves rows 6..12 of lm_hC alone
  // What a thread holds of one landmark tile between its loads and its staging. DEEP (throughput batches): the tiles t + 1 AND t + 2
  // are in flight while the matrix cores work on tile t — with one tile ahead a light tile (a few tile pairs) took a whole
  // memory round trip under load, ~5 us, whatever its arithmetic; two register sets, two workgroups per CU instead of three.
  // (the two register sets are plain local arrays and the tile body a macro over their names: a struct passed to a lambda by
  //  reference left part of it in scratch memory)
  // start frame of a tile from the descriptor's table (scalar loads): no dependent load before the rows can be fetched
  // (the table staged in LDS once and the trust-region scalar read once: a vector-memory load inside the tile loop would make every
  //  staging wait for ALL loads in flight — the counter is in order — and undo the two-deep prefetch)
  __shared__ int s_sf[NF + 1];
  __shared__ double s_rt[NF][12];       // yrows: R_f | t_f = P_f - P_0 of the linearisation point (FrameConst of the pair table)
  const bool yrows = !marg && !d.vis_full;      // the rows are in the compressed form of k_vis<0, false> (gfbe_devutil.h)
  if (t <= NF) s_sf[t] = ds.sf_tile_begin[t];
  if (yrows && t < NF * 12) s_rt[t / 12][t % 12] = d.pc[(((size_t)w * 3 + c.cur) * NPAIR + (t / 12) * (NF + 1)) * PC_DOUBLES + LM_RT_OFF + t % 12];
  const int lm_off_c = ds.lm_off;
  __syncthreads();
  auto sframe_of = [&](int tile) __attribute__((always_inline)) { int sf = s_first; for (int q = s_first + 1; q < s_end; q++) sf += (tile >= s_sf[q]) ? 1 : 0; return sf; };
  // Thread (l, part): landmark l of the tile; part 0 holds the common columns (start pose 6, extrinsic 6, td, gradient) and the tenth
  // observing pose (k = 9: start frame 0 only), parts 1..3 the observing poses k = part - 1, part + 2, part + 5: 20 doubles each.
  // Branch-free and always the same number of vector-memory operations (22 loads per thread, whatever its part and the tile): the
  // compiler's wait-count insertion can then let the younger set's loads stay in flight while the older set is staged
  // (s_waitcnt vmcnt(23+) instead of vmcnt(0)); rows a thread does not need are loaded from a neighbouring valid row and ignored.
#define SCHUR_PV 20
#define SCHUR_PREFETCH(PV, PHLL, PSL, PINFO, PS, TILE)                                                                 \
  {                                                                                                                      \
    const int tile_ = min((TILE), te - 1);                                                                               \
    const int slot_ = lm_off_c + tile_ * LM_TILE + l;                                                                    \
    PS = sframe_of(tile_);                                                                                               \
    const int klast_ = max(NF - 2 - PS, 0);       /* last observing pose index of the tile's start frame */              \
    PINFO = d.lm_info[slot_];                                                                                            \
    PHLL = (marg || spec_sw ? d.lm_Hll : d.lm_sw)[slot_];      /* solve: sqrt(w_l), left by the linearisation */         \
    if (spec_sw) PSL = d.lm_sl[slot_];              /* (speculative batches: Hll and s_l, sqrt(w_l) formed at the staging) */    \
    const double *gl_ = d.lm_gl + slot_;                                                                                 \
    _Pragma("unroll") for (int idx_ = 0; idx_ < SCHUR_PV; idx_++) {                                                      \
      const int u_ = idx_ / 6, q_ = idx_ - 6 * u_;                                                                       \
      const int kk_ = min(part - 1 + 3 * u_, klast_), k9_ = min(9, klast_);                                              \
      /* (yrows: a factor's entry is d (3 doubles); the other three slots of its block take the landmark's x = lm_hC rows 3..5) */ \
      const int q9_ = idx_ - HC - 1;                                                                                     \
      const double *src_ = part == 0 ? (idx_ < HC ? d.lm_hC + (size_t)((idx_ < 6 || hc_full) ? idx_ : 5) * TL + slot_   \
                                                  : (idx_ == HC ? gl_ : ((yrows && q9_ >= 3) ? d.lm_hC + (size_t)q9_ * TL + slot_ \
                                                                                             : d.lm_hP + ((size_t)k9_ * 6 + q9_) * TL + slot_))) \
                                     : ((yrows && q_ >= 3) ? d.lm_hC + (size_t)q_ * TL + slot_ : d.lm_hP + ((size_t)kk_ * 6 + q_) * TL + slot_); \
      PV[idx_] = *src_;                                                                                                  \
    }                                                                                                                    \
  }
  const bool stamp_wg = (w == 0 && grp == 0 && t == 0 && !marg);
  double *stamp = d.timing + 8;
  if (stamp_wg) { stamp[0] = (double)wall_clock64(); stamp[5] = (double)clock64(); }
  const int tstep = sub == 2 ? 2 : d.world;
#define SCHUR_SLOT(Q, ACC)                                                                          \
    if (pI[Q] >= 0 && (compact ? pJ[Q] <= jl : ((pI[Q] <= jl || pI[Q] == 4) && (pJ[Q] <= jl || pJ[Q] == 4)))) {   \
      const double *pa = hs + 16 * pI[Q] + lr + lk * HS_LD, *pb = hs + 16 * pJ[Q] + lr + lk * HS_LD; \
      _Pragma("unroll") for (int hf = 0; hf < 2; hf++) {      /* (operands eight k-steps at a time: 32 instead of 64 registers) */ \
        double va[LM_TILE / 8], vb[LM_TILE / 8];                                                    \
        _Pragma("unroll") for (int kk = 0; kk < LM_TILE / 8; kk++) { va[kk] = pa[4 * (8 * hf + kk) * HS_LD]; vb[kk] = pb[4 * (8 * hf + kk) * HS_LD]; } \
        _Pragma("unroll") for (int kk = 0; kk < LM_TILE / 8; kk++) ACC = __builtin_amdgcn_mfma_f64_16x16x4f64(va[kk], vb[kk], ACC, 0, 0, 0); \
      }                                                                                             \
    }
  // one tile: stage the register set into the LDS panel, refill the set with tile REFILL (if < te), multiply.
  // The tracks of a tile are sorted longest first (pm0: its first landmark, lane 0 of every wave): its rows are zero beyond the pose
  // columns of the first track's last observer, jl = the last 16-column block they reach (absolute layout: block 4 holds the
  // extrinsic / td / gradient columns of every row). Tile pairs outside are products of zeros: skipped (the accumulators keep their bits).
#define SCHUR_RUN_TILE(PV, PHLL, PSL, PINFO, PS, TILE, REFILL)                                                          \
  {                                                                                                                      \
    __syncthreads();                                                                                                     \
    if (stamp_wg && (TILE) == tfirst) stamp[1] = (double)wall_clock64();                                                 \
    const int ps = PS, pm0 = __builtin_amdgcn_readfirstlane(PINFO);                                                      \
    const int jl = __builtin_amdgcn_readfirstlane(compact ? (d.vis_full ? nside - 1 : (6 * (ps - s_first + ((pm0 >> 8) & 0xff) + 1)) >> 4) \
                                                          : (6 * (ps + ((pm0 >> 8) & 0xff) + 1) - 1) >> 4);              \
    {                                                                                                                    \
      const int s = ps, kmax = NF - 1 - s;          /* this tile's start frame */                                        \
      const bool valid = (PINFO >> 24) & 1;                                                                              \
      const int m = (PINFO >> 8) & 0xff;                                                                                 \
      const bool is_const = (PINFO >> 16) & 1;                                                                           \
      double sw = 0.0;                                                                                                   \
      if (valid && m > 0 && (marg || !is_const))                                                                         \
        sw = marg ? sqrt((PHLL > d.opt.marg_eps) ? 1.0 / PHLL : 0.0) : PHLL;                                             \
      if (spec_sw && valid && m > 0 && !is_const) {      /* k_vis<0>'s expression, with the mu of THIS iteration */              \
        const double hs2_ = PSL * PSL * PHLL;                                                                            \
        sw = sqrt(PSL * PSL / (hs2_ + spec_mu * clamp_diag(hs2_)));                                                      \
      }                                                                                                                  \
      double *row = hs + l * HS_LD + coff;                                                                               \
      if (part == 0) {                                                                                                   \
        for (int q = compact ? 6 * s_first : 16 * I0; q < 6 * s; q++) row[q] = 0.0;                                      \
        double blk0[6];                                                                                                  \
        if (yrows) lm_row_block(s_rt[s], PV, PV + 3, true, blk0);                     /* [ D ; Ri^T ((x - t_i) x D) ] */ \
        else { _Pragma("unroll") for (int q = 0; q < 6; q++) blk0[q] = PV[q]; }                                          \
        _Pragma("unroll") for (int q = 0; q < 6; q++) { row[6 * s + q] = sw * blk0[q]; row[T_EX + q] = hc_full ? sw * PV[6 + q] : 0.0; } \
        row[T_TD] = hc_full ? sw * PV[12] : 0.0;                                                                         \
        if (compact) {                                                                                                   \
          row[-coff] = sw * PV[HC];                                     /* column 0: the gradient */                     \
          for (int q = NV + coff; q < min(16 * nside, (int)HS_LD); q++) row[q - coff] = 0.0;     /* behind td, up to the last block any tile of the group multiplies */ \
        } else {                                                                                                         \
          row[NV] = sw * PV[HC];                                                                                         \
          _Pragma("unroll") for (int q = NV + 1; q < HS_LD; q++) row[q] = 0.0;                                           \
        }                                                                                                                \
        if (9 < kmax) {                                                  /* the tenth observing pose (start frame 0) */  \
          const bool written = valid && 9 < m;                                                                           \
          double blk9[6];                                                                                                \
          if (yrows) lm_row_block(s_rt[min(s + 10, NF - 1)], PV + HC + 1, PV + 3, false, blk9);                          \
          else { _Pragma("unroll") for (int q = 0; q < 6; q++) blk9[q] = PV[HC + 1 + q]; }                               \
          _Pragma("unroll") for (int q = 0; q < 6; q++) row[6 * (s + 10) + q] = written ? sw * blk9[q] : 0.0;            \
        }                                                                                                                \
      } else {                                                                                                           \
        _Pragma("unroll") for (int u = 0; u < 3; u++) {                                                                  \
          const int k = part - 1 + 3 * u;                                                                                \
          if (k < kmax) {                                                                                                \
            /* (rows from the landmark's track length on were never written — lm_hP is not cleared at upload: zeros, not products) */ \
            const bool written = valid && k < m;                                                                         \
            double blkk[6];                                                                                              \
            if (yrows) lm_row_block(s_rt[s + 1 + k], PV + u * 6, PV + u * 6 + 3, false, blkk);    /* [ -d ; Rj^T (d x (x - t_j)) ] */ \
            else { _Pragma("unroll") for (int q = 0; q < 6; q++) blkk[q] = PV[u * 6 + q]; }                              \
            _Pragma("unroll") for (int q = 0; q < 6; q++) row[6 * (s + 1 + k) + q] = written ? sw * blkk[q] : 0.0;       \
          }                                                                                                              \
        }                                                                                                                \
      }                                                                                                                  \
    }                                                                                                                    \
    __syncthreads();                                                                                                     \
    if (stamp_wg && (TILE) == tfirst) stamp[2] = (double)wall_clock64();                                                 \
    SCHUR_PREFETCH(PV, PHLL, PSL, PINFO, PS, (REFILL))         /* (past the last tile: a repeat of it, never staged) */ \
    SCHUR_SLOT(0, acc0)                                                                                                  \
    SCHUR_SLOT(1, acc1)                                                                                                  \
    SCHUR_SLOT(2, acc2)                                                                                                  \
    SCHUR_SLOT(3, acc3)                                                                                                  \
  }
  double preA[SCHUR_PV], aHll = 0.0, aSl = 0.0;
  int aInfo = 0, aS = 0;
  SCHUR_PREFETCH(preA, aHll, aSl, aInfo, aS, tfirst)
  // (measured, round 4: TWO tiles ahead — a second register set, the compiler's wait counts letting the younger set stay in flight —
  //  was slower at two and at three workgroups per CU, 183-198 against 169-175 us per 512 windows: the kernel is not bound by the
  //  latency of its loads)
  for (int tile = tfirst; tile < te; tile += tstep) SCHUR_RUN_TILE(preA, aHll, aSl, aInfo, aS, tile, tile + tstep)
#undef SCHUR_RUN_TILE
#undef SCHUR_SLOT
#undef SCHUR_PREFETCH
#undef SCHUR_PV
  if (stamp_wg) { stamp[3] = (double)wall_clock64(); stamp[4] = (double)(te - tb); stamp[6] = (double)clock64(); }
  double *out = d.schur_part + ((size_t)w * d.schur_groups + grp) * SCHUR_STRIDE;
#define SCHUR_OUT(Q, ACC)                                                         \
  if (pI[Q] >= 0) {                                                               \
    double *o = out + (size_t)(compact ? pJ[Q] * (pJ[Q] + 1) / 2 + pI[Q] : schur_pair(pI[Q], pJ[Q])) * 256; \
    _Pragma("unroll") for (int r = 0; r < 4; r++) o[(lk + 4 * r) * 16 + lr] = ACC[r]; \
  }
  SCHUR_OUT(0, acc0)
  SCHUR_OUT(1, acc1)
  SCHUR_OUT(2, acc2)
  SCHUR_OUT(3, acc3)
#undef SCHUR_OUT
}
#ifndef GFBE_SCHUR_WGS
#define GFBE_SCHUR_WGS 4
#endif
__global__ __launch_bounds__(256, GFBE_SCHUR_WGS) void k_schur(BatchDev d0, int marg) {
  const BatchDev d = marg ? d0 : lin_view(d0, d0.ctl[blockIdx.x].lb);
  schur_body(d, marg, blockIdx.x, blockIdx.y);   // group-major dispatch: the heavy first group of every window first
}
// =============================================================================================
// k_linschur (round 6): the linearisation of the visual factors AND the landmark elimination of
// a throughput batch in ONE launch — k_vis<0, false> + k_schur(solve) — one workgroup per (window, group of start frames), four waves.
// The two kernels exchanged every factor's d = G^T w through HBM (lm_hP: 24 B per factor written by one, re-read by the other; 465 MB + 465 MB
// per 2048 windows) and every landmark's [D | x | g_l | sqrt(w_l)]; here a tile's rows go from the evaluating lanes' registers straight
// into the LDS panel the matrix cores multiply. Per landmark tile (64 landmarks of one start frame, lane = landmark in EVERY wave):
//   evaluate   role q takes the observation steps k = q, q + 4, q + 8 (one pose pair each): visual_lin_y per factor, the 7 x 7
//              [Y r]^T [Y r] of the step's 64 factors on the matrix cores through the wave's own 64 x 17 panel (which lives INSIDE the
//              Schur panel's storage: the two are never live together), d_k kept in registers; the wave's share of H_ll, g_l, D, cost
//   barrier A  the four shares summed in ROLE order by every wave for its lane (same bits in all four); sqrt(w_l); role 2 stores the
//              per-landmark outputs k_lm_step / k_candidate read (H_ll, g_l, D, x, s_l, sqrt(w_l)) and the tile's cost
//   stage      every wave writes the H_pl blocks of ITS steps — sqrt(w_l) [ -d ; R_j^T (d x (x - t_j)) ] from registers — into the 64 x 75
//              panel (compact columns, exactly k_schur's), role 3 (the fewest steps) the start pose's block, the gradient column and the zeros
//   barrier B  multiply: the <= 15 tile pairs dealt over the waves as in k_schur, accumulators in registers across the group's tiles
//   barrier C  (the next tile's evaluation overwrites the panel)
// Same quantities as the two kernels; the landmark sums are added in another order (per role, then over the roles), so H_ll, g_l, D move
// in their last bits against k_vis<0> — the small-batch kernel set and the 20-column / sharded paths keep the old kernels, compared at
// tolerance like every other pair of kernel sets (tests/test_gpu_linschur.py).
// d_k still goes to lm_hP (one write, one read by k_lm_step; the slow E rebuild of a mu retry reads it too) — see WRITE_D.
// SPEC: the linearisation at the candidate, into the other set of outputs (vis_body's SPEC); the Schur partial then belongs to that set
// too (schur_part2, lin_view). gate_mu (not SPEC; batches with the second set): the launch in front of every later iteration — only
// the windows whose current set was formed with another mu than the window's (after TrustRegionMinimizer::HandleInvalidStep: mu x 10
// on the same linearisation) are evaluated again; k_visasm records the mu of a set that is about to be solved (WinCtl::sw_mu).
// MEASURED SLOWER than the two kernels (profiles/r6_linschur_ab.txt; gfbe_options.merge_lin_schur is 0 by default): neither kernel
// is bound by the traffic the merge removes.
// =============================================================================================
#ifndef GFBE_LINSCHUR_WGS
#define GFBE_LINSCHUR_WGS 3
#endif
#ifndef GFBE_LINSCHUR_WRITE_D
#define GFBE_LINSCHUR_WRITE_D 1
#endif
#define LS_XLD 17
template <bool SPEC>
__global__ __launch_bounds__(256, GFBE_LINSCHUR_WGS) void k_linschur(BatchDev d0, int head_in, int gate_mu) {
  const int w = blockIdx.x;
  // heavy groups first: {0, 1}, {3, 4, 5}, {2}, {6 .. 10}
  const int grp = blockIdx.y == 1 ? 2 : (blockIdx.y == 2 ? 1 : (int)blockIdx.y);
  const WinCtl &c = d0.ctl[w];
  if (c.done) return;
  if (SPEC) { if (!c.have_step) return; }
  else { if (c.reuse) return; if (gate_mu && c.sw_mu[c.lb] == c.mu) return; }
  const int lbw = SPEC ? 1 - c.lb : c.lb;
  const BatchDev d = lin_view(d0, lbw);
  const WinDesc &ds = d.desc[w];
  const int s_first = schur_group_first(grp, SCHUR_GROUPS), s_end = grp == SCHUR_GROUPS - 1 ? (int)NF : schur_group_first(grp + 1, SCHUR_GROUPS);
  const int tb = ds.sf_tile_begin[s_first], te = ds.sf_tile_begin[s_end];
  if (tb >= te) return;      // (the partial stays the zero of the upload, in both sets)
  __shared__ double hs[LM_TILE * HS_LD + 8];      // the Schur panel; during the evaluation: the four waves' [Y r] panels (4 x 64 x 17 doubles)
  static_assert(4 * LM_TILE * LS_XLD <= LM_TILE * HS_LD + 8, "the evaluation panels live inside the Schur panel");
  __shared__ double Psum[4][5][LM_TILE];
  __shared__ double Pcost[4];
  __shared__ PairConstY pcs[NF];
  __shared__ FrameConst fcs;
  __shared__ double s_rt[NF][12];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  // ROLES, not waves: which observation steps / tile pairs a wave takes rotates with the window (and, for the evaluation, with the tile).
  // Wave q of every workgroup sits on SIMD q, and role 0 is the heaviest one (steps 0, 4, 8; the first pair of every four): without the
  // rotation SIMD 0 of a CU carries the heavy role of all its workgroups. Every sum is taken in ROLE order, so nothing depends on it.
  // (Measured: no effect, 347 -> 342 us per 512 windows.)
  const int rot = __builtin_amdgcn_readfirstlane((int)(((unsigned)w * 0x9E3779B1u) >> 20) + grp);
  const int mr = (wv + rot) & 3;
  const size_t TL = d.tot_lm;
  const int cur = c.cur;      // (every field of the control block is read ONCE, here: a load behind the loop's stores would wait for them)
  const int buf = SPEC ? 1 - cur : cur;
  const double *X = d.x + ((size_t)w * 2 + buf) * NA;
  const double *pcw = d.pc + ((size_t)w * 3 + buf) * NPAIR * PC_DOUBLES;
  const bool chead = SPEC && head_in;
  const double c1 = c.c1, c2 = c.c2;
  const bool first = !SPEC && c.iter == 0;
  const double mu_w = SPEC ? mu_after_accept(c.mu) : c.mu;
  const double sq = d.opt.vis_sqrt_info, delta = d.opt.huber_delta;
  const int jac_scale = d.opt.jacobi_scaling;
  const double td = X[A_TD];
  int sfb[NF + 1];
#pragma unroll
  for (int q = 0; q <= NF; q++) sfb[q] = ds.sf_tile_begin[q];
  const int lm_off_c = ds.lm_off;
  if (t < NF * 12) s_rt[t / 12][t % 12] = pcw[(size_t)(t / 12) * (NF + 1) * PC_DOUBLES + LM_RT_OFF + t % 12];
  // ---- Schur side: this wave's tile pairs (compact panel, pairs numbered column-major: k_schur)
  const int coff = 1 - 6 * s_first;
  const int nside = (1 + 6 * (NF - s_first) + 15) / 16, npairs = nside * (nside + 1) / 2;
  int pI[4], pJ[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int idx = 4 * q + mr;
    pI[q] = -1; pJ[q] = -1;
    if (idx < npairs) { int J = 0; while ((J + 1) * (J + 2) / 2 <= idx) J++; pJ[q] = J; pI[q] = idx - J * (J + 1) / 2; }
  }
  dbl4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  const int lr = lane & 15, lk = lane >> 4;
  double *const xs = hs + (size_t)wave * (LM_TILE * LS_XLD);
  // ---- what a lane holds of a tile before its evaluation (the NEXT tile's is requested before the current one is multiplied)
  int p_info;
  double p_pt[3], p_lam, p_sl = 1.0, p_vl = 0.0, p_yl = 0.0, p_ob[3][2];
#define LS_PREFETCH(TILE)                                                                                           \
  {                                                                                                                  \
    const int er_ = (wv + rot + (TILE)) & 3;                                                                         \
    const int tile_ = min((TILE), te - 1);                                                                           \
    const int slot_ = lm_off_c + tile_ * LM_TILE + lane;                                                             \
    p_info = d.lm_info[slot_];                                                                                       \
    _Pragma("unroll") for (int q = 0; q < 3; q++) p_pt[q] = d.lm_pts[(size_t)q * TL + slot_];                        \
    p_lam = d.lam[(size_t)(chead ? cur : buf) * TL + slot_];                                                         \
    if (chead) { p_vl = d.lm_vl[slot_]; p_yl = d.lm_yl[slot_]; }                                                     \
    if (!first) p_sl = d.lm_sl[slot_];                                                                               \
    _Pragma("unroll") for (int u = 0; u < 3; u++) {                                                                  \
      const int k_ = min(er_ + 4 * u, (int)MAXOBS - 1);                                                              \
      p_ob[u][0] = d.lm_obs[((size_t)k_ * 5 + 0) * TL + slot_];                                                      \
      p_ob[u][1] = d.lm_obs[((size_t)k_ * 5 + 1) * TL + slot_];                                                      \
    }                                                                                                                \
  }
  LS_PREFETCH(tb)
  int cur_sf = -1;
  for (int tile = tb; tile < te; tile++) {
    int sf = s_first;
#pragma unroll
    for (int q = 1; q < NF; q++) sf += (q > s_first && q < s_end && tile >= sfb[q]) ? 1 : 0;
    if (sf != cur_sf) {      // pair records (sf, j), j > sf, and the start frame's own constants (nobody reads them during a multiply)
      const double *src = pcw + (size_t)sf * NF * PC_DOUBLES;
      for (int e = t; e < (NF - 1 - sf) * (int)PCY_DOUBLES; e += 256) {
        const int j = sf + 1 + e / (int)PCY_DOUBLES, q = e % (int)PCY_DOUBLES;
        ((double *)&pcs[j])[q] = src[(size_t)j * PC_DOUBLES + q];
      }
      if (t < (int)FC_DOUBLES) ((double *)&fcs)[t] = src[(size_t)sf * PC_DOUBLES + t];
      cur_sf = sf;
    }
    __syncthreads();      // (C: the previous tile's multiply is over — the panel's storage is free; the records are staged)
    const int er = (wv + rot + tile) & 3;      // this wave's role in the tile's evaluation
    const int slot = lm_off_c + tile * LM_TILE + lane;
    const int info = p_info;
    const bool valid = (info >> 24) & 1;
    const int m = valid ? ((info >> 8) & 0xff) : 0;
    const bool is_const = (info >> 16) & 1;
    int mmax = m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mmax = max(mmax, __shfl_xor(mmax, o, 64));
    mmax = __builtin_amdgcn_readfirstlane(mmax);
    double lam = p_lam;
    const double sl_old = p_sl;
    if (chead) {      // x_cand = x + s_l (c1 v_l + c2 y_l): candidate_tile's arithmetic (every wave forms it, role 0 stores it)
      double d2, n2;
      lam = candidate_lm(lam, sl_old, p_vl, p_yl, c1, c2, valid && !is_const && m > 0, d2, n2);
      if (er == 0) {
        d.lam[(size_t)(1 - cur) * TL + slot] = lam;
        d2 = wave_sum(d2); n2 = wave_sum(n2);
        if (lane == 0) { double *o = d.tile_cand + ((size_t)w * d.max_tiles + tile) * 4; o[1] = d2; o[2] = n2; }
      }
    }
    const double yinv_l = 1.0 / lam;
    const double ycx = p_pt[0] * yinv_l, ycy = p_pt[1] * yinv_l, ycz = p_pt[2] * yinv_l;
    vec3 yf, yx;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      yf[a] = __builtin_fma(fcs.W(a, 0), ycx, __builtin_fma(fcs.W(a, 1), ycy, fcs.W(a, 2) * ycz));
      yx[a] = (yf[a] + fcs.wt[a]) + fcs.dPc[a];      // from the window's origin P_0
    }
    double *xr = xs + lane * LS_XLD;
    xr[7] = 0.0; xr[15] = 0.0;
    double Hq = 0.0, gq = 0.0, Dq[3] = {0.0, 0.0, 0.0}, cq = 0.0, dk[3][3];
#pragma unroll
    for (int u = 0; u < 3; u++) {
      dk[u][0] = 0.0; dk[u][1] = 0.0; dk[u][2] = 0.0;
      const int k = er + 4 * u;
      if (k < mmax) {      // (wave-uniform)
        if (k < m) {
          double r[2], g0[3], g1[3], Jl[2];
          cq += visual_lin_y(pcs[sf + 1 + k], ycx, ycy, ycz, yf, yinv_l, td, p_ob[u][0], p_ob[u][1], 0.0, 0.0, td, sq, delta, r, g0, g1, Jl);
          const vec3 y0 = cross3(mk3(g0[0], g0[1], g0[2]), yx), y1 = cross3(mk3(g1[0], g1[1], g1[2]), yx);
          const double w0 = is_const ? 0.0 : Jl[0], w1 = is_const ? 0.0 : Jl[1];
#pragma unroll
          for (int q = 0; q < 3; q++) dk[u][q] = __builtin_fma(g0[q], w0, g1[q] * w1);      // d = G^T w
          Hq += __builtin_fma(w0, w0, w1 * w1);
          gq += __builtin_fma(w0, r[0], w1 * r[1]);
#pragma unroll
          for (int q = 0; q < 3; q++) Dq[q] += dk[u][q];
#pragma unroll
          for (int q = 0; q < 3; q++) { xr[q] = g0[q]; xr[3 + q] = y0[q]; xr[8 + q] = g1[q]; xr[11 + q] = y1[q]; }
          xr[6] = r[0]; xr[14] = r[1];
          if (GFBE_LINSCHUR_WRITE_D) {
#pragma unroll
            for (int q = 0; q < 3; q++) d.lm_hP[((size_t)k * 6 + q) * TL + slot] = dk[u][q];
          }
        } else {
#pragma unroll
          for (int q = 0; q < 7; q++) { xr[q] = 0.0; xr[8 + q] = 0.0; }
        }
        dbl4_t a0 = {0, 0, 0, 0};
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int blk = 0; blk < LM_TILE / 4 / 8; blk++) {
          double va[8];
#pragma unroll
          for (int v = 0; v < 8; v++) va[v] = xs[(4 * (8 * blk + v) + lk) * LS_XLD + lr];
#pragma unroll
          for (int v = 0; v < 8; v++) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(va[v], va[v], a0, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        const double s0 = a0[0] + __shfl_down(a0[2], 8, 64), s1 = a0[1] + __shfl_down(a0[3], 8, 64);
        double *vo = d.vis_part + (((size_t)w * d.max_tiles + tile) * MAXOBS + k) * VPY_STRIDE;
        if (lr < 7) {
          { const int a = lk; if (a <= lr) vo[7 * a - a * (a - 1) / 2 + lr - a] = s0; }
          { const int a = lk + 4; if (a <= lr) vo[7 * a - a * (a - 1) / 2 + lr - a] = s1; }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    Psum[er][0][lane] = Hq; Psum[er][1][lane] = gq;
#pragma unroll
    for (int q = 0; q < 3; q++) Psum[er][2 + q][lane] = Dq[q];
    cq = wave_sum(cq);
    if (lane == 0) Pcost[er] = cq;
    __syncthreads();      // (A: every wave is done with its evaluation panel; the shares are visible)
    double Hll, gl, Dv[3];
    Hll = ((Psum[0][0][lane] + Psum[1][0][lane]) + Psum[2][0][lane]) + Psum[3][0][lane];
    gl = ((Psum[0][1][lane] + Psum[1][1][lane]) + Psum[2][1][lane]) + Psum[3][1][lane];
#pragma unroll
    for (int q = 0; q < 3; q++) Dv[q] = ((Psum[0][2 + q][lane] + Psum[1][2 + q][lane]) + Psum[2][2 + q][lane]) + Psum[3][2 + q][lane];
    // the landmark's weight in the Schur term (vis_body's expression), once per landmark, the same bits in the four waves
    double sl = 1.0, sw = 0.0;
    if (valid && m > 0 && !is_const) {
      sl = first ? (jac_scale ? 1.0 / (1.0 + sqrt(Hll)) : 1.0) : sl_old;
      const double hs2 = sl * sl * Hll;
      sw = sqrt(sl * sl / (hs2 + mu_w * clamp_diag(hs2)));
    }
    if (er == 2) {      // (a light role: two steps at most)
      if (valid) {
        if (first) d.lm_sl[slot] = sl;
        d.lm_sw[slot] = sw;
        d.lm_Hll[slot] = Hll;
        d.lm_gl[slot] = gl;
#pragma unroll
        for (int q = 0; q < 3; q++) { d.lm_hC[(size_t)q * TL + slot] = Dv[q]; d.lm_hC[(size_t)(3 + q) * TL + slot] = yx[q]; }
      }
      if (lane == 0) {
        const double cost = ((Pcost[0] + Pcost[1]) + Pcost[2]) + Pcost[3];
        if (SPEC) d.tile_cand[((size_t)w * d.max_tiles + tile) * 4] = cost;
        else d.tile_cost[(size_t)w * d.max_tiles + tile] = cost;
        if (SPEC && d.spec && tile == 0) d.ctl[w].sw_mu[lbw] = mu_w;
      }
    }
    // ---- stage the panel row of the lane's landmark: this wave's observing poses; role 3 the common columns
    {
      const int s = sf, kmax = NF - 1 - s;
      double *row = hs + lane * HS_LD + coff;
      const double xl[3] = {yx[0], yx[1], yx[2]};
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const int k = er + 4 * u;
        if (k < kmax) {
          const bool written = valid && k < m;
          double blkk[6];
          lm_row_block(s_rt[s + 1 + k], dk[u], xl, false, blkk);      // [ -d ; Rj^T (d x (x - t_j)) ]
#pragma unroll
          for (int q = 0; q < 6; q++) row[6 * (s + 1 + k) + q] = written ? sw * blkk[q] : 0.0;
        }
      }
      if (er == 3) {
        for (int q = 6 * s_first; q < 6 * s; q++) row[q] = 0.0;
        double blk0[6];
        lm_row_block(s_rt[s], Dv, xl, true, blk0);                      // [ D ; Ri^T ((x - t_i) x D) ]
#pragma unroll
        for (int q = 0; q < 6; q++) { row[6 * s + q] = sw * blk0[q]; row[T_EX + q] = 0.0; }
        row[T_TD] = 0.0;
        row[-coff] = sw * gl;                                           // column 0: the gradient
        for (int q = NV + coff; q < min(16 * nside, (int)HS_LD); q++) row[q - coff] = 0.0;
      }
    }
    const int m0 = (__builtin_amdgcn_readfirstlane(info) >> 8) & 0xff;      // (tracks sorted longest first: the tile's first landmark)
    const int jl = __builtin_amdgcn_readfirstlane((6 * (sf - s_first + m0 + 1)) >> 4);
    __syncthreads();      // (B: the panel is complete)
    LS_PREFETCH(tile + 1)      // (past the last tile: a repeat of it, never used)
#define LS_SLOT(Q, ACC)                                                                                 \
    if (pI[Q] >= 0 && pJ[Q] <= jl) {                                                                    \
      const double *pa = hs + 16 * pI[Q] + lr + lk * HS_LD, *pb = hs + 16 * pJ[Q] + lr + lk * HS_LD;   \
      _Pragma("unroll") for (int hf = 0; hf < 2; hf++) {                                                \
        double va[LM_TILE / 8], vb[LM_TILE / 8];                                                        \
        _Pragma("unroll") for (int kk = 0; kk < LM_TILE / 8; kk++) { va[kk] = pa[4 * (8 * hf + kk) * HS_LD]; vb[kk] = pb[4 * (8 * hf + kk) * HS_LD]; } \
        _Pragma("unroll") for (int kk = 0; kk < LM_TILE / 8; kk++) ACC = __builtin_amdgcn_mfma_f64_16x16x4f64(va[kk], vb[kk], ACC, 0, 0, 0); \
      }                                                                                                 \
    }
    LS_SLOT(0, acc0)
    LS_SLOT(1, acc1)
    LS_SLOT(2, acc2)
    LS_SLOT(3, acc3)
#undef LS_SLOT
  }
#undef LS_PREFETCH
  double *out = d.schur_part + ((size_t)w * SCHUR_GROUPS + grp) * SCHUR_STRIDE;
#define LS_OUT(Q, ACC)                                                            \
  if (pI[Q] >= 0) {                                                               \
    double *o = out + (size_t)(pJ[Q] * (pJ[Q] + 1) / 2 + pI[Q]) * 256;            \
    _Pragma("unroll") for (int r = 0; r < 4; r++) o[(lk + 4 * r) * 16 + lr] = ACC[r]; \
  }
  LS_OUT(0, acc0)
  LS_OUT(1, acc1)
  LS_OUT(2, acc2)
  LS_OUT(3, acc3)
#undef LS_OUT
}

// Small batches, marginalisation: the pair sums (0, j) and the Schur partial of start frame 0 both read what the linearisation of the
// marginalisation set left and nothing of each other: one launch (GFBE_FUSE_SMALL bit 3).
__global__ __launch_bounds__(VP_STRIDE) void k_pairsum_schur_marg(BatchDev d) {
  const int w = blockIdx.x, y = blockIdx.y;
  if (y < NF - 1) pairsum_body(d, 1, w, y);
  else if (threadIdx.x < 256) schur_body(d, 1, w, 0);
}

// =============================================================================================
// k_assemble: every partial into the dense normal equations (fixed summation order =>
// bit-reproducible). Writes H (unscaled, constant dims removed), g, E, eg.
// =============================================================================================
__device__ __forceinline__ int vis_loc(int a, int i, int j) {   // compact column of visual dim a in pair (i,j)
  if (a < 66) { const int f = a / 6; if (f == i) return a - 6 * f; if (f == j) return 6 + a - 6 * f; return -1; }
  if (a < 72) return 12 + (a - 66);
  return 18;
}
// entry (la, lb) of a factor's J^T J (symmetric, row stride C) in its lower triangle: k_dense_tp writes only that half
__device__ __forceinline__ int part_lower(int la, int lb, int C) { return max(la, lb) * C + min(la, lb); }
__device__ __forceinline__ int imu_loc(int a, int i) {          // column of dim a in the IMU factor (i, i+1)
  if (a < 66) { const int f = a / 6; if (f == i) return a - 6 * f; if (f == i + 1) return 15 + a - 6 * f; return -1; }
  if (a >= 73 && a < 172) { const int f = (a - 73) / 9; if (f == i) return 6 + (a - 73 - 9 * f); if (f == i + 1) return 21 + (a - 73 - 9 * f); }
  return -1;
}
__device__ __forceinline__ int wheel_loc(int a, int i) {
  if (a < 66) { const int f = a / 6; if (f == i) return a - 6 * f; if (f == i + 1) return 6 + a - 6 * f; return -1; }
  if (a >= T_EXW && a < T_EXW + 6) return 12 + (a - T_EXW);
  if (a >= T_SX && a <= T_TDW) return 18 + (a - T_SX);
  return -1;
}

__device__ __forceinline__ int plane_loc(int a, int i) {         // column of dim a in the PlaneFactor of pose i
  if (a < 66) { const int f = a / 6; return f == i ? a - 6 * f : -1; }
  if (a >= T_EXW && a < T_EXW + 6) return 6 + (a - T_EXW);
  if (a >= T_PLR && a < T_PLR + 3) return 12 + (a - T_PLR);
  if (a == T_PLZ) return 15;
  return -1;
}
// PlaneFactors' and the PoseAnchorFactor's share of H(a, b) (b == -1: of the gradient g(a)), fixed order
__device__ __forceinline__ double plane_anchor_term(const BatchDev &d, int w, int n_plane, int use_anchor, int a, int b) {
  double s = 0.0;
  if (n_plane > 0) {
    const int fa = a < 66 ? a / 6 : -1, fb = (b >= 0 && b < 66) ? b / 6 : -1;
    if (!(fa >= 0 && fb >= 0 && fa != fb)) {
      const int only = fa >= 0 ? fa : fb;       // a pose dim pins the factor; otherwise every factor touches the entry
      const int i0 = only >= 0 ? only : 0, i1 = only >= 0 ? only : n_plane - 1;
      for (int i = i0; i <= i1 && i < n_plane; i++) {
        const int la = plane_loc(a, i), lb = b >= 0 ? plane_loc(b, i) : 0;
        if (la >= 0 && lb >= 0) s += d.plane_part[((size_t)w * MAX_PLANE + i) * PLANE_PART + (b >= 0 ? la * 16 + lb : 256 + la)];
      }
    }
  }
  if (use_anchor && a < 6 && b < 6) s += d.anchor_part[(size_t)w * ANCHOR_PART + (b >= 0 ? a * 6 + b : 36 + a)];
  return s;
}

__device__ __forceinline__ int vp_off(int a, int b) {   // entry (a <= b) of the 20-column X^T X inside a fused visual partial
  if (a > b) { const int t = a; a = b; b = t; }
  if (b < 16) return a * 16 + b;
  if (a < 16) return 256 + a * 4 + (b - 16);
  return 320 + (a - 16) * 4 + (b - 16);
}
// Frame of a pose / speed-bias tangent dim (-1 for the global blocks).
__device__ __forceinline__ int dim_frame(int a) {
  if (a < 66) return a / 6;
  if (a >= 73 && a < 172) return (a - 73) / 9;
  return -1;
}

// Tables of the window descriptor the assembly consults per entry, staged in LDS once per workgroup.
struct AsmTab {
  int prior_map[ND];
  int imu_of_frame[NF], wheel_of_frame[NF];
  int prior_n, n_wheel, n_plane, use_anchor;
  unsigned char act[ND + 2];
};

// Window-independent part of the assembly of entry e = (a, b), b <= a, of the lower triangle: which (at most
// two) inertial and (at most two) wheel factors reach it and where inside their J^T J blocks. Built once per
// batch by k_asm_table; k_assemble only resolves frame -> factor and the prior column per window.
//   x = a | b << 8
//   y = IMU:   slot0 = (i0 + 1) | off0 << 4 (bits 0..13), slot1 the same in bits 16..29; i + 1 == 0: none
//   z = wheel: same packing (the wheel-global x wheel-global block is handled separately)
#define ASM_NTRI (ND * (ND + 1) / 2)
__device__ __forceinline__ void asm_table_body(int4 *tab, const int bx) {
  const int e = bx * 256 + threadIdx.x;
  if (e >= ASM_NTRI) return;
  int hi, lo_;
  tri_decode(e, hi, lo_);
  const int a = lo_, b = hi;          // gather convention: a <= b
  int4 r = make_int4(hi | (lo_ << 8), 0, 0, 0);
  const int fa = dim_frame(a), fb = dim_frame(b);
  if (fa >= 0 && fb >= 0 && abs(fa - fb) <= 1) {   // IMU factors (lo-1, lo) and (lo, lo+1)
    const int lo = min(fa, fb);
    const int i0 = (fa == fb) ? lo - 1 : -1, i1 = lo;
    if (i0 >= 0) { const int la = imu_loc(a, i0), lb = imu_loc(b, i0); if (la >= 0 && lb >= 0) r.y |= (i0 + 1) | (part_lower(la, lb, 30) << 4); }
    if (i1 <= NF - 2) { const int la = imu_loc(a, i1), lb = imu_loc(b, i1); if (la >= 0 && lb >= 0) r.y |= ((i1 + 1) | (part_lower(la, lb, 30) << 4)) << 16; }
  }
  {
    const bool ga = (a >= T_EXW), gb = (b >= T_EXW);            // wheel extrinsic / intrinsics / td_wheel
    const int wa = (a < 66) ? fa : -1, wb = (b < 66) ? fb : -1;
    int i0 = -1, i1 = -1;
    if (wa >= 0 && wb >= 0) { if (abs(wa - wb) <= 1) { const int lo = min(wa, wb); i0 = (wa == wb) ? lo - 1 : -1; i1 = lo; } }
    else if ((wa >= 0 && gb) || (wb >= 0 && ga)) { const int f = max(wa, wb); i0 = f - 1; i1 = f; }
    if (i0 >= 0 && i0 <= NF - 2) { const int la = wheel_loc(a, i0), lb = wheel_loc(b, i0); if (la >= 0 && lb >= 0) r.z |= (i0 + 1) | (part_lower(la, lb, 22) << 4); }
    if (i1 >= 0 && i1 <= NF - 2) { const int la = wheel_loc(a, i1), lb = wheel_loc(b, i1); if (la >= 0 && lb >= 0) r.z |= ((i1 + 1) | (part_lower(la, lb, 22) << 4)) << 16; }
  }
  tab[e] = r;
}
__global__ __launch_bounds__(256) void k_asm_table(int4 *tab) { asm_table_body(tab, blockIdx.x); }
// The compact table: the entries of the core dims (the first NC (NC + 1) / 2 of the full table) that some factor of a window can reach
// when its prior holds no speed-bias block but SpeedBias[0] — both dims among the 73 visual ones, an inertial or wheel slot, the wheel's
// global block, or both dims in the set a reference-structured prior (and the plane / anchor factors) can couple: the poses, SpeedBias[0],
// the camera extrinsic and td, the wheel extrinsic / intrinsics / td, the plane blocks. One workgroup, order kept (block scans).
__device__ __forceinline__ bool asm_prior_dim(int x) { return x < NV || (x >= T_SB(0) && x < T_SB(1)) || (x >= T_EXW && x < NC); }
__global__ __launch_bounds__(1024) void k_asm_compact(const int4 *full, int4 *compact, int *n_out) {
  __shared__ int wcnt[16], base;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) base = 0;
  __syncthreads();
  const int ntri = NC * (NC + 1) / 2;
  for (int e0 = 0; e0 < ntri; e0 += 1024) {
    const int e = e0 + t;
    int4 r = make_int4(0, 0, 0, 0);
    bool live = false;
    if (e < ntri) {
      r = full[e];
      const int hi = r.x & 255, lo = (r.x >> 8) & 255;
      live = hi < NV || r.y != 0 || r.z != 0 || (lo >= T_EXW && hi <= T_TDW) || (asm_prior_dim(hi) && asm_prior_dim(lo));
    }
    const unsigned long long m = __ballot(live);
    if (lane == 0) wcnt[wv] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int q = 0; q < wv; q++) off += wcnt[q];
    if (live) compact[off + __popcll(m & ((1ull << lane) - 1ull))] = r;
    __syncthreads();
    if (t == 0) { int s2 = 0; for (int q = 0; q < 16; q++) s2 += wcnt[q]; base += s2; }
    __syncthreads();
  }
  if (t == 0) *n_out = base;
}
hipError_t asm_tables_build(int **full, int **compact, int *n_compact, hipStream_t s) {
  int *f = nullptr, *cmp = nullptr, *dn = nullptr;
  hipError_t e = hipMalloc(&f, sizeof(int4) * ASM_NTRI);
  if (e == hipSuccess) e = hipMalloc(&cmp, sizeof(int4) * (NC * (NC + 1) / 2));
  if (e == hipSuccess) e = hipMalloc(&dn, sizeof(int));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_asm_table, dim3((ASM_NTRI + 255) / 256), dim3(256), 0, s, (int4 *)f);
    hipLaunchKernelGGL(k_asm_compact, dim3(1), dim3(1024), 0, s, (const int4 *)f, (int4 *)cmp, dn);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(n_compact, dn, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
  }
  if (dn) (void)hipFree(dn);
  if (e != hipSuccess) {     // (nothing is handed back on failure: the context frees only what it was given)
    if (f) (void)hipFree(f);
    if (cmp) (void)hipFree(cmp);
    return e;
  }
  *full = f; *compact = cmp;
  return e;
}

// entry `idx` of a double array whose base is wave-uniform: the BYTE offset formed in 32 bits, so that the load is [scalar base + 32-bit
// vector offset] instead of a 64-bit address per lane (idx < 2^29)
// a pointer every lane of the wave holds the same value of, moved to scalar registers (a base formed from a LOADED value — the set index
// WinCtl::lb behind lin_view — is a vector value to the compiler, whatever its lanes hold)
template <class T> __device__ __forceinline__ T *uniform_ptr(T *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T *)(((unsigned long long)hi << 32) | lo);
}
// (global address space spelled out: a pointer rebuilt from two scalar halves is a generic one to the compiler — FLAT loads)
typedef __attribute__((address_space(1))) char asm_glb_char;
typedef __attribute__((address_space(1))) double asm_glb_double;
// (the byte offset behind an empty assembly statement: the combiner otherwise turns zext(select(c, x, 0)) into a 64-bit select and the
//  [scalar base + 32-bit offset] form is lost again)
__device__ __forceinline__ double ld_u32(const double *base, unsigned idx) { unsigned bo = idx << 3; asm("" : "+v"(bo)); return *(const asm_glb_double *)((const asm_glb_char *)base + bo); }
__device__ __forceinline__ void st_u32(double *base, unsigned idx, double v) { unsigned bo = idx << 3; asm("" : "+v"(bo)); *(asm_glb_double *)((asm_glb_char *)base + bo) = v; }
__device__ __forceinline__ double gather_g_dense(const BatchDev &d, const AsmTab &tb, const double *Z, int w, int a) {
  const int fa = dim_frame(a);
  const double *p[2] = {Z, Z};
  if (fa >= 0) {
    if (fa >= 1) { const int q = tb.imu_of_frame[fa - 1], la = imu_loc(a, fa - 1); if (q >= 0 && la >= 0) p[0] = d.imu_part + ((size_t)w * MAX_IMU + q) * IMU_PART + 900 + la; }
    { const int q = tb.imu_of_frame[fa], la = imu_loc(a, fa); if (q >= 0 && la >= 0) p[1] = d.imu_part + ((size_t)w * MAX_IMU + q) * IMU_PART + 900 + la; }
  }
  double s = *p[0] + *p[1];
  if (tb.n_wheel > 0 && (a < 66 || a >= T_EXW)) {
    const int i0 = (a < 66) ? fa - 1 : 0, i1 = (a < 66) ? fa : NF - 2;
#pragma unroll
    for (int i = 0; i <= NF - 2; i++) {
      const int q = tb.wheel_of_frame[i], la = wheel_loc(a, i);
      const bool use = i >= i0 && i <= i1 && q >= 0 && la >= 0;
      s += *(use ? d.wheel_part + ((size_t)w * MAX_WHEEL + q) * WHEEL_PART + 484 + la : Z);
    }
  }
  if (tb.prior_n > 0 && tb.prior_map[a] >= 0) s += d.prior_g[(size_t)w * (ND + 2) + tb.prior_map[a]];
  if (tb.n_plane > 0 || tb.use_anchor) s += plane_anchor_term(d, w, tb.n_plane, tb.use_anchor, a, -1);
  return s;
}
// gather_g_dense for throughput batches (k_visasm, end of round 6): the same terms added in the same order, every table lookup and every
// load unconditional at a clamped index / offset 0, the VALUE selected (see asm_H_tp)
__device__ __forceinline__ double gather_g_dense_tp(const BatchDev &d, const AsmTab &tb, int w, int a) {
  const double *imu_w = uniform_ptr(d.imu_part + (size_t)w * MAX_IMU * IMU_PART), *wheel_w = uniform_ptr(d.wheel_part + (size_t)w * MAX_WHEEL * WHEEL_PART);
  const double *pg = uniform_ptr(d.prior_g + (size_t)w * (ND + 2));
  const int fa = dim_frame(a);
  const int q0 = tb.imu_of_frame[max(fa - 1, 0)], q1 = tb.imu_of_frame[max(fa, 0)], pm = tb.prior_map[a];
  int qw[NF - 1];
#pragma unroll
  for (int i = 0; i <= NF - 2; i++) qw[i] = tb.wheel_of_frame[i];
  const int la0 = imu_loc(a, fa - 1), la1 = imu_loc(a, fa);
  const bool u0 = (fa >= 1) & (q0 >= 0) & (la0 >= 0), u1 = (fa >= 0) & (q1 >= 0) & (la1 >= 0);
  const double l0 = ld_u32(imu_w, u0 ? (unsigned)(q0 * IMU_PART + 900 + la0) : 0u), l1 = ld_u32(imu_w, u1 ? (unsigned)(q1 * IMU_PART + 900 + la1) : 0u);
  const bool wsel = tb.n_wheel > 0 && (a < 66 || a >= T_EXW);
  const int i0 = (a < 66) ? fa - 1 : 0, i1 = (a < 66) ? fa : NF - 2;
  bool wu[NF - 1];
  double wl[NF - 1];
#pragma unroll
  for (int i = 0; i <= NF - 2; i++) {
    const int la = wheel_loc(a, i);
    wu[i] = wsel & (i >= i0) & (i <= i1) & (qw[i] >= 0) & (la >= 0);
    wl[i] = ld_u32(wheel_w, wu[i] ? (unsigned)(qw[i] * WHEEL_PART + 484 + la) : 0u);
  }
  const bool up = tb.prior_n > 0 && pm >= 0;
  const double pl = ld_u32(pg, up ? (unsigned)pm : 0u);
  double s = (u0 ? l0 : 0.0) + (u1 ? l1 : 0.0);
  double sw = s;
#pragma unroll
  for (int i = 0; i <= NF - 2; i++) sw += wu[i] ? wl[i] : 0.0;
  s = wsel ? sw : s;               // (gather_g_dense skips the loop for the dims no wheel factor reaches: their sum keeps its bits, a -0.0 included)
  s = up ? s + pl : s;
  if (tb.n_plane > 0 || tb.use_anchor) s += plane_anchor_term(d, w, tb.n_plane, tb.use_anchor, a, -1);
  return s;
}
// E(a,b), a <= b < NVP (b = 73: the gradient column): the Schur partials of the start-frame groups that reach dim a (a group
// reaches the dims from its first start frame's pose on), loads unconditional in flight
__device__ __forceinline__ double gather_E11(const BatchDev &d, const double *Z, int w, int a, int b) {
  // compact panels (schur_body): in the partial of a group whose first start frame is s, dim x sits in column 1 + x - 6 s and the
  // gradient in column 0; entry (r, c), r <= c, of the panel product at tile pair (r >> 4, c >> 4) = slot J (J + 1) / 2 + I
  const double *sp = uniform_ptr(d.schur_part + (size_t)w * d.schur_groups * SCHUR_STRIDE);
  auto off_of = [&](int s) {
    if (!GFBE_SCHUR_COMPACT) return schur_pair(a >> 4, b >> 4) * 256 + (a & 15) * 16 + (b & 15);      // (diagnostics: the absolute layout of rounds 1-3)
    const int ca = 1 + a - 6 * s, r = b == NV ? 0 : ca, cc = b == NV ? ca : 1 + b - 6 * s;
    const int I = r >> 4, J = cc >> 4;
    return (J * (J + 1) / 2 + I) * 256 + (r & 15) * 16 + (cc & 15);
  };
  const int ng = d.schur_groups;
  if (ng == SCHUR_GROUPS) {
    double v[SCHUR_GROUPS];
#pragma unroll
    for (int f = 0; f < SCHUR_GROUPS; f++) {
      const int s = schur_group_first(f, SCHUR_GROUPS);
      const bool use = 6 * s <= a;
      const double ld = ld_u32(sp, use ? (unsigned)(f * SCHUR_STRIDE + off_of(s)) : 0u);      // (scalar base + 32-bit offset; the value is selected, not the pointer)
      v[f] = use ? ld : 0.0;
    }
    double sum = 0.0;
#pragma unroll
    for (int f = 0; f < SCHUR_GROUPS; f++) sum += v[f];
    return sum;
  }
  if (ng == NF) {
    double v[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) { const bool use = 6 * f <= a; const double ld = ld_u32(sp, use ? (unsigned)(f * SCHUR_STRIDE + off_of(f)) : 0u); v[f] = use ? ld : 0.0; }
    double sum = 0.0;
#pragma unroll
    for (int f = 0; f < NF; f++) sum += v[f];
    return sum;
  }
  double v[2 * NF];           // small batches: two partials per start frame
#pragma unroll
  for (int f = 0; f < 2 * NF; f++) { const bool use = 6 * (f >> 1) <= a; const double ld = ld_u32(sp, use ? (unsigned)(f * SCHUR_STRIDE + off_of(f >> 1)) : 0u); v[f] = use ? ld : 0.0; }
  double sum = 0.0;
#pragma unroll
  for (int f = 0; f < 2 * NF; f++) sum += v[f];
  return sum;
}

// k_visblock (one workgroup per window): the visual block of the normal equations.
//   For every start frame i, thread e sums entry e of the fused X^T X partials (k_vis) of the pose pairs
//   (i, i+1..10) over the landmark tiles of that start frame (registers, all loads in flight at once; two thread
//   groups take the tiles alternately) and adds them into a 73 x 74 LDS accumulator (column 73 = gradient).
//   Inside one start frame no two threads of a group touch the same entry; the groups add one after the other,
//   so every entry is summed in one fixed order. The block goes to HBM as vis_H[w][73][74].
// k_assemble (16 workgroups per window): owner-computes over the lower triangle of the 182 x 182 system: visual
//   entry + the <= 5 inertial / wheel / prior contributions through the window-independent table of
//   k_asm_table (independent loads, four entries in flight per thread); E and eg = sums of the start-frame
//   Schur partials.
#define VB_THREADS 1024
#ifndef VB_GROUP
#define VB_GROUP 512
#endif
#define V_LD 74
// SPLIT (small batches): one 512-thread workgroup per (start frame, thread group, window) writes its own block
// vis_Hs[w][2 i + group]; k_assemble adds the 20 blocks in (start frame, group) order — exactly the additions, in exactly the
// order, of the sequential loop of the one-workgroup form (bit-identical), twenty workgroups beside each other instead of ten
// start frames one after the other on the latency path of a single window. sgrp: the thread group of a SPLIT workgroup.
// KEEP: the block stays in the caller's LDS array V (k_visasm assembles from there) instead of going to vis_H.
// DUAL: ONE thread group of 512 takes the tiles of both groups, one group after the other — the same sums added in the same order
// by half the threads (k_visasm: two 512-thread workgroups per CU overlap each other's gather latencies).
// ---- The visual block of a window whose batch has constant extrinsic and td everywhere (k_vis<0, false>, visual_lin_y): a (tile,
// step) partial is the upper triangle of M = sum [Y r]^T [Y r] (7 x 7, VPY doubles), Y = [G | G [P_w - c]x] with the landmark's world
// position taken from the window's origin c = P_0. With T_f = [ I  [P_f - c]x R_f ; 0  -R_f ] (6 x 6, one per FRAME):
//   J_i = Y T_i,  J_j = -Y T_j   =>   H(i, j) = -T_i^T M(i, j) T_j,   H(f, f) = T_f^T (sum of M over every pair with frame f) T_f,
//   g(f) = T_f^T (sum of M's r-column over the pairs (f, j) - over the pairs (i, f)).
// (1) every pair's M summed over the tiles of its start frame, in tile order, by one owner thread per entry — all loads in flight at
//     once; (2) the per-frame sums S_f from LDS; (3) owner-computes over the entries of the pose block: no read-modify-write, no loop
//     over start frames, two block barriers.
// ROW = false (k_visasm: one workgroup per window): all frames, into the LDS block `out` (row stride ld).
// ROW = true (small batches: one workgroup per frame, side by side on a single window's latency path): the six rows of frame
//     `frow` — the pairs that frame is part of — straight into vis_Hs block 0 in global memory. The same sums in the same order.
template <int NT, bool ROW>
__device__ __forceinline__ void visblock_y(const BatchDev &d, const WinDesc &ds, const WinCtl &c, const int w, const int frow, const int *s_tile_begin,
                                           double *out, const int ld) {
  __shared__ double sM[NF * (NF - 1) / 2][VPY];          // pair (i, j) at i (21 - i) / 2 + j - i - 1
  __shared__ double sS[NF][VPY];                         // per frame: the sum of M over its pairs (r-column: signed by the role)
  __shared__ double sTf[NF][18];                         // per frame: [P_f - c]x R_f (9) | -R_f (9)
  const int t = threadIdx.x;
  const double *Z = d.zero;
#if GFBE_DIAG
  double *vstamp = d.timing + (size_t)d.B * 32 + 24;     // phase stamps of window 0 in k_visasm (gfbe_debug_timing(batch, B): slots 24..31)
#define VSTAMP(i) do { if (!ROW && w == 0 && t == 0) vstamp[i] = (double)wall_clock64(); } while (0)
#else
#define VSTAMP(i) do { } while (0)
#endif
  VSTAMP(0);
  {
    const double *pcw = d.pc + ((size_t)w * 3 + c.cur) * NPAIR * PC_DOUBLES;
    const double *X = d.x + ((size_t)w * 2 + c.cur) * NA;
    for (int q = t; q < NF * 18; q += NT) {
      const int f = q / 18, en = q - 18 * f;
      const double *R = pcw + (size_t)(f * NF + f) * PC_DOUBLES + 12;     // FrameConst::R
      double val;
      if (en >= 9) val = -R[en - 9];
      else {        // ([tf]x R)(p, cc) = tf x (column cc of R), component p;  tf = P_f - P_0
        const int pp = en / 3, cc = en - 3 * pp, p1 = (pp + 1) % 3, p2 = (pp + 2) % 3;
        const double d1 = X[A_POSE(f) + p1] - X[A_POSE(0) + p1], d2 = X[A_POSE(f) + p2] - X[A_POSE(0) + p2];
        val = __builtin_fma(d1, R[3 * p2 + cc], -(d2 * R[3 * p1 + cc]));
      }
      sTf[f][en] = val;
    }
    const double *vpy = d.vis_part + (size_t)w * d.max_tiles * MAXOBS * VPY_STRIDE;
    const int npair = ROW ? NF - 1 : NF * (NF - 1) / 2;
    // (one workgroup per window: the tiles' longest tracks — which steps a tile ran — staged once instead of a dependent load in
    //  front of every round of partials; a window with more tiles than the table holds keeps the direct loads)
    __shared__ unsigned char s_tile_m[ROW ? 1 : 256];
    const bool staged_m = !ROW && ds.n_tiles <= 256;
    if (staged_m) {
      for (int tt = t; tt < ds.n_tiles; tt += NT) s_tile_m[tt] = TILE_OWNED(d, tt) ? (unsigned char)((d.lm_info[ds.lm_off + tt * LM_TILE] >> 8) & 0xff) : 0;
      __syncthreads();
    }
    for (int q = t; q < npair * VPY; q += NT) {
      const int pq = q / VPY, en = q - pq * VPY;
      int i, k;
      if (ROW) { const int o = pq < frow ? pq : pq + 1; i = min(frow, o); k = max(frow, o) - i - 1; }
      else { i = 0; int rem = pq; while (rem >= NF - 1 - i) { rem -= NF - 1 - i; i++; } k = rem; }
      const int t0 = s_tile_begin[i], t1 = s_tile_begin[i + 1];
      double sum = 0.0;
      if (!ROW && staged_m && ASM_TP_ON(d, 16)) {
        // (end of round 6, like asm_H_tp: the tiles' step counts read unconditionally at a clamped index — one wait for the four —, the
        //  partials loaded from the window's scalar base at a 32-bit offset, offset 0 for a tile that did not run the step, and the VALUE
        //  selected: the same sums in the same order)
        const double *vb = uniform_ptr(vpy);
        for (int tt = t0; tt < t1; tt += 4) {
          int mt[4];
#pragma unroll
          for (int u = 0; u < 4; u++) mt[u] = s_tile_m[min(tt + u, 255)];
          bool on[4];
          double v[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            on[u] = (tt + u < t1) & (k < mt[u]);
            v[u] = ld_u32(vb, on[u] ? (unsigned)(((tt + u) * MAXOBS + k) * VPY_STRIDE + en) : 0u);
          }
#pragma unroll
          for (int u = 0; u < 4; u++) sum += on[u] ? v[u] : 0.0;
        }
      } else
      for (int tt = t0; tt < t1; tt += 4) {          // (tiles sorted longest first: a tile ran step k iff its first track reaches it)
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const bool on = tt + u < t1 && (staged_m ? k < (int)s_tile_m[min(tt + u, 255)]
                                                        : TILE_OWNED(d, tt + u) && k < ((d.lm_info[ds.lm_off + (tt + u) * LM_TILE] >> 8) & 0xff));
          v[u] = *(on ? vpy + ((size_t)(tt + u) * MAXOBS + k) * VPY_STRIDE + en : Z);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) sum += v[u];
      }
      sM[i * (2 * NF - 1 - i) / 2 + k][en] = sum;
    }
  }
  __syncthreads();
  VSTAMP(1);
  // S_f: entries of the 6 x 6 block summed over both roles of the frame; the r-column (packed entries (p, 6)) with the sign of the role
  for (int q = t; q < (ROW ? 1 : NF) * VPY; q += NT) {
    const int f = ROW ? frow : q / VPY, en = ROW ? q : q - f * VPY;
    const bool rcol = en == 6 || en == 12 || en == 17 || en == 21 || en == 24 || en == 26;     // (p, 6), p = 0..5
    double sum = 0.0;
    for (int j = f + 1; j < NF; j++) sum += sM[f * (2 * NF - 1 - f) / 2 + j - f - 1][en];
    for (int i = 0; i < f; i++) { const double v = sM[i * (2 * NF - 1 - i) / 2 + f - i - 1][en]; sum += rcol ? -v : v; }
    sS[f][en] = sum;
  }
  __syncthreads();
  VSTAMP(2);
  // T_f(p, a): a < 3: delta(p, a); a >= 3: sTf[f][3 p + a - 3]
  auto Mat = [](const double *M, int pp, int qq) { const int lo = min(pp, qq), hi = max(pp, qq); return M[7 * lo - lo * (lo - 1) / 2 + hi - lo]; };
  constexpr int NC66 = NF * 6;
  if (!ROW) {
    // One workgroup for the whole block (k_visasm): a thread per COLUMN of a 6 x 6 block instead of a thread per entry. The entry-wise
    // loop below evaluated sum_pp T_a(pp) (sum_qq M(pp, qq) T_b(qq)) from scratch for every entry, four kinds of entries side by side in
    // a wave and the packed index of M computed per access: ~3 800 vector instructions per wave, which is what bounded the kernel
    // (PMC: 62 M per launch of 2048 windows, no matrix-core work, the memory pipes idle half of the time). Here M's 21 entries
    // sit in registers, P = M T_b(:, lb) is formed once per column (36 multiply-adds) and its six rows finished from it (18): the
    // same products added in the same order, entry for entry — bit-identical to the loop below, which the small batches keep.
    for (int pass = 0; pass < 2; pass++) {          // pass 0: the columns lb = 3..5 (through T_b), pass 1: lb = 0..2 (T_b = identity there)
      for (int u = t; u < NF * NF * 3; u += NT) {
        const int blk = u / 3, jc = u - 3 * blk, fa = blk / NF, fb = blk - NF * fa;
        const double *M = fa == fb ? sS[fa] : sM[min(fa, fb) * (2 * NF - 1 - min(fa, fb)) / 2 + max(fa, fb) - min(fa, fb) - 1];
        double P[6];
        if (pass == 0) {
          double m[6][6], tb[6];
#pragma unroll
          for (int pp = 0; pp < 6; pp++) {
            tb[pp] = sTf[fb][3 * pp + jc];
#pragma unroll
            for (int qq = pp; qq < 6; qq++) { m[pp][qq] = M[7 * pp - pp * (pp - 1) / 2 + qq - pp]; m[qq][pp] = m[pp][qq]; }
          }
#pragma unroll
          for (int pp = 0; pp < 6; pp++) {
            double row = 0.0;
#pragma unroll
            for (int qq = 0; qq < 6; qq++) row = __builtin_fma(m[pp][qq], tb[qq], row);
            P[pp] = row;
          }
        } else {
#pragma unroll
          for (int pp = 0; pp < 6; pp++) P[pp] = Mat(M, pp, jc);
        }
        const int b = 6 * fb + (pass == 0 ? 3 : 0) + jc;
        const bool neg = fa != fb;
#pragma unroll
        for (int la = 0; la < 3; la++) out[(6 * fa + la) * ld + b] = neg ? -P[la] : P[la];
#pragma unroll
        for (int la = 3; la < 6; la++) {
          double z = 0.0;
#pragma unroll
          for (int pp = 0; pp < 6; pp++) z = __builtin_fma(sTf[fa][3 * pp + la - 3], P[pp], z);
          out[(6 * fa + la) * ld + b] = neg ? -z : z;
        }
      }
    }
    for (int a = t; a < NC66; a += NT) {          // the gradient column
      const int fa = a / 6, la = a - 6 * fa;
      const double *M = sS[fa];
      double z = 0.0;
      if (la < 3) z = Mat(M, la, 6);
      else for (int pp = 0; pp < 6; pp++) z = __builtin_fma(sTf[fa][3 * pp + la - 3], Mat(M, pp, 6), z);
      out[a * ld + NV] = z;
    }
    VSTAMP(3);
    return;
  }
  for (int q = t; q < (ROW ? 6 : NC66) * (NC66 + 1); q += NT) {
    const int ar = q / (NC66 + 1), b = q - ar * (NC66 + 1), a = ROW ? 6 * frow + ar : ar;     // b == 66: the gradient column
    const int fa = a / 6, la = a - 6 * fa;
    double z = 0.0;
    if (b == NC66) {
      const double *M = sS[fa];
      if (la < 3) z = Mat(M, la, 6);
      else for (int pp = 0; pp < 6; pp++) z = __builtin_fma(sTf[fa][3 * pp + la - 3], Mat(M, pp, 6), z);
      out[a * ld + NV] = z;
      continue;
    }
    const int fb = b / 6, lb = b - 6 * fb;
    const double *M = fa == fb ? sS[fa] : sM[min(fa, fb) * (2 * NF - 1 - min(fa, fb)) / 2 + max(fa, fb) - min(fa, fb) - 1];
    // entry (a, b) = +-T_fa(:, la)^T M T_fb(:, lb); M is symmetric, so the order of the two frames only decides the sign
    if (la < 3 && lb < 3) z = Mat(M, la, lb);
    else if (la < 3) { for (int qq = 0; qq < 6; qq++) z = __builtin_fma(Mat(M, la, qq), sTf[fb][3 * qq + lb - 3], z); }
    else if (lb < 3) { for (int pp = 0; pp < 6; pp++) z = __builtin_fma(sTf[fa][3 * pp + la - 3], Mat(M, pp, lb), z); }
    else {
      for (int pp = 0; pp < 6; pp++) {
        double row = 0.0;
        for (int qq = 0; qq < 6; qq++) row = __builtin_fma(Mat(M, pp, qq), sTf[fb][3 * qq + lb - 3], row);
        z = __builtin_fma(sTf[fa][3 * pp + la - 3], row, z);
      }
    }
    out[a * ld + b] = fa == fb ? z : -z;
  }
}
#undef VSTAMP

template <bool SPLIT, bool KEEP, bool DUAL>
__device__ __forceinline__ void visblock_body(const BatchDev &d, const int w, const int i_first, const int i_last, const int sgrp, double *V) {
  constexpr int NT = (SPLIT || DUAL) ? VB_GROUP : VB_THREADS;
  const WinDesc &ds = d.desc[w];
  const WinCtl &c = d.ctl[w];
  if (c.done || c.reuse) return;
  // (SPLIT: a start frame without landmarks leaves its block as the upload zeroed it — the structure never changes)
  const bool lead = (i_first == 0) && (!SPLIT || sgrp == 0);   // (SPLIT: this workgroup also carries the cost)
  __shared__ int s_tile_begin[NF + 1];
  const int t = threadIdx.x;
  if (SPLIT && !d.vis_full) {
    // 7 x 7 partials (visblock_y): workgroup v = 2 i_first + sgrp < NF builds the six rows of FRAME v of the visual block, straight into
    // vis_Hs block 0 (the other blocks, and the rows / columns of the inactive extrinsic and td dims, stay as the upload zeroed them)
    const int v = 2 * i_first + sgrp;
    if (v >= NF) return;
    if (t <= NF) s_tile_begin[t] = ds.sf_tile_begin[t];
    __syncthreads();
    visblock_y<NT, true>(d, ds, c, w, v, s_tile_begin, d.vis_Hs + (size_t)w * d.vs_blocks * NV * V_LD, V_LD);
    if (lead && t < 64) {      // robustified visual cost of this linearisation point (as below)
      double cs = 0.0;
      for (int q = t; q < ds.n_tiles; q += 64) cs += d.tile_cost[(size_t)w * d.max_tiles + q];
      cs = wave_sum(cs);
      if (ds.lio_n > 0 && d.rank == 0) for (int q = 0; q < LIOW_WGS; q++) cs += d.lio_part[((size_t)w * LIOW_WGS + q) * LIOW_PART + 27];
      for (int r = 0; r < d.world; r++)
        if (t < XCHG) d.xa[((size_t)w * d.world + r) * XCHG + t] = (r == d.rank && t == 0) ? cs : 0.0;
    }
    return;
  }
  if (SPLIT && !lead && ds.sf_tile_begin[i_first] + sgrp >= ds.sf_tile_begin[i_first + 1]) return;
  const double *Z = d.zero;
  double *stamp = d.timing + (size_t)w * 32;
#define ASTAMP(i) do { if (t == 0) stamp[i] = (double)wall_clock64(); } while (0)
  ASTAMP(6);
  for (int e = t; e < NV * V_LD; e += NT) V[e] = 0.0;
  if (t <= NF) s_tile_begin[t] = ds.sf_tile_begin[t];
  __syncthreads();
  const int grp = SPLIT ? sgrp : (DUAL ? 0 : t / VB_GROUP), e = (SPLIT || DUAL) ? t : t - grp * VB_GROUP;
  // reduced panel (extrinsic and td constant in every window of the batch): k_vis fills only the pose block, the gradient
  // column and r^T r of a partial — 157 of its 336 entries; the others belong to inactive dims and are not fetched
  const bool need = d.vis_full || (e < 256 ? ((e >> 4) < 12 && (e & 15) < 12) : (e < 320 ? (((e - 256) & 3) == 3 && ((e - 256) >> 2) < 12) : e == 335));
  const bool live = e < VP_STRIDE && need;
  // compact column pair (la, lb) of this thread's entry of a fused partial (layout of k_vis: T0 16x16, T1 16x4, T2 4x4)
  int la = 0, lb = 0; bool mirror = false;
  if (e < 256) { la = e >> 4; lb = e & 15; }
  else if (e < 320) { la = (e - 256) >> 2; lb = 16 + ((e - 256) & 3); mirror = true; }
  else if (live) { la = 16 + ((e - 320) >> 2); lb = 16 + ((e - 320) & 3); }
  // target of step k of start frame i: row da, column db with
  //   d = 6 i + l (l < 6: pose i) | 6 (i + 1 + k) + l - 6 (l < 12: pose j) | 54 + l (extrinsic, td, gradient)
  const bool aj = la >= 6 && la < 12, bj = lb >= 6 && lb < 12;
  const int a_i = la < 6 ? 6 : (aj ? 6 : 0), b_i = lb < 6 ? 6 : (bj ? 6 : 0);        // d(d)/d(i)
  const int a_0 = la < 6 ? la : (aj ? la : 54 + la), b_0 = lb < 6 ? lb : (bj ? lb : 54 + lb);   // at i = 0, k = 0 (j = 1)
  const int step0 = (aj ? 6 * V_LD : 0) + (bj ? 6 : 0), step1 = (bj ? 6 * V_LD : 0) + (aj ? 6 : 0);   // per k
  const bool row_ok = live && la != 19, mir_ok = live && mirror && lb != 19;
  const bool uses_j = aj || bj;
  const double *vp = d.vis_part + (size_t)w * d.max_tiles * MAXOBS * VP_STRIDE + e;
  if (!d.vis_full) {
    visblock_y<NT, false>(d, ds, c, w, 0, s_tile_begin, V, V_LD);
    __syncthreads();
  } else
  for (int i = i_first; i <= i_last; i++) {
    const int t0 = s_tile_begin[i], t1 = s_tile_begin[i + 1];
    if (t0 == t1) continue;
    const int nk = NF - 1 - i;
    const int da0 = a_0 + a_i * i, db0 = b_0 + b_i * i;
    const int i0 = da0 * V_LD + db0, i1 = db0 * V_LD + da0;
    // thread group g sums its tiles (g, g + 2, g + 4, ...: two in flight) and adds them into V; group 1 after group 0
    for (int gsel = 0; gsel < 2; gsel++) {
      if (gsel >= t1 - t0) break;
      if (DUAL || grp == gsel) {
        double s[MAXOBS];
#pragma unroll
        for (int k = 0; k < MAXOBS; k++) s[k] = 0.0;
        if (live) {
          for (int tt = t0 + gsel; tt < t1; tt += 4) {
            const double *q0 = vp + (size_t)tt * MAXOBS * VP_STRIDE, *q1 = q0 + (size_t)2 * MAXOBS * VP_STRIDE;
            const bool two = tt + 2 < t1;
            // k_vis runs (and writes) the steps below the tile's longest track — its first landmark: a start-frame group is sorted
            // longest first — and nothing else of vis_part is ever read: the array is not cleared at upload
            // (landmark sharding: the tiles of the other ranks were not evaluated here)
            const int m0 = TILE_OWNED(d, tt) ? (d.lm_info[ds.lm_off + tt * LM_TILE] >> 8) & 0xff : 0;
            const int m1 = (two && TILE_OWNED(d, tt + 2)) ? (d.lm_info[ds.lm_off + (tt + 2) * LM_TILE] >> 8) & 0xff : 0;
            double v0[MAXOBS], v1[MAXOBS];
#pragma unroll
            for (int k = 0; k < MAXOBS; k++) {
              v0[k] = *(k < m0 ? q0 + k * VP_STRIDE : Z);
              v1[k] = *(k < m1 ? q1 + k * VP_STRIDE : Z);
            }
#pragma unroll
            for (int k = 0; k < MAXOBS; k++) { s[k] += v0[k]; s[k] += v1[k]; }
          }
        }
        if (!uses_j) {
          double tot = 0.0;
#pragma unroll
          for (int k = 0; k < MAXOBS; k++) tot += s[k];
          if (row_ok) V[i0] += tot;
          if (mir_ok) V[i1] += tot;
        } else {
          // the targets of this thread are distinct entries: read them all, then write them all
          double o0[MAXOBS], o1[MAXOBS];
#pragma unroll
          for (int k = 0; k < MAXOBS; k++) {
            o0[k] = (row_ok && k < nk) ? V[i0 + k * step0] : 0.0;
            o1[k] = (mir_ok && k < nk) ? V[i1 + k * step1] : 0.0;
          }
#pragma unroll
          for (int k = 0; k < MAXOBS; k++) {
            if (row_ok && k < nk) V[i0 + k * step0] = o0[k] + s[k];
            if (mir_ok && k < nk) V[i1 + k * step1] = o1[k] + s[k];
          }
        }
      }
      if (!DUAL || gsel == 1 || t1 - t0 == 1) __syncthreads();    // (DUAL: both groups are this thread's own additions; one barrier per start frame)
    }
  }
  if (!SPLIT && lead && ds.lio_n > 0 && d.rank == 0 && t < 27) {   // LiDAR factors of pose lio_frame (k_lio_window): 6 x 6 block, gradient (SPLIT: k_assemble adds them)
    double v = 0.0;
    for (int q = 0; q < LIOW_WGS; q++) v += d.lio_part[((size_t)w * LIOW_WGS + q) * LIOW_PART + t];
    const int o = 6 * ds.lio_frame;
    if (t < 21) {
      int a = 0, e = t;
      while (e >= 6 - a) { e -= 6 - a; a++; }
      const int b = a + e;
      V[(o + a) * V_LD + o + b] += v;
      if (a != b) V[(o + b) * V_LD + o + a] += v;
    } else {
      V[(o + t - 21) * V_LD + NV] += v;
    }
  }
  __syncthreads();
  if (!KEEP) {
    double *out = SPLIT ? d.vis_Hs + ((size_t)w * d.vs_blocks + 2 * i_first + sgrp) * NV * V_LD : d.vis_H + (size_t)w * NV * V_LD;
    for (int q = t; q < NV * V_LD; q += NT) out[q] = V[q];
  }
  // robustified visual cost of this linearisation point (this rank's tiles; lanes stride the tiles, fixed tree order)
  if (lead && t < 64) {
    double cs = 0.0;
    for (int q = t; q < ds.n_tiles; q += 64) cs += d.tile_cost[(size_t)w * d.max_tiles + q];
    cs = wave_sum(cs);
    if (ds.lio_n > 0 && d.rank == 0) for (int q = 0; q < LIOW_WGS; q++) cs += d.lio_part[((size_t)w * LIOW_WGS + q) * LIOW_PART + 27];
    for (int r = 0; r < d.world; r++)
      if (t < XCHG) d.xa[((size_t)w * d.world + r) * XCHG + t] = (r == d.rank && t == 0) ? cs : 0.0;
  }
  ASTAMP(7);
#undef ASTAMP
}
__global__ __launch_bounds__(VB_GROUP) void k_visblock_small(BatchDev d0) {
  const BatchDev d = lin_view(d0, d0.ctl[blockIdx.y].lb);
  __shared__ double V[NV * V_LD];
  visblock_body<true, false, false>(d, blockIdx.y, blockIdx.x >> 1, blockIdx.x >> 1, blockIdx.x & 1, V);
}
// Small batches, one launch for both: the Schur partials (k_schur's workgroups: the first four waves of a 512-thread workgroup, the
// others leave at once — s_barrier waits for the surviving waves only) and the start-frame blocks of the visual Hessian read different
// outputs of the linearisation and nothing of each other: side by side instead of one after the other on a single window's
// latency path (k_visblock_small was 6.9 us of a 154 us iteration). Same code, same bits as the two launches.
__global__ __launch_bounds__(VB_GROUP) void k_schur_visblock_small(BatchDev d0) {
  const int w = blockIdx.x, y = blockIdx.y;
  const BatchDev d = lin_view(d0, d0.ctl[w].lb);
  if (y < d.schur_groups) {
    if (threadIdx.x < 256) schur_body(d, 0, w, y);
  } else {
    __shared__ double V[NV * V_LD];
    const int v = y - d.schur_groups;
    visblock_body<true, false, false>(d, w, v >> 1, v >> 1, v & 1, V);
  }
}

#ifndef ASM_THREADS
#define ASM_THREADS 256
#endif
// The assembly of window w by the threads gt, gt + gn, ... of its workgroup(s). vis_w: the visual block [73][74] (vis_H in global
// memory, or the LDS array k_visasm built it in — a generic pointer either way); tb: an LDS table of the calling kernel.
struct AsmCommon {
  bool dense_here, vsplit, lio_on;
  int lio_o, ntri;
  double *H, *g, *E, *eg;
  const int4 *tab;
  const double *imu_w, *wheel_w, *prior_w, *vis_s, *Z;
};
__device__ __forceinline__ void asm_stage_tables(const BatchDev &d, const int w, AsmTab &tb) {     // (the caller's block barrier follows)
  const WinDesc &ds = d.desc[w];
  const int t = threadIdx.x;
  for (int a = t; a < ND; a += blockDim.x) { tb.prior_map[a] = ds.prior_map[a]; tb.act[a] = ds.act[a]; }
  if (t < NF) { tb.imu_of_frame[t] = ds.imu_of_frame[t]; tb.wheel_of_frame[t] = ds.wheel_of_frame[t]; }
  if (t == 0) { tb.prior_n = ds.prior_n; tb.n_wheel = ds.n_wheel; tb.n_plane = ds.n_plane; tb.use_anchor = ds.use_anchor; }
}
// The same staging in two halves (k_visasm, end of round 6): the descriptor entries are REQUESTED before the visual block is gathered and
// written to the LDS tables behind it — their round trip (3.6 us under load, tools/diag_scripts/schur_tile_time.py) rides with the gather's.
struct AsmStagePre { int pm, imuf, wheelf, prior_n, n_wheel, n_plane, use_anchor; unsigned char act; };
__device__ __forceinline__ AsmStagePre asm_stage_load(const BatchDev &d, const int w) {
  const WinDesc &ds = d.desc[w];
  const int t = threadIdx.x, ta = min(t, ND - 1), tf = min(t, NF - 1);
  AsmStagePre p;
  p.pm = ds.prior_map[ta]; p.act = ds.act[ta]; p.imuf = ds.imu_of_frame[tf]; p.wheelf = ds.wheel_of_frame[tf];
  p.prior_n = ds.prior_n; p.n_wheel = ds.n_wheel; p.n_plane = ds.n_plane; p.use_anchor = ds.use_anchor;
  return p;
}
__device__ __forceinline__ void asm_stage_store(AsmTab &tb, const AsmStagePre &p) {     // (a workgroup of >= ND threads; the caller's block barrier follows)
  const int t = threadIdx.x;
  if (t < ND) { tb.prior_map[t] = p.pm; tb.act[t] = p.act; }
  if (t < NF) { tb.imu_of_frame[t] = p.imuf; tb.wheel_of_frame[t] = p.wheelf; }
  if (t == 0) { tb.prior_n = p.prior_n; tb.n_wheel = p.n_wheel; tb.n_plane = p.n_plane; tb.use_anchor = p.use_anchor; }
}
__device__ __forceinline__ AsmCommon asm_common(const BatchDev &d, const int w) {
  const WinDesc &ds = d.desc[w];
  AsmCommon c;
  c.Z = d.zero;
  c.dense_here = (d.rank == 0);   // landmark sharding: the inertial / wheel / prior factors are added once (rank 0)
  c.H = d.H + (size_t)w * ND * ND; c.g = d.g + (size_t)w * ND;
  c.E = d.E + (size_t)w * NV * NV; c.eg = d.eg + (size_t)w * NV;
  c.tab = (const int4 *)d.asm_tab;
  c.imu_w = d.imu_part + (size_t)w * MAX_IMU * IMU_PART; c.wheel_w = d.wheel_part + (size_t)w * MAX_WHEEL * WHEEL_PART;
  c.prior_w = d.prior_H + (size_t)w * ND * ND;
  c.vsplit = d.vis_Hs != nullptr;
  c.lio_on = c.vsplit && ds.lio_n > 0 && d.rank == 0;
  c.lio_o = 6 * ds.lio_frame;
  c.vis_s = c.vsplit ? d.vis_Hs + (size_t)w * d.vs_blocks * NV * V_LD : c.Z;
  c.ntri = d.asm_n;      // (the full table is ordered by the larger dim: a batch without GNSS windows stops after the 187 core dims; or the compact one)
  return c;
}
#define ASM_UNPACK(c)                                                                                                              \
  const bool dense_here = (c).dense_here, vsplit = (c).vsplit, lio_on = (c).lio_on; const int lio_o = (c).lio_o, ntri = (c).ntri;   \
  double *H = (c).H, *g = (c).g, *E = (c).E, *eg = (c).eg; const int4 *tab = (c).tab; const double *Z = (c).Z;                    \
  const double *imu_w = (c).imu_w, *wheel_w = (c).wheel_w, *prior_w = (c).prior_w, *vis_s = (c).vis_s;                            \
  (void)dense_here; (void)vsplit; (void)lio_on; (void)lio_o; (void)ntri; (void)H; (void)g; (void)E; (void)eg; (void)tab; (void)Z;  \
  (void)imu_w; (void)wheel_w; (void)prior_w; (void)vis_s
// H: the lower triangle, entries gt, gt + gn, ... of the table (pre: the first pass's four table entries, fetched by the caller
// before the tables were staged — or nullptr)
// U: entries per thread and pass; VSPLIT: the visual entry is the sum of k_visblock_small's start-frame blocks (small batches)
template <int U, bool VSPLIT>
__device__ __forceinline__ void asm_H(const BatchDev &d, const int w, const double *vis_w, const AsmTab &tb, const AsmCommon &cm, const int gt, const int gn,
                                      const int4 *pre) {
  ASM_UNPACK(cm);
  for (int e0 = gt; e0 < ntri; e0 += U * gn) {
    int4 ent[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int e = e0 + u * gn; ent[u] = (pre && e0 == gt) ? pre[u] : (e < ntri ? tab[e] : make_int4(-1, 0, 0, 0)); }
    const double *p[U][6];
    bool on[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int a = ent[u].x & 255, b = (ent[u].x >> 8) & 255;
      on[u] = ent[u].x >= 0 && tb.act[a] && tb.act[b];
      const int y = (on[u] && dense_here) ? ent[u].y : 0, z = (on[u] && dense_here && tb.n_wheel > 0) ? ent[u].z : 0;
      const int i0 = (y & 15) - 1, i1 = ((y >> 16) & 15) - 1, j0 = (z & 15) - 1, j1 = ((z >> 16) & 15) - 1;
      const int q0 = i0 >= 0 ? tb.imu_of_frame[i0] : -1, q1 = i1 >= 0 ? tb.imu_of_frame[i1] : -1;
      const int r0 = j0 >= 0 ? tb.wheel_of_frame[j0] : -1, r1 = j1 >= 0 ? tb.wheel_of_frame[j1] : -1;
      p[u][0] = q0 >= 0 ? imu_w + q0 * IMU_PART + ((y >> 4) & 1023) : Z;
      p[u][1] = q1 >= 0 ? imu_w + q1 * IMU_PART + ((y >> 20) & 1023) : Z;
      p[u][2] = r0 >= 0 ? wheel_w + r0 * WHEEL_PART + ((z >> 4) & 1023) : Z;
      p[u][3] = r1 >= 0 ? wheel_w + r1 * WHEEL_PART + ((z >> 20) & 1023) : Z;
      const int pa = on[u] && dense_here && tb.prior_n > 0 ? tb.prior_map[b] : -1, pb = on[u] && dense_here && tb.prior_n > 0 ? tb.prior_map[a] : -1;
      p[u][4] = (pa >= 0 && pb >= 0) ? prior_w + (size_t)pa * tb.prior_n + pb : Z;
      p[u][5] = (on[u] && a < NV && !VSPLIT) ? vis_w + b * V_LD + a : Z;
    }
    // small batches: the visual entry is the sum of the start-frame blocks, in start-frame order. Their loads are issued with the
    // factors' (unconditionally: a zero slot with stride 0 where the entry has no visual part), the sums follow below
    double blk[VSPLIT ? U : 1][VS_BLOCKS];
    bool vs[U];
    const int nvb = d.vs_blocks;
    if (VSPLIT) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int a = ent[u].x & 255, b = (ent[u].x >> 8) & 255;
        vs[u] = on[u] && a < NV;
        const double *q = vs[u] ? vis_s + b * V_LD + a : Z;
        const size_t st = vs[u] ? (size_t)NV * V_LD : 0;
#pragma unroll
        for (int f = 0; f < VS_BLOCKS; f++) blk[VSPLIT ? u : 0][f] = *(f < nvb ? q + f * st : Z);     // (7 x 7 partials: block 0 holds the whole visual block)
      }
    }
    double v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = *p[u][0] + *p[u][1] + *p[u][2] + *p[u][3] + *p[u][4] + *p[u][5];
    if (VSPLIT) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int a = ent[u].x & 255, b = (ent[u].x >> 8) & 255;
        if (!vs[u]) continue;
        double sv = 0.0;
#pragma unroll
        for (int f = 0; f < VS_BLOCKS; f++) sv += blk[VSPLIT ? u : 0][f];
        if (lio_on && b >= lio_o && a < lio_o + 6) {   // LiDAR block of pose lio_frame: added last, as the one-workgroup form does
          const int ra = b - lio_o, rb = a - lio_o;      // ra <= rb: packed upper triangle of k_lio_window
          const int e = ra * 6 - ra * (ra - 1) / 2 + (rb - ra);
          double lv = 0.0;
          for (int q = 0; q < LIOW_WGS; q++) lv += d.lio_part[((size_t)w * LIOW_WGS + q) * LIOW_PART + e];
          sv += lv;
        }
        v[u] += sv;
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (ent[u].x < 0) continue;
      const int a = ent[u].x & 255, b = (ent[u].x >> 8) & 255;
      double x = v[u];
      if (b >= T_EXW && a <= T_TDW && on[u] && dense_here && tb.n_wheel > 0) {
        // wheel extrinsic / intrinsic / td_wheel block: every wheel factor contributes (10 loads in flight)
        const int off = part_lower(wheel_loc(b, 0), wheel_loc(a, 0), 22);   // global dims: the column does not depend on the factor
        double ws = 0.0;
#pragma unroll
        for (int i = 0; i <= NF - 2; i++) {
          const int q = tb.wheel_of_frame[i];
          ws += *(q >= 0 ? wheel_w + q * WHEEL_PART + off : Z);
        }
        x += ws;
      }
      if ((tb.n_plane > 0 || tb.use_anchor) && on[u] && dense_here) x += plane_anchor_term(d, w, tb.n_plane, tb.use_anchor, a, b);
      H[(size_t)a * ND + b] = x;   // lower triangle only (b <= a): k_solve never reads the mirror
    }
  }
}
// Throughput batches (k_visasm; end of round 6): the sums of asm_H<U, false> — the same terms in the same order, bit for bit — written
// without a branch per contributor. asm_H's `cond ? tb.x[i] : -1` and `on = x >= 0 && tb.act[a] && tb.act[b]` compile to an EXEC-masked
// region and a wait per LDS lookup (the compiler may not read LDS at an index it cannot prove valid), its `cond ? base + off : Z` to 64-bit
// selects and 64-bit address arithmetic per load: ~100 instructions and five dependent waits per entry. Here every lookup reads a clamped
// index unconditionally (one wait for all of them), every load is [wave-uniform base + 32-bit offset] (offset 0 where the contributor is
// absent) and the VALUE is selected; the visual entry is read from the caller's LDS block as LDS; the next round's table entries are
// requested before the current round's are decoded.
template <int U>
__device__ __forceinline__ void asm_H_tp(const BatchDev &d, const int w, const double *vis_lds, const AsmTab &tb, const AsmCommon &cm, const int gt, const int gn) {
  ASM_UNPACK(cm);
  const bool wheel_on = dense_here && tb.n_wheel > 0, prior_on = dense_here && tb.prior_n > 0;
  const bool pa_on = (tb.n_plane > 0 || tb.use_anchor) && dense_here;
  const int prior_n = tb.prior_n, last = ntri - 1;
  imu_w = uniform_ptr(imu_w); wheel_w = uniform_ptr(wheel_w); prior_w = uniform_ptr(prior_w); H = uniform_ptr(H); tab = uniform_ptr(tab);
  int4 nxt[U];
  typedef int asm_iv4 __attribute__((ext_vector_type(4)));      // (a plain vector type: HIP's int4 is a class, not loadable through an address-space pointer)
  typedef __attribute__((address_space(1))) asm_iv4 glb_iv4;
  const glb_iv4 *gtab = (const glb_iv4 *)tab;
  auto tab_at = [&](int e) __attribute__((always_inline)) { const asm_iv4 v = gtab[e]; return make_int4(v.x, v.y, v.z, v.w); };
#pragma unroll
  for (int u = 0; u < U; u++) nxt[u] = tab_at(min(gt + u * gn, last));
  for (int e0 = gt; e0 < ntri; e0 += U * gn) {
    int4 ent[U];
    bool valid[U];
#pragma unroll
    for (int u = 0; u < U; u++) { ent[u] = nxt[u]; valid[u] = e0 + u * gn < ntri && ent[u].x >= 0; }
#pragma unroll
    for (int u = 0; u < U; u++) nxt[u] = tab_at(min(e0 + (U + u) * gn, last));
    int a[U], b[U], q0[U], q1[U], r0[U], r1[U], pma[U], pmb[U];
    unsigned char acta[U], actb[U];
#pragma unroll
    for (int u = 0; u < U; u++) {      // every LDS lookup of the round, unconditionally at a clamped index
      a[u] = valid[u] ? ent[u].x & 255 : 0; b[u] = valid[u] ? (ent[u].x >> 8) & 255 : 0;
      acta[u] = tb.act[a[u]]; actb[u] = tb.act[b[u]];
      q0[u] = tb.imu_of_frame[min(max((ent[u].y & 15) - 1, 0), NF - 1)]; q1[u] = tb.imu_of_frame[min(max(((ent[u].y >> 16) & 15) - 1, 0), NF - 1)];
      r0[u] = tb.wheel_of_frame[min(max((ent[u].z & 15) - 1, 0), NF - 1)]; r1[u] = tb.wheel_of_frame[min(max(((ent[u].z >> 16) & 15) - 1, 0), NF - 1)];
      pma[u] = tb.prior_map[a[u]]; pmb[u] = tb.prior_map[b[u]];
    }
    bool on[U], use[U][6];
    double ld[U][6];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int y = ent[u].y, z = ent[u].z;
      on[u] = valid[u] & (acta[u] != 0) & (actb[u] != 0);
      use[u][0] = on[u] & dense_here & ((y & 15) != 0) & (q0[u] >= 0);
      use[u][1] = on[u] & dense_here & (((y >> 16) & 15) != 0) & (q1[u] >= 0);
      use[u][2] = on[u] & wheel_on & ((z & 15) != 0) & (r0[u] >= 0);
      use[u][3] = on[u] & wheel_on & (((z >> 16) & 15) != 0) & (r1[u] >= 0);
      use[u][4] = on[u] & prior_on & (pma[u] >= 0) & (pmb[u] >= 0);
      use[u][5] = on[u] & (a[u] < NV);
      ld[u][0] = ld_u32(imu_w, use[u][0] ? (unsigned)(q0[u] * IMU_PART + ((y >> 4) & 1023)) : 0u);
      ld[u][1] = ld_u32(imu_w, use[u][1] ? (unsigned)(q1[u] * IMU_PART + ((y >> 20) & 1023)) : 0u);
      ld[u][2] = ld_u32(wheel_w, use[u][2] ? (unsigned)(r0[u] * WHEEL_PART + ((z >> 4) & 1023)) : 0u);
      ld[u][3] = ld_u32(wheel_w, use[u][3] ? (unsigned)(r1[u] * WHEEL_PART + ((z >> 20) & 1023)) : 0u);
      ld[u][4] = ld_u32(prior_w, use[u][4] ? (unsigned)(pmb[u] * prior_n + pma[u]) : 0u);      // (asm_H: pa = prior_map[b] the row, pb = prior_map[a] the column)
      ld[u][5] = vis_lds[use[u][5] ? b[u] * V_LD + a[u] : 0];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (!valid[u]) continue;
      double x = (use[u][0] ? ld[u][0] : 0.0) + (use[u][1] ? ld[u][1] : 0.0) + (use[u][2] ? ld[u][2] : 0.0) + (use[u][3] ? ld[u][3] : 0.0)
               + (use[u][4] ? ld[u][4] : 0.0) + (use[u][5] ? ld[u][5] : 0.0);
      if (b[u] >= T_EXW && a[u] <= T_TDW && on[u] && wheel_on) {
        // wheel extrinsic / intrinsic / td_wheel block: every wheel factor contributes (10 loads in flight)
        const int off = part_lower(wheel_loc(b[u], 0), wheel_loc(a[u], 0), 22);   // global dims: the column does not depend on the factor
        double ws = 0.0;
#pragma unroll
        for (int i = 0; i <= NF - 2; i++) {
          const int q = tb.wheel_of_frame[i];
          const double wl = ld_u32(wheel_w, q >= 0 ? (unsigned)(q * WHEEL_PART + off) : 0u);
          ws += q >= 0 ? wl : 0.0;
        }
        x += ws;
      }
      if (pa_on && on[u]) x += plane_anchor_term(d, w, tb.n_plane, tb.use_anchor, a[u], b[u]);
      st_u32(H, (unsigned)(a[u] * ND + b[u]), x);   // lower triangle only (b <= a): k_solve never reads the mirror
    }
  }
}
// E (73 x 73, symmetric): entries gt, gt + gn, ... of its triangle
__device__ __forceinline__ void asm_E(const BatchDev &d, const int w, const AsmTab &tb, const AsmCommon &cm, const int gt, const int gn) {
  ASM_UNPACK(cm);
  // (two entries per round: eight loads of the Schur partials in flight; a thread of k_visasm has five or six entries)
  constexpr int NE = NV * (NV + 1) / 2;
  for (int e = gt; e < NE; e += 2 * gn) {
    int a0, b0, a1 = 0, b1 = 0;
    tri_decode(e, a0, b0);
    const bool two = e + gn < NE;
    if (two) tri_decode(e + gn, a1, b1);
    const double g0 = gather_E11(d, Z, w, b0, a0), g1 = gather_E11(d, Z, w, b1, a1);      // (unconditional: selected below)
    const double ev0 = (tb.act[a0] && tb.act[b0]) ? g0 : 0.0, ev1 = (two && tb.act[a1] && tb.act[b1]) ? g1 : 0.0;
    E[a0 * NV + b0] = ev0;
    E[b0 * NV + a0] = ev0;
    if (two) { E[a1 * NV + b1] = ev1; E[b1 * NV + a1] = ev1; }
  }
}
// g and eg: dims gt, gt + gn, ...
__device__ __forceinline__ void asm_g(const BatchDev &d, const int w, const double *vis_w, const AsmTab &tb, const AsmCommon &cm, const int gt, const int gn) {
  ASM_UNPACK(cm);
  for (int a = gt; a < ND; a += gn) {
    double v = 0.0;
    if (tb.act[a]) {
      if (ASM_TP_ON(d, 8)) v = dense_here ? gather_g_dense_tp(d, tb, w, a) : 0.0;
      else v = dense_here ? gather_g_dense(d, tb, Z, w, a) : 0.0;
      if (a < NV) {
        if (!vsplit) v += vis_w[a * V_LD + NV];
        else {
          double sv = 0.0;
          for (int f = 0; f < d.vs_blocks; f++) sv += vis_s[(size_t)f * NV * V_LD + a * V_LD + NV];
          if (lio_on && a >= lio_o && a < lio_o + 6) {
            double lv = 0.0;
            for (int q = 0; q < LIOW_WGS; q++) lv += d.lio_part[((size_t)w * LIOW_WGS + q) * LIOW_PART + 21 + a - lio_o];
            sv += lv;
          }
          v += sv;
        }
      }
    }
    g[a] = v;
    if (a < NV) eg[a] = tb.act[a] ? gather_E11(d, Z, w, a, NV) : 0.0;
  }
}
__device__ __forceinline__ void assemble_body(const BatchDev &d, const int w, const double *vis_w, AsmTab &tb, const int gt, const int gn, const bool staged = false) {
#if GFBE_DIAG
  double *astamp = d.timing + (size_t)d.B * 32 + 24;
#define ASTAMP(i) do { if (w == 0 && gt == 0) astamp[i] = (double)wall_clock64(); } while (0)
#else
#define ASTAMP(i) do { } while (0)
#endif
  if (!staged) {      // (staged: the caller filled tb and passed its barrier)
    asm_stage_tables(d, w, tb);
    __syncthreads();
  }
  ASTAMP(4);
  const AsmCommon cm = asm_common(d, w);
  if (ASM_TP_ON(d, 1)) asm_H_tp<GFBE_ASM_U>(d, w, vis_w, tb, cm, gt, gn);      // (k_visasm: throughput batches, the visual block in the caller's LDS)
  else asm_H<GFBE_ASM_U, false>(d, w, vis_w, tb, cm, gt, gn, nullptr);
  ASTAMP(5);
  asm_E(d, w, tb, cm, gt, gn);      // (E's entries the branch-free way — loop-free triangle decode, 2 / 3 / 6 entries in flight — measured: no gain, profiles/r6_late_experiments.txt)
  ASTAMP(6);
  asm_g(d, w, vis_w, tb, cm, gt, gn);
  ASTAMP(7);
#undef ASTAMP
}
// small batches (a single window's latency): the visual blocks come from k_visblock_small, and every thread has ONE item — the
// workgroups [0, nH) take an entry of H each (the table entry is fetched before the descriptor tables are staged), the next
// ASM_E_WGS an entry of E, the last one g and eg — instead of an H entry, then an E entry, then a gradient entry one after the other:
// the dependent memory round trips of the three parts side by side (14.1 -> 8.2 us per launch for one window).
#define ASM_E_WGS ((NV * (NV + 1) / 2 + ASM_THREADS - 1) / ASM_THREADS)
__global__ __launch_bounds__(ASM_THREADS) void k_assemble(BatchDev d0, int nH) {
  const int w = blockIdx.y, bx = blockIdx.x;
  const BatchDev d = lin_view(d0, d0.ctl[w].lb);
  const WinCtl &c = d.ctl[w];
  if (c.done || c.reuse) return;
  __shared__ AsmTab tb;
  const int gtH = bx * ASM_THREADS + threadIdx.x, gnH = nH * ASM_THREADS;
  int4 pre[1] = {make_int4(-1, 0, 0, 0)};
  if (bx < nH) {
    const int ntri = d.asm_n;
    if (gtH < ntri) pre[0] = ((const int4 *)d.asm_tab)[gtH];
  }
  asm_stage_tables(d, w, tb);
  __syncthreads();
  const AsmCommon cm = asm_common(d, w);
  const double *vis_w = d.vis_H + (size_t)w * NV * V_LD;
  if (bx < nH) asm_H<1, true>(d, w, vis_w, tb, cm, gtH, gnH, pre);
  else if (bx < nH + ASM_E_WGS) asm_E(d, w, tb, cm, (bx - nH) * ASM_THREADS + threadIdx.x, ASM_E_WGS * ASM_THREADS);
  else asm_g(d, w, vis_w, tb, cm, threadIdx.x, ASM_THREADS);
}
// throughput batches: ONE workgroup per window builds the visual block in LDS and assembles from there. (Measured, 1024 windows:
// 16 workgroups of 256 threads per window took 313 us per launch with the SIMDs half empty — the kernel was bound by the
// dispatch and the table staging of its 16384 short workgroups; 2 workgroups 200 us, one 192 us; fused with k_visblock the
// 43 KB block per window neither goes to HBM nor comes back.)
#ifndef GFBE_VISASM_WAVES
#define GFBE_VISASM_WAVES 4      // waves per SIMD the register allocation of k_visasm aims at
#endif
__global__ __launch_bounds__(VB_GROUP, GFBE_VISASM_WAVES) void k_visasm(BatchDev d0) {
  const int w = blockIdx.x;
  const BatchDev d = lin_view(d0, d0.ctl[w].lb);
  const WinCtl &c = d.ctl[w];
  if (c.done || c.reuse) return;
  // (k_linschur batches with the second set: the set that is about to be solved was formed with the window's mu — the record k_linschur's
  //  gate in front of the next iteration compares; written here, a launch later, because the workgroups of ONE k_linschur launch read it)
  if (d.linschur && d.spec && threadIdx.x == 0) d.ctl[w].sw_mu[c.lb] = c.mu;
  __shared__ double V[NV * V_LD];
  __shared__ AsmTab tb;
  static_assert(VB_GROUP >= ND, "asm_stage_store: one table entry per thread");
  if (ASM_TP_ON(d, 4)) {
    const AsmStagePre pre = asm_stage_load(d, w);
    visblock_body<false, true, true>(d, w, 0, NF - 2, 0, V);
    asm_stage_store(tb, pre);
    __syncthreads();
    assemble_body(d, w, V, tb, threadIdx.x, VB_GROUP, true);
  } else {
    visblock_body<false, true, true>(d, w, 0, NF - 2, 0, V);
    __syncthreads();
    assemble_body(d, w, V, tb, threadIdx.x, VB_GROUP);
  }
}

// =============================================================================================
// k_lm_step: back-substitution of the eliminated landmarks and their share of the dogleg scalars.
// =============================================================================================
// One landmark tile (64 lanes): sy / sv = the scaled Gauss-Newton step and Cauchy direction of the visual dims, staged by the caller.
// Per frame f: [ dp_y | u_y = R_f dtheta_y | dp_v | u_v | t_f ] from the scaled steps sy / sv (LDS, already staged; the caller's barrier
// lies between) and the frame constants of the linearisation point — threads 0 .. NF - 1 of the workgroup.
__device__ __forceinline__ void stage_frame_steps(const BatchDev &d, const WinCtl &c, const int w, const int t, const double *sy, const double *sv, double *fs) {
  if (t >= NF) return;
  const double *Rt = d.pc + (((size_t)w * 3 + c.cur) * NPAIR + t * (NF + 1)) * PC_DOUBLES + LM_RT_OFF;
  double *o = fs + t * LM_FS;
#pragma unroll
  for (int q = 0; q < 3; q++) {
    o[q] = sy[6 * t + q]; o[6 + q] = sv[6 * t + q]; o[12 + q] = Rt[9 + q];
    o[3 + q] = __builtin_fma(Rt[3 * q], sy[6 * t + 3], __builtin_fma(Rt[3 * q + 1], sy[6 * t + 4], Rt[3 * q + 2] * sy[6 * t + 5]));
    o[9 + q] = __builtin_fma(Rt[3 * q], sv[6 * t + 3], __builtin_fma(Rt[3 * q + 1], sv[6 * t + 4], Rt[3 * q + 2] * sv[6 * t + 5]));
  }
}
// fs: the frames' steps in the form of lm_row_dot2 (stage_frame_steps below) when the rows are compressed, else nullptr
// lm_step_tile for a single window's launch (k_lm_step_fused, end of round 6), the compressed-row form: the same operations on every landmark in
// the same order, the same bits — with ONE round of loads in front of everything. lm_step_tile below asks for the landmark's record, then
// (its track length known) for its rows one observation step at a time, then for its scalars: two dependent round trips + one per step
// (its ISA: three loads, the arithmetic, s_waitcnt vmcnt(0), the back edge). Seven other waves of a SIMD hide that in a throughput batch
// (a ring of rows in flight there costs a wave per SIMD and gains nothing: 56.3 against 56.2 - 58.5 us per 512 windows,
// profiles/r6_late_experiments.txt); a single window's tile has its SIMD to itself and waits every time. Here the record, D, x, the
// scalars and the rows of the first AHEAD + 1 steps are requested together — rows past a short track's end are allocated and never used
// (lm_hP holds MAXOBS steps per slot) — and the rows AHEAD steps ahead stay in flight while a step is multiplied.
template <int AHEAD>
__device__ __forceinline__ void lm_step_tile_ahead(const BatchDev &d, const WinDesc &ds, const WinCtl &c, const int w, const int tile, const int t, const double *fs) {
  static_assert(AHEAD >= 1 && AHEAD < MAXOBS, "rows of the first AHEAD + 1 observation steps exist for every slot");
  const int s0 = d.tile_start[ds.tile_off + tile];
  const int slot = ds.lm_off + tile * LM_TILE + t;
  const size_t TL = d.tot_lm;
  const int info = d.lm_info[slot];
  double Dx[6], ring[AHEAD + 1][3];
#pragma unroll
  for (int q = 0; q < 6; q++) Dx[q] = d.lm_hC[(size_t)q * TL + slot];
#pragma unroll
  for (int a = 0; a <= AHEAD; a++)
#pragma unroll
    for (int q = 0; q < 3; q++) ring[a][q] = d.lm_hP[((size_t)a * 6 + q) * TL + slot];
  const double sl = d.lm_sl[slot], Hll = d.lm_Hll[slot], gl = d.lm_gl[slot], lam = d.lam[(size_t)c.cur * TL + slot], mu = c.mu;
  const int m = (info >> 8) & 0xff;
  const bool free_lm = ((info >> 24) & 1) && !((info >> 16) & 1) && m > 0;
  double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (free_lm) {
    double hy = 0.0, hv = 0.0;
    lm_row_dot2(fs + s0 * LM_FS, Dx, Dx + 3, hy, hv);
    const int mlast = m - 1;
    for (int k = 0; k < m; k++) {
      double dk[3], oy, ov;
#pragma unroll
      for (int q = 0; q < 3; q++) dk[q] = ring[0][q];
#pragma unroll
      for (int a = 0; a < AHEAD; a++)
#pragma unroll
        for (int q = 0; q < 3; q++) ring[a][q] = ring[a + 1][q];
#pragma unroll
      for (int q = 0; q < 3; q++) ring[AHEAD][q] = d.lm_hP[((size_t)min(k + 1 + AHEAD, mlast) * 6 + q) * TL + slot];
      lm_row_dot2(fs + (s0 + 1 + k) * LM_FS, dk, Dx + 3, oy, ov);
      hy -= oy; hv -= ov;
    }
    const double hll = sl * sl * Hll, d2 = clamp_diag(hll), glt = sl * gl;
    const double yl = (glt - sl * hy) / (hll + mu * d2);
    const double vl = glt / d2;
    d.lm_yl[slot] = yl; d.lm_vl[slot] = vl;
    p[0] = glt * glt / d2;                                   // G2
    p[1] = d2 * yl * yl;                                     // N2
    p[2] = glt * yl;                                         // gy
    p[3] = 2.0 * vl * sl * hv + hll * vl * vl;               // vHv
    p[4] = vl * sl * hy + yl * sl * hv + hll * vl * yl;      // vHy
    p[5] = 2.0 * yl * sl * hy + hll * yl * yl;               // yHy
    p[6] = fabs(gl);                                         // gradient max-norm share
    p[7] = lam * lam;                                        // |x|^2 share
  }
  double *out = d.tile_gram + ((size_t)w * d.max_tiles + tile) * 8;
  for (int q = 0; q < 8; q++) {
    const double r = (q == 6) ? wave_max(p[q]) : wave_sum(p[q]);
    if (t == 0) out[q] = r;
  }
}
__device__ __forceinline__ void lm_step_tile(const BatchDev &d, const WinDesc &ds, const WinCtl &c, const int w, const int tile, const int t,
                                             const double *sy, const double *sv, const double *fs) {
  const int s0 = d.tile_start[ds.tile_off + tile];
  const int slot = ds.lm_off + tile * LM_TILE + t;
  const int info = d.lm_info[slot];
  const int m = (info >> 8) & 0xff;
  const bool free_lm = ((info >> 24) & 1) && !((info >> 16) & 1) && m > 0;
  const size_t TL = d.tot_lm;
  double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (free_lm) {
    double hy = 0.0, hv = 0.0;
    if (fs) {      // compressed rows (k_vis<0, false>): h_l . delta = D . v_i(x) - sum_k d_k . v_jk(x)  (gfbe_devutil.h)
      double Dx[6];
#pragma unroll
      for (int q = 0; q < 6; q++) Dx[q] = d.lm_hC[(size_t)q * TL + slot];
      lm_row_dot2(fs + s0 * LM_FS, Dx, Dx + 3, hy, hv);
      for (int k = 0; k < m; k++) {
        double dk[3], oy, ov;
#pragma unroll
        for (int q = 0; q < 3; q++) dk[q] = d.lm_hP[((size_t)k * 6 + q) * TL + slot];
        lm_row_dot2(fs + (s0 + 1 + k) * LM_FS, dk, Dx + 3, oy, ov);
        hy -= oy; hv -= ov;
      }
    } else if (d.vis_full) {
      for (int q = 0; q < 6; q++) {
        const double hi = d.lm_hC[(size_t)q * TL + slot], he = d.lm_hC[(size_t)(6 + q) * TL + slot];
        hy += hi * sy[6 * s0 + q] + he * sy[T_EX + q];
        hv += hi * sv[6 * s0 + q] + he * sv[T_EX + q];
      }
      { const double ht = d.lm_hC[(size_t)12 * TL + slot]; hy += ht * sy[T_TD]; hv += ht * sv[T_TD]; }
    } else {      // constant extrinsic and td in every window of the batch: their rows of lm_hC are not written and their step is zero —
                  // the sums below add the same terms in the same order (x + 0 * 0 = x) without the seven loads
      for (int q = 0; q < 6; q++) {
        const double hi = d.lm_hC[(size_t)q * TL + slot];
        hy += hi * sy[6 * s0 + q] + 0.0;
        hv += hi * sv[6 * s0 + q] + 0.0;
      }
    }
    if (!fs)
    for (int k = 0; k < m; k++)
      for (int q = 0; q < 6; q++) {
        const double h = d.lm_hP[((size_t)k * 6 + q) * TL + slot];
        hy += h * sy[6 * (s0 + 1 + k) + q];
        hv += h * sv[6 * (s0 + 1 + k) + q];
      }
    const double sl = d.lm_sl[slot], Hll = d.lm_Hll[slot], gl = d.lm_gl[slot];
    const double hll = sl * sl * Hll, d2 = clamp_diag(hll), glt = sl * gl;
    const double yl = (glt - sl * hy) / (hll + c.mu * d2);
    const double vl = glt / d2;
    d.lm_yl[slot] = yl; d.lm_vl[slot] = vl;
    p[0] = glt * glt / d2;                                   // G2
    p[1] = d2 * yl * yl;                                     // N2
    p[2] = glt * yl;                                         // gy
    p[3] = 2.0 * vl * sl * hv + hll * vl * vl;               // vHv
    p[4] = vl * sl * hy + yl * sl * hv + hll * vl * yl;      // vHy
    p[5] = 2.0 * yl * sl * hy + hll * yl * yl;               // yHy
    p[6] = fabs(gl);                                         // gradient max-norm share
    const double lam = d.lam[(size_t)c.cur * TL + slot];
    p[7] = lam * lam;                                        // |x|^2 share
  }
  double *out = d.tile_gram + ((size_t)w * d.max_tiles + tile) * 8;
  for (int q = 0; q < 8; q++) {
    const double r = (q == 6) ? wave_max(p[q]) : wave_sum(p[q]);
    if (t == 0) out[q] = r;
  }
}
__global__ __launch_bounds__(LM_TILE) void k_lm_step(BatchDev d0) {
  const int w = blockIdx.x, tile = blockIdx.y;   // tile-major dispatch (longest tracks first)
  const BatchDev d = lin_view(d0, d0.ctl[w].lb);
  const WinDesc &ds = d.desc[w];
  if (tile >= ds.n_tiles || !TILE_OWNED(d, tile)) return;
  const WinCtl &c = d.ctl[w];
  if (c.done || c.reuse) return;
  __shared__ double sy[NV], sv[NV];
  const int t = threadIdx.x;
  for (int a = t; a < NV; a += LM_TILE) {
    const double s = d.sp[(size_t)w * ND + a];
    sy[a] = s * d.yp[(size_t)w * ND + a];
    sv[a] = s * d.vp[(size_t)w * ND + a];
  }
  __syncthreads();
  __shared__ double fsteps[NF * LM_FS];
  const bool comp = !d.vis_full;      // the rows are in the compressed form of k_vis<0, false>
  if (comp) { stage_frame_steps(d, c, w, t, sy, sv, fsteps); __syncthreads(); }
  lm_step_tile(d, ds, c, w, tile, t, sy, sv, comp ? fsteps : nullptr);
}

// =============================================================================================
// k_step: scalar trust-region logic of one iteration (one thread per window).
// =============================================================================================
// landmark shares of the dogleg scalars: the tiles' partials (k_lm_step), lanes stride the tiles, fixed tree order.
// p = [G2, N2, gy, vHv, vHy, yHy, max |gradient|, |x|^2]; every lane returns the totals.
__device__ __forceinline__ void tile_gram_sum(const BatchDev &d, const WinDesc &ds, int w, int lane, double p[8]) {
#pragma unroll
  for (int k = 0; k < 8; k++) p[k] = 0.0;
  for (int q = lane; q < ds.n_tiles; q += 64) {
    const double *tg = d.tile_gram + ((size_t)w * d.max_tiles + q) * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) p[k] = (k == 6) ? fmax(p[k], tg[k]) : p[k] + tg[k];
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double other = __shfl_xor(p[k], o, 64); p[k] = (k == 6) ? fmax(p[k], other) : p[k] + other; }
  }
}
// landmark sharding: this rank's row of the exchange blocks (the other rows zeroed), all-reduced by the host loop
__global__ __launch_bounds__(64) void k_xchg_gram(BatchDev d) {
  const int w = blockIdx.x, lane = threadIdx.x;
  double p[8];
  tile_gram_sum(d, d.desc[w], w, lane, p);
  for (int r = 0; r < d.world; r++)
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (lane == k) d.xb[((size_t)w * d.world + r) * XCHG + k] = (r == d.rank) ? p[k] : 0.0;
}
__device__ __forceinline__ void tile_cand_sum(const BatchDev &d, const WinDesc &ds, int w, int lane, double &cand, double &d2, double &n2) {
  cand = 0.0; d2 = 0.0; n2 = 0.0;
  for (int q = lane; q < ds.n_tiles; q += 64) {
    const double *o = d.tile_cand + ((size_t)w * d.max_tiles + q) * 4;
    cand += o[0]; d2 += o[1]; n2 += o[2];
  }
}
__global__ __launch_bounds__(64) void k_xchg_cand(BatchDev d) {
  const int w = blockIdx.x, lane = threadIdx.x;
  double cand, d2, n2;
  tile_cand_sum(d, d.desc[w], w, lane, cand, d2, n2);
  cand = wave_sum(cand); d2 = wave_sum(d2); n2 = wave_sum(n2);
  for (int r = 0; r < d.world; r++)
    if (lane < XCHG) d.xc[((size_t)w * d.world + r) * XCHG + lane] = (r != d.rank) ? 0.0 : (lane == 0 ? cand : (lane == 1 ? d2 : (lane == 2 ? n2 : 0.0)));
}
// landmark sharding: inverse depths of the other ranks' tiles are zeroed before the final sum all-reduce
__global__ __launch_bounds__(LM_TILE) void k_lam_mask(BatchDev d) {
  const int w = blockIdx.y, tile = blockIdx.x;
  const WinDesc &ds = d.desc[w];
  if (tile >= ds.n_tiles || TILE_OWNED(d, tile)) return;
  const int slot = ds.lm_off + tile * LM_TILE + threadIdx.x;
  d.lam[slot] = 0.0;
  d.lam[(size_t)d.tot_lm + slot] = 0.0;
}

// c: the scalars of the window's trust-region state — WinCtl itself, or a register copy of them (StepLocal: k_lm_step_fused loads it
// before its workgroup arrives); cg: WinCtl in memory, for the per-iteration records (stores only).
struct StepLocal {
  int done, have_step, reuse, iter, termination, status, invalid_steps;
  double G2, N2, gy, vHv, vHy, yHy, grad_max, x_norm, alpha, radius, c1, c2, step_norm, model_change, cost, mu;
  long long t_start;
};
template <class CT>
__device__ __forceinline__ void step_body(const BatchDev &d, const WinDesc &ds, CT &c, WinCtl &cg, const int w, const int lane) {
  if (c.done) return;
  if (c.have_step == 2) {   // fresh linearisation: fold in the landmark shares (lanes stride the tiles; fixed tree order)
    double p[8];
    if (!d.sharded) tile_gram_sum(d, ds, w, lane, p);
    else {   // landmark sharding: the ranks' shares (k_xchg_gram + all-reduce), combined in rank order
#pragma unroll
      for (int k = 0; k < 8; k++) p[k] = 0.0;
      for (int r = 0; r < d.world; r++) {
        const double *xr = d.xb + ((size_t)w * d.world + r) * XCHG;
#pragma unroll
        for (int k = 0; k < 8; k++) p[k] = (k == 6) ? fmax(p[k], xr[k]) : p[k] + xr[k];
      }
    }
    if (lane == 0) {
      c.G2 += p[0]; c.N2 += p[1]; c.gy += p[2]; c.vHv += p[3]; c.vHy += p[4]; c.yHy += p[5];
      c.grad_max = fmax(c.grad_max, p[6]);
      c.x_norm = sqrt(c.x_norm + p[7]);
      c.alpha = c.G2 / c.vHv;
      c.reuse = 1;
    }
  }
  if (lane != 0) return;
  c.have_step = 0;
  // TrustRegionMinimizer::FinalizeIterationAndCheckIfMinimizerCanContinue
  const int max_it = min(d.opt.max_num_iterations, 15);
  // Solver::Options::max_solver_time_in_seconds (estimator.cpp:3369-3376), against the device's 100 MHz wall clock
  if (d.opt.max_solver_time_in_seconds > 0.0 && (double)((long long)wall_clock64() - c.t_start) * 1e-8 >= d.opt.max_solver_time_in_seconds) { c.done = 1; c.termination = 0; return; }
  if (c.iter >= max_it) { c.done = 1; c.termination = 0; return; }
  if (c.grad_max <= d.opt.gradient_tolerance) { c.done = 1; c.termination = 3; c.status = GFBE_OK; return; }
  if (c.radius < 1e-32) { c.done = 1; c.termination = 4; return; }
  c.iter++;
  const int it = c.iter;
  const double radius = c.radius;
  const double g_norm = sqrt(c.G2), gn_norm = sqrt(c.N2);
  double c1, c2, step_norm;
  if (gn_norm <= radius) { c1 = 0.0; c2 = -1.0; step_norm = gn_norm; }
  else if (g_norm * c.alpha >= radius) { c1 = -radius / g_norm; c2 = 0.0; step_norm = radius; }
  else {
    const double b_dot_a = c.alpha * c.gy;
    const double a_sq = (c.alpha * g_norm) * (c.alpha * g_norm);
    const double bma = a_sq - 2.0 * b_dot_a + c.N2;
    const double cc = b_dot_a - a_sq;
    const double dd = sqrt(cc * cc + bma * (radius * radius - a_sq));
    const double beta = (cc <= 0.0) ? (dd - cc) / bma : (radius * radius - a_sq) / (dd + cc);
    c1 = -c.alpha * (1.0 - beta); c2 = -beta;
    step_norm = sqrt(fmax(0.0, c1 * c1 * c.G2 + 2.0 * c1 * c2 * c.gy + c2 * c2 * c.N2));
  }
  const double model_change = -(c1 * c.G2 + c2 * c.gy) - 0.5 * (c1 * c1 * c.vHv + 2.0 * c1 * c2 * c.vHy + c2 * c2 * c.yHy);
  c.c1 = c1; c.c2 = c2; c.step_norm = step_norm; c.model_change = model_change;
  if (!(model_change > 0.0)) {   // TrustRegionMinimizer::HandleInvalidStep
    cg.accepted[it] = 0; cg.cost_history[it] = c.cost;
    if (++c.invalid_steps >= 5) { c.done = 1; c.termination = 4; c.status = GFBE_NUMERICAL_FAILURE; return; }
    c.mu *= GF_MU_INC; c.reuse = 0;
    return;
  }
  c.invalid_steps = 0;
  c.have_step = 1;
}
__global__ __launch_bounds__(64) void k_step(BatchDev d) {
  GFBE_SMALL_KERNEL_PRIO();
  const int w = blockIdx.x;
  step_body(d, d.desc[w], d.ctl[w], d.ctl[w], w, threadIdx.x);
}

// =============================================================================================
// k_candidate: x_cand = x (+) s * (c1 v + c2 y). Blocks [0, max_tiles) landmarks, block max_tiles
// the dense parameter blocks.
// =============================================================================================
// landmark tile `tile` (64 lanes); returns the lane's candidate inverse depth
__device__ __forceinline__ double candidate_tile(const BatchDev &d, const WinDesc &ds, const WinCtl &c, const int w, const int tile, const int t) {
  const size_t TL = d.tot_lm;
  const int slot = ds.lm_off + tile * LM_TILE + t;
  const int info = d.lm_info[slot];
  const bool valid = (info >> 24) & 1;
  const bool free_lm = valid && !((info >> 16) & 1) && ((info >> 8) & 0xff) > 0;
  const double lam = d.lam[(size_t)c.cur * TL + slot];
  double d2, n2;
  const double lc = candidate_lm(lam, d.lm_sl[slot], d.lm_vl[slot], d.lm_yl[slot], c.c1, c.c2, free_lm, d2, n2);
  d.lam[(size_t)(1 - c.cur) * TL + slot] = lc;
  d2 = wave_sum(d2); n2 = wave_sum(n2);
  if (t == 0) {
    double *o = d.tile_cand + ((size_t)w * d.max_tiles + tile) * 4;
    o[1] = d2; o[2] = n2;
  }
  return lc;
}
// the dense parameter blocks (one wave; sp_cand: NF + 1 PoseRT of LDS; contains a block barrier: call it from every thread of the
// workgroup or from a one-wave workgroup)
__device__ __forceinline__ void candidate_dense(const BatchDev &d, const WinDesc &ds, const WinCtl &c, const int w, const int t, PoseRT *sp_cand,
                                                const bool my_wave) {
  const double *X = d.x + ((size_t)w * 2 + c.cur) * NA;
  double *Y = d.x + ((size_t)w * 2 + 1 - c.cur) * NA;
  if (my_wave) {
    double d2 = 0.0, n2 = 0.0;
    for (int b = t; b < GFBE_BLK_COUNT; b += 64) {
      const int off = blk_tan(b), am = blk_amb(b), gs = blk_gsize(b);
      if (ds.blk_free[b]) {
        double dl[9];
        for (int k = 0; k < blk_lsize(b); k++) {
          const size_t a = (size_t)w * ND + off + k;
          dl[k] = d.sp[a] * (c.c1 * d.vp[a] + c.c2 * d.yp[a]);
          d.step[a] = dl[k];
        }
        if (gs == 7) {
          const unsigned char *mask = (b == GFBE_BLK_EX_CAM) ? ds.ex_cam_mask : (b == GFBE_BLK_EX_WHEEL ? ds.ex_wheel_mask : nullptr);
          pose_plus(X + am, dl, mask, Y + am);
        } else if (gs == 4) {   // para_plane_R: OrientationSubsetParameterization({2}) (estimator.cpp:3122)
          const unsigned char constant[3] = {0, 0, 1};
          orientation_plus(X + am, dl, constant, Y + am);
        } else {
          for (int k = 0; k < gs; k++) Y[am + k] = X[am + k] + dl[k];
        }
        for (int k = 0; k < gs; k++) { const double df = X[am + k] - Y[am + k]; d2 += df * df; n2 += Y[am + k] * Y[am + k]; }
      } else {
        for (int k = 0; k < gs; k++) Y[am + k] = X[am + k];
      }
    }
    d2 = wave_sum(d2); n2 = wave_sum(n2);
    if (t == 0) { d.dense_cand[(size_t)w * 4 + 1] = d2; d.dense_cand[(size_t)w * 4 + 2] = n2; }
    __threadfence_block();
  }
  // the candidate's pose-pair constants, for its cost evaluation and — if it is accepted — the next linearisation
  __syncthreads();
  // (every thread of the workgroup passes the barrier inside; the lanes of the other waves have no pair to form)
  pair_consts_of_state(Y, d.pc + ((size_t)w * 3 + (1 - c.cur)) * NPAIR * PC_DOUBLES, sp_cand, my_wave ? t : NPAIR);
}
__global__ __launch_bounds__(LM_TILE) void k_candidate(BatchDev d) {
  const int w = blockIdx.y;
  const WinDesc &ds = d.desc[w];
  const WinCtl &c = d.ctl[w];
  if (c.done || !c.have_step) return;
  const int t = threadIdx.x;
  if ((int)blockIdx.x < d.max_tiles) {
    const int tile = blockIdx.x;
    if (tile >= ds.n_tiles || !TILE_OWNED(d, tile)) return;
    candidate_tile(d, ds, c, w, tile, t);
  } else {
    __shared__ PoseRT sp_cand[NF + 1];
    candidate_dense(d, ds, c, w, t, sp_cand, true);
  }
}

// Throughput batches: ONE workgroup per window — wave 0 takes the dense blocks, then all four waves walk the landmark tiles
// (a wave per tile, the per-tile sums by the same 64 lanes as in k_candidate: the same bits). The tiles + 1 single-wave workgroups
// per window of k_candidate are 37 k dispatches of a few hundred nanoseconds of work each per launch of 1024 windows.
#ifndef CAND_THREADS
#define CAND_THREADS 256   // (512 / 1024 measured: 36 / 49 us per launch over 512 windows against 41, throughput -1 % / -4 %)
#endif
// the dense half alone (throughput batches, GFBE_FUSE_CAND: the landmark half runs at the head of the cost pass, k_vis<1>)
__global__ __launch_bounds__(LM_TILE) void k_candidate_dense(BatchDev d) {
  GFBE_SMALL_KERNEL_PRIO();
  const int w = blockIdx.x;
  const WinDesc &ds = d.desc[w];
  const WinCtl &c = d.ctl[w];
  if (c.done || !c.have_step) return;
  __shared__ PoseRT sp_cand[NF + 1];
  candidate_dense(d, ds, c, w, threadIdx.x, sp_cand, true);
}
// ... and behind k_step in ONE launch (round 6): both are one wave per window of dependent scalar work, 8 + 25 us per 512 windows as two
// launches. Lane 0's trust-region scalars reach the other lanes of its wave through a workgroup-scope fence; the
// arithmetic is step_body's and candidate_dense's, operand for operand.
#ifndef GFBE_FUSE_STEP_CAND
#define GFBE_FUSE_STEP_CAND 1
#endif
__global__ __launch_bounds__(CAND_THREADS) void k_candidate_window(BatchDev d) {
  const int w = blockIdx.x;
  const WinDesc &ds = d.desc[w];
  const WinCtl &c = d.ctl[w];
  if (c.done || !c.have_step) return;
  const int t = threadIdx.x, wave = t >> 6;
  __shared__ PoseRT sp_cand[NF + 1];
  for (int tile = wave; tile < ds.n_tiles; tile += CAND_THREADS / 64)
    if (TILE_OWNED(d, tile)) candidate_tile(d, ds, c, w, tile, t & 63);
  candidate_dense(d, ds, c, w, t, sp_cand, wave == 0);
}

// Small batches (GFBE_FUSE_SMALL bit 1): k_lm_step whose last workgroup of a window to finish (an arrival counter per window; nobody
// waits for anybody) goes on with k_step's scalar logic and the dense parameter blocks of k_candidate — two dependent launches
// less per iteration on a single window's latency path. The landmark half of k_candidate runs at the head of k_lin_small<1>'s tile
// workgroups. A window that re-uses its linearisation (or is done) has nothing to back-substitute: its workgroups only arrive.
// Returns true in the workgroup that arrived last, after an acquire fence (the others' stores are visible to it).
__device__ __forceinline__ bool arrive_last(int *cnt, const int expected, const int lane) {
  __threadfence();
  int last = 0;
  if (lane == 0) last = atomicAdd(cnt, 1) == expected - 1;
  last = __shfl(last, 0, 64);
  if (!last) return false;
  __threadfence();
  if (lane == 0) *cnt = 0;     // (ready for the next launch)
  return true;
}
// What the tail needs and no workgroup of this launch writes — the trust-region scalars, this lane's parameter blocks with their scale
// / Cauchy / Gauss-Newton entries — is loaded BEFORE the workgroup arrives (by every workgroup: any of them may be the last), so
// that after the arrival only the tiles' dogleg shares are fetched: the tail was five dependent memory round trips longer without.
// The arithmetic is step_body's and candidate_dense's, operand for operand.
struct BlockPre {       // one parameter block of a lane
  int off, am, gs, ls;
  bool on, is_free;
  double sp[9], vp[9], yp[9], X[9];
  unsigned char mask[6];
  bool has_mask;
};
template <int MAXS>
__device__ __forceinline__ void block_preload(const BatchDev &d, const WinDesc &ds, const int w, const double *X, const int b, BlockPre &q) {
  q.on = b < GFBE_BLK_COUNT;
  const int bb = q.on ? b : 0;
  q.off = blk_tan(bb); q.am = blk_amb(bb); q.gs = q.on ? blk_gsize(bb) : 0; q.ls = q.on ? blk_lsize(bb) : 0;
  q.is_free = q.on && ds.blk_free[bb];
  q.has_mask = (bb == GFBE_BLK_EX_CAM || bb == GFBE_BLK_EX_WHEEL);
#pragma unroll
  for (int k = 0; k < 6; k++) q.mask[k] = q.has_mask ? (bb == GFBE_BLK_EX_CAM ? ds.ex_cam_mask[k] : ds.ex_wheel_mask[k]) : 0;
#pragma unroll
  for (int k = 0; k < MAXS; k++) {
    const size_t a = (size_t)w * ND + q.off + min(k, max(q.ls - 1, 0));
    const bool in = q.is_free && k < q.ls;
    q.sp[k] = in ? d.sp[a] : 0.0; q.vp[k] = in ? d.vp[a] : 0.0; q.yp[k] = in ? d.yp[a] : 0.0;
    q.X[k] = k < q.gs ? X[q.am + k] : 0.0;
  }
}
// x_cand of one block from its preloaded entries (candidate_dense's loop body); Yl: the candidate block
template <int MAXS>
__device__ __forceinline__ void block_candidate(const BatchDev &d, const int w, const BlockPre &q, const double c1, const double c2, double *Y, double (&Yl)[9],
                                                double &d2, double &n2) {
  if (!q.on) return;
  if (q.is_free) {
    double dl[9];
#pragma unroll
    for (int k = 0; k < MAXS; k++) {
      dl[k] = q.sp[k] * (c1 * q.vp[k] + c2 * q.yp[k]);
      if (k < q.ls) d.step[(size_t)w * ND + q.off + k] = dl[k];
    }
    if (MAXS >= 7 && q.gs == 7) {
      pose_plus(q.X, dl, q.has_mask ? q.mask : nullptr, Yl);
    } else if (MAXS >= 4 && q.gs == 4) {   // para_plane_R: OrientationSubsetParameterization({2}) (estimator.cpp:3122)
      const unsigned char constant[3] = {0, 0, 1};
      orientation_plus(q.X, dl, constant, Yl);
    } else {
#pragma unroll
      for (int k = 0; k < MAXS; k++) Yl[k] = q.X[k] + dl[k];
    }
#pragma unroll
    for (int k = 0; k < MAXS; k++) if (k < q.gs) { const double df = q.X[k] - Yl[k]; d2 += df * df; n2 += Yl[k] * Yl[k]; }
  } else {
#pragma unroll
    for (int k = 0; k < MAXS; k++) Yl[k] = q.X[k];
  }
#pragma unroll
  for (int k = 0; k < MAXS; k++) if (k < q.gs) Y[q.am + k] = Yl[k];
}
#ifndef GFBE_STEP_CAND_REGS
#define GFBE_STEP_CAND_REGS 1      // k_step_candidate_dense on register copies, like the tail of k_lm_step_fused (0: step_body on WinCtl in memory + candidate_dense)
#endif
__global__ __launch_bounds__(LM_TILE) void k_step_candidate_dense(BatchDev d) {
  GFBE_SMALL_KERNEL_PRIO();
  const int w = blockIdx.x;
  const WinDesc &ds = d.desc[w];
#if GFBE_STEP_CAND_REGS
  // (end of round 6) The form above walks memory one dependent access at a time: step_body reads and writes the window's trust-region
  // scalars in WinCtl field by field, candidate_dense's loop over a block's dims loads sp / vp / yp and stores the step entry per
  // iteration (the store may alias the next loads: load, wait, store, nine times for a speed-bias block), re-reads the candidate it has
  // just written for |dx|^2 and |x|^2, and the pair constants read it a third time: 28 us per launch over 512 windows. Here — the tail of
  // k_lm_step_fused, operand for operand: StepLocal, block_preload, block_candidate, pair_consts_from_staged — everything the wave needs
  // is requested at the start in one round, the scalars live in lane 0's registers, a lane's block in its own, the poses go to the pair
  // constants through LDS. Same operations in the same order.
  WinCtl &c = d.ctl[w];
  const int t = threadIdx.x;
  __shared__ PoseRT sp_cand[NF + 1];
  const int cur = c.cur;
  StepLocal lc;
  lc.done = c.done; lc.have_step = c.have_step; lc.reuse = c.reuse; lc.iter = c.iter; lc.termination = c.termination; lc.status = c.status;
  lc.invalid_steps = c.invalid_steps;
  lc.G2 = c.G2; lc.N2 = c.N2; lc.gy = c.gy; lc.vHv = c.vHv; lc.vHy = c.vHy; lc.yHy = c.yHy; lc.grad_max = c.grad_max; lc.x_norm = c.x_norm;
  lc.alpha = c.alpha; lc.radius = c.radius; lc.c1 = c.c1; lc.c2 = c.c2; lc.step_norm = c.step_norm; lc.model_change = c.model_change;
  lc.cost = c.cost; lc.mu = c.mu; lc.t_start = c.t_start;
  const double *X = d.x + ((size_t)w * 2 + cur) * NA;
  double *Y = d.x + ((size_t)w * 2 + 1 - cur) * NA;
  BlockPre q0, q1;
  block_preload<9>(d, ds, w, X, t, q0);
  block_preload<1>(d, ds, w, X, t + 64, q1);
  step_body(d, ds, lc, c, w, t);
  if (t == 0) {
    c.done = lc.done; c.have_step = lc.have_step; c.reuse = lc.reuse; c.iter = lc.iter; c.termination = lc.termination; c.status = lc.status;
    c.invalid_steps = lc.invalid_steps;
    c.G2 = lc.G2; c.N2 = lc.N2; c.gy = lc.gy; c.vHv = lc.vHv; c.vHy = lc.vHy; c.yHy = lc.yHy; c.grad_max = lc.grad_max; c.x_norm = lc.x_norm;
    c.alpha = lc.alpha; c.c1 = lc.c1; c.c2 = lc.c2; c.step_norm = lc.step_norm; c.model_change = lc.model_change; c.mu = lc.mu;
  }
  const int go = __shfl((lc.done || !lc.have_step) ? 0 : 1, 0, 64);       // (lane 0 ran the scalar logic)
  if (!go) return;
  const double c1 = __shfl(lc.c1, 0, 64), c2 = __shfl(lc.c2, 0, 64);
  double d2 = 0.0, n2 = 0.0, Yl0[9], Yl1[9];
  block_candidate<9>(d, w, q0, c1, c2, Y, Yl0, d2, n2);
  block_candidate<1>(d, w, q1, c1, c2, Y, Yl1, d2, n2);
  d2 = wave_sum(d2); n2 = wave_sum(n2);
  if (t == 0) { d.dense_cand[(size_t)w * 4 + 1] = d2; d.dense_cand[(size_t)w * 4 + 2] = n2; }
  if (t < NF) sp_cand[t] = make_pose(Yl0);
  if (t == GFBE_BLK_EX_CAM) sp_cand[NF] = make_pose(Yl0);
  __syncthreads();
  pair_consts_from_staged(d.pc + ((size_t)w * 3 + (1 - cur)) * NPAIR * PC_DOUBLES, sp_cand, t);
#else
  step_body(d, ds, d.ctl[w], d.ctl[w], w, threadIdx.x);
  __threadfence_block();      // (one wave, one CU, one vector cache: workgroup scope — a device-scope fence writes the XCD's L2 back, measured -4 % end to end)
  __builtin_amdgcn_wave_barrier();
  const WinCtl &c = d.ctl[w];
  if (c.done || !c.have_step) return;
  __shared__ PoseRT sp_cand[NF + 1];
  candidate_dense(d, ds, c, w, threadIdx.x, sp_cand, true);
#endif
}
#ifndef GFBE_LMS_AHEAD
#define GFBE_LMS_AHEAD 3      // k_lm_step_fused: rows of the observation steps in flight ahead of the one being multiplied (lm_step_tile)
#endif
#ifndef GFBE_LMS_STAMP
#define GFBE_LMS_STAMP 0      // diagnostics build: phase stamps of k_lm_step_fused (tools/diag_scripts/lms_stamps.py)
#endif
__global__ __launch_bounds__(LM_TILE) void k_lm_step_fused(BatchDev d0) {
  static_assert(GFBE_BLK_COUNT <= 128 && GFBE_BLK_RCV_DT0 <= 64, "two blocks per lane, the second one a scalar block");
  const int w = blockIdx.x, tile = blockIdx.y;   // tile-major dispatch (longest tracks first)
#if GFBE_LMS_STAMP
  unsigned long long *ms = (unsigned long long *)(d0.timing + (size_t)d0.B * 32);
#define MSTAMP_T(i) do { if (w == 0 && tile == 0 && threadIdx.x == 0) ms[i] = wall_clock64(); } while (0)
#define MSTAMP_L(i) do { if (w == 0 && threadIdx.x == 0) ms[i] = wall_clock64(); } while (0)
#else
#define MSTAMP_T(i) do { } while (0)
#define MSTAMP_L(i) do { } while (0)
#endif
  MSTAMP_T(0);
  const BatchDev d = lin_view(d0, d0.ctl[w].lb);
  const WinDesc &ds = d.desc[w];
  WinCtl &c = d.ctl[w];
  const int t = threadIdx.x;
  __shared__ double sy[NV], sv[NV];
  __shared__ PoseRT sp_cand[NF + 1];
  const int cur = c.cur;
  if (tile < ds.n_tiles && !(c.done || c.reuse)) {
    for (int a = t; a < NV; a += LM_TILE) {
      const double s = d.sp[(size_t)w * ND + a];
      sy[a] = s * d.yp[(size_t)w * ND + a];
      sv[a] = s * d.vp[(size_t)w * ND + a];
    }
    __syncthreads();
    MSTAMP_T(1);
    __shared__ double fsteps[NF * LM_FS];
    const bool comp = !d.vis_full;      // (workgroup-uniform: the barrier below is safe)
    if (comp) { stage_frame_steps(d, c, w, t, sy, sv, fsteps); __syncthreads(); }
    MSTAMP_T(2);
#if GFBE_LMS_AHEAD > 0
    if (comp) lm_step_tile_ahead<GFBE_LMS_AHEAD>(d, ds, c, w, tile, t, fsteps);
    else
#endif
    lm_step_tile(d, ds, c, w, tile, t, sy, sv, comp ? fsteps : nullptr);
    MSTAMP_T(3);
  }
  // ---- preloads of the tail (behind the landmarks' work: requested in front of it — round 5, tools/diag_scripts/lms_stamps.py — their two
  //      dependent levels delayed the tile's own loads, the counters being in order: 17.4 -> 18.7 us per launch)
  StepLocal lc;
  lc.done = c.done; lc.have_step = c.have_step; lc.reuse = c.reuse; lc.iter = c.iter; lc.termination = c.termination; lc.status = c.status;
  lc.invalid_steps = c.invalid_steps;
  lc.G2 = c.G2; lc.N2 = c.N2; lc.gy = c.gy; lc.vHv = c.vHv; lc.vHy = c.vHy; lc.yHy = c.yHy; lc.grad_max = c.grad_max; lc.x_norm = c.x_norm;
  lc.alpha = c.alpha; lc.radius = c.radius; lc.c1 = c.c1; lc.c2 = c.c2; lc.step_norm = c.step_norm; lc.model_change = c.model_change;
  lc.cost = c.cost; lc.mu = c.mu; lc.t_start = c.t_start;
  const double *X = d.x + ((size_t)w * 2 + cur) * NA;
  double *Y = d.x + ((size_t)w * 2 + 1 - cur) * NA;
  BlockPre q0, q1;
  block_preload<9>(d, ds, w, X, t, q0);
  block_preload<1>(d, ds, w, X, t + 64, q1);
  MSTAMP_T(4);
  if (!arrive_last(d.win_cnt + 2 * w, gridDim.y, t)) return;
  MSTAMP_L(5);
  // ---- k_step
  step_body(d, ds, lc, c, w, t);
  MSTAMP_L(6);
  if (t == 0) {
    c.done = lc.done; c.have_step = lc.have_step; c.reuse = lc.reuse; c.iter = lc.iter; c.termination = lc.termination; c.status = lc.status;
    c.invalid_steps = lc.invalid_steps;
    c.G2 = lc.G2; c.N2 = lc.N2; c.gy = lc.gy; c.vHv = lc.vHv; c.vHy = lc.vHy; c.yHy = lc.yHy; c.grad_max = lc.grad_max; c.x_norm = lc.x_norm;
    c.alpha = lc.alpha; c.c1 = lc.c1; c.c2 = lc.c2; c.step_norm = lc.step_norm; c.model_change = lc.model_change; c.mu = lc.mu;
  }
  const int go = __shfl((lc.done || !lc.have_step) ? 0 : 1, 0, 64);       // (lane 0 ran the scalar logic)
  if (!go) return;
  const double c1 = __shfl(lc.c1, 0, 64), c2 = __shfl(lc.c2, 0, 64);
  // ---- the dense parameter blocks of k_candidate
  double d2 = 0.0, n2 = 0.0, Yl0[9], Yl1[9];
  block_candidate<9>(d, w, q0, c1, c2, Y, Yl0, d2, n2);
  block_candidate<1>(d, w, q1, c1, c2, Y, Yl1, d2, n2);
  d2 = wave_sum(d2); n2 = wave_sum(n2);
  if (t == 0) { d.dense_cand[(size_t)w * 4 + 1] = d2; d.dense_cand[(size_t)w * 4 + 2] = n2; }
  // the candidate's pose-pair constants: the poses straight from the lanes that formed them (block id = lane: Pose[t]; the camera extrinsic)
  if (t < NF) sp_cand[t] = make_pose(Yl0);
  if (t == GFBE_BLK_EX_CAM) sp_cand[NF] = make_pose(Yl0);
  __syncthreads();
  MSTAMP_L(7);
  pair_consts_from_staged(d.pc + ((size_t)w * 3 + (1 - cur)) * NPAIR * PC_DOUBLES, sp_cand, t);
  MSTAMP_L(8);
#undef MSTAMP_T
#undef MSTAMP_L
}

// =============================================================================================
// k_accept: candidate cost, tolerances, step acceptance, radius / mu update
// (TrustRegionMinimizer::{ParameterToleranceReached,FunctionToleranceReached,IsStepSuccessful,
//  HandleSuccessfulStep,HandleUnsuccessfulStep}, DoglegStrategy::{StepAccepted,StepRejected}).
// =============================================================================================
// c: the window's trust-region scalars — WinCtl itself or a register copy (AcceptLocal: the fused tail of k_lin_small<1> loads it
// before its workgroup arrives); cg: WinCtl in memory, for the per-iteration records (stores only).
// cslot: where the dense factors' candidate costs are — 1: the slot the cost pass fills; 2: the cost slot of a linearisation (the
// speculative pass: d is the view of the set it wrote, and an accepted step makes that set the current one).
template <class CT>
__device__ __forceinline__ void accept_body(const BatchDev &d, const int w, const int lane, CT &c, WinCtl &cg, const int cslot) {
  const WinDesc &ds = d.desc[w];
  if (c.done || !c.have_step) return;
  double cand = 0.0, d2 = 0.0, n2 = 0.0;
  // the dense factors' candidate costs: a lane holds at most ONE of them (lanes 0.. the inertial factors, 16.. the wheel factors, 32 the prior and
  // the dense blocks' |dx|^2, |x|^2, 40.. the LiDAR workgroups, 48.. the plane factors, 58 the anchor, 59 GNSS) — its address is selected and the
  // value requested BEFORE the tiles' sums (end of round 6: a conditional load per kind was seven dependent round trips of one wave), and added
  // behind them as before: the same sums in the same order
  static_assert(MAX_IMU <= 16 && MAX_WHEEL <= 16 && LIOW_WGS <= 8 && MAX_PLANE <= 10, "accept_body: one dense term per lane");
  const double *tp = nullptr;
  if (lane < ds.n_imu) tp = d.imu_part + ((size_t)w * MAX_IMU + lane) * IMU_PART + IMU_PART - cslot;
  else if (lane >= 16 && lane - 16 < ds.n_wheel) tp = d.wheel_part + ((size_t)w * MAX_WHEEL + lane - 16) * WHEEL_PART + WHEEL_PART - cslot;
  else if (lane == 32) tp = d.prior_g + (size_t)w * (ND + 2) + ND + 2 - cslot;
  else if (ds.lio_n > 0 && lane >= 40 && lane < 40 + LIOW_WGS) tp = d.lio_part + ((size_t)w * LIOW_WGS + lane - 40) * LIOW_PART + 29 - cslot;      // (28: the cost pass's slot; 27: a linearisation's)
  else if (lane >= 48 && lane - 48 < ds.n_plane) tp = d.plane_part + ((size_t)w * MAX_PLANE + lane - 48) * PLANE_PART + PLANE_PART - cslot;
  else if (lane == 58 && ds.use_anchor) tp = d.anchor_part + (size_t)w * ANCHOR_PART + ANCHOR_PART - cslot;
  else if (lane == 59 && ds.gnss_factors) tp = d.gnss_cost + (size_t)w * 2 + 2 - cslot;
  const bool has_term = tp != nullptr;
  const double term = *(has_term ? tp : d.zero);
  const double dc1 = d.dense_cand[(size_t)w * 4 + 1], dc2 = d.dense_cand[(size_t)w * 4 + 2];
  if (!d.sharded) tile_cand_sum(d, ds, w, lane, cand, d2, n2);
  else if (lane < d.world) {   // landmark sharding: the ranks' sums (k_xchg_cand + all-reduce)
    const double *xr = d.xc + ((size_t)w * d.world + lane) * XCHG;
    cand = xr[0]; d2 = xr[1]; n2 = xr[2];
  }
  if (has_term) cand += term;
  if (lane == 32) { d2 += dc1; n2 += dc2; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { cand += __shfl_xor(cand, o, 64); d2 += __shfl_xor(d2, o, 64); n2 += __shfl_xor(n2, o, 64); }
  if (lane != 0) return;
  if (!isfinite(cand)) cand = 1.7976931348623157e308;
  const int it = c.iter;
  c.cand_cost = cand;
  cg.cost_history[it] = c.cost;
  const double step_amb = sqrt(d2);
  if (step_amb <= d.opt.parameter_tolerance * (c.x_norm + d.opt.parameter_tolerance)) {
    c.done = 1; c.termination = 2; c.status = GFBE_OK; c.have_step = 0; return;
  }
  const double cost_change = c.cost - cand;
  if (fabs(cost_change) <= d.opt.function_tolerance * c.cost) {
    c.done = 1; c.termination = 1; c.status = GFBE_OK; c.have_step = 0; return;
  }
  const double quality = cost_change / c.model_change;
  if (quality > d.opt.min_relative_decrease) {
    c.cur = 1 - c.cur;
    c.cost = cand;
    c.x_norm = sqrt(n2);
    cg.accepted[it] = 1; c.num_successful++;
    cg.cost_history[it] = cand;
    if (quality < 0.25) c.radius *= 0.5;
    if (quality > 0.75) c.radius = fmax(c.radius, 3.0 * c.step_norm);
    c.mu = mu_after_accept(c.mu);
    c.reuse = 0;
    if (cslot == 2) c.lb = 1 - c.lb;
  } else {
    cg.accepted[it] = 0;
    c.radius *= 0.5;
    c.reuse = 1;
  }
  c.have_step = 0;
}
__global__ __launch_bounds__(64) void k_accept(BatchDev d, int spec) {      // spec: after the candidate's linearisation (its costs, its set of outputs)
  GFBE_SMALL_KERNEL_PRIO();
  const int w = blockIdx.x;
  if (spec) accept_body(lin_view(d, 1 - d.ctl[w].lb), w, threadIdx.x, d.ctl[w], d.ctl[w], 2);
  else accept_body(d, w, threadIdx.x, d.ctl[w], d.ctl[w], 1);
}

// =============================================================================================
// k_reanchor: double2vector()'s yaw / position gauge fix followed by vector2double()
// (estimator.cpp:2501-2555, 2341-2362): the result is what optimization() leaves in para_*.
// =============================================================================================
__global__ __launch_bounds__(64) void k_reanchor(BatchDev d) {
  const int w = blockIdx.x;
  WinCtl &c = d.ctl[w];
  if (threadIdx.x == 0) c.t_solved = (long long)wall_clock64();
  const double *X0 = d.x0 + (size_t)w * NA;
  const double *X = d.x + ((size_t)w * 2 + c.cur) * NA;
  double *Y = d.xout + (size_t)w * NA;
  const int t = threadIdx.x;
  for (int q = t; q < NA; q += 64) Y[q] = X[q];
  __syncthreads();
  const mat3 R0 = qrot(ldq(X0 + A_POSE(0) + 3));
  const mat3 R00 = qrot(ldq(X + A_POSE(0) + 3));
  const vec3 o0 = rot_to_ypr_deg(R0), o00 = rot_to_ypr_deg(R00);
  mat3 rd = yaw_rot_deg(o0[0] - o00[0]);
  if (fabs(fabs(o0[1]) - 90.0) < 1.0 || fabs(fabs(o00[1]) - 90.0) < 1.0) rd = mul(R0, transp(R00));
  if (t < NF) {
    const mat3 Ri = mul(rd, qrot(qnormalize(ldq(X + A_POSE(t) + 3))));
    const vec3 Pi = add(mv(rd, sub(ld3(X + A_POSE(t)), ld3(X + A_POSE(0)))), ld3(X0 + A_POSE(0)));
    const vec3 Vi = mv(rd, ld3(X + A_SB(t)));
    const quat q = rot2quat(Ri);
    double *p = Y + A_POSE(t);
    p[0] = Pi[0]; p[1] = Pi[1]; p[2] = Pi[2]; p[3] = q.x; p[4] = q.y; p[5] = q.z; p[6] = q.w;
    double *v = Y + A_SB(t);
    v[0] = Vi[0]; v[1] = Vi[1]; v[2] = Vi[2];
  } else if (t == NF) {
    const quat q = rot2quat(qrot(ldq(X + A_EX + 3)));
    Y[A_EX + 3] = q.x; Y[A_EX + 4] = q.y; Y[A_EX + 5] = q.z; Y[A_EX + 6] = q.w;
  } else if (t == NF + 1) {
    const quat q = rot2quat(qrot(qnormalize(ldq(X + A_EXW + 3))));
    Y[A_EXW + 3] = q.x; Y[A_EXW + 4] = q.y; Y[A_EXW + 5] = q.z; Y[A_EXW + 6] = q.w;
  } else if (t == NF + 2) {   // estimator.cpp:3383-3386: para_yaw_enu_local back into (-pi, pi]
    double yaw = X[A_YAW];
    if (isfinite(yaw) && fabs(yaw) < 1e6) {   // (the reference's loops would not terminate on a non-finite or absurd value: left alone)
      while (yaw > 3.14159265358979323846) yaw -= 2.0 * 3.14159265358979323846;
      while (yaw < -3.14159265358979323846) yaw += 2.0 * 3.14159265358979323846;
    }
    Y[A_YAW] = yaw;
  }
  // pose-pair constants of the re-anchored state: the marginalisation linearises its visual factors there
  __shared__ PoseRT sp_anch[NF + 1];
  __threadfence_block();
  __syncthreads();
  pair_consts_of_state(Y, d.pc + ((size_t)w * 3 + 2) * NPAIR * PC_DOUBLES, sp_anch, t);
}

// =============================================================================================
// k_expand (host upload): the factors cross PCIe compactly — fobs[record][5], pair-major record order — and are scattered
// here into the ELL rows lm_obs[k][.][slot] the evaluation kernels read; lm_rec[k][slot] is the record position itself:
// inside a start-frame group the landmarks are sorted longest track first, so the factors of pair (s, s+1+k) are a prefix
// of the group and the record of (slot, k) is pair_begin + position in the group.
// =============================================================================================
__device__ __forceinline__ void expand_body(const BatchDev &d, const int w, const int bx) {
  const WinDesc &ds = d.desc[w];
  const int rel = bx * 256 + threadIdx.x;
  if (rel >= ds.lm_slots) return;
  const int slot = ds.lm_off + rel;
  const int info = d.lm_info[slot];
  if (!((info >> 24) & 1)) return;
  const int s = info & 0xff, m = (info >> 8) & 0xff;
  const int pos = rel - ds.sf_tile_begin[s] * LM_TILE;
  const size_t TL = d.tot_lm;
  // A batch whose windows ALL hold td constant (and the camera extrinsic: !vis_full) stores every observation already shifted to the
  // window's td — p' = p - (td - td_obs) v, what each factor evaluation would compute (projectionTwoFrameOneCamFactor.cpp:60-61), the
  // same fused multiply-add — and td in place of the observation's own td, so that the kernels that still apply the shift (the
  // marginalisation's 20-column panel, gfbe_eval_factors) subtract an exact zero: the linearisation and the candidate-cost pass then
  // read two of an observation's five doubles and three of a landmark's six (k_vis<0, false>, k_vis<1>: both wait for memory).
  const bool shift = !d.vis_full;
  const double tdw = d.x0[(size_t)w * NA + A_TD];
  if (shift) {
    const double dti = tdw - d.lm_pts[5 * TL + slot];
    d.lm_pts[0 * TL + slot] = __builtin_fma(-dti, d.lm_pts[3 * TL + slot], d.lm_pts[0 * TL + slot]);
    d.lm_pts[1 * TL + slot] = __builtin_fma(-dti, d.lm_pts[4 * TL + slot], d.lm_pts[1 * TL + slot]);
    d.lm_pts[5 * TL + slot] = tdw;
  }
  for (int k = 0; k < m; k++) {
    const int rec = ds.pair_begin[s * NF + s + 1 + k] + pos;
    d.lm_rec[(size_t)k * TL + slot] = rec;
    double *ob = d.lm_obs + (size_t)k * 5 * TL + slot;
    if (d.obs_compact) {      // the host shifted the observation (upload_one); velocities only where the marginalisation reads them
      const double *f = d.fobs + ((size_t)ds.rec_off + rec) * 2;
      ob[0] = f[0]; ob[TL] = f[1];
      if (s == 0) {           // (no pointer is formed for the others: a window without frame-0 records has vel_off at the end of fvel)
        const double *v = d.fvel + ((size_t)ds.vel_off + rec) * 3;
        ob[2 * TL] = v[0]; ob[3 * TL] = v[1];
      } else { ob[2 * TL] = 0.0; ob[3 * TL] = 0.0; }
      ob[4 * TL] = tdw;
      continue;
    }
    const double *f = d.fobs + ((size_t)ds.rec_off + rec) * 5;
    const double dtj = shift ? tdw - f[4] : 0.0;
    ob[0] = shift ? __builtin_fma(-dtj, f[2], f[0]) : f[0];
    ob[TL] = shift ? __builtin_fma(-dtj, f[3], f[1]) : f[1];
    ob[2 * TL] = f[2]; ob[3 * TL] = f[3];
    ob[4 * TL] = shift ? tdw : f[4];
  }
}
__global__ __launch_bounds__(256) void k_expand(BatchDev d) { expand_body(d, blockIdx.y, blockIdx.x); }
// Small batches (one window per call is the reference's pattern): the four preparation steps of an upload are independent of each
// other — the landmark rows, the factors' square-root informations, the prior's J0^T J0, the assembly table — and run as ONE launch
// (workgroup ranges), not four one after the other on the latency path of gfbe_solve_window. (Throughput batches keep them apart:
// the factorisations' 256 VGPRs would cap the occupancy of the others.)
__global__ __launch_bounds__(256) void k_upload_small(BatchDev d, int n_expand) {
  int bx = blockIdx.x;
  const int ne = d.B * n_expand, nf = d.B * PREP_FACT_WGS, np = d.B * PREP_PRIOR_WGS;
  if (bx < ne) { expand_body(d, bx / n_expand, bx % n_expand); return; }
  bx -= ne;
  if (bx < nf) { prep_body(d, bx / PREP_FACT_WGS, bx % PREP_FACT_WGS); return; }
  bx -= nf;
  if (bx < np) prep_prior_body(d, bx / PREP_PRIOR_WGS, bx % PREP_PRIOR_WGS);
}
// Small batches (round 5): what an upload enqueued BEFORE k_upload_small as four commands of the copy path — two clears, the host-to-device
// copy of the staged upload region, the spread of the priors' compact J0 rows — in one kernel that READS THE PINNED HOST BUFFER itself
// (444 KB per 2k-landmark window across PCIe by 128 workgroups) and writes the slab: between a copy command and a kernel the stream
// idles ~9 us, between two kernels nothing (gpurun_out timeline of gfbe_solve_window: 83 us before the first kernel of the solve,
// 15 of them the copy itself). src / dst: the staged region (16-byte units); z0 / z1: the ranges to clear; J0 rows: `nrow` rows of
// `row` doubles at `jsrc` (inside the HOST buffer) to a stride of `jstride` doubles at `jdst`.
enum { INGEST_WGS = 128 };
__global__ __launch_bounds__(256) void k_ingest_small(const uint4 *src, uint4 *dst, size_t n16, uint4 *z0, size_t z0n16, uint4 *z1, size_t z1n16,
                                                      const double *jsrc, double *jdst, int nrow, size_t row, size_t jstride) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
  for (size_t i = t; i < n16; i += nt) dst[i] = src[i];
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = t; i < z0n16; i += nt) z0[i] = z;
  for (size_t i = t; i < z1n16; i += nt) z1[i] = z;
  for (int r = 0; r < nrow; r++)
    for (size_t i = t; i < row; i += nt) jdst[(size_t)r * jstride + i] = jsrc[(size_t)r * row + i];
}
void launch_ingest_small(const void *src_host, void *dst, size_t bytes, void *z0, size_t z0_bytes, void *z1, size_t z1_bytes,
                         const double *jsrc_host, double *jdst, int nrow, size_t row, size_t jstride, hipStream_t s) {
  hipLaunchKernelGGL(k_ingest_small, dim3(INGEST_WGS), dim3(256), 0, s, (const uint4 *)src_host, (uint4 *)dst, (bytes + 15) / 16,
                     (uint4 *)z0, z0_bytes / 16, (uint4 *)z1, z1_bytes / 16, jsrc_host, jdst, nrow, row, jstride);
}
void launch_upload_small(const BatchDev &d, int with_expand, hipStream_t s) {
  const int n_expand = (with_expand && d.max_tiles > 0) ? (d.max_tiles * LM_TILE + 255) / 256 : 0;
  const int total = d.B * (n_expand + PREP_FACT_WGS + PREP_PRIOR_WGS);
  hipLaunchKernelGGL(k_upload_small, dim3(total), dim3(256), 0, s, d, n_expand);
}

// =============================================================================================
// k_gather: everything gfbe_batch_download hands back, packed for ONE device-to-host copy:
// WinCtl | re-anchored state | new prior's block table, x0, r0 per window at a fixed stride; para_Feature in ABI order
// (scatter through lm_abi); the new prior's J0 (n x n) at host-known offsets.
// =============================================================================================
enum { GATHER_WGS = 8 };
__global__ __launch_bounds__(256) void k_gather(BatchDev d, int margin_flag) {
  const int w = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x, nt = GATHER_WGS * 256;
  const WinDesc &ds = d.desc[w];
  const WinCtl &c = d.ctl[w];
  double *fix = d.dl_fix + (size_t)w * DL_FIX;
  const int *meta = d.mmeta + (size_t)w * (4 + 3 * GFBE_MAX_PRIOR_BLOCKS);
  const bool marg = margin_flag != GFBE_MARGIN_NONE && c.marg_ran;
  {
    const double *src = (const double *)&c;
    for (int q = t; q < (int)(sizeof(WinCtl) / 8); q += nt) fix[q] = src[q];
    for (int q = t; q < NA; q += nt) fix[DL_OFF_X + q] = d.xout[(size_t)w * NA + q];
    int *mi = (int *)(fix + DL_OFF_META);
    for (int q = t; q < 4 + 3 * GFBE_MAX_PRIOR_BLOCKS; q += nt) mi[q] = marg ? meta[q] : 0;
    if (marg && meta[0] == 1) {
      for (int q = t; q < PRIOR_X0; q += nt) fix[DL_OFF_X0 + q] = d.mx0[(size_t)w * PRIOR_X0 + q];
      for (int q = t; q < meta[1]; q += nt) fix[DL_OFF_R0 + q] = d.mr0[(size_t)w * ND + q];
    }
  }
  {
    const size_t TL = d.tot_lm;
    double *feat = d.dl_feat + d.dl_feat_off[w];
    for (int q = t; q < ds.lm_slots; q += nt) {
      const int slot = ds.lm_off + q, abi = d.lm_abi[slot];
      if (abi >= 0) feat[abi] = d.lam[(size_t)c.cur * TL + slot];
    }
  }
  if (marg && meta[0] == 1) {
    const long long n2 = (long long)meta[1] * meta[1], cap = d.dl_j0_off[w + 1] - d.dl_j0_off[w];
    if (n2 <= cap) {
      const double *J = d.mJ0 + (size_t)w * ND * ND;
      double *o = d.dl_J0 + d.dl_j0_off[w];
      for (long long q = t; q < n2; q += nt) o[q] = J[q];
    } else if (t == 0) {
      ((int *)(fix + DL_OFF_META))[0] = -2;   // (never: the host's bound of n comes from the same block tables)
    }
  }
}

// =============================================================================================
// launchers
// =============================================================================================

void launch_expand(const BatchDev &d, hipStream_t s) {
  if (d.max_tiles > 0) hipLaunchKernelGGL(k_expand, dim3((d.max_tiles * LM_TILE + 255) / 256, d.B), dim3(256), 0, s, d);
}
void launch_gather(const BatchDev &d, int margin_flag, hipStream_t s) {
  hipLaunchKernelGGL(k_gather, dim3(GATHER_WGS, d.B), dim3(256), 0, s, d, margin_flag);
}
void launch_prep(const BatchDev &d, hipStream_t s) {
  hipLaunchKernelGGL(k_prep, dim3(d.B, PREP_FACT_WGS), dim3(256), 0, s, d);
  hipLaunchKernelGGL(k_prep_prior, dim3(d.B, PREP_PRIOR_WGS), dim3(256), 0, s, d);
}
void launch_reset(const BatchDev &d, hipStream_t s) {
  const int slots = d.max_tiles * LM_TILE;
  hipLaunchKernelGGL(k_reset, dim3((slots + 255) / 256 > 0 ? (slots + 255) / 256 : 1, d.B), dim3(256), 0, s, d);
}
void launch_vis(const BatchDev &d, int mode, hipStream_t s, int write_records, int spec) {
  if (d.max_tiles == 0) return;
  const dim3 g(d.B, d.max_tiles), b(LM_TILE);
  if (GFBE_VIS_CHUNK && mode == 0 && !d.vis_full && !write_records && d.B >= DENSE_SPLIT_MIN_B && d.world == 1) {
    // throughput batches on the 7 x 7 panel: a wave per (window, start frame, chunk of VIS_CHUNK tiles) — k_vis_chunk
    const int nsub = (d.max_sf_tiles + VIS_CHUNK - 1) / VIS_CHUNK;
    const dim3 gc(d.B, NF * nsub);
    if (spec) hipLaunchKernelGGL(k_vis_chunk<true>, gc, b, 0, s, d, GFBE_FUSE_CAND ? 1 : 0, nsub);
    else hipLaunchKernelGGL(k_vis_chunk<false>, gc, b, 0, s, d, 0, nsub);
    return;
  }
  if (mode == 0 && spec) {      // (the candidate linearised; its tiles form the candidate inverse depths first, as the cost pass's do)
    const int head = (d.B >= DENSE_SPLIT_MIN_B && GFBE_FUSE_CAND) ? 1 : 0;
    if (d.vis_full) hipLaunchKernelGGL((k_vis<0, true, true>), g, b, 0, s, d, head);
    else hipLaunchKernelGGL((k_vis<0, false, true>), g, b, 0, s, d, head);
    return;
  }
  // (reduced panel: only when the camera extrinsic and td are constant in EVERY window of the batch and no records are asked for)
  if (mode == 0 && (d.vis_full || write_records)) hipLaunchKernelGGL((k_vis<0, true>), g, b, 0, s, d, write_records);
  else if (mode == 0) hipLaunchKernelGGL((k_vis<0, false>), g, b, 0, s, d, 0);
  else if (mode == 1) hipLaunchKernelGGL((k_vis<1, true>), g, b, 0, s, d, (d.B >= DENSE_SPLIT_MIN_B && GFBE_FUSE_CAND) ? 1 : 0);   // (1: the tiles form their candidate inverse depths first)
  else if (d.B < DENSE_SPLIT_MIN_B) hipLaunchKernelGGL((k_vis_split<2, true>), dim3(d.B, d.max_tiles * LIN_SMALL_KS), b, 0, s, d, write_records);
  else hipLaunchKernelGGL((k_vis<2, true>), g, b, 0, s, d, write_records);
}
// dynamic LDS of a k_lin_small workgroup: the KS panels of a tile ([J | r] rows, 21 doubles wide; 17 for the 7 x 7 form) or the prior's chunk of J0
static size_t lin_small_lds(bool split, bool full) {
  const size_t panels = split ? (size_t)LIN_SMALL_KS * LM_TILE * (full ? (size_t)XS_LD : 17) : 0;
  return sizeof(double) * std::max(panels, (size_t)PRIOR_CHUNK);
}
bool lin_small_takes_gnss(const BatchDev &d, int mode) { return d.any_gnss && (mode == 1 || mode == 3) && d.gnss_max_obs <= (int)GN_ITEM_MAX_OBS; }
void launch_lin_small(const BatchDev &d, int mode, hipStream_t s, int fuse) {
  // (+ one workgroup per window for the GNSS factors at the candidate: gnss_candidate_item)
  const dim3 g(d.B, d.max_tiles + MAX_IMU + MAX_WHEEL + 1 + (d.any_plane ? MAX_PLANE + 1 : 0) + (lin_small_takes_gnss(d, mode) ? 1 : 0)), b(LIN_SMALL_THREADS);
  if (mode == 2) hipLaunchKernelGGL((k_lin_small<2, true>), g, b, lin_small_lds(true, true), s, d, 0);   // (the marginalisation set: k_vis_split<2> + k_dense mode 2 in one launch)
  else if (mode == 0 && d.vis_full) hipLaunchKernelGGL((k_lin_small<0, true>), g, b, lin_small_lds(true, true), s, d, 0);
  else if (mode == 0) hipLaunchKernelGGL((k_lin_small<0, false>), g, b, lin_small_lds(true, false), s, d, 0);
  else if (mode == 3 && d.vis_full) hipLaunchKernelGGL((k_lin_small<3, true>), g, b, lin_small_lds(true, true), s, d, fuse);   // (speculative: the candidate linearised)
  else if (mode == 3) hipLaunchKernelGGL((k_lin_small<3, false>), g, b, lin_small_lds(true, false), s, d, fuse);
  else hipLaunchKernelGGL((k_lin_small<1, true>), g, b, lin_small_lds(false, true), s, d, fuse);
}
hipError_t lin_small_init_device() {   // per device, from gfbe_create (see kernels_init_device)
  hipError_t e = hipSuccess;
  auto set = [&](const void *k, size_t bytes) { if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); };
  set((const void *)k_lin_small<2, true>, lin_small_lds(true, true));
  set((const void *)k_lin_small<0, true>, lin_small_lds(true, true));
  set((const void *)k_lin_small<0, false>, lin_small_lds(true, false));
  set((const void *)k_lin_small<3, true>, lin_small_lds(true, true));
  set((const void *)k_lin_small<3, false>, lin_small_lds(true, false));
  set((const void *)k_lin_small<1, true>, lin_small_lds(false, true));
  return e;
}
void launch_pair_schur_marg(const BatchDev &d, hipStream_t s) {
  if (d.max_tiles == 0) { launch_pair(d, 1, s); return; }
  hipLaunchKernelGGL(k_pairsum_schur_marg, dim3(d.B, NF), dim3(VP_STRIDE), 0, s, d);
}
void launch_pair(const BatchDev &d, int marg, hipStream_t s) {
  hipLaunchKernelGGL(k_pairsum, dim3(marg ? NF - 1 : NF * (NF - 1) / 2, d.B), dim3(VP_STRIDE), 0, s, d, marg);
}
void launch_dense_factors(const BatchDev &d, int mode, int debug_out, hipStream_t s, int spec) {
  const int nf = MAX_IMU + MAX_WHEEL + 1 + (d.any_plane ? MAX_PLANE + 1 : 0);   // (+ PlaneFactors and the PoseAnchorFactor)
  if (d.B < DENSE_SPLIT_MIN_B) {
    hipLaunchKernelGGL(k_dense<true>, dim3(nf, d.B), dim3(64), 0, s, d, mode, debug_out, 0, spec);
    return;
  }
  if (mode != 3) hipLaunchKernelGGL(k_dense_raw, dim3(MAX_IMU + MAX_WHEEL, (d.B + 63) / 64), dim3(64), 0, s, d, mode, spec);
  if (GFBE_DENSE_TP && mode <= 1 && !debug_out) {
    // the linearisation and the candidate cost of a throughput batch: factor slots of four windows per workgroup + the priors
    hipLaunchKernelGGL(k_dense_tp, dim3(DTP_PRIOR0, (d.B + 3) / 4), dim3(256), 0, s, d, mode, spec);
    const int lds_n = d.prior_n_max <= PRIOR_LDS_N ? d.prior_n_max : 0;
    hipLaunchKernelGGL(k_prior_tp, dim3(d.B), dim3(256), prior_tp_lds(lds_n), s, d, mode, lds_n, spec);
    if (d.any_plane) hipLaunchKernelGGL(k_dense<false>, dim3(MAX_PLANE + 1, d.B), dim3(64), 0, s, d, mode, 0, DTP_PRIOR0 + 1, spec);   // PlaneFactors, PoseAnchorFactor
    return;
  }
  hipLaunchKernelGGL(k_dense<false>, dim3(nf, d.B), dim3(64), 0, s, d, mode, debug_out, 0, spec);
}
void launch_schur(const BatchDev &d, int marg, hipStream_t s, int with_visblock) {
  if (d.max_tiles == 0) return;
  if (with_visblock && !marg && d.vis_Hs) hipLaunchKernelGGL(k_schur_visblock_small, dim3(d.B, d.schur_groups + (d.vis_full ? (int)VS_BLOCKS : (int)NF)), dim3(VB_GROUP), 0, s, d);
  else hipLaunchKernelGGL(k_schur, dim3(d.B, marg ? 1 : d.schur_groups), dim3(256), 0, s, d, marg);
}
void launch_linschur(const BatchDev &d, int spec, int gate_mu, hipStream_t s) {
  if (d.max_tiles == 0) return;
  const dim3 g(d.B, SCHUR_GROUPS), b(256);
  const int head = GFBE_FUSE_CAND ? 1 : 0;      // (the candidate's pass: its tiles form the candidate inverse depths first, as the cost pass's do)
  if (spec) hipLaunchKernelGGL(k_linschur<true>, g, b, 0, s, d, head, 0);
  else hipLaunchKernelGGL(k_linschur<false>, g, b, 0, s, d, 0, gate_mu);
}
void launch_xchg_gram(const BatchDev &d, hipStream_t s) { hipLaunchKernelGGL(k_xchg_gram, dim3(d.B), dim3(64), 0, s, d); }
void launch_xchg_cand(const BatchDev &d, hipStream_t s) { hipLaunchKernelGGL(k_xchg_cand, dim3(d.B), dim3(64), 0, s, d); }
void launch_lam_mask(const BatchDev &d, hipStream_t s) {
  if (d.max_tiles > 0) hipLaunchKernelGGL(k_lam_mask, dim3(d.max_tiles, d.B), dim3(LM_TILE), 0, s, d);
}
void launch_lio_window(const BatchDev &d, int mode, hipStream_t s, int spec) {
  if (mode == 0) hipLaunchKernelGGL(k_lio_window<0>, dim3(LIOW_WGS, d.B), dim3(256), 0, s, d, spec);
  else hipLaunchKernelGGL(k_lio_window<1>, dim3(LIOW_WGS, d.B), dim3(256), 0, s, d, 0);
}
void launch_visblock(const BatchDev &d, hipStream_t s) {       // (throughput batches: part of k_visasm, launch_assemble)
  if (d.vis_Hs) hipLaunchKernelGGL(k_visblock_small, dim3(d.vis_full ? (int)VS_BLOCKS : (int)NF, d.B), dim3(VB_GROUP), 0, s, d);
}
void launch_assemble(const BatchDev &d, hipStream_t s) {
  if (d.vis_Hs) {
    const int nH = (d.asm_n + ASM_THREADS - 1) / ASM_THREADS;      // one entry of H per thread
    hipLaunchKernelGGL(k_assemble, dim3(nH + ASM_E_WGS + 1, d.B), dim3(ASM_THREADS), 0, s, d, nH);
  }
  else hipLaunchKernelGGL(k_visasm, dim3(d.B), dim3(VB_GROUP), 0, s, d);
}
// Landmark sharding: the partial reduced system of a window, packed for the all-reduce — the lower triangle of H over the dims
// in use, g, the lower triangle of E, eg and the ranks' cost rows: sys_pack_doubles() per window (20.6k doubles = 164 KB for a
// batch without GNSS blocks, against 60.5k + 5.4k for the square arrays). dir 0: gather into sys_pack; 1: scatter the sums back
// (E into both triangles: the solve kernels read it as a full matrix).
__host__ __device__ inline int sys_pack_doubles(int nu, int world) { return nu * (nu + 1) / 2 + nu + NV * (NV + 1) / 2 + NV + world * XCHG; }
__global__ __launch_bounds__(256) void k_sys_pack(BatchDev d, int dir) {
  const int w = blockIdx.y, nthr = gridDim.x * 256, t0 = blockIdx.x * 256 + threadIdx.x;
  const int nu = d.nu, T = nu * (nu + 1) / 2, TE = NV * (NV + 1) / 2, nx = d.world * XCHG;
  double *P = d.sys_pack + (size_t)w * sys_pack_doubles(nu, d.world);
  double *H = d.H + (size_t)w * ND * ND, *g = d.g + (size_t)w * ND, *E = d.E + (size_t)w * NV * NV, *eg = d.eg + (size_t)w * NV;
  double *xa = d.xa + (size_t)w * nx;
  for (int e = t0; e < T + nu + TE + NV + nx; e += nthr) {
    double *src;
    double *mirror = nullptr;
    if (e < T) { int a, b; tri_decode(e, a, b); src = H + (size_t)a * ND + b; }
    else if (e < T + nu) src = g + (e - T);
    else if (e < T + nu + TE) { int a, b; tri_decode(e - T - nu, a, b); src = E + a * NV + b; mirror = E + b * NV + a; }
    else if (e < T + nu + TE + NV) src = eg + (e - T - nu - TE);
    else src = xa + (e - T - nu - TE - NV);
    if (dir == 0) P[e] = *src;
    else { const double v = P[e]; *src = v; if (mirror) *mirror = v; }
  }
}
void launch_sys_pack(const BatchDev &d, int dir, hipStream_t s) { hipLaunchKernelGGL(k_sys_pack, dim3(8, d.B), dim3(256), 0, s, d, dir); }
size_t sys_pack_doubles_host(int nu, int world) { return (size_t)sys_pack_doubles(nu, world); }
void launch_lm_step(const BatchDev &d, hipStream_t s, int fuse) {
  if (d.max_tiles == 0) return;
  if (fuse) hipLaunchKernelGGL(k_lm_step_fused, dim3(d.B, d.max_tiles), dim3(LM_TILE), 0, s, d);
  else hipLaunchKernelGGL(k_lm_step, dim3(d.B, d.max_tiles), dim3(LM_TILE), 0, s, d);
}
static bool step_candidate_fused(const BatchDev &d) { return GFBE_FUSE_STEP_CAND && LM_TILE == 64 && d.B >= DENSE_SPLIT_MIN_B && GFBE_FUSE_CAND && d.max_tiles > 0; }
void launch_step(const BatchDev &d, hipStream_t s) {
  if (step_candidate_fused(d)) hipLaunchKernelGGL(k_step_candidate_dense, dim3(d.B), dim3(LM_TILE), 0, s, d);      // (launch_candidate has nothing left to do)
  else hipLaunchKernelGGL(k_step, dim3(d.B), dim3(64), 0, s, d);
}
void launch_candidate(const BatchDev &d, hipStream_t s) {
  if (step_candidate_fused(d)) return;
  if (d.B >= DENSE_SPLIT_MIN_B && GFBE_FUSE_CAND && d.max_tiles > 0) hipLaunchKernelGGL(k_candidate_dense, dim3(d.B), dim3(LM_TILE), 0, s, d);
  else if (d.B >= DENSE_SPLIT_MIN_B) hipLaunchKernelGGL(k_candidate_window, dim3(d.B), dim3(CAND_THREADS), 0, s, d);
  else hipLaunchKernelGGL(k_candidate, dim3(d.max_tiles + 1, d.B), dim3(LM_TILE), 0, s, d);
}
void launch_accept(const BatchDev &d, hipStream_t s, int spec) { hipLaunchKernelGGL(k_accept, dim3(d.B), dim3(64), 0, s, d, spec); }
void launch_reanchor(const BatchDev &d, hipStream_t s) { hipLaunchKernelGGL(k_reanchor, dim3(d.B), dim3(64), 0, s, d); }

}  // namespace gfd

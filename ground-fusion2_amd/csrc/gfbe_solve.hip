// gfbe_solve.hip — the dense part of DoglegStrategy::ComputeStep (Ceres 1.14, restated: DESIGN.md section 2) for one window per
// workgroup: Jacobi scaling, the mu-regularised Schur-reduced system, its Cholesky factorisation, the Gauss-Newton step and the
// dense shares of the dogleg scalars. Reference call site: estimator.cpp:3364-3379 (ceres::Solve, DENSE_SCHUR + DOGLEG).
#include "gfbe_devutil.h"

namespace gfd {

// =============================================================================================
// k_solve: one workgroup per window. Jacobi scaling (iteration 0), D = sqrt(clamp(diag)), the
// mu-regularised reduced system, packed Cholesky in LDS, Gauss-Newton step y_p, dense shares of
// the dogleg scalars. (Ceres 1.14 DoglegStrategy::ComputeStep / ComputeGaussNewtonStep.)
// =============================================================================================
// 768 threads = 12 waves = 3 per SIMD: 170 VGPRs per lane instead of the 128 of a 1024-thread workgroup. Measured on one box
// (tests/diag_variants.py, one window / 1024 resident windows): 1024 threads 83.6 us / 48.6k solves/s, 512 threads 83.2 /
// 49.2-49.6k (faster under load: less scratch traffic from the out-of-line phases, but the tile build takes 21 instead of 15 us),
// 768 threads 77.8 / 50.4-50.7k. Inlining the phases back is slower at every size (the back-substitution alone 9 -> 17 us).
#ifndef SOLVE_THREADS
#define SOLVE_THREADS 768
#endif
#ifndef SOLVE_WAVES_PER_EU
#define SOLVE_WAVES_PER_EU 3
#endif
#ifndef BUILD_UNROLL
#define BUILD_UNROLL 6
#endif
#ifndef GFBE_SOLVE_INLINE
#define GFBE_SOLVE_INLINE 0
#endif
#if GFBE_SOLVE_INLINE
#define GFBE_SOLVE_FN __forceinline__
#else
#define GFBE_SOLVE_FN __noinline__
#endif
#ifndef GFBE_CHOL_STAMP
#define GFBE_CHOL_STAMP 0   // diagnostics: per-panel time stamps into the NEXT window's timing slots (single-window runs only)
#endif
#ifndef GFBE_SOLVE_ESYM
#define GFBE_SOLVE_ESYM 1
#endif
#define TB 16                          // tile edge of the blocked Cholesky
typedef double dbl4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int tile_idx(int I, int J) { return I * (I + 1) / 2 + J; }   // J <= I
// Element (r, c) of a 16x16 LDS tile. The column is XOR-swizzled with the row so that the column-wise
// accesses of the panel solve / MFMA operand loads (16 lanes, same column, 16 rows) hit 16 different banks
// instead of two (a row stride of 16 doubles = 32 dwords is the worst case for the 64-bank LDS).
__device__ __forceinline__ int tsw(int r, int c) { return r * TB + (c ^ r); }

// Rebuild E for a new mu directly from the landmark rows (slow path: only after a failed Cholesky). own_only: the tiles of this
// rank (landmark sharding; the ranks' parts are summed by an all-reduce).
__device__ void rebuild_E(const BatchDev &d, const WinDesc &ds, int w, double mu, double *E, double *eg, bool own_only) {
  const size_t TL = d.tot_lm;
  for (int e = threadIdx.x; e < NV * NV + NV; e += blockDim.x) {
    const bool isg = e >= NV * NV;
    const int a = isg ? e - NV * NV : e / NV, b = isg ? 0 : e % NV;
    double acc = 0.0;
    if (ds.act[a] && (isg || ds.act[b]) && (isg || a <= b)) {
      for (int tile = 0; tile < ds.n_tiles; tile++) {
        const int s = d.tile_start[ds.tile_off + tile];
        if (6 * s > a) break;   // tiles are ordered by start frame
        if (own_only && !TILE_OWNED(d, tile)) continue;
        for (int l = 0; l < LM_TILE; l++) {
          const int slot = ds.lm_off + tile * LM_TILE + l;
          const int info = d.lm_info[slot];
          const int m = (info >> 8) & 0xff;
          if (!((info >> 24) & 1) || ((info >> 16) & 1) || m == 0) continue;
          const double sl = d.lm_sl[slot], hs2 = sl * sl * d.lm_Hll[slot];
          const double wl = sl * sl / (hs2 + mu * clamp_diag(hs2));
          auto hval = [&](int x) -> double {
            if (x >= T_EX) return d.lm_hC[(size_t)(x == T_TD ? 12 : 6 + x - T_EX) * TL + slot];
            const int f = x / 6, q = x % 6;
            if (f == s) return d.lm_hC[(size_t)q * TL + slot];
            const int k = f - s - 1;
            if (k < 0 || k >= m) return 0.0;
            return d.lm_hP[((size_t)k * 6 + q) * TL + slot];
          };
          acc += wl * hval(a) * (isg ? d.lm_gl[slot] : hval(b));
        }
      }
    }
    if (isg) eg[a] = acc;
    else if (a <= b) { E[a * NV + b] = acc; E[b * NV + a] = acc; }
  }
  __syncthreads();
}
// landmark sharding: this rank's part of E | eg at the retry's mu for the windows that retry, zeros for the others
__global__ __launch_bounds__(1024) void k_rebuild_E_shard(BatchDev d) {
  const int w = blockIdx.x;
  const WinCtl &c = d.ctl[w];
  double *Er = d.Er + (size_t)w * (NV * NV + NV);
  if (c.done || !c.lin_retry) { for (int e = threadIdx.x; e < NV * NV + NV; e += blockDim.x) Er[e] = 0.0; return; }
  rebuild_E(d, d.desc[w], w, c.mu, Er, Er + NV * NV, true);
}

// 1/sqrt(d) from v_rsq_f64 + two Newton steps (full FP64 accuracy without the long IEEE sqrt/div sequences).
__device__ __forceinline__ double rsqrt_refined(double d) {
  double r = __builtin_amdgcn_rsq(d);
  const double hd = 0.5 * d;
  r = r * __builtin_fma(-hd * r, r, 1.5);
  r = r * __builtin_fma(-hd * r, r, 1.5);
  return r;
}
__device__ __forceinline__ double lane_bcast(double v, int src) {   // src is wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// Lower Cholesky of one 16 x 16 LDS tile by a single wave, replaced IN PLACE by the inverse of its factor, W = L^-1
// (lower triangular, the upper triangle written as zeros). With W the panel step X L^T = A becomes the matrix-core product
// X = A W^T (no 16-step substitution per row any more) and the back-substitution a 16 x 16 matrix-vector product per panel.
//   A single wave issues one instruction every ~5 cycles, so the tile step is bound by its instruction count, not by latencies.
//   Lane i (of every 16-lane row of the wave) keeps row i of the FULL symmetric tile (the tiles are built and updated symmetric)
//   and row i of the inverse being formed in registers. At step k the pivot row travels by DPP: `row_newbcast:k` hands lane
//   k's register to its whole row inside the FMA itself (64-bit DPP; no v_readlane, no SGPR round trip):
//       d = a_kk (v_mov_b64_dpp), inv = 1/sqrt(d), t_i = a_ik / d
//       a_ij -= t_i a_kj  (j > k)        u_ic -= t_i u_kc  (c <= k)       one v_fmac_f64_dpp each
//   where u = diag(L) W is the unscaled inverse (u starts as the identity); lane i scales its row by 1/L_ii at the end.
//   Column k of the tile dies at step k and column k of u is born there: 17 live doubles per lane throughout.
// Returns false on a bad pivot (not positive or not finite).
// zrow >= 0: row `zrow` of L (the right-hand side row of the LAST diagonal tile: z of the last partial panel) is written to zout
// as it is formed (entries q < zrow are meaningful).
// (out of line — inlined, its live registers push the 128-VGPR kernel into scratch — with LDS-typed pointers: a generic
// pointer would turn every tile access into a FLAT instruction. A VALU result needs two wait states before a DPP instruction
// reads it; the compiler's hazard recogniser does not look into inline assembly, hence the s_nop in front of the pivot move:
// everything else a DPP operand reads was written at least one pivot chain earlier.)
typedef __attribute__((address_space(3))) double lds_double;
typedef __attribute__((address_space(3))) int lds_int;
template <int KK>
__device__ __forceinline__ void dpp_fmac(double &acc, double m) {   // acc += m * (lane KK of the 16-lane row)'s acc
  asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "n"(KK));
}
// Pivot step K. (Measured, profiles/ubench/dp_issue_rate_mi355x.txt: a lone wave issues one FP64 instruction every ~5.4 cycles,
// a dependent one every 9, v_rsq_f64 26: the step is bound by its ~33 instructions; weaving the updates of step K-1 into the
// rsq / Newton chain of step K by hand — every instruction a volatile asm — came out slower than the compiler's schedule.)
template <int K>
__device__ __forceinline__ void chol_inv_step(double (&row)[TB], double (&u)[TB], int li, double &myinv, int zrow, lds_double *zout) {
  double dkk;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(dkk) : "v"(row[K]), "n"(K));
  const double inv = rsqrt_refined(dkk);
  const double lik = row[K] * inv;          // L[i][k] for i > k
  double m = (li > K) ? -(lik * inv) : 0.0; // -a_ik / d for the rows below the pivot; rows <= k are final
  myinv = (li == K) ? inv : myinv;
  u[K] = (li == K) ? 1.0 : 0.0;             // (column K of the unscaled inverse starts here: rows above K never touch it)
  asm volatile("" : "+v"(u[K]), "+v"(m));   // materialised HERE: a VALU write needs two wait states before a DPP instruction reads
                                            // the register, and the compiler's hazard recogniser does not look into inline assembly
  // (the last diagonal tile only — a scalar branch — and every lane stores the same value: a lane-masked store would rewrite
  // EXEC in every step, and a DPP instruction needs five wait states after an EXEC write)
  if (zrow >= 0) zout[K] = lane_bcast(lik, zrow);
#pragma unroll
  for (int j = K + 1; j < TB; j++) dpp_fmac<K>(row[j], m);
#pragma unroll
  for (int c = 0; c <= K; c++) dpp_fmac<K>(u[c], m);
}
__device__ __forceinline__ bool chol_inv_tile16(lds_double *T, int lane, int zrow, lds_double *zout, double *stamp = nullptr) {
  double row[TB], u[TB];
  const int li = lane & 15;
  if (stamp && lane == 0) stamp[21] = (double)wall_clock64();
  int lio = li;
  asm volatile("" : "+v"(lio));               // (opaque: 16 loop-invariant tile addresses hoisted out of the panel loop would be spilled)
#pragma unroll
  for (int q = 0; q < TB; q++) row[q] = T[tsw(lio, q)];
  double myinv = 0.0;
  chol_inv_step<0>(row, u, li, myinv, zrow, zout);   chol_inv_step<1>(row, u, li, myinv, zrow, zout);
  chol_inv_step<2>(row, u, li, myinv, zrow, zout);   chol_inv_step<3>(row, u, li, myinv, zrow, zout);
  chol_inv_step<4>(row, u, li, myinv, zrow, zout);   chol_inv_step<5>(row, u, li, myinv, zrow, zout);
  chol_inv_step<6>(row, u, li, myinv, zrow, zout);   chol_inv_step<7>(row, u, li, myinv, zrow, zout);
  chol_inv_step<8>(row, u, li, myinv, zrow, zout);   chol_inv_step<9>(row, u, li, myinv, zrow, zout);
  chol_inv_step<10>(row, u, li, myinv, zrow, zout); chol_inv_step<11>(row, u, li, myinv, zrow, zout);
  chol_inv_step<12>(row, u, li, myinv, zrow, zout); chol_inv_step<13>(row, u, li, myinv, zrow, zout);
  chol_inv_step<14>(row, u, li, myinv, zrow, zout); chol_inv_step<15>(row, u, li, myinv, zrow, zout);
  if (stamp && lane == 0) stamp[22] = (double)wall_clock64();
  if (lane < TB) {
#pragma unroll
    for (int q = 0; q < TB; q++) T[tsw(lio, q)] = u[q] * myinv;     // (u[q] is an exact zero for q > i: column q starts as e_q and rows < q never touch it)
  }
  if (stamp && lane == 0) stamp[23] = (double)wall_clock64();
  // a pivot that is not positive and finite turns its 1/sqrt into inf or NaN (and everything after it into NaN): one test of
  // every lane's own 1 / L_ii at the end instead of a test per pivot inside the chain
  return __ballot(!((myinv > 0.0) && (myinv < 1.0e300))) == 0ull;
}

// The tile build of k_solve, out of line (its own register allocation: six tiles in flight per thread group). Returns this
// thread's share of v^T S v.
// (address-space-typed pointers: through generic ones every load here would be a FLAT instruction)
typedef __attribute__((address_space(3))) short lds_short;
typedef __attribute__((address_space(1))) double glb_double;
__device__ GFBE_SOLVE_FN double solve_build_tiles(lds_double *smem, const lds_short *perm, const lds_double *ys, const glb_double *H, const glb_double *E,
                                                 const glb_double *eg, const glb_double *gsp, const glb_double *gDp, const glb_double *ggts, double mu,
                                                 int n, int ntile_all, int t) {
  double vsv = 0.0;      // v^T S v, summed over the tile entries as they are built (off-diagonal tiles stand for both triangles)
  for (int te0 = t >> 8; te0 < ntile_all; te0 += BUILD_UNROLL * (SOLVE_THREADS >> 8)) {   // BUILD_UNROLL tiles per thread group in flight
    const int r = (t & 255) >> 4, cc = t & 15;
    double hv[BUILD_UNROLL], ev[BUILD_UNROLL];
    int aa[BUILD_UNROLL], bb[BUILD_UNROLL], kind[BUILD_UNROLL];
    bool offdiag[BUILD_UNROLL];
#pragma unroll
    for (int u = 0; u < BUILD_UNROLL; u++) {
      const int te = te0 + u * (SOLVE_THREADS >> 8);
      int I, J;
      tri_decode(te, I, J);
      const int ia = I * TB + r, ib = J * TB + cc;
      offdiag[u] = I != J;
      kind[u] = 0; aa[u] = 0; bb[u] = 0; hv[u] = 0.0; ev[u] = 0.0;
      if (te < ntile_all) {
        if (ia < n && ib < n) {
          const int a = perm[ia], b = perm[ib];
          aa[u] = a; bb[u] = b; kind[u] = 1;
          hv[u] = H[(size_t)max(a, b) * ND + min(a, b)];   // H holds its lower triangle
#if GFBE_SOLVE_ESYM
          if (a < NV && b < NV) ev[u] = E[max(a, b) * NV + min(a, b)];   // (lower triangle, like H: the diagonal tiles come out exactly symmetric)
#else
          if (a < NV && b < NV) ev[u] = E[a * NV + b];
#endif
        } else if (ia == n && ib < n) { bb[u] = perm[ib]; kind[u] = 2; }
        else if (ib == n && ia < n) { bb[u] = perm[ia]; kind[u] = 2; }
        else kind[u] = (ia == ib) ? (ia == n ? 4 : 3) : 5;
      }
    }
#pragma unroll
    for (int u = 0; u < BUILD_UNROLL; u++) {
      const int te = te0 + u * (SOLVE_THREADS >> 8);
      if (te >= ntile_all) continue;
      double v;
      if (kind[u] == 1) {
        v = hv[u];
        if (aa[u] < NV && bb[u] < NV) v -= ev[u];
        v *= ys[aa[u]] * ys[bb[u]];
        if (aa[u] == bb[u]) { const double dp = gDp[aa[u]]; v += mu * dp * dp; }
        vsv = __builtin_fma(v * ys[ND + aa[u]], ys[ND + bb[u]] * (offdiag[u] ? 2.0 : 1.0), vsv);
      } else if (kind[u] == 2) v = ggts[bb[u]] - (bb[u] < NV ? gsp[bb[u]] * eg[bb[u]] : 0.0);
      else v = kind[u] == 4 ? 1e200 : (kind[u] == 3 ? 1.0 : 0.0);
      smem[(size_t)te * (TB * TB) + tsw(r, cc)] = v;
    }
  }
  return vsv;
}

// The factorisation loop of k_solve, out of line: inside this function the only live state is a handful of indices, so the
// register-resident tile step (chol_inv_tile16: 64 VGPRs of tile and inverse rows) is inlined without spilling and without a
// call per panel; the kernel around it saves its own registers once.
template <int NWAVES>
__device__ GFBE_SOLVE_FN void chol_factor_all(lds_double *smem, int nt, int n, int t, lds_double *zlast, lds_int *flag, double *stamp) {
  const int lane = t & 63, wave = t >> 6;
#define CF_STAMP(i) do { if (t == 0) stamp[i] = (double)wall_clock64(); } while (0)
  const int lr = lane & 15, lk = lane >> 4;
  for (int P = -1; P < nt; P++) {
    if (P == 0) CF_STAMP(17);
    if (*flag) break;
    if (P >= 0) {
      const lds_double *Wp = smem + (size_t)tile_idx(P, P) * (TB * TB);
      for (int I = P + 1 + wave; I < nt; I += NWAVES) {
        lds_double *tip = smem + (size_t)tile_idx(I, P) * (TB * TB);
        dbl4 acc = {0.0, 0.0, 0.0, 0.0};
        double va[4], vb[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { va[q] = tip[tsw(lr, q * 4 + lk)]; vb[q] = Wp[tsw(lr, q * 4 + lk)]; }
#pragma unroll
        for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[kk], vb[kk], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; q++) tip[tsw(lk + 4 * q, lr)] = acc[q];
      }
      __syncthreads();
    }
    if (P == 0) CF_STAMP(18);
    // trailing tiles (I, J), P < J <= I. Wave 0 takes tile (P+1, P+1) and its factor-and-invert — the critical chain — and
    // nothing else; the other tiles go round-robin over waves 1..15.
    const int nrem = nt - 1 - P;
    const int ntr = P < 0 ? 1 : nrem * (nrem + 1) / 2;
    for (int e = (wave == 0 ? 0 : wave); e < ntr; e += (wave == 0 ? ntr : NWAVES - 1)) {
      int ii = 0, rr = e;
      while (rr > ii) { rr -= ii + 1; ii++; }
      const int I = P + 1 + ii, J = P + 1 + rr;
      lds_double *C = smem + (size_t)tile_idx(I, J) * (TB * TB);
      if (P >= 0) {
        const lds_double *LI = smem + (size_t)tile_idx(I, P) * (TB * TB), *LJ = smem + (size_t)tile_idx(J, P) * (TB * TB);
        dbl4 acc;
        double va[4], vb[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { acc[q] = C[tsw(lk + 4 * q, lr)]; va[q] = -LI[tsw(lr, q * 4 + lk)]; vb[q] = LJ[tsw(lr, q * 4 + lk)]; }
#pragma unroll
        for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[kk], vb[kk], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; q++) C[tsw(lk + 4 * q, lr)] = acc[q];
      }
      if (e == 0 && P + 1 < nt) {   // tile (P+1, P+1) is final now: factorise and invert it here (wave 0), ahead of the block barrier
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        if (!chol_inv_tile16(C, lane, P + 2 == nt ? n % TB : -1, zlast, P == 0 ? stamp : nullptr) && lane == 0) *flag = 1;
      }
    }
    if (P == 0) CF_STAMP(20);
    __syncthreads();
    if (P == 0) CF_STAMP(19);
#if GFBE_CHOL_STAMP
    if (P >= 0 && P < 12) { if (t == 0) stamp[32 + P] = (double)wall_clock64(); if (t == 0 && P == 0) stamp[31] = stamp[17]; }
#endif
  }
#undef CF_STAMP
}

__global__ __launch_bounds__(SOLVE_THREADS, SOLVE_WAVES_PER_EU) void k_solve(BatchDev d, int retry_pass) {
  const int w = blockIdx.x;
  const WinDesc &ds = d.desc[w];
  WinCtl &c = d.ctl[w];
  if (c.done || c.reuse) return;
  if (retry_pass && !c.lin_retry) return;       // (landmark sharding: second factorisation of the windows whose first one failed)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ short perm[ND + TB];     // (16-bit: the 160 KB of LDS are full — 78 tiles of 2 KB for a fully active window)
  __shared__ double red[16], ys[2 * ND + TB];
  __shared__ double s_zz, s_vSv, zlast[TB];
  __shared__ int flag, s_nact;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double *H = d.H + (size_t)w * ND * ND, *g = d.g + (size_t)w * ND;
  double *gsp = d.sp + (size_t)w * ND, *gDp = d.Dp + (size_t)w * ND, *ggts = d.gts + (size_t)w * ND;
  double *gvp = d.vp + (size_t)w * ND, *gyp = d.yp + (size_t)w * ND;
  const bool first = (c.iter == 0);
  double *stamp = d.timing + (size_t)w * 32;
#define STAMP(i) do { if (t == 0) stamp[i] = (double)wall_clock64(); } while (0)
  STAMP(0);

  // active-dim list by a wave-level prefix count (dims 0..191 live in waves 0..2)
  __shared__ int wcount[4];
  {
    const bool on = (t < ND) && ds.act[t];
    const unsigned long long m = __ballot(on);
    if (t < 192 && lane == 0) wcount[wave] = __popcll(m);
    __syncthreads();
    if (t < 192) {
      int base = 0;
      for (int q = 0; q < wave; q++) base += wcount[q];
      if (on) perm[base + __popcll(m & ((1ull << lane) - 1ull))] = t;
      if (t == 0) s_nact = wcount[0] + wcount[1] + wcount[2];
    }
    __syncthreads();
    for (int a = s_nact + t; a < ND + TB; a += blockDim.x) perm[a] = -1;
  }
  if (first && t == 0) {   // total cost of the first linearisation point (fixed order)
    double cost = 0.0;
    for (int r = 0; r < d.world; r++) cost += d.xa[((size_t)w * d.world + r) * XCHG];   // visual cost (k_visblock; summed over the ranks)
    for (int q = 0; q < ds.n_imu; q++) cost += d.imu_part[((size_t)w * MAX_IMU + q) * IMU_PART + IMU_PART - 2];
    for (int q = 0; q < ds.n_wheel; q++) cost += d.wheel_part[((size_t)w * MAX_WHEEL + q) * WHEEL_PART + WHEEL_PART - 2];
    cost += d.prior_g[(size_t)w * (ND + 2) + ND];
    for (int q = 0; q < ds.n_plane; q++) cost += d.plane_part[((size_t)w * MAX_PLANE + q) * PLANE_PART + PLANE_PART - 2];
    if (ds.use_anchor) cost += d.anchor_part[(size_t)w * ANCHOR_PART + ANCHOR_PART - 2];
    c.cost = cost; c.initial_cost = cost; c.cost_history[0] = cost;
  }
  // Jacobi scaling (iteration 0 only), D = sqrt(clamp(diag)), scaled gradient, Cauchy direction
  double g2 = 0.0, gmax = 0.0, xn2 = 0.0;
  for (int a = t; a < ND; a += blockDim.x) {
    double s = 1.0, dp = 1.0, gt = 0.0, v = 0.0;
    if (ds.act[a]) {
      const double haa = H[(size_t)a * ND + a];
      s = first ? (d.opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(haa)) : 1.0) : gsp[a];
      const double d2 = clamp_diag(s * s * haa);
      dp = sqrt(d2); gt = s * g[a]; v = gt / d2;
      g2 += gt * gt / d2;
      gmax = fmax(gmax, fabs(g[a]));
    }
    if (first) gsp[a] = s;
    gDp[a] = dp; ggts[a] = gt; gvp[a] = v;
  }
  {
    const double *X = d.x + ((size_t)w * 2 + c.cur) * NA;
    for (int b = t; b < GFBE_BLK_COUNT; b += blockDim.x)
      if (ds.blk_free[b]) for (int k = 0; k < blk_gsize(b); k++) { const double v = X[blk_amb(b) + k]; xn2 += v * v; }
  }
  {   // (the tile area is free until the build: scratch of the combined reduction)
    const double pv[3] = {g2, gmax, xn2};
    block_reduce_multi<3>(pv, 0x2u, smem);
    g2 = smem[48]; gmax = smem[49]; xn2 = smem[50];
  }
  __syncthreads();
  STAMP(1);
  const int n = s_nact;                 // active dims
  const int na = n + 1;                 // + the right-hand side as an extra row (forward substitution for free)
  const int nt = (na + TB - 1) / TB;    // tiles per side
  const double *E = retry_pass ? d.Er + (size_t)w * (NV * NV + NV) : d.E + (size_t)w * NV * NV;
  const double *eg = retry_pass ? E + NV * NV : d.eg + (size_t)w * NV;

  double mu = c.mu;
  bool solved = false, e_valid = true;
  while (mu < GF_MAX_MU) {
    if (!e_valid) {
      if (d.world > 1) {
        // landmark sharding: E for the larger mu needs every rank's landmarks — hand the window to the retry pass (rebuild on
        // all ranks, one all-reduce, k_solve again); a second failure is a failed linear solve
        if (!retry_pass) { if (t == 0) { c.lin_retry = 1; c.mu = mu; } return; }
        break;
      }
      rebuild_E(d, ds, w, mu, d.E + (size_t)w * NV * NV, d.eg + (size_t)w * NV, false);
    }
    // ---- augmented, scaled, regularised, Schur-reduced system in 16x16 LDS tiles (lower triangle of tiles)
    //      [ S    rhs ]   S = s H s + mu D^2 - s E s   rhs = gt - s eg
    //      [ rhs' big ]
    const int ntile_all = nt * (nt + 1) / 2;
    // scaling and right-hand side staged in LDS (ys is free until the back-substitution)
    // scaling and the Cauchy direction v staged in LDS (ys is free until the back-substitution; the right-hand side row
    // gt - s eg is read where it is placed: one tile row)
    for (int a = t; a < ND; a += blockDim.x) { ys[a] = gsp[a]; ys[ND + a] = gvp[a]; }
    __syncthreads();
    double vsv = solve_build_tiles((lds_double *)smem, (const lds_short *)perm, (const lds_double *)ys, (const glb_double *)H, (const glb_double *)E,
                                   (const glb_double *)eg, (const glb_double *)gsp, (const glb_double *)gDp, (const glb_double *)ggts, mu, n, ntile_all, t);
    vsv = block_sum(vsv, red);
    if (t == 0) { flag = 0; s_vSv = vsv; }
    __syncthreads();
    STAMP(2);
    // ---- blocked right-looking Cholesky on the FP64 matrix cores. Per panel P:
    //        W_P = L_PP^-1 in place of the diagonal tile (one wave, chol_inv_tile16)
    //        L_IP = A_IP W_P^T for every tile below it (one v_mfma_f64_16x16x4_f64 chain per tile; the right-hand side row is one
    //        of them: forward substitution for free)
    //        trailing tiles (I, J) -= L_IP L_JP^T; the wave that updates tile (P+1, P+1) first factorises and inverts it right
    //        away, while the other waves are still updating: only panel 0 pays for its own diagonal tile.
    //      Two block barriers per panel.
    // (the loop starts at P = -1 — no panel yet, wave 0 factorises tile (0, 0) — so that chol_inv_tile16 is inlined once:
    // two copies of its 64 live registers do not fit the 128-VGPR budget of a 1024-thread workgroup)
    chol_factor_all<(SOLVE_THREADS >> 6)>((lds_double *)smem, nt, n, t, (lds_double *)zlast, (lds_int *)&flag, stamp);
    bool ok = (flag == 0);
    if (d.test_fail_chol_iter > 0 && c.iter + 1 == d.test_fail_chol_iter && e_valid && !retry_pass) ok = false;   // fault injection: first attempt of that iteration
    STAMP(3);
    if (ok) {
      // z = L^-1 rhs sits in row n of the factor; y^T S y = |z|^2. Backward substitution y = L^-T z by ONE wave without
      // block barriers: for panel P, lane (c, part) gathers sum_{I > P} L(I,P)^T y_I for column c over its four rows of every
      // tile, the four parts meet through two shuffles, and y_P = W_P^T (z_P - sum) is a 16 x 16 matrix-vector product.
      double zz = 0.0;
      for (int i = t; i < n + TB; i += blockDim.x) {
        // (the entries of the last diagonal tile were saved before the tile became its own inverse)
        const double z = i < n ? (i / TB == n / TB ? zlast[i % TB] : smem[(size_t)tile_idx(n / TB, i / TB) * (TB * TB) + tsw(n % TB, i % TB)]) : 0.0;
        ys[i] = z;                                   // (zeros behind n: the rows of the last tile past the system)
        zz += z * z;
      }
      zz = block_sum(zz, red);
      if (t == 0) s_zz = zz;
      __syncthreads();
      if (wave == 0) {
        const int cI = lane & 15, part = lane >> 4;
        const int np = (n - 1) / TB;
        for (int P = np; P >= 0; P--) {
          const double *Wt = smem + (size_t)tile_idx(P, P) * (TB * TB);
          const int r0 = P * TB;
          double s0 = 0.0, s1 = 0.0;
          for (int I = P + 1; I <= np; I++) {
            const double *Tip = smem + (size_t)tile_idx(I, P) * (TB * TB);
            const int rI = I * TB + 4 * part;
            s0 = __builtin_fma(Tip[tsw(4 * part, cI)], ys[rI], s0);
            s1 = __builtin_fma(Tip[tsw(4 * part + 1, cI)], ys[rI + 1], s1);
            s0 = __builtin_fma(Tip[tsw(4 * part + 2, cI)], ys[rI + 2], s0);
            s1 = __builtin_fma(Tip[tsw(4 * part + 3, cI)], ys[rI + 3], s1);
          }
          double sacc = s0 + s1;
          sacc += __shfl_xor(sacc, 16, 64);
          sacc += __shfl_xor(sacc, 32, 64);
          const double tc = ys[r0 + cI] - sacc;       // (entries past n: 0 - 0)
          // y_j = sum_i W[i][j] t_i: lane (j, part) takes rows i = 4 part .. 4 part + 3; t_i comes from lane i
          double yj = 0.0;
#pragma unroll
          for (int h = 0; h < 4; h++) {
            const int i = 4 * part + h;
            yj = __builtin_fma(Wt[tsw(i, cI)], __shfl(tc, i, 64), yj);
          }
          yj += __shfl_xor(yj, 16, 64);
          yj += __shfl_xor(yj, 32, 64);
          __builtin_amdgcn_wave_barrier();
          if (lane < TB && r0 + lane < n) ys[r0 + lane] = yj;
          __threadfence_block();
          __builtin_amdgcn_wave_barrier();
        }
      }
      __syncthreads();
      // y back to the tangent dims: inactive dims get 0 (written by the threads of the padding entries of perm), every dim once
      int bad = 0;
      for (int i = t; i < n; i += blockDim.x) { const double y = ys[i]; gyp[perm[i]] = y; if (!isfinite(y)) bad = 1; }
      for (int a = t; a < ND; a += blockDim.x) if (!ds.act[a]) gyp[a] = 0.0;
      if (bad) flag = 1;
      __syncthreads();
      ok = (flag == 0);
    }
    __syncthreads();
    if (ok) { solved = true; break; }
    mu *= GF_MU_INC;
    e_valid = false;
  }
  if (!solved) {
    if (t == 0) { c.done = 1; c.termination = 4; c.status = GFBE_NUMERICAL_FAILURE; c.lin_fail = 1; c.mu = mu; }
    return;
  }
  STAMP(4);
  // dense shares of the dogleg scalars v^T Ht v, v^T Ht y, y^T Ht y with Ht = s H s (the landmark shares come from k_lm_step).
  // The factorised system is S = Ht + mu D^2 - Et (Et = s E s) and S y = rhs, y^T S y = |z|^2, so
  //   y^T Ht y = |z|^2   - mu y^T D^2 y + y^T Et y        v^T Ht y = v^T rhs - mu v^T D^2 y + v^T Et y
  //   v^T Ht v = v^T S v - mu v^T D^2 v + v^T Et v        (v^T S v: summed while the tiles were built)
  // — one pass over the 73 x 73 block E instead of a second pass over the 182 x 182 block H.
  double n2 = 0.0, gyv = 0.0, vrhs = 0.0, vDv = 0.0, vDy = 0.0, vEv = 0.0, vEy = 0.0, yEy = 0.0;
  for (int a = t; a < ND; a += blockDim.x) { ys[a] = gsp[a] * gvp[a]; ys[ND + a] = gsp[a] * gyp[a]; }   // s v, s y (original dims)
  __syncthreads();
  for (int a = t; a < ND; a += blockDim.x) {
    const double d2 = gDp[a] * gDp[a], y = gyp[a], v = gvp[a];
    n2 += d2 * y * y;
    gyv += ggts[a] * y;
    vDv += d2 * v * v;
    vDy += d2 * v * y;
    vrhs += v * (ggts[a] - (a < NV ? gsp[a] * eg[a] : 0.0));
  }
  for (int e = t; e < NV * NV; e += blockDim.x) {
    const int a = e / NV, b = e - a * NV;
    const double ev = E[e];                       // (rows / columns of inactive dims are zero in E)
    vEv = __builtin_fma(ev * ys[a], ys[b], vEv);
    vEy = __builtin_fma(ev * ys[a], ys[ND + b], vEy);
    yEy = __builtin_fma(ev * ys[ND + a], ys[ND + b], yEy);
  }
  {   // (the tiles are dead after the back-substitution: scratch of the combined reduction)
    const double gv[8] = {n2, gyv, vrhs, vDv, vDy, vEv, vEy, yEy};
    block_reduce_multi<8>(gv, 0u, smem);
    n2 = smem[128]; gyv = smem[129]; vrhs = smem[130]; vDv = smem[131]; vDy = smem[132]; vEv = smem[133]; vEy = smem[134]; yEy = smem[135];
  }
  if (t == 0) {
    const double zz = s_zz, vSv = s_vSv;
    c.mu = mu;
    c.G2 = g2; c.N2 = n2; c.gy = gyv;
    c.vHv = vSv - mu * vDv + vEv;
    c.vHy = vrhs - mu * vDy + vEy;
    c.yHy = zz - mu * n2 + yEy;
    c.grad_max = gmax;
    c.x_norm = xn2;        // dense share; k_step adds the landmarks and takes the square root
    c.have_step = 2;       // "fresh linearisation" marker consumed by k_step
    c.lin_retry = 0;
  }
  STAMP(5);
#undef STAMP
}


// =============================================================================================
// k_solve_chain: the same step as k_solve with the speed-bias blocks eliminated FIRST.
//
// After the landmark elimination the reduced system S (n ~ 175) has a dense part — the poses the landmarks couple, the
// extrinsics and the odometer intrinsics, ~76 dims — and the eleven speed-bias blocks, which IMUFactor (k, k+1) couples
// only with their chain neighbours and with the poses of frames k, k+1 (the prior adds SpeedBias[0] x everything it kept:
// MarginalizationFactor of estimator.cpp:3400-3433 never keeps another speed-bias block; the host checks that and hands any
// other structure to the monolithic k_solve). k_solve factorises all of S as 12 x 12 tiles: twelve 16-pivot tile steps on one
// wave's critical chain, 78 tiles = the whole 160 KB of a CU's LDS. Here, with the variable order [SB_10, SB_9, ..., SB_0, dense]:
//
//   chain, block k = 10 .. 0 (ONE wave, 9 pivots per block, all in registers through 64-bit DPP row broadcasts):
//       W_k = chol(A_k)^-1                    A_k: the block's 9 x 9 diagonal block (already downdated by block k + 1)
//       Yc_k = W_k C_k                        C_k = S(SB_k, SB_k-1): the row operations of the factorisation applied to C_k's
//       A_k-1 -= Yc_k^T Yc_k                                         columns as well, like the identity that becomes W_k
//   wide rows (two waves behind it, one LANE per column of the dense part + the right-hand side; lane-local recurrence):
//       R_k' = R_k - Yc_k+1^T Yr_k+1          R_k = S(SB_k, dense | rhs)
//       Yr_k = W_k R_k'                       (column nd of Yr_k = z_k, the forward-substituted right-hand side of the block)
//   dense update (five waves behind those, FP64 matrix cores, accumulators in registers over all blocks):
//       D -= sum_k Yr_k^T Yr_k                (augmented with the right-hand side row, like k_solve's tiles)
//   then the blocked Cholesky of k_solve on the 5 x 5 tiles of D (chol_factor_all), its back-substitution, and the chain's:
//       x_k = W_k^T (z_k - Yr_k x_dense - Yc_k x_k-1),  k = 0 .. 10.
//
// The three stages run as a pipeline, one block barrier per chain block. Fill stays inside the structure: Yr_k only reaches the
// poses of frames >= k - 1 (lo_k) besides the extrinsic / intrinsic columns, which the column lanes and the tile products skip.
// 15 tiles + chain blocks + a two-block ring of Yr = ~67 KB of LDS: two workgroups per CU; the Yr rows (76 KB per window)
// go to HBM / L2 for the back-substitution. Inactive speed-bias dims (constant block, window still filling up, USE_IMU = 0)
// are identity rows of their block.
// =============================================================================================
#define S2_THREADS 512
#define S2_WAVES (S2_THREADS >> 6)
#define S2_COL_WAVES 2                       // waves 1..2: the wide-row recurrence (lane = dense column)
enum { CH_NB = 9, CH_NC = NF, CH_ROWS = CH_NB * CH_NC, CH_RW = 96, CH_BLK = CH_NB * CH_NB, RING_ROWS = 12, RING_LD = 112,
       S2_MAX_NT = 6, S2_MAX_TILES = S2_MAX_NT * (S2_MAX_NT + 1) / 2 };
__host__ __device__ inline bool dim_in_chain(int a) { return a >= T_SB(0) && a < T_SB(0) + CH_ROWS; }

template <int KK>
__device__ __forceinline__ void dpp_fmac2(double &acc, double bc, double m) {   // acc += m * (lane KK of the 16-lane row)'s bc
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bc), "v"(m), "n"(KK));
}
// Pivot K of a chain block: chol_inv_step with CH_NB columns and the block's coupling row cr riding along (see chol_inv_step for
// the hazards handled by hand).
template <int K>
__device__ __forceinline__ void chain_pivot(double (&row)[CH_NB], double (&u)[CH_NB], double (&cr)[CH_NB], int li, double &myinv) {
  double dkk;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(dkk) : "v"(row[K]), "n"(K));
  const double inv = rsqrt_refined(dkk);
  const double lik = row[K] * inv;
  double m = (li > K) ? -(lik * inv) : 0.0;
  myinv = (li == K) ? inv : myinv;
  u[K] = (li == K) ? 1.0 : 0.0;
  asm volatile("" : "+v"(u[K]), "+v"(m));
#pragma unroll
  for (int j = K + 1; j < CH_NB; j++) dpp_fmac<K>(row[j], m);
#pragma unroll
  for (int c = 0; c <= K; c++) dpp_fmac<K>(u[c], m);
#pragma unroll
  for (int j = 0; j < CH_NB; j++) dpp_fmac<K>(cr[j], m);
}
template <int K> struct ChainPivots {
  static __device__ __forceinline__ void run(double (&row)[CH_NB], double (&u)[CH_NB], double (&cr)[CH_NB], int li, double &myinv) {
    chain_pivot<K>(row, u, cr, li, myinv);
    if constexpr (K + 1 < CH_NB) ChainPivots<K + 1>::run(row, u, cr, li, myinv);
  }
};
template <int M> struct ChainDowndate {   // arow -= Yc^T Yc, one row m of Yc at a time: arow[j] -= Yc[m][i] * Yc[m][j]
  static __device__ __forceinline__ void run(double (&arow)[CH_NB], const double (&ycs)[CH_NB], const double (&nyc)[CH_NB]) {
#pragma unroll
    for (int j = 0; j < CH_NB; j++) dpp_fmac2<M>(arow[j], ycs[j], nyc[M]);
    if constexpr (M + 1 < CH_NB) ChainDowndate<M + 1>::run(arow, ycs, nyc);
  }
};
// One block of the chain by one wave (every 16-lane row of the wave does the same work; lane i < 9 of a row owns row i).
//   A: the block's diagonal block (in: downdated A_k; out: W_k = L_k^-1, lower triangular)     Cb: C_k in, Yc_k out (k > 0)
//   An: A_k-1, downdated in place (k > 0)
// Returns false on a pivot that is not positive and finite.
__device__ __forceinline__ bool chain_block(lds_double *A, lds_double *Cb, lds_double *An, int lane, int has_next) {
  const int li = lane & 15;
  int lio = li < CH_NB ? li : 0;
  asm volatile("" : "+v"(lio));
  double row[CH_NB], u[CH_NB], cr[CH_NB];
  const bool mine = li < CH_NB;
#pragma unroll
  for (int q = 0; q < CH_NB; q++) {
    const double a = A[lio * CH_NB + q], cc = has_next ? Cb[lio * CH_NB + q] : 0.0;
    row[q] = mine ? a : 0.0; cr[q] = mine ? cc : 0.0; u[q] = 0.0;
  }
  double myinv = 0.0;
  // (definitions pinned here: a VALU write needs two wait states before a DPP instruction reads the register and the compiler
  //  does not look into inline assembly — it could otherwise sink a definition next to its first DPP use; EXEC may have been
  //  rewritten by masked code: five wait states before the first DPP instruction)
#pragma unroll
  for (int q = 0; q < CH_NB; q++) asm volatile("" : "+v"(row[q]), "+v"(cr[q]));
  asm volatile("s_nop 4" ::: "memory");
  ChainPivots<0>::run(row, u, cr, li, myinv);
  double ycs[CH_NB];
#pragma unroll
  for (int q = 0; q < CH_NB; q++) { u[q] *= myinv; ycs[q] = cr[q] * myinv; }
  if (lane < CH_NB) {
#pragma unroll
    for (int q = 0; q < CH_NB; q++) { A[lio * CH_NB + q] = u[q]; if (has_next) Cb[lio * CH_NB + q] = ycs[q]; }
  }
  const bool ok = __ballot(mine && !((myinv > 0.0) && (myinv < 1.0e300))) == 0ull;
  if (has_next) {
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    double arow[CH_NB], nyc[CH_NB];
#pragma unroll
    for (int q = 0; q < CH_NB; q++) { arow[q] = An[lio * CH_NB + q]; nyc[q] = -Cb[q * CH_NB + lio]; }   // row i of A_k-1, column i of Yc_k
#pragma unroll
    for (int q = 0; q < CH_NB; q++) asm volatile("" : "+v"(ycs[q]), "+v"(nyc[q]), "+v"(arow[q]));
    asm volatile("s_nop 4" ::: "memory");
    ChainDowndate<0>::run(arow, ycs, nyc);
    if (lane < CH_NB) {
#pragma unroll
      for (int q = 0; q < CH_NB; q++) An[lio * CH_NB + q] = arow[q];
    }
  }
  __threadfence_block();
  return ok;
}

typedef __attribute__((address_space(3))) unsigned char lds_uchar;
// ---- the three roles of the chain pipeline (k_solve_chain). Each is out of line — its own register allocation: the chain wave
// keeps ~45 doubles of block rows, a column lane its recurrence state, the matrix-core waves their accumulators — and each
// passes exactly CH_NC + 2 block barriers.
__device__ __noinline__ void chain_role(lds_double *Ach, lds_double *Cch, lds_int *flag, int lane) {
  for (int s = 0; s <= CH_NC + 1; s++) {
    const int k = CH_NC - 1 - s;
    if (k >= 0) {
      if (!chain_block(Ach + k * CH_BLK, Cch + k * CH_BLK, Ach + (k > 0 ? k - 1 : 0) * CH_BLK, lane, k > 0) && lane == 0) *flag = 1;
    }
    __syncthreads();
  }
}
// Wide rows: lane j owns column j of the dense part (j == n: the right-hand side). Returns the lane's share of v^T S v over the
// coupling entries it loads; the right-hand side lane leaves |z_chain|^2 in *zzc_out.
__device__ __noinline__ double column_role(const lds_double *Ach, const lds_double *Cch, lds_double *ring, const lds_double *ys, const lds_short *perm,
                                           const lds_int *s_lo, const lds_uchar *chact, lds_double *zzc_out, const glb_double *H, const glb_double *ggts,
                                           glb_double *gY, int n, int j) {
  const int na = n + 1;
  const bool col_on = j < na;
  const int bj = col_on && j < n ? perm[j] : 0;          // tangent dim of the column
  const double sbj = col_on && j < n ? ys[bj] : 0.0, vbj = col_on && j < n ? ys[ND + bj] : 0.0;
  double yprev[CH_NB], rpre[CH_NB], zzc = 0.0, vsv = 0.0;
#pragma unroll
  for (int i = 0; i < CH_NB; i++) { yprev[i] = 0.0; rpre[i] = 0.0; }
  // R_k[i][j] = S(SB_k dim i, column j): scaled H entry (the speed-bias rows lie outside E), or the scaled gradient for j == n
  auto load_R = [&](int k, double (&r)[CH_NB]) {
    const bool reach = col_on && j >= s_lo[k];
#pragma unroll
    for (int i = 0; i < CH_NB; i++) {
      const int a = T_SB(k) + i;
      double v = 0.0;
      if (reach && chact[a - T_SB(0)]) v = (j < n) ? H[(size_t)max(a, bj) * ND + min(a, bj)] * ys[a] * sbj : ggts[a];
      r[i] = v;
    }
  };
  load_R(CH_NC - 1, rpre);
  for (int s = 0; s <= CH_NC + 1; s++) {
    const int k = CH_NC - s;
    if (k >= 0 && k < CH_NC) {
      double r[CH_NB], y[CH_NB];
#pragma unroll
      for (int i = 0; i < CH_NB; i++) r[i] = rpre[i];
      if (k > 0) load_R(k - 1, rpre);                     // (in flight during this block's arithmetic)
      if (j < n) {
#pragma unroll
        for (int i = 0; i < CH_NB; i++) vsv = __builtin_fma(r[i] * ys[ND + T_SB(k) + i], 2.0 * vbj, vsv);
      }
      if (k + 1 < CH_NC) {
        const lds_double *Yc = Cch + (k + 1) * CH_BLK;    // Yc_k+1[m][i]
#pragma unroll
        for (int m = 0; m < CH_NB; m++)
#pragma unroll
          for (int i = 0; i < CH_NB; i++) r[i] = __builtin_fma(-Yc[m * CH_NB + i], yprev[m], r[i]);
      }
      const lds_double *Wk = Ach + k * CH_BLK;
#pragma unroll
      for (int i = 0; i < CH_NB; i++) {
        double sacc = 0.0;
#pragma unroll
        for (int m = 0; m <= i; m++) sacc = __builtin_fma(Wk[i * CH_NB + m], r[m], sacc);
        y[i] = sacc;
      }
      if (col_on) {
        lds_double *rg = ring + (size_t)(k & 1) * RING_ROWS * RING_LD + j;
#pragma unroll
        for (int i = 0; i < CH_NB; i++) { rg[i * RING_LD] = y[i]; gY[(size_t)(k * CH_NB + i) * CH_RW + j] = y[i]; yprev[i] = y[i]; }
        if (j == n) {
#pragma unroll
          for (int i = 0; i < CH_NB; i++) zzc = __builtin_fma(y[i], y[i], zzc);
        }
      }
    }
    __syncthreads();
  }
  if (col_on && j == n) *zzc_out = zzc;
  return vsv;
}
// Dense update D -= sum_k Yr_k^T Yr_k: wave gw takes the tiles e = gw, gw + 5, ... of the lower triangle (<= 5 of the 21 tiles of a
// 6 x 6 grid), accumulators in registers over all blocks, the blocks' Yr rows from the ring.
#define S2_UPD_WAVES (S2_WAVES - 1 - S2_COL_WAVES)
#define S2_UPD_TPW ((S2_MAX_TILES + S2_UPD_WAVES - 1) / S2_UPD_WAVES)
__device__ __noinline__ void gemm_role(lds_double *tiles, const lds_double *ring, const lds_int *s_lo, int ntile_all, int gw, int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  dbl4 acc[S2_UPD_TPW];
  int tI[S2_UPD_TPW], tJ[S2_UPD_TPW];
#pragma unroll
  for (int q = 0; q < S2_UPD_TPW; q++) {
    acc[q] = dbl4{0.0, 0.0, 0.0, 0.0};
    const int e = gw + S2_UPD_WAVES * q;
    tI[q] = -1; tJ[q] = 0;
    if (e < ntile_all) tri_decode(e, tI[q], tJ[q]);
  }
  for (int s = 0; s <= CH_NC + 1; s++) {
    const int k = CH_NC + 1 - s;
    if (k >= 0 && k < CH_NC) {
      const lds_double *rg = ring + (size_t)(k & 1) * RING_ROWS * RING_LD;
      const int lo = s_lo[k];
#pragma unroll
      for (int q = 0; q < S2_UPD_TPW; q++) {
        if (tI[q] < 0 || TB * (tJ[q] + 1) <= lo) continue;   // (columns below lo_k are zero in Yr_k)
        double va[3], vb[3];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) { va[kk] = rg[(4 * kk + lk) * RING_LD + TB * tI[q] + lr]; vb[kk] = rg[(4 * kk + lk) * RING_LD + TB * tJ[q] + lr]; }
#pragma unroll
        for (int kk = 0; kk < 3; kk++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[kk], vb[kk], acc[q], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < S2_UPD_TPW; q++) {
    if (tI[q] < 0) continue;
    lds_double *C = tiles + (size_t)tile_idx(tI[q], tJ[q]) * (TB * TB);
#pragma unroll
    for (int r = 0; r < 4; r++) C[tsw(lk + 4 * r, lr)] -= acc[q][r];
  }
}

// Dense tiles of the chain solve: entry (r, cc) of tile te of the augmented, scaled, regularised, Schur-reduced dense part
// (the k_solve tile layout with perm = the dense dims only). tb / nth: index and number of the building threads.
// Returns the thread's share of v^T S v over these entries.
__device__ __forceinline__ double chain_build_dense(double *tiles, const short *perm, const double *ys, const double *H, const double *E,
                                                    const double *eg, const double *gsp, const double *gDp, const double *ggts, double mu,
                                                    int n, int ntile_all, int tb, int nth) {
  double vsv = 0.0;
  constexpr int U = 4;
  for (int i0 = tb; i0 < ntile_all * (TB * TB); i0 += U * nth) {
    double hv[U], ev[U];
    int aa[U], bb[U], kind[U];
    bool offd[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int idx = i0 + u * nth;
      kind[u] = -1; aa[u] = 0; bb[u] = 0; hv[u] = 0.0; ev[u] = 0.0; offd[u] = false;
      if (idx >= ntile_all * (TB * TB)) continue;
      const int te = idx >> 8, r = (idx >> 4) & 15, cc = idx & 15;
      int I, J;
      tri_decode(te, I, J);
      const int ia = I * TB + r, ib = J * TB + cc;
      offd[u] = I != J;
      kind[u] = 0;
      if (ia < n && ib < n) {
        const int a = perm[ia], b = perm[ib];
        aa[u] = a; bb[u] = b; kind[u] = 1;
        hv[u] = H[(size_t)max(a, b) * ND + min(a, b)];
        if (a < NV && b < NV) ev[u] = E[max(a, b) * NV + min(a, b)];
      } else if (ia == n && ib < n) { bb[u] = perm[ib]; kind[u] = 2; }
      else if (ib == n && ia < n) { bb[u] = perm[ia]; kind[u] = 2; }
      else kind[u] = (ia == ib) ? (ia == n ? 4 : 3) : 5;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (kind[u] < 0) continue;
      const int idx = i0 + u * nth;
      const int te = idx >> 8, r = (idx >> 4) & 15, cc = idx & 15;
      double v;
      if (kind[u] == 1) {
        v = hv[u];
        if (aa[u] < NV && bb[u] < NV) v -= ev[u];
        v *= ys[aa[u]] * ys[bb[u]];
        if (aa[u] == bb[u]) { const double dp = gDp[aa[u]]; v += mu * dp * dp; }
        vsv = __builtin_fma(v * ys[ND + aa[u]], ys[ND + bb[u]] * (offd[u] ? 2.0 : 1.0), vsv);
      } else if (kind[u] == 2) v = ggts[bb[u]] - (bb[u] < NV ? gsp[bb[u]] * eg[bb[u]] : 0.0);
      else v = kind[u] == 4 ? 1e200 : (kind[u] == 3 ? 1.0 : 0.0);
      tiles[(size_t)te * (TB * TB) + tsw(r, cc)] = v;
    }
  }
  return vsv;
}

__global__ __launch_bounds__(S2_THREADS, 4) void k_solve_chain(BatchDev d, int retry_pass) {
  const int w = blockIdx.x;
  const WinDesc &ds = d.desc[w];
  WinCtl &c = d.ctl[w];
  if (c.done || c.reuse) return;
  if (retry_pass && !c.lin_retry) return;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ short perm[ND + TB];
  __shared__ double red[16], ys[2 * ND + TB];
  __shared__ double s_zz, s_zzc, s_vSv, zlast[TB], xs[CH_ROWS], tch[CH_ROWS + 16];
  __shared__ int flag, s_nact, s_nch, s_lo[CH_NC + 1];
  __shared__ unsigned char chact[CH_ROWS + 1];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double *H = d.H + (size_t)w * ND * ND, *g = d.g + (size_t)w * ND;
  double *gsp = d.sp + (size_t)w * ND, *gDp = d.Dp + (size_t)w * ND, *ggts = d.gts + (size_t)w * ND;
  double *gvp = d.vp + (size_t)w * ND, *gyp = d.yp + (size_t)w * ND;
  double *gY = d.solveY + (size_t)w * CH_ROWS * CH_RW;
  const bool first = (c.iter == 0);
  double *stamp = d.timing + (size_t)w * 32;
#define STAMP(i) do { if (t == 0) stamp[i] = (double)wall_clock64(); } while (0)
  STAMP(0);
  // LDS carve-up: dense tiles | chain diagonal blocks (A_k -> W_k) | chain couplings (C_k -> Yc_k) | two-block ring of Yr
  double *tiles = smem;
  double *Ach = smem + (size_t)d.solve_ntile * (TB * TB);
  double *Cch = Ach + CH_NC * CH_BLK;
  double *ring = Cch + CH_NC * CH_BLK;

  // dense dims (active, not in the chain) by a wave-level prefix count (dims 0..191 live in waves 0..2); chain activity flags
  __shared__ int wcount[4];
  {
    const bool act_t = (t < ND) && ds.act[t];
    const bool on = act_t && !dim_in_chain(t);
    const unsigned long long m = __ballot(on);
    const unsigned long long mc = __ballot(act_t && dim_in_chain(t));
    if (t < 192 && lane == 0) wcount[wave] = __popcll(m);
    if (t < ND && dim_in_chain(t)) chact[t - T_SB(0)] = act_t ? 1 : 0;
    if (t == 0) s_nch = 0;
    __syncthreads();
    if (t < 192) {
      int base = 0;
      for (int q = 0; q < wave; q++) base += wcount[q];
      if (on) perm[base + __popcll(m & ((1ull << lane) - 1ull))] = t;
      if (t == 0) s_nact = wcount[0] + wcount[1] + wcount[2];
      if (lane == 0 && mc) atomicAdd(&s_nch, __popcll(mc));
    }
    __syncthreads();
    for (int a = s_nact + t; a < ND + TB; a += blockDim.x) perm[a] = -1;
    // lo_k: first dense column the wide row of block k can reach — the poses of frames >= k - 1 (perm is ascending)
    if (t <= CH_NC) {
      int lo = 0;
      if (t >= 2 && t < CH_NC) { const int lim = 6 * (t - 1); for (int q = 0; q < s_nact && perm[q] < lim; q++) lo++; }
      s_lo[t] = (t == CH_NC) ? 0 : lo;
    }
  }
  if (first && t == 0) {   // total cost of the first linearisation point (fixed order)
    double cost = 0.0;
    for (int r = 0; r < d.world; r++) cost += d.xa[((size_t)w * d.world + r) * XCHG];
    for (int q = 0; q < ds.n_imu; q++) cost += d.imu_part[((size_t)w * MAX_IMU + q) * IMU_PART + IMU_PART - 2];
    for (int q = 0; q < ds.n_wheel; q++) cost += d.wheel_part[((size_t)w * MAX_WHEEL + q) * WHEEL_PART + WHEEL_PART - 2];
    cost += d.prior_g[(size_t)w * (ND + 2) + ND];
    for (int q = 0; q < ds.n_plane; q++) cost += d.plane_part[((size_t)w * MAX_PLANE + q) * PLANE_PART + PLANE_PART - 2];
    if (ds.use_anchor) cost += d.anchor_part[(size_t)w * ANCHOR_PART + ANCHOR_PART - 2];
    c.cost = cost; c.initial_cost = cost; c.cost_history[0] = cost;
  }
  // Jacobi scaling (iteration 0 only), D = sqrt(clamp(diag)), scaled gradient, Cauchy direction
  double g2 = 0.0, gmax = 0.0, xn2 = 0.0;
  for (int a = t; a < ND; a += blockDim.x) {
    double s = 1.0, dp = 1.0, gt = 0.0, v = 0.0;
    if (ds.act[a]) {
      const double haa = H[(size_t)a * ND + a];
      s = first ? (d.opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(haa)) : 1.0) : gsp[a];
      const double d2 = clamp_diag(s * s * haa);
      dp = sqrt(d2); gt = s * g[a]; v = gt / d2;
      g2 += gt * gt / d2;
      gmax = fmax(gmax, fabs(g[a]));
    }
    if (first) gsp[a] = s;
    gDp[a] = dp; ggts[a] = gt; gvp[a] = v;
    ys[a] = s; ys[ND + a] = v;            // staged for the builds (ys is free until the back-substitution)
  }
  {
    const double *X = d.x + ((size_t)w * 2 + c.cur) * NA;
    for (int b = t; b < GFBE_BLK_COUNT; b += blockDim.x)
      if (ds.blk_free[b]) for (int k = 0; k < blk_gsize(b); k++) { const double v = X[blk_amb(b) + k]; xn2 += v * v; }
  }
  {
    const double pv[3] = {g2, gmax, xn2};
    block_reduce_multi<3>(pv, 0x2u, smem);
    g2 = smem[48]; gmax = smem[49]; xn2 = smem[50];
  }
  __syncthreads();
  STAMP(1);
  const int n = s_nact;                 // dense dims
  const int na = n + 1;                 // + the right-hand side row / column
  const int nt = (na + TB - 1) / TB;
  const int ntile_all = nt * (nt + 1) / 2;
  const bool chain_on = s_nch > 0;
  const double *E = retry_pass ? d.Er + (size_t)w * (NV * NV + NV) : d.E + (size_t)w * NV * NV;
  const double *eg = retry_pass ? E + NV * NV : d.eg + (size_t)w * NV;

  double mu = c.mu;
  bool solved = false, e_valid = true;
  while (mu < GF_MAX_MU) {
    if (!e_valid) {
      if (d.world > 1) {
        if (!retry_pass) { if (t == 0) { c.lin_retry = 1; c.mu = mu; } return; }
        break;
      }
      rebuild_E(d, ds, w, mu, d.E + (size_t)w * NV * NV, d.eg + (size_t)w * NV, false);
      for (int a = t; a < ND; a += blockDim.x) { ys[a] = gsp[a]; ys[ND + a] = gvp[a]; }   // (ys held the previous attempt's solution)
      __syncthreads();
    }
    // ---- build: chain blocks (identity rows for inactive dims), dense tiles, the ring's padding rows
    double vsv = 0.0;
    if (chain_on) {
      for (int e = t; e < 2 * CH_NC * CH_BLK; e += S2_THREADS) {
        const bool isC = e >= CH_NC * CH_BLK;
        const int ee = isC ? e - CH_NC * CH_BLK : e;
        const int k = ee / CH_BLK, i = (ee % CH_BLK) / CH_NB, j = ee % CH_NB;
        double v = 0.0;
        if (!isC) {
          const int a = T_SB(k) + i, b = T_SB(k) + j;
          if (chact[a - T_SB(0)] && chact[b - T_SB(0)]) {
            v = H[(size_t)max(a, b) * ND + min(a, b)] * ys[a] * ys[b];
            if (i == j) { const double dp = gDp[a]; v += mu * dp * dp; }
            vsv = __builtin_fma(v * ys[ND + a], ys[ND + b], vsv);
          } else if (i == j) v = 1.0;
          Ach[ee] = v;
        } else {
          if (k > 0) {
            const int a = T_SB(k) + i, b = T_SB(k - 1) + j;      // a > b: lower triangle of H
            if (chact[a - T_SB(0)] && chact[b - T_SB(0)]) {
              v = H[(size_t)a * ND + b] * ys[a] * ys[b];
              vsv = __builtin_fma(v * ys[ND + a], 2.0 * ys[ND + b], vsv);
            }
          }
          Cch[ee] = v;
        }
      }
      for (int e = t; e < 2 * RING_ROWS * RING_LD; e += S2_THREADS) ring[e] = 0.0;
    }
    vsv += chain_build_dense(tiles, perm, ys, H, E, eg, gsp, gDp, ggts, mu, n, ntile_all, t, S2_THREADS);
    if (t == 0) { flag = 0; s_zzc = 0.0; }
    __syncthreads();
    STAMP(2);
    if (chain_on) {
      // ---- the three-stage pipeline over the chain blocks, one block barrier per block (the roles are separate functions with
      //      their own register allocation; every role passes the same CH_NC + 2 barriers):
      //   step s: wave 0 factorises block NC-1-s | waves 1..2 form Yr of block NC-s | waves 3..7 add Yr^T Yr of block NC+1-s
      if (wave == 0) chain_role((lds_double *)Ach, (lds_double *)Cch, (lds_int *)&flag, lane);
      else if (wave <= S2_COL_WAVES)
        vsv += column_role((const lds_double *)Ach, (const lds_double *)Cch, (lds_double *)ring, (const lds_double *)ys, (const lds_short *)perm,
                           (const lds_int *)s_lo, (const lds_uchar *)chact, (lds_double *)&s_zzc, (const glb_double *)H, (const glb_double *)ggts,
                           (glb_double *)gY, n, t - 64);
      else gemm_role((lds_double *)tiles, (const lds_double *)ring, (const lds_int *)s_lo, ntile_all, wave - 1 - S2_COL_WAVES, lane);
    }
    vsv = block_sum(vsv, red);
    if (t == 0) s_vSv = vsv;
    __syncthreads();
    STAMP(15);
    chol_factor_all<S2_WAVES>((lds_double *)tiles, nt, n, t, (lds_double *)zlast, (lds_int *)&flag, stamp);
    bool ok = (flag == 0);
    if (d.test_fail_chol_iter > 0 && c.iter + 1 == d.test_fail_chol_iter && e_valid && !retry_pass) ok = false;
    STAMP(3);
    if (ok) {
      double zz = 0.0;
      for (int i = t; i < n + TB; i += blockDim.x) {
        const double z = i < n ? (i / TB == n / TB ? zlast[i % TB] : tiles[(size_t)tile_idx(n / TB, i / TB) * (TB * TB) + tsw(n % TB, i % TB)]) : 0.0;
        ys[i] = z;
        zz += z * z;
      }
      zz = block_sum(zz, red);
      if (t == 0) s_zz = zz + s_zzc;
      __syncthreads();
      if (wave == 0) {
        const int cI = lane & 15, part = lane >> 4;
        const int np = (n - 1) / TB;
        for (int P = np; P >= 0; P--) {
          const double *Wt = tiles + (size_t)tile_idx(P, P) * (TB * TB);
          const int r0 = P * TB;
          double s0 = 0.0, s1 = 0.0;
          for (int I = P + 1; I <= np; I++) {
            const double *Tip = tiles + (size_t)tile_idx(I, P) * (TB * TB);
            const int rI = I * TB + 4 * part;
            s0 = __builtin_fma(Tip[tsw(4 * part, cI)], ys[rI], s0);
            s1 = __builtin_fma(Tip[tsw(4 * part + 1, cI)], ys[rI + 1], s1);
            s0 = __builtin_fma(Tip[tsw(4 * part + 2, cI)], ys[rI + 2], s0);
            s1 = __builtin_fma(Tip[tsw(4 * part + 3, cI)], ys[rI + 3], s1);
          }
          double sacc = s0 + s1;
          sacc += __shfl_xor(sacc, 16, 64);
          sacc += __shfl_xor(sacc, 32, 64);
          const double tc = ys[r0 + cI] - sacc;
          double yj = 0.0;
#pragma unroll
          for (int h = 0; h < 4; h++) {
            const int i = 4 * part + h;
            yj = __builtin_fma(Wt[tsw(i, cI)], __shfl(tc, i, 64), yj);
          }
          yj += __shfl_xor(yj, 16, 64);
          yj += __shfl_xor(yj, 32, 64);
          __builtin_amdgcn_wave_barrier();
          if (lane < TB && r0 + lane < n) ys[r0 + lane] = yj;
          __threadfence_block();
          __builtin_amdgcn_wave_barrier();
        }
      }
      __syncthreads();
      // ---- the chain's back-substitution: tch = z_k - Yr_k x_dense for all 99 rows (every wave a share of the rows, the Yr
      //      rows read back from HBM / L2), then x_k = W_k^T (tch_k - Yc_k x_k-1) block by block on one wave
      if (chain_on) {
        constexpr int RPW = (CH_ROWS + S2_WAVES - 1) / S2_WAVES;   // rows per wave
        double part[RPW];
        const double x0 = lane < n ? ys[lane] : 0.0, x1 = lane + 64 < n ? ys[lane + 64] : 0.0;
#pragma unroll
        for (int q = 0; q < RPW; q++) {
          const int r = wave * RPW + q;
          part[q] = 0.0;
          if (r < CH_ROWS) {
            const double *yr = gY + (size_t)r * CH_RW;
            const double a0 = lane < n ? yr[lane] : 0.0, a1 = lane + 64 < n ? yr[lane + 64] : 0.0;
            part[q] = __builtin_fma(a0, x0, a1 * x1);
          }
        }
#pragma unroll
        for (int q = 0; q < RPW; q++) {
          const int r = wave * RPW + q;
          const double sacc = wave_sum(part[q]);
          if (lane == 0 && r < CH_ROWS) tch[r] = gY[(size_t)r * CH_RW + n] - sacc;
        }
        __syncthreads();
        if (wave == 0) {
          const int i = lane < CH_NB ? lane : 0;
          for (int k = 0; k < CH_NC; k++) {
            double uu = tch[k * CH_NB + i];
            if (k > 0) {
              const double *Yc = Cch + k * CH_BLK;
#pragma unroll
              for (int m = 0; m < CH_NB; m++) uu = __builtin_fma(-Yc[i * CH_NB + m], xs[(k - 1) * CH_NB + m], uu);
            }
            const double *Wk = Ach + k * CH_BLK;
            double xv = 0.0;
#pragma unroll
            for (int m = 0; m < CH_NB; m++) xv = __builtin_fma(Wk[m * CH_NB + i], __shfl(uu, m, 64), xv);   // (W_k is lower triangular: the entries above the diagonal are stored zeros)
            if (lane < CH_NB) xs[k * CH_NB + lane] = xv;
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
          }
        }
        __syncthreads();
      }
      // y back to the tangent dims: inactive dims get 0, every dim written once
      int bad = 0;
      for (int i = t; i < n; i += blockDim.x) { const double y = ys[i]; gyp[perm[i]] = y; if (!isfinite(y)) bad = 1; }
      for (int a = t; a < ND; a += blockDim.x) {
        if (dim_in_chain(a)) { const double y = chain_on ? xs[a - T_SB(0)] : 0.0; gyp[a] = y; if (!isfinite(y)) bad = 1; }
        else if (!ds.act[a]) gyp[a] = 0.0;
      }
      if (bad) flag = 1;
      __syncthreads();
      ok = (flag == 0);
    }
    __syncthreads();
    if (ok) { solved = true; break; }
    mu *= GF_MU_INC;
    e_valid = false;
  }
  if (!solved) {
    if (t == 0) { c.done = 1; c.termination = 4; c.status = GFBE_NUMERICAL_FAILURE; c.lin_fail = 1; c.mu = mu; }
    return;
  }
  STAMP(4);
  // dense shares of the dogleg scalars (see k_solve)
  double n2 = 0.0, gyv = 0.0, vrhs = 0.0, vDv = 0.0, vDy = 0.0, vEv = 0.0, vEy = 0.0, yEy = 0.0;
  for (int a = t; a < ND; a += blockDim.x) { ys[a] = gsp[a] * gvp[a]; ys[ND + a] = gsp[a] * gyp[a]; }
  __syncthreads();
  for (int a = t; a < ND; a += blockDim.x) {
    const double d2 = gDp[a] * gDp[a], y = gyp[a], v = gvp[a];
    n2 += d2 * y * y;
    gyv += ggts[a] * y;
    vDv += d2 * v * v;
    vDy += d2 * v * y;
    vrhs += v * (ggts[a] - (a < NV ? gsp[a] * eg[a] : 0.0));
  }
  for (int e = t; e < NV * NV; e += blockDim.x) {
    const int a = e / NV, b = e - a * NV;
    const double ev = E[e];
    vEv = __builtin_fma(ev * ys[a], ys[b], vEv);
    vEy = __builtin_fma(ev * ys[a], ys[ND + b], vEy);
    yEy = __builtin_fma(ev * ys[ND + a], ys[ND + b], yEy);
  }
  {
    const double gv[8] = {n2, gyv, vrhs, vDv, vDy, vEv, vEy, yEy};
    block_reduce_multi<8>(gv, 0u, smem);
    n2 = smem[128]; gyv = smem[129]; vrhs = smem[130]; vDv = smem[131]; vDy = smem[132]; vEv = smem[133]; vEy = smem[134]; yEy = smem[135];
  }
  if (t == 0) {
    const double zz = s_zz, vSv = s_vSv;
    c.mu = mu;
    c.G2 = g2; c.N2 = n2; c.gy = gyv;
    c.vHv = vSv - mu * vDv + vEv;
    c.vHy = vrhs - mu * vDy + vEy;
    c.yHy = zz - mu * n2 + yEy;
    c.grad_max = gmax;
    c.x_norm = xn2;
    c.have_step = 2;
    c.lin_retry = 0;
  }
  STAMP(5);
#undef STAMP
}

static size_t solve_smem_bytes() { const int nt = (ND + 1 + TB - 1) / TB; return sizeof(double) * (size_t)(nt * (nt + 1) / 2) * TB * TB; }
static size_t chain_smem_bytes(int ntile) { return sizeof(double) * ((size_t)ntile * TB * TB + 2 * CH_NC * CH_BLK + 2 * RING_ROWS * RING_LD); }
// dense tiles a window with these active dims needs in k_solve_chain (host side of the kernel's own count)
int solve_chain_tiles(const unsigned char *act) {
  int n = 0;
  for (int a = 0; a < ND; a++) if (act[a] && !dim_in_chain(a)) n++;
  const int nt = (n + 1 + TB - 1) / TB;
  return nt * (nt + 1) / 2;
}
// Per-DEVICE kernel attributes (dynamic LDS above the 64 KB default): set by gfbe_create for the context's device, so that
// contexts on several GPUs of one process all get them (a process-wide "done" flag would cover the first device only).
hipError_t kernels_init_device() {
  const hipError_t e = hipFuncSetAttribute((const void *)k_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_smem_bytes());
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void *)k_solve_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_smem_bytes(S2_MAX_TILES));
}
void launch_solve(const BatchDev &d, hipStream_t s, int retry_pass) {
  if (d.solve_mono) hipLaunchKernelGGL(k_solve, dim3(d.B), dim3(SOLVE_THREADS), solve_smem_bytes(), s, d, retry_pass);
  else hipLaunchKernelGGL(k_solve_chain, dim3(d.B), dim3(S2_THREADS), chain_smem_bytes(d.solve_ntile), s, d, retry_pass);
}
void launch_rebuild_E_shard(const BatchDev &d, hipStream_t s) { hipLaunchKernelGGL(k_rebuild_E_shard, dim3(d.B), dim3(1024), 0, s, d); }

}  // namespace gfd

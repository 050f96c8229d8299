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
// (tools/diag_variants.py, one window / 1024 resident windows): 1024 threads 83.6 us / 48.6k solves/s, 512 threads 83.2 /
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
#define BRM_T t      // (every kernel of this file names its thread index t; only k_solve_chain's may differ from threadIdx.x)
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
__device__ void rebuild_E(const BatchDev &d0, const WinDesc &ds, int w, double mu, double *E, double *eg, bool own_only) {
  const BatchDev d = lin_view(d0, d0.ctl[w].lb);      // (speculative batches: the current set of the linearisation's outputs)
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
            if (x >= T_EX) return d.vis_full ? d.lm_hC[(size_t)(x == T_TD ? 12 : 6 + x - T_EX) * TL + slot] : 0.0;
            const int f = x / 6, q = x % 6;
            const int k = f - s - 1;
            if (f != s && (k < 0 || k >= m)) return 0.0;
            if (d.vis_full) return f == s ? d.lm_hC[(size_t)q * TL + slot] : d.lm_hP[((size_t)k * 6 + q) * TL + slot];
            // compressed rows (k_vis<0, false>): the block of frame f from d (or D), x and the frame's constants (gfbe_devutil.h)
            double dv[3], xl[3], blk[6];
            for (int u = 0; u < 3; u++) {
              dv[u] = f == s ? d.lm_hC[(size_t)u * TL + slot] : d.lm_hP[((size_t)k * 6 + u) * TL + slot];
              xl[u] = d.lm_hC[(size_t)(3 + u) * TL + slot];
            }
            lm_row_block(d.pc + (((size_t)w * 3 + d.ctl[w].cur) * NPAIR + f * (NF + 1)) * PAIR_CONST_DOUBLES + LM_RT_OFF, dv, xl, f == s, blk);
            return blk[q];
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

// The same tile step with the 1/sqrt chain of pivot K + 1 taken OFF the critical path (round 5). The chain of a pivot — the DPP move
// of the diagonal entry, v_rsq_f64 (26 cycles), two Newton steps: eleven DEPENDENT instructions at ~9 cycles each — stood between
// the updates of two pivots, with nothing else to issue (a lone wave issues an independent FP64 instruction every ~5.4 cycles, a
// dependent one every 9). Pivot K + 1 only needs a(K+1, K+1) after the update of pivot K — ONE of its ~16 row updates. So: that
// update first, then the move and the chain of pivot K + 1 are STARTED, and the other updates of pivot K follow in the instruction
// stream; the chain's instructions are ordinary (schedulable) code, the updates volatile assembly in a fixed order: the compiler's
// scheduler spreads the chain between them. The unscaled-inverse half of the updates (columns 0..K of u) is dealt over the wave's
// four 16-lane rows like the passive columns of chain_block — row group g keeps the columns c with c mod 4 == g — which leaves
// (15 - K) + ceil((K + 1) / 4) updates per pivot instead of 16; the groups' columns are gathered when the tile is stored.
// Same operations on every entry as chol_inv_tile16, in the same order: bit-identical results (profiles/ubench/chol_tile_check2.hip).
#ifndef GFBE_TILE_PIPELINED
#define GFBE_TILE_PIPELINED 1
#endif
// One stage of the 1/sqrt chain (rsqrt_refined, the same operations): the value is pinned where the stage is placed in the
// instruction stream — an empty volatile assembly statement keeps its place among the (volatile) updates around it.
template <int S>
__device__ __forceinline__ void rsq_stage(const double d, double &nhd, double &r, double &t) {
  // (volatile assembly: the stage stays exactly where it is placed among the updates. An ordinary statement pinned by an empty
  //  assembly statement is either clustered with the other stages by the scheduler — pin that only reads — or, pin that redefines,
  //  followed by a wait state per use, the hazard recogniser knowing nothing about the definition)
  const double c15 = 1.5;
  if (S == 0) asm volatile("v_rsq_f64 %0, %2\n\tv_mul_f64 %1, %2, -0.5" : "=&v"(r), "=&v"(nhd) : "v"(d));
  else if (S == 1) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t) : "v"(nhd), "v"(r));
  else if (S == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(t) : "v"(r), "v"(c15));
  else if (S == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r) : "v"(t));
  else if (S == 4) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t) : "v"(nhd), "v"(r));
  else if (S == 5) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(t) : "v"(r), "v"(c15));
  else if (S == 6) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r) : "v"(t));
}
// stage S of the next pivot's chain goes behind update number RSQ_AT[S] of the current pivot (v_rsq_f64 takes 26 cycles, a dependent
// FP64 instruction 9, an update issues in ~5.4): whatever does not fit between the updates follows them
__device__ constexpr int rsq_at(int S) { return S == 0 ? 0 : 3 + 2 * S; }
template <int K>
__device__ __forceinline__ void chol_inv_step_p(double (&row)[TB], double (&u)[TB / 4], int li, int g, double &myinv, double &inv, int zrow, lds_double *zout) {
  const double lik = row[K] * inv;          // L[i][k] for i > k
  double m = (li > K) ? -(lik * inv) : 0.0; // -a_ik / d for the rows below the pivot; rows <= k are final
  myinv = (li == K) ? inv : myinv;
  // column K of the unscaled inverse starts as e_K in the group that keeps it (rows above K never touch it)
  u[K / 4] = (g == (K & 3) && li == K) ? 1.0 : u[K / 4];
  asm volatile("" : "+v"(u[K / 4]), "+v"(m));   // materialised HERE (two wait states before a DPP instruction reads a VALU result)
  if (zrow >= 0) zout[K] = lane_bcast(lik, zrow);
  constexpr bool NEXT = K + 1 < TB;
  double dkk = 1.0, nhd = 0.0, r = inv, t = 0.0;
  if (NEXT) {
    dpp_fmac<K>(row[K + 1 < TB ? K + 1 : K], m);
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(dkk) : "v"(row[K + 1 < TB ? K + 1 : K]), "n"((K + 1) & 15));
    rsq_stage<0>(dkk, nhd, r, t);
  }
  constexpr int NR = K + 2 < TB ? TB - K - 2 : 0, NU = K / 4 + 1, NF = NR + NU;      // the other updates of this pivot: rows, then this group's inverse columns
#pragma unroll
  for (int f = 0; f < NF; f++) {
    if (f < NR) dpp_fmac<K>(row[K + 2 + (f < NR ? f : 0)], m);
    else dpp_fmac<K>(u[f - NR < NU ? (f >= NR ? f - NR : 0) : 0], m);      // (this group's columns 4 q + g <= K; a register whose column is > K holds an exact zero in every row)
    if (NEXT) {
      if (f + 1 == rsq_at(1)) rsq_stage<1>(dkk, nhd, r, t);
      if (f + 1 == rsq_at(2)) rsq_stage<2>(dkk, nhd, r, t);
      if (f + 1 == rsq_at(3)) rsq_stage<3>(dkk, nhd, r, t);
      if (f + 1 == rsq_at(4)) rsq_stage<4>(dkk, nhd, r, t);
      if (f + 1 == rsq_at(5)) rsq_stage<5>(dkk, nhd, r, t);
      if (f + 1 == rsq_at(6)) rsq_stage<6>(dkk, nhd, r, t);
    }
  }
  if (NEXT) {      // the stages that did not fit between the updates
    if (NF < rsq_at(1)) rsq_stage<1>(dkk, nhd, r, t);
    if (NF < rsq_at(2)) rsq_stage<2>(dkk, nhd, r, t);
    if (NF < rsq_at(3)) rsq_stage<3>(dkk, nhd, r, t);
    if (NF < rsq_at(4)) rsq_stage<4>(dkk, nhd, r, t);
    if (NF < rsq_at(5)) rsq_stage<5>(dkk, nhd, r, t);
    if (NF < rsq_at(6)) rsq_stage<6>(dkk, nhd, r, t);
    inv = r;
  }
}
__device__ __forceinline__ bool chol_inv_tile16_p(lds_double *T, int lane, int zrow, lds_double *zout, double *stamp = nullptr) {
  double row[TB], u[TB / 4];
  const int li = lane & 15, g = lane >> 4;
  if (stamp && lane == 0) stamp[21] = (double)wall_clock64();
  int lio = li;
  asm volatile("" : "+v"(lio));               // (opaque: 16 loop-invariant tile addresses hoisted out of the panel loop would be spilled)
#pragma unroll
  for (int q = 0; q < TB; q++) row[q] = T[tsw(lio, q)];
#pragma unroll
  for (int q = 0; q < TB / 4; q++) u[q] = 0.0;
  double myinv = 0.0, inv, d00;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(d00) : "v"(row[0]));
  inv = rsqrt_refined(d00);
  chol_inv_step_p<0>(row, u, li, g, myinv, inv, zrow, zout);   chol_inv_step_p<1>(row, u, li, g, myinv, inv, zrow, zout);
  chol_inv_step_p<2>(row, u, li, g, myinv, inv, zrow, zout);   chol_inv_step_p<3>(row, u, li, g, myinv, inv, zrow, zout);
  chol_inv_step_p<4>(row, u, li, g, myinv, inv, zrow, zout);   chol_inv_step_p<5>(row, u, li, g, myinv, inv, zrow, zout);
  chol_inv_step_p<6>(row, u, li, g, myinv, inv, zrow, zout);   chol_inv_step_p<7>(row, u, li, g, myinv, inv, zrow, zout);
  chol_inv_step_p<8>(row, u, li, g, myinv, inv, zrow, zout);   chol_inv_step_p<9>(row, u, li, g, myinv, inv, zrow, zout);
  chol_inv_step_p<10>(row, u, li, g, myinv, inv, zrow, zout); chol_inv_step_p<11>(row, u, li, g, myinv, inv, zrow, zout);
  chol_inv_step_p<12>(row, u, li, g, myinv, inv, zrow, zout); chol_inv_step_p<13>(row, u, li, g, myinv, inv, zrow, zout);
  chol_inv_step_p<14>(row, u, li, g, myinv, inv, zrow, zout); chol_inv_step_p<15>(row, u, li, g, myinv, inv, zrow, zout);
  if (stamp && lane == 0) stamp[22] = (double)wall_clock64();
  // lane (g, i) stores the columns 4 q + g of row i of W (exact zeros above the diagonal: a column starts as e_c and rows < c never touch it)
#pragma unroll
  for (int q = 0; q < TB / 4; q++) T[tsw(lio, 4 * q + g)] = u[q] * myinv;
  if (stamp && lane == 0) stamp[23] = (double)wall_clock64();
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
        vsv = __builtin_fma(v * ys[NC + aa[u]], ys[NC + bb[u]] * (offdiag[u] ? 2.0 : 1.0), vsv);
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
// (VAR: an instance of its own for k_solve_chain_wide — an out-of-line function shared by kernels of different launch bounds is compiled to the
//  most permissive of them, and the two-workgroups-per-CU kernels then exceed their register budget)
template <int NWAVES, int VAR = 0>
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
#if GFBE_TILE_PIPELINED
        if (!chol_inv_tile16_p(C, lane, P + 2 == nt ? n % TB : -1, zlast, P == 0 ? stamp : nullptr) && lane == 0) *flag = 1;
#else
        if (!chol_inv_tile16(C, lane, P + 2 == nt ? n % TB : -1, zlast, P == 0 ? stamp : nullptr) && lane == 0) *flag = 1;
#endif
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
  __shared__ short perm[NC + TB];     // (16-bit: the 160 KB of LDS are full — 78 tiles of 2 KB for a fully active window)
  __shared__ double red[16], ys[2 * NC + TB];   // (NC: this kernel only ever sees batches without GNSS dims, d.nu == NC — with ND the
                                                // static LDS would push the 78 tiles past the CU's 160 KB)
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
    const bool on = (t < NC) && ds.act[t];
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
    for (int a = s_nact + t; a < NC + TB; a += blockDim.x) perm[a] = -1;
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
  for (int a = t; a < NC; a += blockDim.x) {
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
    block_reduce_multi<3>(pv, 0x2u, smem, BRM_T);
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
  int att = 0;      // factorisation attempts of this linearisation so far (landmark sharding: the pass index counts them)
  while (mu < GF_MAX_MU) {
    if (!e_valid) {
      if (d.sharded) {
        // landmark sharding: E for the larger mu needs every rank's landmarks — hand the window to the retry pass (rebuild on
        // all ranks, one all-reduce, k_solve again); a second failure is a failed linear solve
        if (retry_pass < min(max(d.opt.sharded_mu_retries, 0), 8)) { if (t == 0) { c.lin_retry = 1; c.mu = mu; } return; }     // (on to pass retry_pass + 1)
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
    for (int a = t; a < NC; a += blockDim.x) { ys[a] = gsp[a]; ys[NC + a] = gvp[a]; }
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
    if (d.test_fail_chol_iter > 0 && c.iter + 1 == d.test_fail_chol_iter && (d.sharded ? retry_pass : att) < max(d.opt.test_fail_chol_count, 1)) ok = false;   // fault injection: first attempt of that iteration
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
      for (int a = t; a < NC; a += blockDim.x) if (!ds.act[a]) gyp[a] = 0.0;
      if (bad) flag = 1;
      __syncthreads();
      ok = (flag == 0);
    }
    __syncthreads();
    if (ok) { solved = true; break; }
    mu *= GF_MU_INC;
    e_valid = false;
    att++;
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
  for (int a = t; a < NC; a += blockDim.x) { ys[a] = gsp[a] * gvp[a]; ys[NC + a] = gsp[a] * gyp[a]; }   // s v, s y (original dims)
  __syncthreads();
  for (int a = t; a < NC; a += blockDim.x) {
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
    vEy = __builtin_fma(ev * ys[a], ys[NC + b], vEy);
    yEy = __builtin_fma(ev * ys[NC + a], ys[NC + b], yEy);
  }
  {   // (the tiles are dead after the back-substitution: scratch of the combined reduction)
    const double gv[8] = {n2, gyv, vrhs, vDv, vDy, vEv, vEy, yEy};
    block_reduce_multi<8>(gv, 0u, smem, BRM_T);
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
//   chain, block k = 10 .. 0 (wave 0, 9 pivots per block, in registers through 64-bit DPP row broadcasts — chain_block):
//       W_k = chol(A_k)^-1                    A_k: the block's 9 x 9 diagonal block (already downdated by block k + 1)
//       Yc_k = W_k C_k                        C_k = S(SB_k, SB_k-1): the row operations of the factorisation applied to C_k's
//       A_k-1 -= Yc_k^T Yc_k                                         columns as well, like the identity that becomes W_k
//   wide rows (waves 1..3 one step behind, FP64 matrix cores, 16 dense columns per tile; wide_role):
//       R_k' = R_k - Yc_k+1^T Yr_k+1          R_k = S(SB_k, dense | rhs)     [one 16x16x4 chain: A = -Yc^T, B = Yr_k+1, C = R_k]
//       Yr_k = W_k R_k'                       (column nd of Yr_k = z_k)      [one more: A = W_k, B = R_k' — the accumulator
//                                                                             layout of the first IS the B layout of the second]
//   dense update (the same waves, one more step behind; accumulators in registers over all blocks):
//       D -= sum_k Yr_k^T Yr_k                (augmented with the right-hand side row, like k_solve's tiles; the wave that forms
//                                              a column tile of Yr_k holds the operand of its tile row in registers)
//   then the blocked Cholesky of k_solve on the 5 x 5 tiles of D (chol_factor_all), its back-substitution, and the chain's:
//       x_k = W_k^T (z_k - Yr_k x_dense) - G_k x_k-1,  k = 0 .. 10,   G_k = W_k^T Yc_k formed inside the pipeline,
//   i.e. one 9 x 9 matrix-vector product per block on the sequential path (DPP row broadcasts, no LDS round trip).
//
// One block barrier per chain block. Fill stays inside the structure: Yr_k only reaches the poses of frames >= k - 1 (lo_k)
// besides the extrinsic / intrinsic columns; column tiles and tile products below lo_k are skipped. 15 tiles + chain blocks +
// a two-block ring of Yr = ~67 KB of LDS and 4 waves: two workgroups per CU; the Yr rows (80 KB per window, transposed) go
// to HBM / L2 for the back-substitution. Inactive speed-bias dims (constant block, window still filling up, USE_IMU = 0)
// are identity rows of their block.
// =============================================================================================
#define S2_THREADS 256
#ifndef GFBE_ROLE_INLINE
#define GFBE_ROLE_INLINE 0
#endif
#ifndef GFBE_WIDE_PREFETCH2
#define GFBE_WIDE_PREFETCH2 0      // k_solve_chain's wide rows: the rows of S two blocks ahead instead of one (measured: 86.9 against 84.4 us per 512
                                   // windows — the pipeline does not wait for those loads)
#endif
#ifndef GFBE_SOLVE_WIDE
#define GFBE_SOLVE_WIDE 1          // batches with GNSS dims on k_solve_chain_wide where they fit (0: k_solve_big for all of them, rounds 3-5)
#endif
#ifndef GFBE_CHAIN_PRIO
#define GFBE_CHAIN_PRIO 0          // k_solve_chain: s_setprio 3 for wave 0 and GFBE_CHAIN_PRIO - 1 for the wide waves (0: the priorities are left alone)
#endif
#ifndef GFBE_CHAIN_SIMD_ROLES
#define GFBE_CHAIN_SIMD_ROLES 0    // k_solve_chain's waves numbered by the SIMD they sit on (measured: the pipeline 33.3 -> 31.5 us beside a second
                                   // workgroup, the dense Cholesky 15.8 -> 19.2: 89.6 against 85.0 us per 512 windows; tools/diag_scripts/hwid)
#endif
#if GFBE_ROLE_INLINE
#define GFBE_ROLE_FN __forceinline__
#else
#define GFBE_ROLE_FN __noinline__
#endif
#define S2_WAVES (S2_THREADS >> 6)
enum { CH_NB = 9, CH_NC = NF, CH_ROWS = CH_NB * CH_NC, CH_BLK = CH_NB * CH_NB, RING_ROWS = 12,
       S2_MAX_NT = 6, S2_MAX_TILES = S2_MAX_NT * (S2_MAX_NT + 1) / 2, GYT_LD = 104, GYT_COLS = TB * S2_MAX_NT,
       S2_WIDE_NT = 9, S2_WIDE_TILES = S2_WIDE_NT * (S2_WIDE_NT + 1) / 2 };      // k_solve_chain_wide (round 6): the GNSS dims as dense columns
// row stride of the Yr ring: >= 16 x tile columns and = 16 (mod 32) doubles, so that the two row groups a half-wave reads as a
// matrix-core operand fall into disjoint LDS banks (five tile columns -> 80, six -> 112)
__host__ __device__ inline int chain_ring_ld(int ntile) { return ntile <= 15 ? 80 : 112; }
enum { CHAIN_RING_LD_WIDE = 176 };      // nine tile columns (k_solve_chain_wide): 144 + 32, = 16 (mod 32)
__host__ __device__ inline bool dim_in_chain(int a) { return a >= T_SB(0) && a < T_SB(0) + CH_ROWS; }
typedef __attribute__((address_space(3))) unsigned char lds_uchar;

template <int KK>
__device__ __forceinline__ void dpp_fmac2(double &acc, double bc, double m) {   // acc += m * (lane KK of the 16-lane row)'s bc
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bc), "v"(m), "n"(KK));
}
// One block of the chain by one wave. Lane (g, i) — 16-lane row g of the wave, lane i of the row — owns row i of the block:
// every row group keeps the 9 entries of the diagonal block A_k (the pivot column must be in every group), and the 18
// "passive" columns that only ride along — the identity that becomes W_k and the coupling block C_k that becomes Yc_k — are
// dealt over the four groups: register q of group g holds passive column 4 q + g (columns 0..8: W, 9..17: Yc). A pivot is then
// ~14 instructions of the 1/sqrt chain + (8 - K) + 5 row updates (v_fmac_f64_dpp, the pivot row travels inside the FMA) instead
// of (8 - K) + 18. The downdate of A_k-1 is dealt the same way: group g forms the columns of A_k-1 whose Yc columns it holds.
// Hazards the compiler does not see inside inline assembly: chol_inv_step.
//   A: in A_k (downdated), out W_k = L_k^-1 (lower triangular, zeros above the diagonal)     Cb: in C_k, out Yc_k (has_next)
//   An: A_k-1, downdated in place (has_next). Returns false on a pivot that is not positive and finite.
#define CH_NP 5          // passive registers per lane: ceil(18 / 4)
template <int K>
__device__ __forceinline__ void chain_pivot(double (&row)[CH_NB], double (&p)[CH_NP], int li, double &myinv) {
  double dkk;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(dkk) : "v"(row[K]), "n"(K));
  const double inv = rsqrt_refined(dkk);
  const double lik = row[K] * inv;
  double m = (li > K) ? -(lik * inv) : 0.0;
  myinv = (li == K) ? inv : myinv;
  asm volatile("" : "+v"(m));
#pragma unroll
  for (int j = K + 1; j < CH_NB; j++) dpp_fmac<K>(row[j], m);
#pragma unroll
  for (int q = 0; q < CH_NP; q++) dpp_fmac<K>(p[q], m);
}
template <int K> struct ChainPivots {
  static __device__ __forceinline__ void run(double (&row)[CH_NB], double (&p)[CH_NP], int li, double &myinv) {
    chain_pivot<K>(row, p, li, myinv);
    if constexpr (K + 1 < CH_NB) ChainPivots<K + 1>::run(row, p, li, myinv);
  }
};
template <int M> struct ChainDowndate {   // a[q] -= Yc[m][i] * Yc[m][j_q], one row m of Yc at a time (q = 2..4: the Yc columns of this group)
  static __device__ __forceinline__ void run(double (&a)[3], const double (&p)[CH_NP], const double (&nyc)[CH_NB]) {
#pragma unroll
    for (int q = 0; q < 3; q++) dpp_fmac2<M>(a[q], p[2 + q], nyc[M]);
    if constexpr (M + 1 < CH_NB) ChainDowndate<M + 1>::run(a, p, nyc);
  }
};
__device__ __forceinline__ bool chain_block(lds_double *A, lds_double *Cb, lds_double *An, int lane, int has_next) {
  const int li = lane & 15, g = lane >> 4;
  const bool mine = li < CH_NB;
  int lio = mine ? li : 0;
  asm volatile("" : "+v"(lio));
  double row[CH_NB], p[CH_NP];
#pragma unroll
  for (int q = 0; q < CH_NB; q++) { const double a = A[lio * CH_NB + q]; row[q] = mine ? a : 0.0; }
#pragma unroll
  for (int q = 0; q < CH_NP; q++) {
    const int idx = 4 * q + g;                                         // passive column of this register
    double v = (idx == li) ? 1.0 : 0.0;                                // columns 0..8: the identity
    if (idx >= CH_NB) { const double cv = (has_next && idx < 2 * CH_NB) ? Cb[lio * CH_NB + min(idx, 2 * CH_NB - 1) - CH_NB] : 0.0; v = cv; }
    p[q] = mine ? v : 0.0;
  }
  double myinv = 0.0;
  // (definitions pinned: a VALU write needs two wait states before a DPP instruction reads the register and the compiler does not
  //  look into inline assembly; EXEC may have been rewritten by masked code: five wait states before the first DPP instruction)
#pragma unroll
  for (int q = 0; q < CH_NB; q++) asm volatile("" : "+v"(row[q]));
#pragma unroll
  for (int q = 0; q < CH_NP; q++) asm volatile("" : "+v"(p[q]));
  asm volatile("s_nop 4" ::: "memory");
  ChainPivots<0>::run(row, p, li, myinv);
#pragma unroll
  for (int q = 0; q < CH_NP; q++) p[q] *= myinv;
#pragma unroll
  for (int q = 0; q < CH_NP; q++) {
    const int idx = 4 * q + g;
    lds_double *dst = idx < CH_NB ? A + lio * CH_NB + idx : Cb + lio * CH_NB + (idx - CH_NB);
    if (mine && (idx < CH_NB || (has_next && idx < 2 * CH_NB))) *dst = p[q];
  }
  const bool ok = __ballot(mine && !((myinv > 0.0) && (myinv < 1.0e300))) == 0ull;
  if (has_next) {
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    double a[3], nyc[CH_NB];
    int jq[3];
#pragma unroll
    for (int q = 0; q < 3; q++) { jq[q] = 4 * (2 + q) + g - CH_NB; const bool on = jq[q] >= 0 && jq[q] < CH_NB; a[q] = on ? An[lio * CH_NB + (on ? jq[q] : 0)] : 0.0; }
#pragma unroll
    for (int q = 0; q < CH_NB; q++) nyc[q] = -Cb[q * CH_NB + lio];       // column i of Yc_k
#pragma unroll
    for (int q = 0; q < 3; q++) asm volatile("" : "+v"(a[q]));
#pragma unroll
    for (int q = 0; q < CH_NB; q++) asm volatile("" : "+v"(nyc[q]));
#pragma unroll
    for (int q = 0; q < CH_NP; q++) asm volatile("" : "+v"(p[q]));
    asm volatile("s_nop 4" ::: "memory");
    ChainDowndate<0>::run(a, p, nyc);
#pragma unroll
    for (int q = 0; q < 3; q++) if (mine && jq[q] >= 0 && jq[q] < CH_NB) An[lio * CH_NB + jq[q]] = a[q];
  }
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
  return ok;
}

// Block barrier of the pipeline: LDS traffic only. __syncthreads() also drains the vector-memory counter — every step would wait
// for its Yr stores to reach L2 and for the NEXT block's prefetched coupling rows (~2 us per step measured).
#ifndef GFBE_CHAIN_STAMP
#define GFBE_CHAIN_STAMP 0      // diagnostics build (tools/diag_chain.py): per-step time stamps of the pipeline roles into the NEXT window's timing slots
#endif
#if GFBE_CHAIN_STAMP
#define RSTAMP(cond, i) do { if (cond) rstamp[i] = (double)wall_clock64(); } while (0)
#else
#define RSTAMP(cond, i) do { } while (0)
#endif
#define CH_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// The chain's back-substitution along one segment, one wave: x_k0 is final; x_k = a_k - G_k x_(k - dk) for the cnt blocks k = k0 + dk,
// k0 + 2 dk, ... (a_k in xs on entry, x_k on exit; G_k in block k's coupling slot). One 9 x 9 matrix-vector product per block on the
// sequential path, the vector travelling by DPP row broadcasts.
__device__ __forceinline__ void chain_backsub(const lds_double *Cch, lds_double *xs, int k0, int dk, int cnt, int lane) {
  const int li = lane & 15;
  const int lio = li < CH_NB ? li : 0;
  double x = xs[k0 * CH_NB + lio];
  double gn[CH_NB], gc[CH_NB];
#pragma unroll
  for (int m = 0; m < CH_NB; m++) gn[m] = -Cch[(k0 + dk) * CH_BLK + lio * CH_NB + m];
  for (int j = 1; j <= cnt; j++) {
    const int k = k0 + dk * j;
    double acc = xs[k * CH_NB + lio];
#pragma unroll
    for (int m = 0; m < CH_NB; m++) gc[m] = gn[m];
    if (j < cnt) {
#pragma unroll
      for (int m = 0; m < CH_NB; m++) gn[m] = -Cch[(k + dk) * CH_BLK + lio * CH_NB + m];
    }
#pragma unroll
    for (int m = 0; m < CH_NB; m++) asm volatile("" : "+v"(gc[m]));
    asm volatile("" : "+v"(x), "+v"(acc));
    asm volatile("s_nop 4" ::: "memory");
    dpp_fmac2<0>(acc, x, gc[0]); dpp_fmac2<1>(acc, x, gc[1]); dpp_fmac2<2>(acc, x, gc[2]);
    dpp_fmac2<3>(acc, x, gc[3]); dpp_fmac2<4>(acc, x, gc[4]); dpp_fmac2<5>(acc, x, gc[5]);
    dpp_fmac2<6>(acc, x, gc[6]); dpp_fmac2<7>(acc, x, gc[7]); dpp_fmac2<8>(acc, x, gc[8]);
    x = acc;
    if (lane < CH_NB) xs[k * CH_NB + lane] = x;
  }
}
// ---- the roles of the chain pipeline (k_solve_chain, k_solve_chain_tw). Each is out of line — its own register allocation — and each
// passes exactly ChainCfg<TW>::STEPS block barriers.
//
// Chain SEGMENTS. Classic (TW = 0, k_solve_chain): ONE segment, blocks CH_NC-1 down to 0 — eleven sequential 9-pivot steps.
// Twisted (TW = 1, k_solve_chain_tw: the latency variant of small batches): the block-tridiagonal chain is eliminated from BOTH ends
// at once by two chain waves — segment 0: blocks CH_NC-1 down to CH_MID, segment 1: blocks 0 up to CH_MID-1 — and the middle block
// is the last one of segment 0, downdated by both of its neighbours (segment 1 leaves its downdate in `Amid`): six sequential steps
// instead of eleven. Every segment has its own wide waves (wide_role_tw); the middle block's wide row takes the second product
// -Yc'_(CH_MID-1)^T Yr_(CH_MID-1) of the other segment's last block.
// The coupling slot of block k holds S(SB_k, successor of k in its segment): SB_k-1 in segment 0, SB_k+1 in segment 1.
enum { CH_MID = 5 };
template <int TW> struct ChainCfg {
  static constexpr int NB0 = TW ? CH_NC - CH_MID : CH_NC;     // blocks of segment 0 (descending from CH_NC - 1)
  static constexpr int NB1 = TW ? CH_MID : 0;                  // blocks of segment 1 (ascending from 0)
  static constexpr int STEPS = TW ? NB0 + 1 : NB0 + 2;         // block barriers of the pipeline (chain | wide rows | classic: the dense update one more step behind)
};
template <int TW, int SEG, int VAR = 0>
__device__ GFBE_ROLE_FN void chain_role(lds_double *Ach, lds_double *Cch, lds_double *Amid, lds_int *flag, int lane, double *rstamp) {
  constexpr int NB = SEG == 0 ? ChainCfg<TW>::NB0 : ChainCfg<TW>::NB1;
  for (int s = 0; s < ChainCfg<TW>::STEPS; s++) {
    RSTAMP(lane == 0 && SEG == 0, 32 + s);
    if (s < NB) {
      const int k = SEG == 0 ? CH_NC - 1 - s : s;
      const bool has_next = SEG == 0 ? s + 1 < NB : true;
      lds_double *An = SEG == 0 ? Ach + (k > 0 ? k - 1 : 0) * CH_BLK : (k + 1 == CH_MID ? Amid : Ach + (k + 1) * CH_BLK);
      if (TW && SEG == 0 && k == CH_MID) {          // the middle block: segment 1's downdate (written one barrier ago)
        for (int e = lane; e < CH_BLK; e += 64) Ach[k * CH_BLK + e] += Amid[e];
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
      }
      if (!chain_block(Ach + k * CH_BLK, Cch + k * CH_BLK, An, lane, has_next) && lane == 0) *flag = 1;
    }
    RSTAMP(lane == 0 && SEG == 0 && s < 6, 46 + s);
    CH_LDS_BARRIER();
  }
}
// Wide rows and the dense update, waves 1..3 (wv = 0..2): wave wv forms the column tiles wv and wv + 3 of Yr (16 dense columns each,
// the right-hand side is column n) and adds every third tile of the dense update D -= sum_k Yr_k^T Yr_k one step later, both
// operands from the two-block ring of Yr in LDS. Lane (lr, lk) holds rows lk, lk + 4, lk + 8 of column 16 c + lr — the operand
// layout of v_mfma_f64_16x16x4_f64 for every product here and the layout of its result, so Yr_k+1 never leaves the registers.
// A lone wave issues ONE instruction every ~5.4 cycles whatever it is (profiles/ubench/dp_issue_rate_mi355x.txt), so the loop is
// written for instruction count and without divergent branches: pointers into H, into the chain blocks and into the transposed
// Yr rows advance by a per-lane constant per block; lanes outside a 9 x 9 block read a zero slot of LDS (stride 0) and write to a
// dump slot. (Inactive speed-bias dims: H holds exact zeros there.)
// Returns the lane's share of v^T S v over the coupling entries it loads; |z_chain|^2 goes to *zzc_out.
#define S2_WIDE_WAVES 3
enum { CH_ZERO = 96 };     // doubles behind the chain blocks: a zero slot (first half: parked operand reads reach 36 doubles in) and a dump slot
// MAXNT: tile columns of the dense part the instance holds (6: k_solve_chain; 9: k_solve_chain_wide) — NU column tiles of Yr and TPW dense tiles per wave
template <int MAXNT>
__device__ GFBE_ROLE_FN double wide_role(lds_double *tiles, lds_double *Ach, lds_double *Cch, lds_double *ring, lds_double *zslot, const lds_double *sS,
                                         const lds_double *vS, const lds_double *rS, const lds_short *perm, const lds_int *s_lo, lds_double *zzc_out,
                                         const glb_double *H, glb_double *gYT, int n_, int nt_, int ring_ld_, int wv_, int lane, double *rstamp) {
  // (wave-uniform values in scalar registers: the compiler cannot see that they are uniform — they come from LDS / the thread
  //  index — and would turn every branch on them into an EXEC-masked region)
  const int n = __builtin_amdgcn_readfirstlane(n_), nt = __builtin_amdgcn_readfirstlane(nt_);
  const int ring_ld = __builtin_amdgcn_readfirstlane(ring_ld_), wv = __builtin_amdgcn_readfirstlane(wv_);
  constexpr int NU = (MAXNT + S2_WIDE_WAVES - 1) / S2_WIDE_WAVES, S2_TPW = (MAXNT * (MAXNT + 1) / 2 + S2_WIDE_WAVES - 1) / S2_WIDE_WAVES, GYTC = TB * MAXNT;
  const int lr = lane & 15, lk = lane >> 4, na = n + 1;
  const bool in01 = lr < CH_NB, in2 = lr < CH_NB && lk == 0;
  // ---- per-lane state of the two column tiles
  bool on[NU];
  int jc[NU];
  double sbj[NU], vbj2[NU], mrhs[NU];
  const glb_double *pH[NU];        // S-source entry (SB_k row lk, column): H[a * ND + b] for b below the speed-bias dims, H[b * ND + a] above
  long dH[NU];
  glb_double *pY[NU], *pY8[NU];     // transposed Yr rows: rows lk, lk + 4 | row 8 (lk == 0) — or the dump column
  lds_double *pR[NU];              // ring slot (row lk, column 16 c + lr)
#pragma unroll
  for (int u = 0; u < NU; u++) {
    const int c = wv + S2_WIDE_WAVES * u;
    on[u] = c < nt;
    jc[u] = TB * c + lr;
    const bool col_on = on[u] && jc[u] < na, dense_col = on[u] && jc[u] < n;
    const int bj = dense_col ? perm[jc[u]] : 0;
    sbj[u] = dense_col ? sS[bj] : 0.0; vbj2[u] = dense_col ? 2.0 * vS[bj] : 0.0;
    mrhs[u] = (on[u] && jc[u] == n) ? 1.0 : 0.0;
    const int a0 = T_SB(CH_NC - 1) + lk;
    dH[u] = bj < T_SB(0) ? ND : 1;
    pH[u] = H + (bj < T_SB(0) ? (size_t)a0 * ND + bj : (size_t)bj * ND + a0);
    glb_double *col = gYT + (size_t)(col_on ? jc[u] : GYTC - 1) * GYT_LD + (CH_NC - 1) * CH_NB;    // (column 95 is never a system column: the dump)
    pY[u] = col + lk;
    pY8[u] = (col_on && lk == 0) ? col + 8 : gYT + (size_t)(GYTC - 1) * GYT_LD + (CH_NC - 1) * CH_NB + 8;
    pR[u] = ring + lk * ring_ld + min(TB * c, ring_ld - TB) + lr;
  }
  double yv[NU][3], rpre[NU][3], rnxt[NU][3], vsv = 0.0, zzc = 0.0;
#pragma unroll
  for (int u = 0; u < NU; u++)
#pragma unroll
    for (int kk = 0; kk < 3; kk++) { yv[u][kk] = 0.0; rpre[u][kk] = 0.0; rnxt[u][kk] = 0.0; }
  // rows lk, lk + 4, lk + 8 of block k (the third only counts for lk == 0: the others read into the next block and are masked)
  // (row 8 + lk exists for lk == 0 only: the other lanes' third load lands in the next block or above the diagonal, on entries nobody
  //  writes — H is not cleared at upload — and is replaced by zero, not multiplied by it)
  auto load_R = [&](int u, double (&r)[3]) {
    const double r2 = pH[u][8 * dH[u]];
    r[0] = pH[u][0]; r[1] = pH[u][4 * dH[u]]; r[2] = lk == 0 ? r2 : 0.0;
    pH[u] -= CH_NB * dH[u];
  };
#pragma unroll
  for (int u = 0; u < NU; u++) if (on[u]) load_R(u, rpre[u]);
#if GFBE_WIDE_PREFETCH2
  // (round 6) TWO blocks ahead: beside a second workgroup on the CU — and the batch's other parts' kernels — a load takes longer than a step
  // of the pipeline (stamps, B = 1 / 256 / 512: the pipeline 19.6 / 24.8 / 33.3 us), so the rows requested one step ahead paced the steps
#pragma unroll
  for (int u = 0; u < NU; u++) if (on[u]) load_R(u, rnxt[u]);
#endif
  // operand pointers into the chain blocks (block CH_NC - 1 first; per-lane stride, 0 for the lanes parked on the zero slot)
  //   a2[kk] = W_k[lr][4 kk + lk]        a1[kk] = -Yc_k+1[4 kk + lk][lr]      (G: W_kG[4 kk + lk][lr], Yc_kG[4 kk + lk][lr])
  const int sW01 = in01 ? CH_BLK : 0, sW2 = in2 ? CH_BLK : 0;
  const lds_double *pW01 = in01 ? Ach + (CH_NC - 1) * CH_BLK + lr * CH_NB + lk : zslot;
  const lds_double *pW2 = in2 ? Ach + (CH_NC - 1) * CH_BLK + lr * CH_NB + 8 : zslot;
  const lds_double *pC01 = in01 ? Cch + (CH_NC - 1) * CH_BLK + lk * CH_NB + lr : zslot;      // (block k + 1 = CH_NC - 1 is first used at the second step)
  const lds_double *pC2 = in2 ? Cch + (CH_NC - 1) * CH_BLK + 8 * CH_NB + lr : zslot;
  const lds_double *pT01 = in01 ? Ach + (CH_NC - 1) * CH_BLK + lk * CH_NB + lr : zslot;      // W transposed access for G (block kG)
  const lds_double *pT2 = in2 ? Ach + (CH_NC - 1) * CH_BLK + 8 * CH_NB + lr : zslot;
  lds_double *pG01 = in01 ? Cch + (CH_NC - 1) * CH_BLK + lk * CH_NB + lr : zslot + CH_ZERO / 2;   // G output (block kG), or the dump half of the slot
  lds_double *pG2 = in2 ? Cch + (CH_NC - 1) * CH_BLK + 8 * CH_NB + lr : zslot + CH_ZERO / 2;
  const double m2 = lk == 0 ? 1.0 : 0.0;     // (row 8 + lk exists for lk == 0 only)
  // ---- dense update: tiles e = wv, wv + 3, ... of the lower triangle
  dbl4 acc[S2_TPW];
  int oI[S2_TPW], oJ[S2_TPW], tJ1[S2_TPW];
#pragma unroll
  for (int q = 0; q < S2_TPW; q++) {
    acc[q] = dbl4{0.0, 0.0, 0.0, 0.0};
    const int e = wv + S2_WIDE_WAVES * q;
    int I = 0, J = 0;
    const bool v = e < nt * (nt + 1) / 2;
    if (v) tri_decode(e, I, J);
    oI[q] = TB * I; oJ[q] = TB * J; tJ1[q] = v ? TB * (J + 1) : 0;     // (tJ1 = 0: never above lo_k >= 0 -> skipped)
  }
  const int ro = lk * ring_ld + lr, rblk = RING_ROWS * ring_ld;
  for (int s = 0; s <= CH_NC + 1; s++) {
    // (1) dense update with Yr of block kg (in the ring since the previous step)
    const int kg = CH_NC + 1 - s;
    if (kg >= 0 && kg < CH_NC) {
      const lds_double *rg = ring + (kg & 1) * rblk + ro;
      const int lo = __builtin_amdgcn_readfirstlane(s_lo[kg]);
#pragma unroll
      for (int q = 0; q < S2_TPW; q++) {
        if (tJ1[q] <= lo) continue;                               // (columns below lo_kg are zero in Yr_kg; wave-uniform)
        double a[3], b[3];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) { a[kk] = rg[4 * kk * ring_ld + oI[q]]; b[kk] = rg[4 * kk * ring_ld + oJ[q]]; }
#pragma unroll
        for (int kk = 0; kk < 3; kk++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], acc[q], 0, 0, 0);
      }
    }
    // (2) Yr of block k: R' = R_k - Yc_k+1^T Yr_k+1, Yr_k = W_k R'
    const int k = CH_NC - s;
    if (k >= 0 && k < CH_NC) {
      const int lo = __builtin_amdgcn_readfirstlane(s_lo[k]);
      double a1[3], a2[3];
      a2[0] = pW01[0]; a2[1] = pW01[4]; a2[2] = pW2[0];
      if (k + 1 < CH_NC) { a1[0] = -pC01[0]; a1[1] = -pC01[4 * CH_NB]; a1[2] = -pC2[0]; pC01 -= sW01; pC2 -= sW2; }
      else { a1[0] = 0.0; a1[1] = 0.0; a1[2] = 0.0; }
      pW01 -= sW01; pW2 -= sW2;
      double sk[3], vk[3], rk[3];
#pragma unroll
      for (int kk = 0; kk < 3; kk++) { const int a = T_SB(k) + min(lk + 4 * kk, CH_NB - 1); sk[kk] = sS[a]; vk[kk] = vS[a]; rk[kk] = rS[a]; }
      sk[2] *= m2; rk[2] *= m2;
#pragma unroll
      for (int u = 0; u < NU; u++) {
        if (!on[u]) continue;                                      // (wave-uniform)
        const bool zero_tile = TB * (wv + S2_WIDE_WAVES * u + 1) <= lo;      // (below the reach of block k: Yr_k is zero here; wave-uniform)
        const double reach = (jc[u] >= lo && jc[u] < na) ? 1.0 : 0.0;
        const double m1 = reach * sbj[u], mr = reach * mrhs[u];
        double r[3];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) r[kk] = __builtin_fma(rpre[u][kk] * sk[kk], m1, mr * rk[kk]);
#if GFBE_WIDE_PREFETCH2
#pragma unroll
        for (int kk = 0; kk < 3; kk++) rpre[u][kk] = rnxt[u][kk];
        load_R(u, rnxt[u]);                                        // (block k - 2; the loads of the two steps after block 1 read valid, unused rows of H)
#else
        load_R(u, rpre[u]);                                        // (block k - 1; in flight during this block's products. The loads of
                                                                   //  the step after block 0 read valid, unused rows of H)
#endif
        if (!zero_tile) {
#pragma unroll
          for (int kk = 0; kk < 3; kk++) vsv = __builtin_fma(r[kk] * vk[kk], vbj2[u], vsv);
          dbl4 rp = {r[0], r[1], r[2], 0.0};
#pragma unroll
          for (int kk = 0; kk < 3; kk++) rp = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], yv[u][kk], rp, 0, 0, 0);
          dbl4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kk = 0; kk < 3; kk++) y = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[kk], rp[kk], y, 0, 0, 0);
#pragma unroll
          for (int kk = 0; kk < 3; kk++) yv[u][kk] = y[kk];
          lds_double *rgw = pR[u] + (k & 1) * rblk;
#pragma unroll
          for (int kk = 0; kk < 3; kk++) rgw[4 * kk * ring_ld] = yv[u][kk];
#pragma unroll
          for (int kk = 0; kk < 3; kk++) zzc = __builtin_fma(yv[u][kk] * mrhs[u], yv[u][kk], zzc);
        } else {
#pragma unroll
          for (int kk = 0; kk < 3; kk++) yv[u][kk] = 0.0;
        }
        pY[u][0] = yv[u][0]; pY[u][4] = yv[u][1]; pY8[u][0] = yv[u][2];
        pY[u] -= CH_NB; pY8[u] -= CH_NB;
      }
    }
    // (3) G_kG = W_kG^T Yc_kG in place of Yc_kG, one step after its last reader (the wide rows of block kG - 1): the chain's
    //     back-substitution then needs one matrix-vector product per block
    const int kG = CH_NC - s + 2;
    if (wv == S2_WIDE_WAVES - 1 && kG >= 1 && kG < CH_NC) {
      const int off = (kG - (CH_NC - 1));                          // (<= 0) blocks below the first one
      const lds_double *t01 = pT01 + off * sW01, *t2 = pT2 + off * sW2, *c01 = pG01 + off * sW01, *c2 = pG2 + off * sW2;
      const double ag0 = t01[0], ag1 = t01[4 * CH_NB], ag2 = t2[0], bg0 = c01[0], bg1 = c01[4 * CH_NB], bg2 = c2[0];
      dbl4 gq = {0.0, 0.0, 0.0, 0.0};
      gq = __builtin_amdgcn_mfma_f64_16x16x4f64(ag0, in01 ? bg0 : 0.0, gq, 0, 0, 0);
      gq = __builtin_amdgcn_mfma_f64_16x16x4f64(ag1, in01 ? bg1 : 0.0, gq, 0, 0, 0);
      gq = __builtin_amdgcn_mfma_f64_16x16x4f64(ag2, in2 ? bg2 : 0.0, gq, 0, 0, 0);
      lds_double *g01 = pG01 + off * sW01, *g2 = pG2 + off * sW2;
      g01[0] = gq[0]; g01[4 * CH_NB] = gq[1]; g2[0] = gq[2];
    }
    RSTAMP(wv == 1 && lane == 0 && s < 12, 52 + s);
    CH_LDS_BARRIER();
  }
  // dense tiles -= the accumulated products
#pragma unroll
  for (int q = 0; q < S2_TPW; q++) {
    if (tJ1[q] == 0) continue;
    lds_double *C = tiles + (size_t)tile_idx(oI[q] / TB, oJ[q] / TB) * (TB * TB);
#pragma unroll
    for (int r = 0; r < 4; r++) C[tsw(lk + 4 * r, lr)] -= acc[q][r];
  }
  // |z_chain|^2: the four lanes (lk = 0..3) of the right-hand side column
  zzc += __shfl_xor(zzc, 16, 64);
  zzc += __shfl_xor(zzc, 32, 64);
#pragma unroll
  for (int u = 0; u < NU; u++) if (mrhs[u] != 0.0 && lk == 0) *zzc_out = zzc;
  return vsv;
}

// ---- k_solve_chain_tw: the wide rows of one segment, three waves (wv = 0..2). Wave wv forms the column tiles wv and wv + 3 of Yr, block
// after block, one step behind its segment's chain wave: R' = R_k - Yc_prev^T Yr_prev (Yr_prev still in the registers, in the operand
// layout), Yr_k = W_k R'. Unlike wide_role it does NOT add the dense update step by step: a window alone on the GPU is bound by the
// latency of these per-step chains (LDS operands -> 3 + 3 dependent matrix-core instructions -> stores -> barrier), so the step carries
// nothing else — every row of Yr stays in LDS (`Yall`, row 9 k + i, 99 rows + a zero row: 100 = 25 x 4) and the dense update
// D -= Yr^T Yr is ONE product over all 100 rows after the pipeline, dealt over all eight waves (dense_update_tw: 25 instead of 33
// matrix-core instructions per tile, no ring, no transposed copy of Yr in global memory: the chain's back-substitution reads Yall).
// No column skipping either (the ascending segment's rows are full — the prior couples SpeedBias[0] with everything it kept — and
// set the pace): entries outside a block's reach come out as the exact zeros they are.
// The middle block (last of segment 0) takes the second product -Yc'^T Yr of segment 1's last block, read back from Yall.
// Row stride 81: ODD, so that a thread per row walks its row without bank conflicts (the back-substitution's 99 dot products), while
// the matrix-core operand pattern (two rows x 16 consecutive doubles per half-wave) still only collides on two of its 64 banks.
enum { YALL_ROWS = 104, YALL_LD = 81 };     // 99 rows of Yr | row 99: zeros (the 25th group of four) | rows 100..103: dump rows of the masked third store
template <int SEG>
__device__ GFBE_ROLE_FN double wide_role_tw(lds_double *Ach, lds_double *Cch, lds_double *Yall, lds_double *zslot, const lds_double *sS, const lds_double *vS,
                                            const lds_double *rS, const lds_short *perm, lds_double *zzc_out, const glb_double *H, int n_, int nt_, int ld_,
                                            int wv_, int lane, double *rstamp) {
  constexpr int NB = SEG == 0 ? ChainCfg<1>::NB0 : ChainCfg<1>::NB1;
  constexpr int KF = SEG == 0 ? CH_NC - 1 : 0, SG = SEG == 0 ? -1 : 1;
  const int n = __builtin_amdgcn_readfirstlane(n_), nt = __builtin_amdgcn_readfirstlane(nt_);
  const int ld = __builtin_amdgcn_readfirstlane(ld_), wv = __builtin_amdgcn_readfirstlane(wv_);
  const int lr = lane & 15, lk = lane >> 4, na = n + 1;
  const bool in01 = lr < CH_NB, in2 = lr < CH_NB && lk == 0;
  bool on[2];
  int jc[2];
  double sbj[2], vbj2[2], mrhs[2], mcol[2];
  const glb_double *pH[2];        // S-source entry (SB_k row lk, column): H[a * ND + b] for b below the speed-bias dims, H[b * ND + a] above
  long dH[2];
  lds_double *pR[2], *pR8[2];     // Yall slots: (row 9 k + lk, column 16 c + lr) | row 9 k + 8 (lk == 0) or a dump row
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int c = wv + S2_WIDE_WAVES * u;
    on[u] = c < nt;
    jc[u] = TB * c + lr;
    const bool dense_col = on[u] && jc[u] < n;
    const int bj = dense_col ? perm[jc[u]] : 0;
    sbj[u] = dense_col ? sS[bj] : 0.0; vbj2[u] = dense_col ? 2.0 * vS[bj] : 0.0;
    mrhs[u] = (on[u] && jc[u] == n) ? 1.0 : 0.0;
    mcol[u] = (on[u] && jc[u] < na) ? 1.0 : 0.0;
    const int a0 = T_SB(KF) + lk;
    dH[u] = bj < T_SB(0) ? ND : 1;
    pH[u] = H + (bj < T_SB(0) ? (size_t)a0 * ND + bj : (size_t)bj * ND + a0);
    const int cc = min(TB * c, ld - TB) + lr;
    pR[u] = Yall + (KF * CH_NB + lk) * ld + cc;
    pR8[u] = lk == 0 ? Yall + (KF * CH_NB + 8) * ld + cc : Yall + (100 + lk) * ld + cc;
  }
  const int sR = SG * CH_NB * ld, sR8 = lk == 0 ? SG * CH_NB * ld : 0;
  double yv[2][3], rpre[2][3], rnxt[2][3], vsv = 0.0, zzc = 0.0;
#pragma unroll
  for (int u = 0; u < 2; u++)
#pragma unroll
    for (int kk = 0; kk < 3; kk++) { yv[u][kk] = 0.0; rpre[u][kk] = 0.0; rnxt[u][kk] = 0.0; }
  // rows lk, lk + 4, lk + 8 of block k (row 8 + lk exists for lk == 0 only: the other lanes' third load lands in the next block or above
  // the diagonal, on entries nobody writes, and is replaced by zero, not multiplied by it)
  auto load_R = [&](int u, double (&r)[3]) {
    const double r2 = pH[u][8 * dH[u]];
    r[0] = pH[u][0]; r[1] = pH[u][4 * dH[u]]; r[2] = lk == 0 ? r2 : 0.0;
    pH[u] += SG * CH_NB * dH[u];
  };
#pragma unroll
  for (int u = 0; u < 2; u++) if (on[u]) load_R(u, rpre[u]);
  const int sW01 = in01 ? SG * CH_BLK : 0, sW2 = in2 ? SG * CH_BLK : 0;
  const lds_double *pW01 = in01 ? Ach + KF * CH_BLK + lr * CH_NB + lk : zslot;
  const lds_double *pW2 = in2 ? Ach + KF * CH_BLK + lr * CH_NB + 8 : zslot;
  const lds_double *pC01 = in01 ? Cch + KF * CH_BLK + lk * CH_NB + lr : zslot;      // (the previous block's Yc: first used at the segment's second block)
  const lds_double *pC2 = in2 ? Cch + KF * CH_BLK + 8 * CH_NB + lr : zslot;
  const double m2 = lk == 0 ? 1.0 : 0.0;
  for (int s = 0; s < ChainCfg<1>::STEPS; s++) {
    const int pk = s - 1;
    if (pk >= 0 && pk < NB) {
      const int k = KF + SG * pk;
      double a1[3], a2[3];
      a2[0] = pW01[0]; a2[1] = pW01[4]; a2[2] = pW2[0];
      if (pk > 0) { a1[0] = -pC01[0]; a1[1] = -pC01[4 * CH_NB]; a1[2] = -pC2[0]; pC01 += sW01; pC2 += sW2; }
      else { a1[0] = 0.0; a1[1] = 0.0; a1[2] = 0.0; }
      pW01 += sW01; pW2 += sW2;
      const bool mid = SEG == 0 && k == CH_MID;                  // (wave-uniform: the middle block has a second predecessor)
      double b1[3] = {0.0, 0.0, 0.0};
      if (mid) {
        const lds_double *cm01 = in01 ? Cch + (CH_MID - 1) * CH_BLK + lk * CH_NB + lr : zslot, *cm2 = in2 ? Cch + (CH_MID - 1) * CH_BLK + 8 * CH_NB + lr : zslot;
        b1[0] = -cm01[0]; b1[1] = -cm01[4 * CH_NB]; b1[2] = -cm2[0];
      }
      double sk[3], vk[3], rk[3];
#pragma unroll
      for (int kk = 0; kk < 3; kk++) { const int a = T_SB(k) + min(lk + 4 * kk, CH_NB - 1); sk[kk] = sS[a]; vk[kk] = vS[a]; rk[kk] = rS[a]; }
      sk[2] *= m2; rk[2] *= m2;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (!on[u]) continue;                                      // (wave-uniform)
        const double m1 = mcol[u] * sbj[u], mr = mrhs[u];
        double r[3];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) r[kk] = __builtin_fma(rpre[u][kk] * sk[kk], m1, mr * rk[kk]);
        load_R(u, rpre[u]);                                        // (the next block; in flight during this block's products)
#pragma unroll
        for (int kk = 0; kk < 3; kk++) vsv = __builtin_fma(r[kk] * vk[kk], vbj2[u], vsv);
        dbl4 rp = {r[0], r[1], r[2], 0.0};
#pragma unroll
        for (int kk = 0; kk < 3; kk++) rp = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], yv[u][kk], rp, 0, 0, 0);
        if (mid) {                                                 // - Yc'^T Yr of segment 1's last block, from Yall (its ninth row by lk == 0 only)
          const lds_double *yo = Yall + ((CH_MID - 1) * CH_NB + lk) * ld + min(TB * (wv + S2_WIDE_WAVES * u), ld - TB) + lr;
          const double y8 = yo[8 * ld];
          const double yb[3] = {yo[0], yo[4 * ld], lk == 0 ? y8 : 0.0};
#pragma unroll
          for (int kk = 0; kk < 3; kk++) rp = __builtin_amdgcn_mfma_f64_16x16x4f64(b1[kk], yb[kk], rp, 0, 0, 0);
        }
        dbl4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 3; kk++) y = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[kk], rp[kk], y, 0, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 3; kk++) yv[u][kk] = y[kk];
        pR[u][0] = yv[u][0]; pR[u][4 * ld] = yv[u][1]; pR8[u][0] = yv[u][2];
        pR[u] += sR; pR8[u] += sR8;
#pragma unroll
        for (int kk = 0; kk < 3; kk++) zzc = __builtin_fma(yv[u][kk] * mrhs[u], yv[u][kk], zzc);
      }
    }
    RSTAMP(SEG == 0 && wv == 1 && lane == 0 && s < 12, 52 + s);
    RSTAMP(SEG == 1 && wv == 1 && lane == 0 && s >= 1 && s < 7, 39 + s);
    CH_LDS_BARRIER();
  }
  // |z|^2 of the segment's chain rows: the four lanes (lk = 0..3) of the right-hand side column
  zzc += __shfl_xor(zzc, 16, 64);
  zzc += __shfl_xor(zzc, 32, 64);
#pragma unroll
  for (int u = 0; u < 2; u++) if (mrhs[u] != 0.0 && lk == 0) *zzc_out = zzc;
  return vsv;
}
// After the pipeline of k_solve_chain_tw, every wave: (1) the dense update D -= Yr^T Yr over all 100 rows of Yall, tiles dealt round robin;
// (2) G_k = W_k^T Yc_k in place of Yc_k for the ten blocks that have a successor (the chain's back-substitution then needs one 9 x 9
// matrix-vector product per block).
__device__ __forceinline__ void dense_update_tw(lds_double *tiles, const lds_double *Yall, lds_double *Ach, lds_double *Cch, const lds_double *zslot,
                                                const lds_int *s_lo, int nt, int ld, int wave, int lane, int nwaves) {
  const int lr = lane & 15, lk = lane >> 4;
  const int ntile = nt * (nt + 1) / 2;
  // Tiles by RANK, heaviest first: column J of the tile grid from the last one down, rows I from the last one down. A tile of column J
  // only needs the rows of Yr whose blocks reach a dense column below 16 (J + 1) — the descending segment's blocks k > CH_MID start at
  // s_lo[k] (everything below is an exact zero), and their rows are the LAST ones of Yall: rows 0 .. 9 (kmax + 1) - 1, in groups of
  // four. Wave w takes the tiles of rank w and ntile - 1 - w (a heavy one with a light one) side by side — independent accumulators:
  // no matrix-core instruction waits for the one before it —, the operands of the next five groups in flight while five are multiplied.
  auto tile_of_rank = [&](int q, int &I, int &J, int &ng) {
    J = nt - 1;
    while (q >= nt - J) { q -= nt - J; J--; }
    I = nt - 1 - q;
    int kmax = CH_NC - 1;
    while (kmax > CH_MID && s_lo[kmax] >= TB * (J + 1)) kmax--;
    ng = __builtin_amdgcn_readfirstlane((CH_NB * (kmax + 1) + 3) / 4);      // groups of four rows (<= 25)
  };
  for (int q0 = wave; 2 * q0 < ntile; q0 += nwaves) {
    const int q1 = ntile - 1 - q0;
    const bool two = q1 > q0;                                      // (wave-uniform; the middle rank of an odd count stands alone)
    int I0, J0, ng0, I1 = 0, J1 = 0, ng1 = 0;
    tile_of_rank(q0, I0, J0, ng0);
    if (two) tile_of_rank(q1, I1, J1, ng1);
    const lds_double *pa0 = Yall + lk * ld + min(TB * I0, ld - TB) + lr, *pb0 = Yall + lk * ld + min(TB * J0, ld - TB) + lr;
    const lds_double *pa1 = Yall + lk * ld + min(TB * I1, ld - TB) + lr, *pb1 = Yall + lk * ld + min(TB * J1, ld - TB) + lr;
    dbl4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    double a0[5], b0[5], a1[5], b1[5], na0[5], nb0[5], na1[5], nb1[5];
#pragma unroll
    for (int q = 0; q < 5; q++) { a0[q] = pa0[4 * q * ld]; b0[q] = pb0[4 * q * ld]; a1[q] = pa1[4 * q * ld]; b1[q] = pb1[4 * q * ld]; }
#pragma unroll
    for (int g = 0; g < 5; g++) {
      if (5 * g >= ng0) break;                                     // (rank q0 is the heavier tile: ng1 <= ng0)
      if (g < 4) {
#pragma unroll
        for (int q = 0; q < 5; q++) {
          const int o = 4 * (5 * (g + 1) + q) * ld;
          na0[q] = pa0[o]; nb0[q] = pb0[o]; na1[q] = pa1[o]; nb1[q] = pb1[o];
        }
      }
#pragma unroll
      for (int q = 0; q < 5; q++) {
        if (5 * g + q < ng0) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b0[q], acc0, 0, 0, 0);
        if (5 * g + q < ng1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b1[q], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 5; q++) { a0[q] = na0[q]; b0[q] = nb0[q]; a1[q] = na1[q]; b1[q] = nb1[q]; }
    }
    lds_double *C0 = tiles + (size_t)tile_idx(I0, J0) * (TB * TB);
#pragma unroll
    for (int r = 0; r < 4; r++) C0[tsw(lk + 4 * r, lr)] -= acc0[r];
    if (two) {
      lds_double *C1 = tiles + (size_t)tile_idx(I1, J1) * (TB * TB);
#pragma unroll
      for (int r = 0; r < 4; r++) C1[tsw(lk + 4 * r, lr)] -= acc1[r];
    }
  }
  const bool in01 = lr < CH_NB, in2 = lr < CH_NB && lk == 0;
  for (int q = nwaves - 1 - wave; q < CH_NC - 1; q += nwaves) {       // (from the last wave down: the first ones carry the heaviest tile pairs)
    const int k = q < CH_MID ? q : q + 1;                          // (every block but the middle one)
    const lds_double *t01 = in01 ? Ach + k * CH_BLK + lk * CH_NB + lr : zslot, *t2 = in2 ? Ach + k * CH_BLK + 8 * CH_NB + lr : zslot;
    lds_double *c01 = Cch + k * CH_BLK + lk * CH_NB + lr, *c2 = Cch + k * CH_BLK + 8 * CH_NB + lr;
    const double ag0 = t01[0], ag1 = t01[4 * CH_NB], ag2 = t2[0];
    const double bg0 = in01 ? c01[0] : 0.0, bg1 = in01 ? c01[4 * CH_NB] : 0.0, bg2 = in2 ? c2[0] : 0.0;
    dbl4 gq = {0.0, 0.0, 0.0, 0.0};
    gq = __builtin_amdgcn_mfma_f64_16x16x4f64(ag0, bg0, gq, 0, 0, 0);
    gq = __builtin_amdgcn_mfma_f64_16x16x4f64(ag1, bg1, gq, 0, 0, 0);
    gq = __builtin_amdgcn_mfma_f64_16x16x4f64(ag2, bg2, gq, 0, 0, 0);
    __builtin_amdgcn_wave_barrier();                               // (every lane has read its entries of Yc_k before anyone overwrites them)
    if (in01) { c01[0] = gq[0]; c01[4 * CH_NB] = gq[1]; }
    if (in2) c2[0] = gq[2];
  }
}

// TW = 0: k_solve_chain (four waves: the chain role, three wide waves; two workgroups per CU — throughput batches).
// TW = 1: k_solve_chain_tw (eight waves: two chain roles, three wide waves per segment; one workgroup per CU — small batches, where one
//         window's latency counts: the chain is eliminated from both ends, ChainCfg above).
// MAXNT: tile columns of the dense part the instance holds — 6: k_solve_chain / k_solve_chain_tw (the 187 core dims); 9 (round 6,
// k_solve_chain_wide): a window with GNSS blocks, whose 58 extra dims are dense columns like the poses (SpeedBias[k] couples with them
// through the Doppler rows of the pseudo-range factors: wide rows, as with the wheel extrinsic) — 45 tiles, one workgroup per CU
template <int TW, int MAXNT = S2_MAX_NT>
__device__ __forceinline__ void solve_chain_body(const BatchDev &d, int retry_pass) {
  constexpr int NWAVES = TW ? 2 * S2_WAVES : S2_WAVES;
  constexpr int MAXTILES = MAXNT * (MAXNT + 1) / 2;
  constexpr bool WIDE = MAXNT > S2_MAX_NT;
  constexpr int PW = WIDE ? 4 : 3;        // waves whose threads own a tangent dim that can be dense (dims 0..191: the core; 0..245 with the GNSS blocks)
  static_assert(!(TW && WIDE), "the two-ended kernel holds the core dims only");
  const int w = blockIdx.x;
  const WinDesc &ds = d.desc[w];
  WinCtl &c = d.ctl[w];
#if GFBE_CHAIN_SIMD_ROLES
  // (round 6) the waves numbered by the SIMD they sit on: wave 0 — the chain role and every serial section — of BOTH workgroups of a CU on
  // SIMD 0, the wide waves pairwise on SIMDs 1..3. The dispatcher gives a workgroup one wave per SIMD and rotates the start, so that wave 0
  // of one workgroup shares its SIMD with a wide wave of the other one — whose FP64 matrix-core instructions hold the FP64 vector pipe the
  // chain's dependent instructions need (the pipeline: 19.6 us alone, 33.3 beside a second workgroup). Everything below uses the logical
  // index t, the block sums included: the results do not depend on the placement. (Not one wave per SIMD: the physical numbering.)
  int t_;
  {
    const int tp = threadIdx.x, wp = tp >> 6;
    __shared__ int simd_of[S2_WAVES];
    int lw = wp;
    if (!TW) {
      const int simd = (int)(__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3u);      // HW_REG_HW_ID, SIMD_ID = bits 5:4
      if ((tp & 63) == 0) simd_of[wp] = simd;
      __syncthreads();
      int m = 0;
      for (int q = 0; q < S2_WAVES; q++) m |= 1 << simd_of[q];
      lw = (m == 15) ? simd : wp;
    }
    t_ = lw * 64 + (tp & 63);
  }
  const int t = t_, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
#else
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
#endif
#if GFBE_CHAIN_PRIO
  // (round 6) instruction-arbitration priority: the kernel is a chain of dependent instructions on single waves — wave 0 above all — that
  // shares its SIMDs with the second workgroup of the CU and, in a split batch, with the streaming kernels of the other parts
  if (!TW) { if (wave == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(GFBE_CHAIN_PRIO - 1); }
#endif
  const double *H = d.H + (size_t)w * ND * ND, *g = d.g + (size_t)w * ND;
  double *gsp = d.sp + (size_t)w * ND, *gDp = d.Dp + (size_t)w * ND, *ggts = d.gts + (size_t)w * ND;
  double *gvp = d.vp + (size_t)w * ND, *gyp = d.yp + (size_t)w * ND;
  double *gYT = WIDE ? d.solveS + (size_t)w * BIG_LD * BIG_LD : d.solveY + (size_t)w * GYT_COLS * GYT_LD;      // (wide: the batch is a k_solve_big batch — its scratch)
  const double *E = retry_pass ? d.Er + (size_t)w * (NV * NV + NV) : d.E + (size_t)w * NV * NV;
  const double *eg = retry_pass ? E + NV * NV : d.eg + (size_t)w * NV;
  // ---- level 1 of the kernel's global loads: everything whose address needs nothing but the window index is requested HERE, before
  // the first dependent use of any of it (the control block's early-exit test included) — a kernel that follows another one finds
  // nothing in its caches, and each LEVEL of dependent loads costs 1-2 us whatever is loaded (round 5: the prologue and the build
  // were five such levels, 11 us of a 56 us launch; now two — this one and the entries that need `perm`). The workgroup has at least
  // 256 threads: thread a owns tangent dim a (ND = 246) and parameter block a. Nothing loaded here is used by a window / a dim / a
  // block it does not belong to (clamped indices, values selected afterwards).
  const int ta = min(t, ND - 1), tb = min(t, GFBE_BLK_COUNT - 1);
  const bool l_act = ds.act[ta], l_free = ds.blk_free[tb];
  const double l_haa = H[(size_t)ta * ND + ta], l_g = g[ta], l_sp = gsp[ta], l_eg = eg[min(ta, NV - 1)];
  const int c_done = c.done, c_reuse = c.reuse, c_lin_retry = c.lin_retry, c_iter = c.iter, c_cur = c.cur;
  if (c_done || c_reuse) return;
  if (retry_pass && !c_lin_retry) return;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ short perm[ND + TB];
  __shared__ double ys[ND + TB];                                 // ys: the solution of the dense part (tile order)
  __shared__ double sS[ND], vS[ND], dS[ND], gS[ND], rS[ND], yT[ND];   // per tangent dim: Jacobi scale, Cauchy direction, D, scaled gradient, right-hand side, GN step
  __shared__ double s_zz, s_zzc[2], s_vSv, zlast[TB], xs[CH_ROWS + 16], tch[CH_ROWS + 16], cterm[64];
  __shared__ double s_keep[4];
  __shared__ int flag, s_nch, s_lo[CH_NC + 1];
  __shared__ unsigned long long s_mask0;
  __shared__ unsigned char chact[CH_ROWS + 1];
  const bool first = (c_iter == 0);
  double *stamp = d.timing + (size_t)w * 32;
#define STAMP(i) do { if (t == 0) stamp[i] = (double)wall_clock64(); } while (0)
  STAMP(0);
  // the terms of the first linearisation point's cost (wave 3, summed below): requested with level 1
  double l_term = 0.0;
  if (first && wave == 3) {
    if (lane < ds.n_imu) l_term = d.imu_part[((size_t)w * MAX_IMU + lane) * IMU_PART + IMU_PART - 2];
    else if (lane >= 16 && lane - 16 < ds.n_wheel) l_term = d.wheel_part[((size_t)w * MAX_WHEEL + lane - 16) * WHEEL_PART + WHEEL_PART - 2];
    else if (lane == 32) l_term = d.prior_g[(size_t)w * (ND + 2) + ND];
    else if (lane == 33) {   // the ranks' visual-cost shares, in rank order (any world size gfbe_set_allreduce accepts: k_solve / k_solve_big loop the same way)
      for (int r = 0; r < d.world; r++) l_term += d.xa[((size_t)w * d.world + r) * XCHG];
    }
    else if (lane >= 44 && lane - 44 < ds.n_plane) l_term = d.plane_part[((size_t)w * MAX_PLANE + lane - 44) * PLANE_PART + PLANE_PART - 2];
    else if (lane == 58 && ds.use_anchor) l_term = d.anchor_part[(size_t)w * ANCHOR_PART + ANCHOR_PART - 2];
    else if (WIDE && lane == 59 && ds.gnss_factors) l_term = d.gnss_cost[(size_t)w * 2];      // (k_solve_big's order: visual, inertial, wheel, prior, plane, anchor, GNSS)
  }
  // LDS carve-up: dense tiles | chain diagonal blocks (A_k -> W_k) | chain couplings (C_k -> Yc_k -> G_k) | zero / dump slot | two-block ring of Yr
  double *tiles = smem;
  double *Ach = smem + (size_t)d.solve_ntile * (TB * TB);
  double *Cch = Ach + CH_NC * CH_BLK;
  double *zslot = Cch + CH_NC * CH_BLK;
  double *ring = zslot + CH_ZERO;
  double *Yall = ring;                                                      // (twisted: every row of Yr instead of the ring, then the middle block's second downdate)
  double *Amid = Yall + YALL_ROWS * YALL_LD;
  if (t < CH_ZERO) zslot[t] = 0.0;
  if (TW && t < YALL_LD) Yall[99 * YALL_LD + t] = 0.0;      // (row 99: the padding of the 25th group of four rows)

  // dense dims (active, not in the chain) by a wave-level prefix count (dims 0..191 live in waves 0..2); chain activity flags
  __shared__ int wcount[4];
  {
    const bool act_t = (t < ND) && l_act;
    const bool on = act_t && !dim_in_chain(t);
    const unsigned long long m = __ballot(on);
    const unsigned long long mc = __ballot(act_t && dim_in_chain(t));
    if (t < 64 * PW && lane == 0) wcount[wave] = __popcll(m);
    if (t == 0) { s_nch = 0; s_mask0 = m; }
    if (t < ND && dim_in_chain(t)) chact[t - T_SB(0)] = act_t ? 1 : 0;
    __syncthreads();
    if (t < 64 * PW) {
      int base = 0;
      for (int q = 0; q < wave; q++) base += wcount[q];
      if (on) perm[base + __popcll(m & ((1ull << lane) - 1ull))] = t;
      if (lane == 0 && mc) atomicAdd(&s_nch, __popcll(mc));
    }
    const int nact = WIDE ? wcount[0] + wcount[1] + wcount[2] + wcount[3] : wcount[0] + wcount[1] + wcount[2];
    for (int a = nact + t; a < ND + TB; a += blockDim.x) perm[a] = -1;
    // lo_k: first dense column the wide row of block k can reach — the poses of frames >= k - 1 (pose dims are 0..65: wave 0's mask)
    if (t <= CH_NC) s_lo[t] = (t >= 2 && t < CH_NC) ? __popcll(s_mask0 & ((1ull << (6 * (t - 1))) - 1ull)) : 0;
  }
  __syncthreads();        // (perm is complete: the build's entries can be requested)
  const int n = __builtin_amdgcn_readfirstlane(WIDE ? wcount[0] + wcount[1] + wcount[2] + wcount[3] : wcount[0] + wcount[1] + wcount[2]);      // dense dims (wave-uniform: kept in scalar registers)
  const int na = n + 1;                 // + the right-hand side row / column
  const int nt = (na + TB - 1) / TB;
  const int ntile_all = nt * (nt + 1) / 2;
  // ---- level 2: the entries of H and E the build needs (through perm), the chain blocks, the parameter blocks of |x|^2 — all in flight
  // while the per-dim quantities below are formed from level 1
  constexpr int NQ = (2 * CH_NC + 2) / 3;
  int ar[MAXNT], bc[MAXNT];
  double hv[MAXTILES], ev[MAXTILES], hc[NQ];
  auto chain_partner = [](int k) -> int { return TW ? (k > CH_MID ? k - 1 : (k < CH_MID ? k + 1 : k)) : (k > 0 ? k - 1 : 0); };
  // (every array entry is defined by every thread — zero first, the loads inside wave-uniform branches)
  {
    const int wv = wave, tt = t;
    const int r = (tt >> 4) & 15, cc = tt & 15;
    const int nm1 = max(n - 1, 0);
#pragma unroll
    for (int I = 0; I < MAXNT; I++) { ar[I] = max((int)perm[min(I * TB + r, nm1)], 0); bc[I] = max((int)perm[min(I * TB + cc, nm1)], 0); }
#pragma unroll
    for (int te = 0; te < MAXTILES; te++) { hv[te] = 0.0; ev[te] = 0.0; }
#pragma unroll
    for (int q = 0; q < NQ; q++) hc[q] = 0.0;
    if (wv < TB * TB / 64) {      // the dense tiles' threads
      // (through clamped indices, selected afterwards: straight-line code, every load in flight at once)
      int I = 0, J = 0;
#pragma unroll
      for (int te = 0; te < MAXTILES; te++) {
        if (te < ntile_all) {                                   // (wave-uniform)
          const int hi = max(ar[I], bc[J]), lo = min(ar[I], bc[J]);
          hv[te] = H[(size_t)hi * ND + lo];
          ev[te] = E[min(hi, NV - 1) * NV + min(lo, NV - 1)];
        }
        if (++J > I) { J = 0; I++; }
      }
    }
    if (TW ? wv >= TB * TB / 64 : true) {      // the chain blocks' threads (twisted: the second half of the workgroup, beside the dense tiles)
      const int tc = TW ? tt - TB * TB : tt;
      const int tcc = tc < 3 * CH_BLK ? tc : 0;
      const int grp = tcc / CH_BLK, e81 = tcc - CH_BLK * grp, ei = e81 / CH_NB, ej = e81 - CH_NB * ei;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const int kb = grp + 3 * q;                     // block slot: 0..10 diagonal blocks, 11..21 couplings
        const bool isC = kb >= CH_NC;
        const int k = isC ? kb - CH_NC : kb;
        const int a = T_SB(min(k, CH_NC - 1)) + ei, b = (isC ? T_SB(chain_partner(min(k, CH_NC - 1))) : T_SB(min(k, CH_NC - 1))) + ej;
        hc[q] = H[(size_t)max(a, b) * ND + min(a, b)];
      }
    }
  }
  double xv[9];
  {
    const double *X = d.x + ((size_t)w * 2 + c_cur) * NA;
    const int gs = (t < GFBE_BLK_COUNT && l_free) ? blk_gsize(tb) : 0, xa0 = blk_amb(tb);
#pragma unroll
    for (int k = 0; k < 9; k++) xv[k] = k < gs ? X[xa0 + k] : 0.0;
  }
  if (first && wave == 3) {   // total cost of the first linearisation point: the terms side by side, summed in lane order
    cterm[lane] = l_term;
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      double cost = 0.0;
      for (int q = 33; q < 44; q++) cost += cterm[q];      // visual cost (k_visblock; the ranks' shares)
      for (int q = 0; q < 33; q++) cost += cterm[q];
      for (int q = 44; q < 64; q++) cost += cterm[q];
      c.cost = cost; c.initial_cost = cost; c.cost_history[0] = cost;
    }
  }
  // Jacobi scaling (iteration 0 only), D = sqrt(clamp(diag)), scaled gradient, Cauchy direction
  double g2 = 0.0, gmax = 0.0, xn2 = 0.0;
  if (t < ND) {
    const int a = t;
    double s = 1.0, dp = 1.0, gt = 0.0, v = 0.0;
    if (l_act) {
      const double haa = l_haa;
      s = first ? (d.opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(haa)) : 1.0) : l_sp;
      const double d2 = clamp_diag(s * s * haa);
      dp = sqrt(d2); gt = s * l_g; v = gt / d2;
      g2 += gt * gt / d2;
      gmax = fmax(gmax, fabs(l_g));
    }
    if (first) gsp[a] = s;
    gDp[a] = dp; ggts[a] = gt; gvp[a] = v;
    sS[a] = s; vS[a] = v; dS[a] = dp; gS[a] = gt;      // staged in LDS: the builds and the dogleg sums never wait for HBM
    rS[a] = gt - (a < NV ? s * l_eg : 0.0);            // right-hand side of the reduced system
  }
#pragma unroll
  for (int k = 0; k < 9; k++) xn2 += xv[k] * xv[k];    // (|x|^2 over the free parameter blocks: a block per thread, its entries in order; absent ones add an exact zero)
  {
    const double pv[3] = {g2, gmax, xn2};
    block_reduce_multi<3>(pv, 0x2u, smem, BRM_T);
    if (t == 0) { s_keep[0] = smem[48]; s_keep[1] = smem[49]; s_keep[2] = smem[50]; }   // (needed at the very end: parked in LDS, not in registers)
  }
  __syncthreads();
  STAMP(1);
  const bool chain_on = __builtin_amdgcn_readfirstlane(s_nch) > 0;

  double mu = c.mu;
  bool solved = false;
  int att = 0;      // factorisation attempts of this linearisation so far (landmark sharding: the pass index counts them)
  double vsv = 0.0;
  {
    // ---- build: the arithmetic and the LDS stores (the entries were requested before the per-dim quantities were formed, or just above).
    //      Threads 0..255: entry (r, cc) = (t / 16, t % 16) of EVERY dense tile — the tangent dims of the thread's row r and column
    //      cc of each tile row / column and their scale / direction entries are looked up once. The chain blocks: the same threads
    //      (twisted: the other half of the workgroup).
    {
      if (t < TB * TB) {
        const int r = t >> 4, cc = t & 15;
        double sa[MAXNT], va[MAXNT], sb[MAXNT], vb[MAXNT], dd[MAXNT];
#pragma unroll
        for (int I = 0; I < MAXNT; I++) { sa[I] = sS[ar[I]]; va[I] = vS[ar[I]]; sb[I] = sS[bc[I]]; vb[I] = vS[bc[I]]; dd[I] = dS[ar[I]]; }
#if GFBE_CHAIN_STAMP
        if (t == 0) stamp[8] = (double)wall_clock64();
#endif
        {
          int I = 0, J = 0;
#pragma unroll
          for (int te = 0; te < MAXTILES; te++) {
            if (te < ntile_all) {
              const int a = ar[I], b = bc[J];
              double v = (hv[te] - ((a < NV && b < NV) ? ev[te] : 0.0)) * (sa[I] * sb[J]);
              if (I == J && r == cc) v = __builtin_fma(mu * dd[I], dd[I], v);          // (a == b: the diagonal)
              double q = v * va[I] * (vb[J] * (I != J ? 2.0 : 1.0));
              if (I == nt - 1) {       // (wave-uniform: the last tile row holds the right-hand side row and the padding)
                const int ia = I * TB + r, ib = J * TB + cc;
                if (ia >= n || ib >= n) {
                  q = 0.0;
                  if (ia == n && ib < n) v = rS[b];
                  else if (ib == n && ia < n) v = rS[a];
                  else v = (ia == ib) ? (ia == n ? 1e200 : 1.0) : 0.0;
                }
              }
              vsv += q;
              tiles[(size_t)te * (TB * TB) + tsw(r, cc)] = v;
            }
            if (++J > I) { J = 0; I++; }
          }
        }
      }
      const int tc = TW ? t - TB * TB : t;          // (twisted: the second half of the workgroup, beside the dense tiles)
      if (chain_on && tc >= 0 && tc < 3 * CH_BLK) {
        // chain blocks: the 22 blocks A_0..A_10, C_0..C_10 (C_k = S(SB_k, SB_k-1); slot 0 of the couplings is unused — twisted: C_k =
        // S(SB_k, SB_k+1) below the middle block, whose own slot is the unused one) are dealt over three thread groups of 81 — a thread
        // keeps its entry (i, j) and takes every third block (inactive dims: H holds exact zeros there, the diagonal becomes 1)
        const int grp = tc / CH_BLK, e81 = tc - CH_BLK * grp, ei = e81 / CH_NB, ej = e81 - CH_NB * ei;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          const int kb = grp + 3 * q;
          if (kb >= 2 * CH_NC) continue;
          const bool isC = kb >= CH_NC;
          const int k = isC ? kb - CH_NC : kb;
          const int a = T_SB(k) + ei, b = (isC ? T_SB(chain_partner(k)) : T_SB(k)) + ej;
          double v = hc[q] * sS[a] * sS[b];
          if (isC) { if (k == (TW ? CH_MID : 0)) v = 0.0; }
          else if (ei == ej) v = chact[a - T_SB(0)] ? __builtin_fma(mu * dS[a], dS[a], v) : 1.0;
          vsv = __builtin_fma(v * vS[a], vS[b] * (isC ? 2.0 : 1.0), vsv);
          (isC ? Cch : Ach)[k * CH_BLK + e81] = v;
        }
      }
    }
  }
  // (The attempts of the mu ladder: the system of the first attempt is built before the loop, the next one's at the END of a failed
  //  attempt — the ~100 registers of entries between the request and the build never live across the roles or the loop's back edge.)
  if (mu < GF_MAX_MU) for (;;) {
    // (the tiles and chain blocks of this attempt are in LDS: build_compute, before the loop / at the end of the failed attempt)
    if (t == 0) { flag = 0; s_zzc[0] = 0.0; s_zzc[1] = 0.0; }
    if (TW && t < CH_BLK) Amid[t] = 0.0;
#if GFBE_CHAIN_STAMP
    if (t == 0) stamp[11] = (double)wall_clock64();
#endif
    __syncthreads();
    STAMP(2);
    if (chain_on) {
      // ---- the pipeline over the chain blocks, one block barrier per block (the roles are separate functions with their own
      //      register allocation; every role passes the same CH_NC + 2 barriers):
      //   step s: wave 0 factorises block NC-1-s | waves 1..3 form Yr of block NC-s and add Yr^T Yr of block NC+1-s
if (!TW) {
        if (wave == 0) chain_role<0, 0, WIDE ? 1 : 0>((lds_double *)Ach, (lds_double *)Cch, (lds_double *)Amid, (lds_int *)&flag, lane, stamp);
        else
          vsv += wide_role<MAXNT>((lds_double *)tiles, (lds_double *)Ach, (lds_double *)Cch, (lds_double *)ring, (lds_double *)zslot, (const lds_double *)sS,
                           (const lds_double *)vS, (const lds_double *)rS, (const lds_short *)perm, (const lds_int *)s_lo, (lds_double *)&s_zzc[0],
                           (const glb_double *)H, (glb_double *)gYT, n, nt, WIDE ? (int)CHAIN_RING_LD_WIDE : chain_ring_ld(d.solve_ntile), wave - 1, lane, stamp);
      } else {
        // waves 0, 1: the chain roles of the two segments; 2..4: segment 0's wide waves; 5..7: segment 1's
#define WIDE_ARGS(seg) (lds_double *)Ach, (lds_double *)Cch, (lds_double *)Yall, (lds_double *)zslot, (const lds_double *)sS, (const lds_double *)vS, \
                       (const lds_double *)rS, (const lds_short *)perm, (lds_double *)&s_zzc[seg], (const glb_double *)H, n, nt, (int)YALL_LD
        if (wave == 0) chain_role<TW, 0>((lds_double *)Ach, (lds_double *)Cch, (lds_double *)Amid, (lds_int *)&flag, lane, stamp);
        else if (wave == 1) chain_role<TW, 1>((lds_double *)Ach, (lds_double *)Cch, (lds_double *)Amid, (lds_int *)&flag, lane, stamp);
        else if (wave < 2 + S2_WIDE_WAVES) vsv += wide_role_tw<0>(WIDE_ARGS(0), wave - 2, lane, stamp);
        else vsv += wide_role_tw<1>(WIDE_ARGS(1), wave - 2 - S2_WIDE_WAVES, lane, stamp);
#undef WIDE_ARGS
        dense_update_tw((lds_double *)tiles, (const lds_double *)Yall, (lds_double *)Ach, (lds_double *)Cch, (const lds_double *)zslot,
                        (const lds_int *)s_lo, nt, (int)YALL_LD, wave, lane, NWAVES);
      }
    }
    __syncthreads();      // (the thread's share of v^T S v is summed with |z|^2 below: one reduction instead of two on this path)
    STAMP(15);
    chol_factor_all<NWAVES, WIDE ? 1 : 0>((lds_double *)tiles, nt, n, t, (lds_double *)zlast, (lds_int *)&flag, stamp);
    bool ok = (flag == 0);
    if (d.test_fail_chol_iter > 0 && c.iter + 1 == d.test_fail_chol_iter && (d.sharded ? retry_pass : att) < max(d.opt.test_fail_chol_count, 1)) ok = false;
    STAMP(3);
    if (ok) {
      double zz = 0.0;
      for (int i = t; i < n + TB; i += blockDim.x) {
        const double z = i < n ? (i / TB == n / TB ? zlast[i % TB] : tiles[(size_t)tile_idx(n / TB, i / TB) * (TB * TB) + tsw(n % TB, i % TB)]) : 0.0;
        ys[i] = z;
        zz += z * z;
      }
      {
        const double zv[2] = {zz, vsv};
        block_reduce_multi<2>(zv, 0u, cterm, BRM_T);      // (same wave-order sums as two block_sum calls; cterm: free since the prologue)
        if (t == 0) { s_zz = cterm[32] + (s_zzc[0] + s_zzc[1]); s_vSv = cterm[33]; }
      }
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
      STAMP(16);
      // ---- the chain's back-substitution: t_r = z_r - (Yr x_dense)_r for the 99 chain rows (lane = row, the transposed Yr rows
      //      read back coalesced from HBM / L2, all loads in flight), a_k = W_k^T t_k, then x_k = a_k - G_k x_k-1 on one wave
      if (chain_on) {
        if (TW) {
          // t_r = z_r - (Yr x_dense)_r from the rows of Yall, a thread per row (odd row stride: no bank conflicts), eight loads in flight
          if (t < 128) {
            const int r = t < CH_ROWS ? t : CH_ROWS - 1;
            const double *row = Yall + r * YALL_LD;
            double s0 = 0.0, s1 = 0.0;
            int j = 0;
            for (; j + 8 <= n; j += 8) {
              double v[8];
#pragma unroll
              for (int q = 0; q < 8; q++) v[q] = row[j + q];
#pragma unroll
              for (int q = 0; q < 8; q += 2) { s0 = __builtin_fma(v[q], ys[j + q], s0); s1 = __builtin_fma(v[q + 1], ys[j + q + 1], s1); }
            }
            for (; j < n; j++) s0 = __builtin_fma(row[j], ys[j], s0);
            if (t < CH_ROWS) tch[t] = row[n] - (s0 + s1);
          }
        } else if (t < 128) {
          const int r = t < CH_ROWS ? t : CH_ROWS - 1;
          const double *col = gYT + r;
          double s0 = 0.0, s1 = 0.0;
          int j = 0;
          for (; j + 8 <= n; j += 8) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = col[(size_t)(j + q) * GYT_LD];
#pragma unroll
            for (int q = 0; q < 8; q += 2) { s0 = __builtin_fma(v[q], ys[j + q], s0); s1 = __builtin_fma(v[q + 1], ys[j + q + 1], s1); }
          }
          for (; j < n; j++) s0 = __builtin_fma(col[(size_t)j * GYT_LD], ys[j], s0);
          if (t < CH_ROWS) tch[t] = col[(size_t)n * GYT_LD] - (s0 + s1);
        }
        __syncthreads();
        if (t < CH_ROWS) {
          const int k = t / CH_NB, i = t % CH_NB;
          double a = 0.0;
#pragma unroll
          for (int m = 0; m < CH_NB; m++) a = __builtin_fma(Ach[k * CH_BLK + m * CH_NB + i], tch[k * CH_NB + m], a);
          xs[t] = a;
        }
        __syncthreads();
        // classic: x_0 is final, x_k = a_k - G_k x_k-1 upwards. Twisted: x_CH_MID is final (the block eliminated last); segment 0's blocks
        // upwards from it on wave 0, segment 1's downwards on wave 1 (x_k = a_k - G'_k x_k+1), side by side.
        if (wave == 0) chain_backsub((const lds_double *)Cch, (lds_double *)xs, TW ? CH_MID : 0, 1, TW ? CH_NC - 1 - CH_MID : CH_NC - 1, lane);
        else if (TW && wave == 1) chain_backsub((const lds_double *)Cch, (lds_double *)xs, CH_MID, -1, CH_MID, lane);
        __syncthreads();
      }
      // y back to the tangent dims: inactive dims get 0, every dim written once
      int bad = 0;
      for (int i = t; i < n; i += blockDim.x) { const double y = ys[i]; const int a = perm[i]; gyp[a] = y; yT[a] = y; if (!isfinite(y)) bad = 1; }
      for (int a = t; a < ND; a += blockDim.x) {
        if (dim_in_chain(a)) { const double y = chain_on ? xs[a - T_SB(0)] : 0.0; gyp[a] = y; yT[a] = y; if (!isfinite(y)) bad = 1; }
        else if (!ds.act[a]) { gyp[a] = 0.0; yT[a] = 0.0; }
      }
      if (bad) flag = 1;
      __syncthreads();
      ok = (flag == 0);
    }
    __syncthreads();
    if (ok) { solved = true; break; }
    mu *= GF_MU_INC;
    att++;
    if (!(mu < GF_MAX_MU)) break;
    if (d.sharded) {
      if (retry_pass < min(max(d.opt.sharded_mu_retries, 0), 8)) { if (t == 0) { c.lin_retry = 1; c.mu = mu; } return; }     // (on to pass retry_pass + 1)
      break;
    }
    rebuild_E(d, ds, w, mu, d.E + (size_t)w * NV * NV, d.eg + (size_t)w * NV, false);
    for (int a = t; a < ND; a += blockDim.x) rS[a] = gS[a] - (a < NV ? sS[a] * eg[a] : 0.0);   // (eg was rebuilt with E)
    __syncthreads();
    // ---- the system of the next attempt, built in place (the cold path: loads and arithmetic in one piece, like rounds 3-4)
    //      Threads 0..255: entry (r, cc) = (t / 16, t % 16) of EVERY dense tile — the tangent dims of the thread's row r and column
    //      cc of each tile row / column and their scale / direction entries are looked up once. Waves 4..5: the chain blocks.
    vsv = 0.0;
    {
      int tt = t;
      asm volatile("" : "+v"(tt));     // (opaque: the entry addresses are invariant in the mu-retry loop — hoisted out of it they are all spilled)
      if (tt < TB * TB) {
        const int r = tt >> 4, cc = tt & 15;
        const int nm1 = max(n - 1, 0);
        int ar[MAXNT], bc[MAXNT];
        double sa[MAXNT], va[MAXNT], sb[MAXNT], vb[MAXNT], dd[MAXNT];
#pragma unroll
        for (int I = 0; I < MAXNT; I++) {
          ar[I] = max((int)perm[min(I * TB + r, nm1)], 0); bc[I] = max((int)perm[min(I * TB + cc, nm1)], 0);
          sa[I] = sS[ar[I]]; va[I] = vS[ar[I]]; sb[I] = sS[bc[I]]; vb[I] = vS[bc[I]]; dd[I] = dS[ar[I]];
        }
        // (loads unconditional, through clamped indices, selected afterwards: straight-line code, every load in flight at once)
        double hv[MAXTILES], ev[MAXTILES];
        {
          int I = 0, J = 0;
#pragma unroll
          for (int te = 0; te < MAXTILES; te++) {
            hv[te] = 0.0; ev[te] = 0.0;
            if (te < ntile_all) {                                   // (wave-uniform)
              const int hi = max(ar[I], bc[J]), lo = min(ar[I], bc[J]);
              hv[te] = H[(size_t)hi * ND + lo];
              ev[te] = E[min(hi, NV - 1) * NV + min(lo, NV - 1)];
            }
            if (++J > I) { J = 0; I++; }
          }
        }
#if GFBE_CHAIN_STAMP
        if (t == 0) stamp[8] = (double)wall_clock64();
#endif
        {
          int I = 0, J = 0;
#pragma unroll
          for (int te = 0; te < MAXTILES; te++) {
            if (te < ntile_all) {
              const int a = ar[I], b = bc[J];
              double v = (hv[te] - ((a < NV && b < NV) ? ev[te] : 0.0)) * (sa[I] * sb[J]);
              if (I == J && r == cc) v = __builtin_fma(mu * dd[I], dd[I], v);          // (a == b: the diagonal)
              double q = v * va[I] * (vb[J] * (I != J ? 2.0 : 1.0));
              if (I == nt - 1) {       // (wave-uniform: the last tile row holds the right-hand side row and the padding)
                const int ia = I * TB + r, ib = J * TB + cc;
                if (ia >= n || ib >= n) {
                  q = 0.0;
                  if (ia == n && ib < n) v = rS[b];
                  else if (ib == n && ia < n) v = rS[a];
                  else v = (ia == ib) ? (ia == n ? 1e200 : 1.0) : 0.0;
                }
              }
              vsv += q;
              tiles[(size_t)te * (TB * TB) + tsw(r, cc)] = v;
            }
            if (++J > I) { J = 0; I++; }
          }
        }
      }
      const int tc = TW ? tt - TB * TB : tt;          // (twisted: the second half of the workgroup, beside the dense tiles)
      if (chain_on && tc >= 0 && tc < 3 * CH_BLK) {
        // chain blocks: the 22 blocks A_0..A_10, C_0..C_10 (C_k = S(SB_k, SB_k-1); slot 0 of the couplings is unused — twisted: C_k =
        // S(SB_k, SB_k+1) below the middle block, whose own slot is the unused one) are dealt over three thread groups of 81 — a thread
        // keeps its entry (i, j) and takes every third block (inactive dims: H holds exact zeros there, the diagonal becomes 1)
        const int grp = tc / CH_BLK, e81 = tc - CH_BLK * grp, ei = e81 / CH_NB, ej = e81 - CH_NB * ei;
        auto partner = [](int k) -> int { return TW ? (k > CH_MID ? k - 1 : (k < CH_MID ? k + 1 : k)) : (k > 0 ? k - 1 : 0); };
        double hc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          const int kb = grp + 3 * q;                     // block slot: 0..10 diagonal blocks, 11..21 couplings
          const bool isC = kb >= CH_NC;
          const int k = isC ? kb - CH_NC : kb;
          const int a = T_SB(min(k, CH_NC - 1)) + ei, b = (isC ? T_SB(partner(min(k, CH_NC - 1))) : T_SB(min(k, CH_NC - 1))) + ej;
          hc[q] = H[(size_t)max(a, b) * ND + min(a, b)];
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          const int kb = grp + 3 * q;
          if (kb >= 2 * CH_NC) continue;
          const bool isC = kb >= CH_NC;
          const int k = isC ? kb - CH_NC : kb;
          const int a = T_SB(k) + ei, b = (isC ? T_SB(partner(k)) : T_SB(k)) + ej;
          double v = hc[q] * sS[a] * sS[b];
          if (isC) { if (k == (TW ? CH_MID : 0)) v = 0.0; }
          else if (ei == ej) v = chact[a - T_SB(0)] ? __builtin_fma(mu * dS[a], dS[a], v) : 1.0;
          vsv = __builtin_fma(v * vS[a], vS[b] * (isC ? 2.0 : 1.0), vsv);
          (isC ? Cch : Ach)[k * CH_BLK + e81] = v;
        }
      }
    }
  }
  if (!solved) {
    if (t == 0) { c.done = 1; c.termination = 4; c.status = GFBE_NUMERICAL_FAILURE; c.lin_fail = 1; c.mu = mu; }
    return;
  }
  STAMP(4);
  // dense shares of the dogleg scalars (see k_solve), from the LDS copies of the vectors; E: three threads per row, the row's 25
  // entries of a thread in flight at once
  double n2 = 0.0, gyv = 0.0, vrhs = 0.0, vDv = 0.0, vDy = 0.0, vEv = 0.0, vEy = 0.0, yEy = 0.0;
  for (int a = t; a < NV; a += blockDim.x) { ys[a] = sS[a] * vS[a]; ys[NVP + a] = sS[a] * yT[a]; }   // s v, s y of the visual dims
  __syncthreads();
  for (int a = t; a < ND; a += blockDim.x) {
    const double d2 = dS[a] * dS[a], y = yT[a], v = vS[a];
    n2 += d2 * y * y;
    gyv += gS[a] * y;
    vDv += d2 * v * v;
    vDy += d2 * v * y;
    vrhs += v * rS[a];
  }
  if (t < 3 * NV) {
    const int a = t / 3, b0 = t - 3 * a;
    constexpr int NE = (NV + 2) / 3;
    double ev[NE];
#pragma unroll
    for (int q = 0; q < NE; q++) { const int b = b0 + 3 * q; ev[q] = b < NV ? E[a * NV + b] : 0.0; }   // (rows / columns of inactive dims are zero in E)
    double Ev = 0.0, Ey = 0.0;
#pragma unroll
    for (int q = 0; q < NE; q++) { const int b = min(b0 + 3 * q, NV - 1); Ev = __builtin_fma(ev[q], ys[b], Ev); Ey = __builtin_fma(ev[q], ys[NVP + b], Ey); }
    vEv = ys[a] * Ev; vEy = ys[a] * Ey; yEy = ys[NVP + a] * Ey;
  }
  {
    const double gv[8] = {n2, gyv, vrhs, vDv, vDy, vEv, vEy, yEy};
    block_reduce_multi<8>(gv, 0u, smem, BRM_T);
    n2 = smem[128]; gyv = smem[129]; vrhs = smem[130]; vDv = smem[131]; vDy = smem[132]; vEv = smem[133]; vEy = smem[134]; yEy = smem[135];
  }
  if (t == 0) {
    const double zz = s_zz, vSv = s_vSv;
    c.mu = mu;
    c.G2 = s_keep[0]; c.N2 = n2; c.gy = gyv;
    c.vHv = vSv - mu * vDv + vEv;
    c.vHy = vrhs - mu * vDy + vEy;
    c.yHy = zz - mu * n2 + yEy;
    c.grad_max = s_keep[1];
    c.x_norm = s_keep[2];
    c.have_step = 2;
    c.lin_retry = 0;
  }
  STAMP(5);
#undef STAMP
}
#ifndef GFBE_CHAIN_MINBLOCKS
#define GFBE_CHAIN_MINBLOCKS 2      // workgroups per CU the register allocation of k_solve_chain aims at (2: up to 256 VGPRs; its LDS allows no more than two)
#endif
__global__ __launch_bounds__(S2_THREADS, GFBE_CHAIN_MINBLOCKS) void k_solve_chain(BatchDev d, int retry_pass) { solve_chain_body<0>(d, retry_pass); }
__global__ __launch_bounds__(2 * S2_THREADS, 2) void k_solve_chain_tw(BatchDev d, int retry_pass) { solve_chain_body<1>(d, retry_pass); }
// (round 6) a batch with GNSS dims on the chain kernel: nine tile columns of dense dims (<= 142 + the right-hand side), 45 tiles + ring +
// chain blocks = 138 KB of dynamic LDS — one workgroup per CU, its four waves with the whole register file
__global__ __launch_bounds__(S2_THREADS, 1) void k_solve_chain_wide(BatchDev d, int retry_pass) { solve_chain_body<0, S2_WIDE_NT>(d, retry_pass); }



// =============================================================================================
// k_solve_big: the same step as k_solve for windows whose reduced system does not fit a CU's LDS — the GNSS blocks (anchor,
// 44 receiver clock biases, 11 drifts) bring the dense part to as many as 246 dims: 136 tiles of 2 KB. The scaled, regularised,
// Schur-reduced system and its factor live in global memory (BatchDev::solveS, 512 KB per window, L2 resident for a handful of
// windows); the factorisation is left-looking over 16-column panels:
//     panel j (rows 16 j .. of the lower triangle) is loaded into LDS, downdated by the earlier panels k < j staged through LDS
//     one at a time (P -= L(., k) L(j, k)^T), its diagonal tile is factorised and inverted by one wave (chol_inv_tile16, the
//     register-resident step of k_solve), the rows below become L = P W^T, and the panel goes back to global memory with W in
//     place of the diagonal tile.
// The right-hand side is row n of the matrix (forward substitution for free), the back-substitution runs panel by panel from
// the last. GNSS is an optional path of the reference (gnss_enable: 0 in every shipped yaml): this kernel favours plain,
// structure-agnostic code — any set of active dims, any prior — over the last microsecond (k_solve / k_solve_chain keep the
// batches without GNSS dims).
// =============================================================================================
#define BIG_THREADS 512
enum { BIG_WAVES = BIG_THREADS / 64, BIG_BLD = 244 };    // BIG_BLD: LDS row stride of the staged tile row (up to 15 tiles of 16 columns)
static size_t big_smem_bytes() { return sizeof(double) * ((size_t)(BIG_WAVES + 1) * TB * TB + TB * BIG_BLD); }   // a tile per wave (layout changes, reductions) + the diagonal tile + a tile row

// The factorisation loop of k_solve_big, out of line (its own register allocation: two accumulator tiles, four k-steps of operands
// in flight and — at another time — the 64 registers of the tile step).
//   Left-looking blocked Cholesky on the FP64 matrix cores, operands straight from global memory (L2): for panel j every wave takes
//   the tiles (I, j), I = j + wave, j + wave + 8 and subtracts sum_k L(I, k) L(j, k)^T — 16x16x4 chains, no barrier and no LDS
//   inside the k loop; the operands of four k-steps are loaded together (a lone workgroup has nothing else to hide an L2 round
//   trip behind: one trip per four steps instead of one per step). Wave 0's first tile is the diagonal one, which it factorises
//   and inverts in LDS (chol_inv_tile16); after one block barrier every wave turns its tiles into L(I, j) = P W^T and stores them.
//   Two block barriers per panel. Tw: a tile of the calling wave's own, for the accumulator -> A-operand layout change.
enum { BIG_KCH = 4 };
#ifndef GFBE_BIG_STAMP
#define GFBE_BIG_STAMP 0    // diagnostics: phase stamps of panel 10 into the window's timing slots 8..13
#endif
// the k loop of one panel for a wave with one (TWO = false) or two tiles; B operands from the staged tile row in LDS
template <bool TWO>
__device__ __forceinline__ void big_kloop(const glb_double *Sa0, const glb_double *Sa1, const lds_double *Bj, int j, dbl4 &acc0, dbl4 &acc1) {
  for (int k0 = 0; k0 < j; k0 += BIG_KCH) {
    double va[BIG_KCH][4], vb[BIG_KCH][4], vc[BIG_KCH][4];
#pragma unroll
    for (int u = 0; u < BIG_KCH; u++) {
      const int k = min(k0 + u, j - 1);        // (clamped: the steps past the panel re-load the last one and are not multiplied)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        va[u][q] = -Sa0[TB * k + q];
        vc[u][q] = TWO ? -Sa1[TB * k + q] : 0.0;
        vb[u][q] = Bj[TB * k + q];
      }
    }
#pragma unroll
    for (int u = 0; u < BIG_KCH; u++) {
      if (k0 + u < j) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u][kk], vb[u][kk], acc0, 0, 0, 0);
        if (TWO) {
#pragma unroll
          for (int kk = 0; kk < 4; kk++) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(vc[u][kk], vb[u][kk], acc1, 0, 0, 0);
        }
      }
    }
  }
}
// Brow: [16][BIG_BLD] LDS copy of tile row j (columns 0 .. 16 j - 1), shared by all waves
__device__ __noinline__ void big_factor(glb_double *S, int nt, int n, int lane, int wave, lds_double *Dgl, lds_double *Tw, lds_double *Brow, lds_double *zlast,
                                        lds_int *flag, double *stamp) {
  const int lr = lane & 15, lk = lane >> 4, t = wave * 64 + lane;
#define FSTAMP(i) do { if (GFBE_BIG_STAMP && j == 10 && wave == 0 && lane == 0) stamp[i] = (double)wall_clock64(); } while (0)
  for (int j = 0; j < nt; j++) {
    FSTAMP(8);
    const int I0 = j + wave, I1 = j + wave + BIG_WAVES;
    dbl4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    // tile row j is the B operand of every product of the panel: staged once (coalesced), read from LDS by all waves
    for (int e = t; e < TB * TB * j; e += BIG_THREADS) {
      const int r = e / (TB * j), cc = e - r * (TB * j);
      Brow[r * BIG_BLD + cc] = S[(size_t)(TB * j + r) * BIG_LD + cc];
    }
    if (I0 < nt) {
      const glb_double *Sc0 = S + (size_t)(TB * I0 + lk) * BIG_LD + TB * j + lr;    // accumulator layout: row lk + 4 q, column lr
      const glb_double *Sc1 = S + (size_t)(TB * min(I1, nt - 1) + lk) * BIG_LD + TB * j + lr;
#pragma unroll
      for (int q = 0; q < 4; q++) { acc0[q] = Sc0[(size_t)4 * q * BIG_LD]; acc1[q] = Sc1[(size_t)4 * q * BIG_LD]; }
    }
    __syncthreads();
    if (I0 < nt) {
      // operands: lane (lr, lk) takes the FOUR CONSECUTIVE entries 4 lk .. 4 lk + 3 of its row of a 16 x 16 tile — one 32-byte load —
      // and feeds entry kk to the kk-th 16x16x4 instruction: the k index of a product runs 4 lk + kk instead of 4 kk + lk, the
      // same permutation for both operands, so the sum over the sixteen k's is the same sum in another order. (With the
      // natural 4 kk + lk mapping every 8-byte load picked 32 bytes out of sixteen different 128-byte lines, four times over:
      // the k loop was bound by the vector L1's line rate, 6.6 us per four-step chunk with all eight waves loading.)
      const glb_double *Sa0 = S + (size_t)(TB * I0 + lr) * BIG_LD + 4 * lk, *Sa1 = S + (size_t)(TB * min(I1, nt - 1) + lr) * BIG_LD + 4 * lk;
      const lds_double *Bj = Brow + lr * BIG_BLD + 4 * lk;
      if (I1 < nt) big_kloop<true>(Sa0, Sa1, Bj, j, acc0, acc1);
      else big_kloop<false>(Sa0, Sa1, Bj, j, acc0, acc1);
    }
    FSTAMP(9);
    if (wave == 0) {     // I0 == j: the diagonal tile
#pragma unroll
      for (int q = 0; q < 4; q++) Dgl[tsw(lk + 4 * q, lr)] = acc0[q];
      __threadfence_block();
      __builtin_amdgcn_wave_barrier();
#if GFBE_TILE_PIPELINED
      if (!chol_inv_tile16_p(Dgl, lane, j == nt - 1 ? n % TB : -1, zlast) && lane == 0) *flag = 1;
#else
      if (!chol_inv_tile16(Dgl, lane, j == nt - 1 ? n % TB : -1, zlast) && lane == 0) *flag = 1;
#endif
    }
    FSTAMP(10);
    __syncthreads();
    FSTAMP(11);
    if (*flag) break;
    // L(I, j) = P(I, j) W^T: the accumulator goes through the wave's LDS tile into the A-operand layout, B = W^T
    double wb[4];
#pragma unroll
    for (int q = 0; q < 4; q++) wb[q] = Dgl[tsw(lr, q * 4 + lk)];
    if (wave == 0) {     // W in place of the diagonal tile (zeros above the diagonal)
#pragma unroll
      for (int q = 0; q < 4; q++) S[(size_t)(TB * j + lk + 4 * q) * BIG_LD + TB * j + lr] = Dgl[tsw(lk + 4 * q, lr)];
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int I = u == 0 ? I0 : I1;
      if (I < nt && I != j) {
        const dbl4 pacc = u == 0 ? acc0 : acc1;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; q++) Tw[tsw(lk + 4 * q, lr)] = pacc[q];
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        double pa[4];
#pragma unroll
        for (int q = 0; q < 4; q++) pa[q] = Tw[tsw(lr, q * 4 + lk)];
        dbl4 out = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; kk++) out = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[kk], wb[kk], out, 0, 0, 0);
        glb_double *So = S + (size_t)(TB * I + lk) * BIG_LD + TB * j + lr;
#pragma unroll
        for (int q = 0; q < 4; q++) So[(size_t)4 * q * BIG_LD] = out[q];
      }
    }
    FSTAMP(12);
    __syncthreads();
    FSTAMP(13);
  }
#undef FSTAMP
}

__global__ __launch_bounds__(BIG_THREADS) void k_solve_big(BatchDev d, int retry_pass) {
  const int w = blockIdx.x;
  const WinDesc &ds = d.desc[w];
  WinCtl &c = d.ctl[w];
  if (c.done || c.reuse) return;
  if (retry_pass && !c.lin_retry) return;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *Pn = smem, *Dg = smem + BIG_WAVES * TB * TB;
  __shared__ short perm[ND + TB];
  __shared__ double red[16], ys[2 * ND + TB], zlast[TB];
  __shared__ double s_zz, s_vSv;
  __shared__ int flag, s_nact, wcount[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double *H = d.H + (size_t)w * ND * ND, *g = d.g + (size_t)w * ND;
  double *gsp = d.sp + (size_t)w * ND, *gDp = d.Dp + (size_t)w * ND, *ggts = d.gts + (size_t)w * ND;
  double *gvp = d.vp + (size_t)w * ND, *gyp = d.yp + (size_t)w * ND;
  double *S = d.solveS + (size_t)w * BIG_LD * BIG_LD;
  const bool first = (c.iter == 0);
  double *stamp = d.timing + (size_t)w * 32;       // phase stamps (diagnostics: gfbe_debug_timing)
#define BSTAMP(i) do { if (t == 0) stamp[i] = (double)wall_clock64(); } while (0)
  BSTAMP(0);
  // active-dim list by a wave-level prefix count (dims 0..255 live in waves 0..3)
  {
    const bool on = (t < ND) && ds.act[t];
    const unsigned long long m = __ballot(on);
    if (t < 256 && lane == 0) wcount[wave] = __popcll(m);
    __syncthreads();
    if (t < 256) {
      int base = 0;
      for (int q = 0; q < wave; q++) base += wcount[q];
      if (on) perm[base + __popcll(m & ((1ull << lane) - 1ull))] = t;
      if (t == 0) s_nact = wcount[0] + wcount[1] + wcount[2] + wcount[3];
    }
    __syncthreads();
    for (int a = s_nact + t; a < ND + TB; a += blockDim.x) perm[a] = -1;
  }
  if (first && t == 0) {   // total cost of the first linearisation point (fixed order)
    double cost = 0.0;
    for (int r = 0; r < d.world; r++) cost += d.xa[((size_t)w * d.world + r) * XCHG];
    for (int q = 0; q < ds.n_imu; q++) cost += d.imu_part[((size_t)w * MAX_IMU + q) * IMU_PART + IMU_PART - 2];
    for (int q = 0; q < ds.n_wheel; q++) cost += d.wheel_part[((size_t)w * MAX_WHEEL + q) * WHEEL_PART + WHEEL_PART - 2];
    cost += d.prior_g[(size_t)w * (ND + 2) + ND];
    for (int q = 0; q < ds.n_plane; q++) cost += d.plane_part[((size_t)w * MAX_PLANE + q) * PLANE_PART + PLANE_PART - 2];
    if (ds.use_anchor) cost += d.anchor_part[(size_t)w * ANCHOR_PART + ANCHOR_PART - 2];
    if (ds.gnss_factors) cost += d.gnss_cost[(size_t)w * 2];
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
  {
    const double pv[3] = {g2, gmax, xn2};
    block_reduce_multi<3>(pv, 0x2u, smem, BRM_T);
    g2 = smem[48]; gmax = smem[49]; xn2 = smem[50];
  }
  __syncthreads();
  BSTAMP(1);
  const int n = s_nact, na = n + 1, nt = (na + TB - 1) / TB, npad = nt * TB;
  const double *E = retry_pass ? d.Er + (size_t)w * (NV * NV + NV) : d.E + (size_t)w * NV * NV;
  const double *eg = retry_pass ? E + NV * NV : d.eg + (size_t)w * NV;
  double mu = c.mu;
  bool solved = false, e_valid = true;
  int att = 0;      // factorisation attempts of this linearisation so far (landmark sharding: the pass index counts them)
  while (mu < GF_MAX_MU) {
    if (!e_valid) {
      if (d.sharded) {
        if (retry_pass < min(max(d.opt.sharded_mu_retries, 0), 8)) { if (t == 0) { c.lin_retry = 1; c.mu = mu; } return; }     // (on to pass retry_pass + 1)
        break;
      }
      rebuild_E(d, ds, w, mu, d.E + (size_t)w * NV * NV, d.eg + (size_t)w * NV, false);
    }
    for (int a = t; a < ND; a += blockDim.x) { ys[a] = gsp[a]; ys[ND + a] = gvp[a]; }
    __syncthreads();
    // ---- [ S rhs ; rhs' big ], S = s H s + mu D^2 - s E s, rhs = gt - s eg: lower triangle, diagonal tiles in full
    double vsv = 0.0;
    for (int te = wave; te < nt * (nt + 1) / 2; te += BIG_WAVES) {     // a tile per wave and pass, four entries per lane in flight
      int I, J;
      tri_decode(te, I, J);
      const int ib = J * TB + (lane & 15);
      double hv[4], ev[4];
      int aa[4], kind[4];
      const int b = ib < n ? perm[ib] : -1;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int ia = I * TB + (lane >> 4) + 4 * q;
        kind[q] = 0; aa[q] = 0; hv[q] = 0.0; ev[q] = 0.0;
        if (ia < n && ib < n) {
          const int a = perm[ia], hi = max(a, b), lo = min(a, b);
          aa[q] = a; kind[q] = 1;
          hv[q] = H[(size_t)hi * ND + lo];
          if (hi < NV) ev[q] = E[hi * NV + lo];
        } else if ((ia == n && ib < n) || (ib == n && ia < n)) { aa[q] = perm[min(ia, ib)]; kind[q] = 2; }
        else kind[q] = ia == ib ? (ia == n ? 4 : 3) : 5;
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int ia = I * TB + (lane >> 4) + 4 * q;
        double v;
        if (kind[q] == 1) {
          v = (hv[q] - ev[q]) * (ys[aa[q]] * ys[b]);
          if (aa[q] == b) { const double dp = gDp[b]; v += mu * dp * dp; }
          if (ib <= ia) vsv = __builtin_fma(v * ys[ND + aa[q]], ys[ND + b] * (ia != ib ? 2.0 : 1.0), vsv);
        } else if (kind[q] == 2) v = ggts[aa[q]] - (aa[q] < NV ? gsp[aa[q]] * eg[aa[q]] : 0.0);
        else v = kind[q] == 4 ? 1e200 : (kind[q] == 3 ? 1.0 : 0.0);
        S[(size_t)ia * BIG_LD + ib] = v;
      }
    }
    vsv = block_sum(vsv, red);
    if (t == 0) { flag = 0; s_vSv = vsv; }
    __syncthreads();
    BSTAMP(2);
    big_factor((glb_double *)S, nt, n, lane, wave, (lds_double *)Dg, (lds_double *)(Pn + wave * TB * TB), (lds_double *)(Dg + TB * TB), (lds_double *)zlast,
               (lds_int *)&flag, stamp);
    BSTAMP(3);
    bool ok = (flag == 0);
    if (d.test_fail_chol_iter > 0 && c.iter + 1 == d.test_fail_chol_iter && (d.sharded ? retry_pass : att) < max(d.opt.test_fail_chol_count, 1)) ok = false;   // fault injection: first attempt of that iteration
    if (ok) {
      // z = L^-1 rhs is row n of the factor (its last partial panel was saved before the tile became its own inverse); y = L^-T z
      double zz = 0.0;
      for (int i = t; i < npad; i += blockDim.x) {
        const double z = i < n ? (i / TB == n / TB ? zlast[i % TB] : S[(size_t)n * BIG_LD + i]) : 0.0;
        ys[i] = z;
        zz += z * z;
      }
      zz = block_sum(zz, red);
      if (t == 0) s_zz = zz;
      __syncthreads();
      for (int P = (n - 1) / TB; P >= 0; P--) {
        const int p0 = TB * P;
        if (wave == 0) {      // y_P = W_P^T t_P: lane j < 16 sums its column
          double yj = 0.0;
          if (lane < TB) for (int i = lane; i < TB; i++) yj = __builtin_fma(S[(size_t)(p0 + i) * BIG_LD + p0 + lane], ys[p0 + i], yj);
          __builtin_amdgcn_wave_barrier();
          if (lane < TB) ys[p0 + lane] = p0 + lane < n ? yj : 0.0;
        }
        __syncthreads();
        for (int r = t; r < p0; r += blockDim.x) {                 // z_r -= sum_c L(p0 + c, r) y_(p0 + c) for the rows above
          double s = ys[r];
#pragma unroll
          for (int cc = 0; cc < TB; cc++) s = __builtin_fma(-S[(size_t)(p0 + cc) * BIG_LD + r], ys[p0 + cc], s);
          ys[r] = s;
        }
        __syncthreads();
      }
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
    att++;
  }
  if (!solved) {
    if (t == 0) { c.done = 1; c.termination = 4; c.status = GFBE_NUMERICAL_FAILURE; c.lin_fail = 1; c.mu = mu; }
    return;
  }
  BSTAMP(4);
  // dense shares of the dogleg scalars (the identities of k_solve: one pass over E instead of a second pass over H)
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
    block_reduce_multi<8>(gv, 0u, smem, BRM_T);
    n2 = smem[128]; gyv = smem[129]; vrhs = smem[130]; vDv = smem[131]; vDy = smem[132]; vEv = smem[133]; vEy = smem[134]; yEy = smem[135];
  }
  if (t == 0) {
    c.mu = mu;
    c.G2 = g2; c.N2 = n2; c.gy = gyv;
    c.vHv = s_vSv - mu * vDv + vEv;
    c.vHy = vrhs - mu * vDy + vEy;
    c.yHy = s_zz - mu * n2 + yEy;
    c.grad_max = gmax;
    c.x_norm = xn2;
    c.have_step = 2;
    c.lin_retry = 0;
  }
  BSTAMP(5);
#undef BSTAMP
}

// k_solve's LDS budget: 78 tiles of 2 KB + its static arrays must fit the 160 KB of a CU (it did by 200 bytes in round 2 and
// stopped fitting when the GNSS blocks widened ND: the static arrays are sized by the core dims since). The sum below mirrors the
// __shared__ declarations of the kernel; hipFuncSetAttribute fails at gfbe_create if the real figure exceeds the limit.
static_assert(((NC + 1 + TB - 1) / TB) * (((NC + 1 + TB - 1) / TB) + 1) / 2 * TB * TB * sizeof(double)      // dynamic: the tiles
              + sizeof(short) * (NC + TB) + sizeof(double) * (16 + 2 * NC + TB + 2 + TB) + sizeof(int) * 6      // perm, red, ys, s_zz, s_vSv, zlast, flags
              + 128 /* alignment padding */ <= 160 * 1024, "k_solve: tiles + static LDS exceed a CU's 160 KB");
static size_t solve_smem_bytes() { const int nt = (NC + 1 + TB - 1) / TB;   /* (k_solve never sees the GNSS dims: those batches take k_solve_big) */ return sizeof(double) * (size_t)(nt * (nt + 1) / 2) * TB * TB; }
#ifndef GFBE_CHAIN_LDS_PAD
#define GFBE_CHAIN_LDS_PAD 0       // (measurement) bytes of dynamic LDS k_solve_chain asks for beyond its layout: 24576 leaves ONE workgroup per CU
#endif
static size_t chain_smem_bytes(int ntile, bool tw = false) {     // tw: k_solve_chain_tw — every row of Yr instead of the two-block ring, and the middle block's second downdate
  return sizeof(double) * ((size_t)ntile * TB * TB + 2 * CH_NC * CH_BLK + CH_ZERO + (tw ? (size_t)YALL_ROWS * YALL_LD + CH_BLK : (size_t)2 * RING_ROWS * (ntile > S2_MAX_TILES ? (int)CHAIN_RING_LD_WIDE : chain_ring_ld(ntile))));
}
bool solve_chain_tw_fits(int ntile) { return ntile <= 15; }      // (six tile columns — a free camera extrinsic — would need 169 KB of LDS: those windows keep k_solve_chain)
size_t solve_chain_scratch_doubles() { return (size_t)GYT_COLS * GYT_LD; }
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
  const hipError_t e2 = hipFuncSetAttribute((const void *)k_solve_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(chain_smem_bytes(S2_MAX_TILES) + GFBE_CHAIN_LDS_PAD));
  if (e2 != hipSuccess) return e2;
  const hipError_t e3 = hipFuncSetAttribute((const void *)k_solve_chain_tw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_smem_bytes(15, true));
  if (e3 != hipSuccess) return e3;
  const hipError_t e4 = hipFuncSetAttribute((const void *)k_solve_chain_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_smem_bytes(S2_WIDE_TILES));
  if (e4 != hipSuccess) return e4;
  return hipFuncSetAttribute((const void *)k_solve_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_smem_bytes());
}
// a batch with GNSS dims takes the chain kernel when every window's dense part fits nine tile columns (column 143 of the transposed Yr rows
// is the dump column: n + 1 <= 143) — else k_solve_big
bool solve_chain_wide_fits(int n_dense_max) { return GFBE_SOLVE_WIDE && n_dense_max + 1 <= TB * S2_WIDE_NT - 1; }
void launch_solve(const BatchDev &d, hipStream_t s, int retry_pass) {
  if (d.solve_big && d.solve_wide) hipLaunchKernelGGL(k_solve_chain_wide, dim3(d.B), dim3(S2_THREADS), chain_smem_bytes(S2_WIDE_TILES), s, d, retry_pass);
  else if (d.solve_big) hipLaunchKernelGGL(k_solve_big, dim3(d.B), dim3(BIG_THREADS), big_smem_bytes(), s, d, retry_pass);
  else if (d.solve_mono) hipLaunchKernelGGL(k_solve, dim3(d.B), dim3(SOLVE_THREADS), solve_smem_bytes(), s, d, retry_pass);
  else if (d.solve_tw) hipLaunchKernelGGL(k_solve_chain_tw, dim3(d.B), dim3(2 * S2_THREADS), chain_smem_bytes(d.solve_ntile, true), s, d, retry_pass);
  else hipLaunchKernelGGL(k_solve_chain, dim3(d.B), dim3(S2_THREADS), chain_smem_bytes(d.solve_ntile) + GFBE_CHAIN_LDS_PAD, s, d, retry_pass);
}
void launch_rebuild_E_shard(const BatchDev &d, hipStream_t s) { hipLaunchKernelGGL(k_rebuild_E_shard, dim3(d.B), dim3(1024), 0, s, d); }

}  // namespace gfd

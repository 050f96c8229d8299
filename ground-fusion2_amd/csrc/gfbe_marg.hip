// gfbe_marg.hip — marginalisation of the departing frame on the device.
//
// Reference: Estimator::optimization() marginalisation branches
//   Ground-Fusion++/vins_estimator/src/estimator/estimator.cpp:3394-3693
// and MarginalizationInfo::{preMarginalize, marginalize, getParameterBlocks}
//   Ground-Fusion++/vins_estimator/src/factor/marginalization_factor.cpp:119-330.
//
// The factor set is linearised by the same kernels as the solve (k_vis<2>, k_pair, k_dense mode 2/3,
// k_schur marg) at the re-anchored state; k_marg then does, per window in one workgroup:
//   A, b assembly  ->  eliminate the inverse depths of the landmarks starting at frame 0 (diagonal
//   block, done by k_schur)  ->  eliminate Pose[0] + SpeedBias[0] (or Pose[9]) through an
//   eigen-decomposition pseudo-inverse with eps = 1e-8  ->  A' = V S V^T, J0 = sqrt(S) V^T,
//   r0 = 1/sqrt(S) V^T b'  (marginalization_factor.cpp:278-302).
// The reference eigen-decomposes the whole (15 + #landmarks) block Amm at once; eliminating the
// diagonal landmark block first is the same Schur complement whenever every eigenvalue of Amm is
// above eps (then pinv == inverse) — landmarks with H_ll <= eps are dropped like the reference's
// thresholding does.
#include "gfbe_device.h"
#include "gfbe_factors.h"

namespace gfd {

// device functions defined in gfbe_kernels.hip are not visible here; re-declare the small index
// helpers locally (the marginalisation parity tests of tests/test_gpu_parity.py exercise both copies).
__device__ __forceinline__ int m_vis_loc(int a, int j) {   // pair (0, j)
  if (a < 66) { const int f = a / 6; if (f == 0) return a; if (f == j) return 6 + a - 6 * f; return -1; }
  if (a < 72) return 12 + (a - 66);
  if (a == 72) return 18;
  return -1;
}
__device__ __forceinline__ int m_imu_loc(int a) {          // IMU factor (0, 1)
  if (a < 6) return a;
  if (a < 12) return 15 + a - 6;
  if (a >= 73 && a < 82) return 6 + a - 73;
  if (a >= 82 && a < 91) return 21 + a - 82;
  return -1;
}
__device__ __forceinline__ int m_wheel_loc(int a) {        // wheel factor (0, 1)
  if (a < 6) return a;
  if (a < 12) return 6 + a - 6;
  if (a >= T_EXW && a < T_EXW + 6) return 12 + (a - T_EXW);
  if (a >= T_SX && a <= T_TDW) return 18 + (a - T_SX);
  return -1;
}

__device__ __forceinline__ int m_plane_loc(int a) {        // PlaneFactor of pose 0
  if (a < 6) return a;
  if (a >= T_EXW && a < T_EXW + 6) return 6 + (a - T_EXW);
  if (a >= T_PLR && a < T_PLR + 3) return 12 + (a - T_PLR);
  if (a == T_PLZ) return 15;
  return -1;
}

__device__ __forceinline__ int m_gnss_loc(int a) {         // the frame-0 GNSS factors (k_gnss mode 2: P0 V0 P1 V1 | dt0 ddt0 | dt1 ddt1 | yaw | anc)
  if (a < 3) return a;
  if (a >= T_SB(0) && a < T_SB(0) + 3) return 3 + (a - T_SB(0));
  if (a >= T_POSE(1) && a < T_POSE(1) + 3) return 6 + (a - T_POSE(1));
  if (a >= T_SB(1) && a < T_SB(1) + 3) return 9 + (a - T_SB(1));
  if (a >= T_DT(0, 0) && a < T_DT(0, 0) + 4) return 12 + (a - T_DT(0, 0));
  if (a == T_DDT(0)) return 16;
  if (a >= T_DT(1, 0) && a < T_DT(1, 0) + 4) return 17 + (a - T_DT(1, 0));
  if (a == T_DDT(1)) return 21;
  if (a == T_YAW) return 22;
  if (a >= T_ANC && a < T_ANC + 3) return 23 + (a - T_ANC);
  return -1;
}

__device__ __forceinline__ int m_schur_off(int a, int b) {   // a <= b: offset inside a start-frame partial
  const int I = a >> 4, J = b >> 4;
  return (I * 5 - I * (I - 1) / 2 + (J - I)) * 256 + (a & 15) * 16 + (b & 15);
}

__device__ __forceinline__ int m_vp_off(int a, int b) {   // entry (a <= b) of the 20-column X^T X inside a fused visual partial
  if (a > b) { const int t = a; a = b; b = t; }
  if (b < 16) return a * 16 + b;
  if (a < 16) return 256 + a * 4 + (b - 16);
  return 320 + (a - 16) * 4 + (b - 16);
}

// 1/sqrt(d) from v_rsq_f64 + two Newton steps (full FP64 accuracy; the same sequence as rsqrt_refined of gfbe_solve.hip)
__device__ __forceinline__ double rsqrt_refined_m(double d) {
  double r = __builtin_amdgcn_rsq(d);
  const double hd = 0.5 * d;
  r = r * __builtin_fma(-hd * r, r, 1.5);
  r = r * __builtin_fma(-hd * r, r, 1.5);
  return r;
}

struct MargShared {
  int touched[GFBE_BLK_COUNT];
  int keep_id[GFBE_MAX_PRIOR_BLOCKS];
  int n_keep, n, m;
  int drop_dim[32];       // tangent dims being eliminated densely (15 for OLD, + 5 receiver clock dims with GNSS; 6 for SECOND_NEW)
  int keep_dim[ND];       // tangent dim of kept column k
  int use_imu, use_wheel, use_plane, use_gnss;
  int passthrough;
  int sweeps;
};

// Parallel one-sided (Hestenes) Jacobi on a symmetric n x n matrix held column-major in G
// (G is overwritten by A V); V accumulates the rotations. lambda_j = v_j . g_j.
__device__ void jacobi_eig(double *G, double *V, int n, int ld, double *lam, int *conv_flag, int *sweeps_out) {
  const int t = threadIdx.x, nt = blockDim.x;
  // scale of the matrix: the largest squared column norm (computed redundantly per 16-lane group)
  __shared__ double s_scale;
  if (t == 0) s_scale = 0.0;
  __syncthreads();
  {
    double mx = 0.0;
    for (int j = t; j < n; j += nt) { double a = 0.0; for (int i = 0; i < n; i++) a += G[(size_t)j * ld + i] * G[(size_t)j * ld + i]; mx = fmax(mx, a); }
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o, 64));
    if ((t & 63) == 0 && mx > 0.0) atomicMax((unsigned long long *)&s_scale, (unsigned long long)__double_as_longlong(mx));
  }
  __syncthreads();
  // columns whose norm (= |eigenvalue|) is both below roundoff of the largest one and far below eps
  const double tiny2 = fmax(s_scale * 1e-30, 1e-24);
  const int np = (n + 1) & ~1;                 // even number of "players"; player n (if odd) is a bye
  const int grp = t >> 4, gl = t & 15, ngrp = nt >> 4;
  for (int e = t; e < n * n; e += nt) V[(e / n) * ld + (e % n)] = ((e / n) == (e % n)) ? 1.0 : 0.0;
  __syncthreads();
  for (int sweep = 0; sweep < 40; sweep++) {
    if (t == 0) *conv_flag = 0;
    __syncthreads();
    for (int r = 0; r < np - 1; r++) {
      for (int k = grp; k < np / 2; k += ngrp) {
        int p, q;
        if (k == 0) { p = np - 1; q = r; }
        else { p = (r + k) % (np - 1); q = (r - k + (np - 1)) % (np - 1); }
        if (p >= n || q >= n) continue;
        if (p > q) { const int tt = p; p = q; q = tt; }
        double *gp = G + (size_t)p * ld, *gq = G + (size_t)q * ld;
        double a = 0.0, b = 0.0, c = 0.0;
        for (int i = gl; i < n; i += 16) { const double x = gp[i], y = gq[i]; a += x * x; b += y * y; c += x * y; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 16); b += __shfl_xor(b, o, 16); c += __shfl_xor(c, o, 16); }
        // converged pair: orthogonal to working precision, or both columns are numerically null
        // relative to the matrix scale (their eigenvalues are far below the eps threshold anyway)
        if (fabs(c) <= 1e-14 * sqrt(a * b) || c == 0.0 || (a <= tiny2 && b <= tiny2)) continue;
        if (gl == 0) *conv_flag = 1;
        const double zeta = (b - a) / (2.0 * c);
        const double tn = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / sqrt(1.0 + tn * tn), sn = cs * tn;
        double *vp = V + (size_t)p * ld, *vq = V + (size_t)q * ld;
        for (int i = gl; i < n; i += 16) {
          const double x = gp[i], y = gq[i];
          gp[i] = cs * x - sn * y; gq[i] = sn * x + cs * y;
          const double u = vp[i], v = vq[i];
          vp[i] = cs * u - sn * v; vq[i] = sn * u + cs * v;
        }
      }
      __syncthreads();
    }
    const int any = *conv_flag;
    __syncthreads();
    if (sweeps_out && t == 0) *sweeps_out = sweep + 1;
    if (!any) break;
  }
  for (int j = t; j < n; j += nt) {
    double s = 0.0;
    for (int i = 0; i < n; i++) s += V[(size_t)j * ld + i] * G[(size_t)j * ld + i];
    lam[j] = s;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------------
// Divide & conquer for the symmetric tridiagonal matrix (round 6; VERDICT round 5 item 4: the implicit QL below is a sequential scalar
// recurrence — ~7000 rotations of ~230 ns on ONE lane for the 86-dim prior, 1.6 of the eigen mode's 2.0 ms). Cuppen's method with the
// Gu-Eisenstat vectors and LAPACK's deflation rules (dstedc / dlaed2 / dlaed4's problem, restated for a workgroup):
//   tear     T = diag(T_1, T_2, ...) + sum_b |beta_b| u_b u_b^T at EVERY even index: leaves of size 2 (closed form), log2(n / 2) levels
//   merge    two neighbours with eigen-decompositions (D_1, Q_1), (D_2, Q_2) and the torn coupling beta: D = [D_1; D_2], z = [last row
//            of Q_1; sign(beta) first row of Q_2], rho = |beta| |z|^2 — the eigenvalues of D + rho z z^T are the roots of the secular
//            equation 1 + rho sum_j z_j^2 / (d_j - lambda), one in every gap between consecutive poles; its eigenvectors are
//            (D - lambda_i)^-1 zhat, with zhat recomputed from the computed roots (Gu / Eisenstat: orthogonal to working accuracy whatever
//            the accuracy of the roots); poles with a negligible z_j, and pairs of poles a Givens rotation can merge without a visible
//            change, are deflated first (their eigenpairs pass through) — what makes clustered spectra safe.
// ALL merges of a level run side by side: every phase below is one pass of the workgroup over the n entries / roots / rows, each finding
// its merge from its index (blocks of a level are unions of 2^level leaves), with a block barrier between the phases:
//   M1 z and D, M2 rank sort of D inside each merge, M3 (ONE thread per merge) norms, tolerance, the deflation scan — the Givens rotations
//   are only recorded —, compaction of the surviving poles, M3b rotations applied to the rows, M4 one secular root per thread (the
//   "middle way" rational iteration from the nearer pole, safeguarded by bisection: ~5 iterations; the root is kept as origin pole +
//   offset, so that every difference d_j - lambda_i is formed without cancellation), M5 zhat, M6 norms, M7 the vectors into Vt
//   (block-diagonal, transposed), M8 Q <- Q V in place, a WAVE per row (the row's old entries are read from LDS by all lanes before the
//   wave overwrites them).
// Q: n x n (row stride ld): on return column j = eigenvector j of T (unsorted), lam (= wk[0 .. n)) its eigenvalue. Vt: n x n scratch.
// dd / ee are not changed. wk: 12 n doubles. Everything in LDS. A numpy prototype of exactly this structure (scratch-free of LAPACK)
// reaches 3e-15 |T| in residual and orthogonality on random, graded (1e16-conditioned), split, clustered and Wilkinson matrices
// (tests/dc_eig_np.py, tests/test_oracle_numpy.py::test_divide_and_conquer_*); the device code is compared with the QL path and the oracle.
// ---------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) double mlds_double;
typedef __attribute__((address_space(3))) int mlds_int;
__device__ __forceinline__ void dc_geom(const int j, const int L, const int n, int &q, int &a, int &b, int &c) {
  q = j >> (L + 2); a = q << (L + 2); b = a + (2 << L); c = min(a + (4 << L), n);
}
// root i of 1 + rho sum_j z_j^2 / (dl_j - lambda) between the poles i and i + 1 (the last one: beyond pole k - 1): lambda = dl[o] + mu.
// EIGHT lanes per root (sub = the lane's place in its group of eight consecutive lanes): lane sub holds the poles sub, sub + 8, ... (at
// most DC_TERMS: n <= MARG_LDS_N) and their z^2 in registers for the whole iteration, the four sums of an evaluation are reduced over
// the group by an xor butterfly — every lane of the group ends with the same bits, so the iteration's branches agree inside a group.
// (One thread per root walking k poles through LDS with two divisions each: 500 of the 780 us of the first version.)
#define DC_TERMS 12
__device__ __forceinline__ double dc_lane_bcast(double v, int src) {   // src is wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
__device__ __forceinline__ double dc_group_sum(double v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
  return v;
}
__device__ __forceinline__ void dc_secular_root8(const int i, const int k, const int sub, const mlds_double *dl, const mlds_double *zl, const double rho,
                                                 int &o_out, double &mu_out) {
  const double EPSD = 2.220446049250313e-16;
  double dj[DC_TERMS], z2[DC_TERMS];
#pragma unroll
  for (int u = 0; u < DC_TERMS; u++) {
    const int j = sub + 8 * u;
    const double zj = j < k ? zl[j] : 0.0;
    dj[u] = j < k ? dl[j] : 1e300;      // (a pole far away with no weight: its terms are exact zeros)
    z2[u] = zj * zj;
  }
  const bool last = i == k - 1;
  const double di = dl[i], dn = last ? di : dl[i + 1];
  double lo, hi, mu, dorig, guess = 0.0;
  int o;
  if (!last) {
    const double mid = 0.5 * (dn - di);
    double f = 0.0;
#pragma unroll
    for (int u = 0; u < DC_TERMS; u++) if (8 * u < k) f += z2[u] / ((dj[u] - di) - mid);      // (8 u < k: uniform — the poles of a small merge sit in the first terms)
    f = 1.0 + rho * dc_group_sum(f);
    o = f >= 0.0 ? i : i + 1;
    lo = f >= 0.0 ? 0.0 : -mid; hi = f >= 0.0 ? mid : 0.0;
    // first guess (dlaed4's idea): the two poles next to the root kept as they are, every other term frozen at the midpoint —
    // C + rho z_i^2 / (D_i - x) + rho z_(i+1)^2 / (D_(i+1) - x) = 0 in the origin's coordinates; the root inside the bracket, if there is one
    {
      const double zi = zl[i], zn = zl[i + 1], S = rho * zi * zi, R = rho * zn * zn;
      const double Di = o == i ? 0.0 : -(dn - di), Dj = o == i ? dn - di : 0.0, xm = o == i ? mid : -mid;
      const double C = f - S / (Di - xm) - R / (Dj - xm);
      const double qa = C, qb = -(C * (Di + Dj) + S + R), qc = C * Di * Dj + S * Dj + R * Di;
      double x0 = xm;
      if (qa != 0.0) {
        double disc = qb * qb - 4.0 * qa * qc;
        if (disc < 0.0) disc = 0.0;
        const double sq = sqrt(disc), qq = -0.5 * (qb + (qb >= 0.0 ? sq : -sq));
        const double x1 = qq / qa, x2 = qq != 0.0 ? qc / qq : x1;
        x0 = (lo < x1 && x1 < hi) ? x1 : x2;
      } else if (qb != 0.0) x0 = -qc / qb;
      guess = (lo < x0 && x0 < hi) ? x0 : 0.5 * (lo + hi);
    }
  } else {
    double z2s = 0.0;
#pragma unroll
    for (int u = 0; u < DC_TERMS; u++) z2s += z2[u];
    o = i; lo = 0.0; hi = rho * dc_group_sum(z2s);
  }
  dorig = o == i ? di : dn;
  mu = last ? 0.5 * (lo + hi) : guess;
  const double Dpi = di - dorig, Dpj = dn - dorig;      // the two poles next to the root, from the origin
  for (int it = 0; it < 80; it++) {
    double psi = 0.0, dpsi = 0.0, phi = 0.0, dphi = 0.0;
#pragma unroll
    for (int u = 0; u < DC_TERMS; u++) {
      if (8 * u < k) {
        const double rden = 1.0 / ((dj[u] - dorig) - mu), tq = z2[u] * rden, dq = tq * rden;
        const bool left = sub + 8 * u <= i;
        psi += left ? tq : 0.0; dpsi += left ? dq : 0.0;
        phi += left ? 0.0 : tq; dphi += left ? 0.0 : dq;
      }
    }
    psi = rho * dc_group_sum(psi); dpsi = rho * dc_group_sum(dpsi);
    phi = rho * dc_group_sum(phi); dphi = rho * dc_group_sum(dphi);
    const double g = 1.0 + psi + phi;
    const double err = 8.0 * EPSD * (1.0 + fabs(psi) + fabs(phi)) + fabs(mu) * (dpsi + dphi) * EPSD;
    if (!(fabs(g) > err)) break;
    if (g > 0.0) hi = mu; else lo = mu;
    double x;
    if (!last) {
      // the two-pole model: psi ~ sc + S / (Di - x), phi ~ rc + R / (Dj - x) through value and slope at mu (x: the step from mu)
      const double Di = Dpi - mu, Dj = Dpj - mu;
      const double S = dpsi * Di * Di, sc = psi - dpsi * Di, R = dphi * Dj * Dj, rc = phi - dphi * Dj;
      const double cst = 1.0 + sc + rc;
      const double qa = cst, qb = -(cst * (Di + Dj) + S + R), qc = cst * Di * Dj + S * Dj + R * Di;
      if (qa == 0.0) x = qb != 0.0 ? -qc / qb : 0.0;
      else {
        double disc = qb * qb - 4.0 * qa * qc;
        if (disc < 0.0) disc = 0.0;
        const double sq = sqrt(disc), qq = -0.5 * (qb + (qb >= 0.0 ? sq : -sq));
        const double x1 = qq / qa, x2 = qq != 0.0 ? qc / qq : x1;
        x = (Di < x1 && x1 < Dj) ? x1 : x2;
      }
    } else {
      const double Dl = -mu;                               // the last pole, seen from mu
      const double S = dpsi * Dl * Dl, sc = psi - dpsi * Dl, cst = 1.0 + sc;
      x = cst != 0.0 ? Dl + S / cst : 0.0;                 // cst + S / (Dl - x) = 0
    }
    double nw = mu + x;
    if (!(lo < nw && nw < hi)) nw = 0.5 * (lo + hi);      // (also catches a NaN)
    if (nw == mu) break;
    mu = nw;
  }
  o_out = o; mu_out = mu;
}
__device__ void tridiag_dc(mlds_double *Q, mlds_double *Vt, const int n, const int ld, const mlds_double *dd, const mlds_double *ee, mlds_double *wk, double *stamp = nullptr) {
  // (diagnostics build: time per phase, summed over the levels, into stamp[13 ..]: leaves, M1+M2, M3, M3b+M4, M5, M6, M7, M8)
  double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = (stamp && threadIdx.x == 0) ? (long long)wall_clock64() : 0;
#define DCSTAMP(i) do { if (stamp && threadIdx.x == 0) { const long long now_ = (long long)wall_clock64(); ph[i] += (double)(now_ - tprev); tprev = now_; } } while (0)
  const int t = threadIdx.x, nt = blockDim.x;
  mlds_double *lam = wk, *Ds = lam + n, *zs = Ds + n, *dl = zs + n, *zl = dl + n, *mu = zl + n, *zh = mu + n, *rc = zh + n, *rs = rc + n;
  mlds_int *col = (mlds_int *)(rs + n), *cidx = col + n, *orig = cidx + n, *rp = orig + n, *rq = rp + n;
  __shared__ int s_k[64], s_nrot[64], s_trig[64];
  __shared__ double s_rho[64], s_tol[64];
  const double EPSD = 2.220446049250313e-16;
  for (int e = t; e < n * n; e += nt) Q[(size_t)(e / n) * ld + (e % n)] = 0.0;
  __syncthreads();
  // ---- leaves of size 2 (the last one 1 when n is odd) of the torn matrix: closed form
  for (int q = t; 2 * q < n; q += nt) {
    const int a = 2 * q;
    const double da = dd[a] - (a > 0 ? fabs(ee[a - 1]) : 0.0);
    if (a + 1 < n) {
      const double dc = dd[a + 1] - (a + 2 < n ? fabs(ee[a + 1]) : 0.0), off = ee[a];
      double l0 = da, l1 = dc, cs = 1.0, sn = 0.0;
      if (off != 0.0) {
        const double th = (dc - da) / (2.0 * off);
        const double tt = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
        cs = 1.0 / sqrt(tt * tt + 1.0); sn = tt * cs;
        l0 = da - tt * off; l1 = dc + tt * off;
      }
      lam[a] = l0; lam[a + 1] = l1;
      Q[(size_t)a * ld + a] = cs; Q[(size_t)a * ld + a + 1] = sn; Q[(size_t)(a + 1) * ld + a] = -sn; Q[(size_t)(a + 1) * ld + a + 1] = cs;
    } else {
      lam[a] = da; Q[(size_t)a * ld + a] = 1.0;
    }
  }
  __syncthreads();
  DCSTAMP(0);
  for (int L = 0; (2 << L) < n; L++) {
    // ---- M1: D and z of every merge (unsorted, in dl / zl)
    for (int j = t; j < n; j += nt) {
      int q, a, b, c; dc_geom(j, L, n, q, a, b, c);
      if (b < n) { dl[j] = lam[j]; zl[j] = j < b ? Q[(size_t)(b - 1) * ld + j] : (ee[b - 1] >= 0.0 ? Q[(size_t)b * ld + j] : -Q[(size_t)b * ld + j]); }
    }
    __syncthreads();
    // ---- M2: rank sort inside each merge
    for (int j = t; j < n; j += nt) {
      int q, a, b, c; dc_geom(j, L, n, q, a, b, c);
      if (b < n) {
        const double dj = dl[j];
        int r = 0, i = a;
        for (; i + 4 <= c; i += 4) {
          double di[4];
#pragma unroll
          for (int v = 0; v < 4; v++) di[v] = dl[i + v];
#pragma unroll
          for (int v = 0; v < 4; v++) r += (di[v] < dj || (di[v] == dj && i + v < j)) ? 1 : 0;
        }
        for (; i < c; i++) { const double di = dl[i]; r += (di < dj || (di == dj && i < j)) ? 1 : 0; }
        Ds[a + r] = dj; zs[a + r] = zl[j]; col[a + r] = j;
      }
    }
    __syncthreads();
    DCSTAMP(1);
    // ---- M3: deflation. The scan over a merge's poles is sequential only through the Givens rotations that merge close poles — and
    // those are rare: a thread per (sorted) entry normalises z, forms the tolerance (every thread of a merge sums the same numbers in
    // the same order: the same bits), flags negligible z (M3a), then tests its pair (predecessor among the surviving poles, itself) for
    // a rotation (M3b); a merge without any is compacted by counting (M3c), one with some by ONE thread's scan (dc_scan_merge).
    for (int j = t; j < n; j += nt) {
      int q, a, b, c; dc_geom(j, L, n, q, a, b, c);
      if (b < n) {
        double zn2 = 0.0, dmax = 0.0, zmx = 0.0;
        {
          int r = a;
          for (; r + 4 <= c; r += 4) {
            double zv[4], dv[4];
#pragma unroll
            for (int v = 0; v < 4; v++) { zv[v] = zs[r + v]; dv[v] = Ds[r + v]; }
#pragma unroll
            for (int v = 0; v < 4; v++) { zn2 += zv[v] * zv[v]; dmax = fmax(dmax, fabs(dv[v])); zmx = fmax(zmx, fabs(zv[v])); }
          }
          for (; r < c; r++) { const double zr = zs[r]; zn2 += zr * zr; dmax = fmax(dmax, fabs(Ds[r])); zmx = fmax(zmx, fabs(zr)); }
        }
        const double inv = 1.0 / sqrt(zn2), rho = fabs(ee[b - 1]) * zn2;
        const double tol = 8.0 * EPSD * fmax(dmax, zmx * inv);
        const double zr = zs[j] * inv;
        mu[j] = zr;                                            // (normalised z by sorted position: mu is free until M4)
        orig[j] = (rho * fabs(zr) <= tol) ? 1 : 0;             // (flags: orig is free until M4)
        if (j == a) { s_rho[q] = rho; s_tol[q] = tol; s_trig[q] = 0; s_nrot[q] = 0; }
      } else if (j == a) { s_k[q] = -1; s_nrot[q] = 0; s_trig[q] = 0; }
    }
    __syncthreads();
    for (int j = t; j < n; j += nt) {
      int q, a, b, c; dc_geom(j, L, n, q, a, b, c);
      if (b < n && !orig[j]) {
        int p = j - 1;
        while (p >= a && orig[p]) p--;
        if (p >= a) {
          const double zp = mu[p], zr = mu[j], tau = sqrt(zp * zp + zr * zr);
          if (fabs((Ds[j] - Ds[p]) * (zr / tau) * (-zp / tau)) <= s_tol[q]) s_trig[q] = 1;
        }
      }
    }
    __syncthreads();
    for (int j = t; j < n; j += nt) {
      int q, a, b, c; dc_geom(j, L, n, q, a, b, c);
      if (b >= n) continue;
      if (!s_trig[q]) {
        if (orig[j]) lam[col[j]] = Ds[j];
        else {
          int pos = 0, r = a;
          for (; r + 8 <= j; r += 8) {
            int fl[8];
#pragma unroll
            for (int v = 0; v < 8; v++) fl[v] = orig[r + v];
#pragma unroll
            for (int v = 0; v < 8; v++) pos += fl[v] ? 0 : 1;
          }
          for (; r < j; r++) pos += orig[r] ? 0 : 1;
          dl[a + pos] = Ds[j]; zl[a + pos] = mu[j]; cidx[a + pos] = col[j];
        }
        if (j == a) { int k = 0; for (int r = a; r < c; r++) k += orig[r] ? 0 : 1; s_k[q] = k; }
      } else if (j == a) {
        // the sequential scan of LAPACK's dlaed2 (rotations recorded, applied in M3b)
        const double rho = s_rho[q], tol = s_tol[q];
        for (int r = a; r < c; r++) zs[r] = mu[r];
        int k = 0, nrot = 0, prev = -1;
        for (int r = a; r < c; r++) {
          if (rho * fabs(zs[r]) <= tol) { lam[col[r]] = Ds[r]; continue; }
          if (prev >= 0) {
            const double zp = zs[prev], zr = zs[r], tau = sqrt(zp * zp + zr * zr);
            const double cg = zr / tau, sg = -zp / tau, dp = Ds[prev], dr = Ds[r];
            if (fabs((dr - dp) * cg * sg) <= tol) {      // the two poles merge: z_prev <- 0 by a rotation of their columns
              zs[r] = tau; zs[prev] = 0.0;
              rp[a + nrot] = col[prev]; rq[a + nrot] = col[r]; rc[a + nrot] = cg; rs[a + nrot] = sg; nrot++;
              Ds[prev] = dp * cg * cg + dr * sg * sg;
              Ds[r] = dp * sg * sg + dr * cg * cg;
              lam[col[prev]] = Ds[prev];
            } else {
              dl[a + k] = Ds[prev]; zl[a + k] = zs[prev]; cidx[a + k] = col[prev]; k++;
            }
          }
          prev = r;
        }
        if (prev >= 0) { dl[a + k] = Ds[prev]; zl[a + k] = zs[prev]; cidx[a + k] = col[prev]; k++; }
        s_k[q] = k; s_nrot[q] = nrot;
      }
    }
    __syncthreads();
    DCSTAMP(2);
    // ---- M3b: the recorded rotations, a thread per row
    for (int j = t; j < n; j += nt) {
      int q, a, b, c; dc_geom(j, L, n, q, a, b, c);
      const int nrot = s_nrot[q];
      for (int u = 0; u < nrot; u++) {
        const int p = rp[a + u], pq = rq[a + u];
        const double cg = rc[a + u], sg = rs[a + u];
        const double x = Q[(size_t)j * ld + p], y = Q[(size_t)j * ld + pq];
        Q[(size_t)j * ld + p] = cg * x + sg * y;
        Q[(size_t)j * ld + pq] = cg * y - sg * x;
      }
    }
    // ---- M4: the secular roots, eight lanes per root
    for (int j0 = 0; j0 < n; j0 += nt >> 3) {
      const int j = j0 + (t >> 3), sub = t & 7;
      int q = 0, a = 0, b = 0, c = 0, k = 0, i = 0;
      if (j < n) { dc_geom(j, L, n, q, a, b, c); k = s_k[q]; i = j - a; }
      if (j < n && i < k) {      // (all eight lanes of a group take the same branch)
        int o = 0; double m_ = 0.0;
        if (k == 1) m_ = s_rho[q] * zl[a] * zl[a];
        else dc_secular_root8(i, k, sub, dl + a, zl + a, s_rho[q], o, m_);
        if (sub == 0) { orig[j] = o; mu[j] = m_; Ds[j] = dl[a + o]; lam[cidx[j]] = dl[a + o] + m_; }      // (Ds: the origin pole of root j from here on)
      }
    }
    __syncthreads();
    DCSTAMP(3);
    // ---- M5: zhat (Gu / Eisenstat), a thread per pole (four independent partial products: the LDS loads of four roots in flight)
    for (int j = t; j < n; j += nt) {
      int q, a, b, c; dc_geom(j, L, n, q, a, b, c);
      const int k = s_k[q], jj = j - a;
      if (jj < k && k >= 2) {
        const double dj = dl[j];
        double pr[4] = {(Ds[j] - dj) + mu[j], 1.0, 1.0, 1.0};
        int i = 0;
        for (; i + 4 <= k; i += 4) {
          double nu[4], de[4];
#pragma unroll
          for (int v = 0; v < 4; v++) { nu[v] = (Ds[a + i + v] - dj) + mu[a + i + v]; de[v] = dl[a + i + v] - dj; }
#pragma unroll
          for (int v = 0; v < 4; v++) if (i + v != jj) pr[v] *= nu[v] / de[v];
        }
        for (; i < k; i++) if (i != jj) pr[0] *= ((Ds[a + i] - dj) + mu[a + i]) / (dl[a + i] - dj);
        const double zv = sqrt(fabs((pr[0] * pr[1]) * (pr[2] * pr[3])) / s_rho[q]);
        zh[j] = zl[j] >= 0.0 ? zv : -zv;
      }
    }
    __syncthreads();
    DCSTAMP(4);
    // ---- M6: 1 / |v_i|, a thread per root (into zs: free since M3)
    for (int j = t; j < n; j += nt) {
      int q, a, b, c; dc_geom(j, L, n, q, a, b, c);
      const int k = s_k[q], i = j - a;
      if (i < k && k >= 2) {
        const double dorig = Ds[j], m_ = mu[j];
        double s2[4] = {0.0, 0.0, 0.0, 0.0};
        int u = 0;
        for (; u + 4 <= k; u += 4) {
          double v[4];
#pragma unroll
          for (int x = 0; x < 4; x++) v[x] = zh[a + u + x] / ((dl[a + u + x] - dorig) - m_);
#pragma unroll
          for (int x = 0; x < 4; x++) s2[x] += v[x] * v[x];
        }
        for (; u < k; u++) { const double v = zh[a + u] / ((dl[a + u] - dorig) - m_); s2[0] += v * v; }
        zs[j] = 1.0 / sqrt((s2[0] + s2[1]) + (s2[2] + s2[3]));
      }
    }
    __syncthreads();
    DCSTAMP(5);
    // ---- M7: the vectors, transposed and block-diagonal: Vt[pole][root]
    for (int e = t; e < n * n; e += nt) {
      const int pj = e / n, ri = e - pj * n;
      int q, a, b, c; dc_geom(pj, L, n, q, a, b, c);
      const int k = s_k[q];
      if (k >= 2 && pj - a < k && ri >= a && ri - a < k)
        Vt[(size_t)pj * ld + ri] = zh[pj] / ((dl[pj] - Ds[ri]) - mu[ri]) * zs[ri];
    }
    __syncthreads();
    DCSTAMP(6);
    // ---- M8: Q <- Q V on the columns of the surviving poles, in place: a 4 x 4 tile of (rows, roots) per thread in registers — a merge
    // starts at a multiple of four, so no tile straddles two merges —, sixteen products per eight LDS loads; every tile is formed before
    // any is written (a block barrier in between). (A wave per row with the products along its lanes: 94 us summed over the levels.)
    {
      const int nt4 = (n + 3) >> 2;
      for (int e0 = 0; e0 < nt4 * nt4; e0 += nt) {
        const int e = e0 + t;
        double acc[4][4];
#pragma unroll
        for (int x = 0; x < 4; x++)
#pragma unroll
          for (int y = 0; y < 4; y++) acc[x][y] = 0.0;
        int r0 = 0, i0 = 0, a = 0, c = 0, k = 0;
        bool on = false;
        if (e < nt4 * nt4) {
          const int ty = e / nt4, tx = e - ty * nt4;
          r0 = 4 * ty; i0 = 4 * tx;
          int q, b; dc_geom(r0, L, n, q, a, b, c);
          k = s_k[q];
          on = b < n && k >= 2 && i0 >= a && i0 - a < k;
        }
        if (on) {
          const mlds_double *qr[4];
#pragma unroll
          for (int x = 0; x < 4; x++) qr[x] = Q + (size_t)min(r0 + x, c - 1) * ld;
          const mlds_double *vt = Vt + (size_t)a * ld + i0;
          for (int u0 = 0; u0 < k; u0 += 2) {
            const int u1 = min(u0 + 1, k - 1);
            const int ca = cidx[a + u0], cb = cidx[a + u1];
            double xa[4], xb[4], va[4], vb[4];
#pragma unroll
            for (int x = 0; x < 4; x++) { xa[x] = qr[x][ca]; xb[x] = qr[x][cb]; }
#pragma unroll
            for (int y = 0; y < 4; y++) { va[y] = vt[(size_t)u0 * ld + y]; vb[y] = vt[(size_t)u1 * ld + y]; }
            const double wb = u0 + 1 < k ? 1.0 : 0.0;
#pragma unroll
            for (int x = 0; x < 4; x++)
#pragma unroll
              for (int y = 0; y < 4; y++) acc[x][y] = __builtin_fma(xb[x] * wb, vb[y], __builtin_fma(xa[x], va[y], acc[x][y]));
          }
        }
        __syncthreads();
        if (on) {
#pragma unroll
          for (int y = 0; y < 4; y++) {
            if (i0 + y - a < k) {
              const int co = cidx[i0 + y];
#pragma unroll
              for (int x = 0; x < 4; x++) if (r0 + x < c) Q[(size_t)(r0 + x) * ld + co] = acc[x][y];
            }
          }
        }
      }
    }
    __syncthreads();
    DCSTAMP(7);
  }
  if (stamp && threadIdx.x == 0) for (int q = 0; q < 8; q++) stamp[13 + q] = ph[q];
#undef DCSTAMP
}

// ---------------------------------------------------------------------------------------------------------------------------
// Symmetric eigen-decomposition the way Eigen's SelfAdjointEigenSolver — the reference's (marginalization_factor.cpp:281) — does it:
// Householder tridiagonalisation, then implicit-shift QL on the tridiagonal matrix with the rotations accumulated into the
// eigenvectors (round 5: the one-sided Jacobi above needs up to 40 sweeps on the 1e14-conditioned A' of a window — 7.7 ms for the
// 86-dim prior, six times the rest of the call; this takes 2.2 ms: 0.22 + 0.14 + 1.9 for the three phases below).
//   phase 1  A = Q T Q^T: n - 2 reflectors H_k = I - beta v v^T, each a symmetric matrix-vector product and a rank-2 update of the
//            trailing block by the whole workgroup (four block barriers per reflector); v stays in column k of A below the diagonal
//   phase 2  Q = H_0 ... H_(n-3) accumulated backwards into Z (row-major Q), then transposed in place: row j of Z = column j of Q
//   phase 3  QL with implicit Wilkinson shifts (tqli / tql2): the rotations of a sweep are a sequential scalar recurrence — ONE lane
//            runs it (1/sqrt by v_rsq_f64 + two Newton steps, the next entries of d / e fetched one step ahead: ~250 ns per
//            rotation, ~11 us per sweep of the 86-dim prior, ~165 sweeps) and leaves (c, s) in LDS; the eigenvector rows are updated by the threads 64.. (one per component, the
//            rotated column carried in a register) WHILE the lane already runs the next sweep (double-buffered lists).
// A: n x n, row stride ld, symmetric (both triangles), destroyed. Z: n x n, row stride ld: row j = eigenvector j (unit length),
// lam[j] its eigenvalue (unsorted). wk: 6 n + (blockDim / 128) n doubles of scratch (LDS when it fits). Returns 0, or 1 when an
// eigenvalue has not converged after 60 sweeps (the caller falls back to the Jacobi).
// (PD: the pointer type of the matrices and the scratch — address-space-3 pointers when everything is in LDS, the n <= MARG_LDS_N of
//  every prior the reference builds: through generic pointers every access of the sequential lane and of the rotation loop is a
//  FLAT instruction with twice the latency, 3.0 instead of 1.x ms measured.)
#ifndef GFBE_EIG_DC
#define GFBE_EIG_DC 1      // phase 3 of the eigen-decomposition of an in-LDS A': divide & conquer (0: the implicit QL iteration)
#endif
#ifndef GFBE_EIG_NOAPPLY
#define GFBE_EIG_NOAPPLY 0      // (timing experiment: the scalar lane alone)
#endif
// park (round 6; LDS instantiation only): an n x n global scratch. When given, phase 3 is the divide & conquer above instead of the QL
// iteration: Z (= Q of the tridiagonalisation) is parked there, tridiag_dc runs on the two LDS matrices, and the eigenvectors
// Q_dc^T Z are written to `park` (row j = eigenvector j, row stride n) — return value 2; a non-finite eigenvalue falls back to the QL.
template <typename PD>
__device__ int tridiag_ql_eig(PD A, PD Z, int n, int ld, double *lam, PD wk, double *stamp = nullptr, double *park = nullptr) {
#define ESTAMP(i) do { if (stamp && threadIdx.x == 0) stamp[i] = (double)wall_clock64(); } while (0)
  ESTAMP(8);
  const int t = threadIdx.x, nt = blockDim.x, lane = t & 63;
  const int NPART = nt >> 7;                                  // partial sums per row of a matrix-vector product (threads / 128)
  PD dd = wk, ee = dd + n, vv = ee + n, pv = vv + n, cs = pv + n, pp = cs + 2 * n;     // cs | pp: the two lists of (c, s) pairs of phase 3
  __shared__ double s_beta, s_K;
  __shared__ int s_rng[2][3], s_fail;                         // per list: first rotation, last rotation (hi < lo: none), finished
  if (n == 1) { if (t == 0) { lam[0] = A[0]; Z[0] = 1.0; } __syncthreads(); return 0; }
  // ---- phase 1
  for (int k = 0; k + 2 < n; k++) {
    const int m = n - k - 1;                                  // rows k + 1 .. n - 1
    if (t < 64) {
      double sig = 0.0;
      for (int i = 1 + lane; i < m; i += 64) { const double x = A[(size_t)(k + 1 + i) * ld + k]; sig += x * x; }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sig += __shfl_xor(sig, o, 64);
      const double x0 = A[(size_t)(k + 1) * ld + k];
      double alpha = x0, beta = 0.0, v0 = 0.0;
      if (sig > 0.0) {
        const double nrm = sqrt(x0 * x0 + sig);
        alpha = x0 > 0.0 ? -nrm : nrm;
        v0 = x0 - alpha;
        beta = 2.0 / (v0 * v0 + sig);
      }
      for (int i = lane; i < m; i += 64) vv[i] = beta == 0.0 ? 0.0 : (i == 0 ? v0 : A[(size_t)(k + 1 + i) * ld + k]);
      if (lane == 0) { dd[k] = A[(size_t)k * ld + k]; ee[k] = alpha; lam[k] = beta; s_beta = beta; A[(size_t)(k + 1) * ld + k] = v0; }
    }
    __syncthreads();
    const double beta = s_beta;
    if (beta != 0.0) {                                        // (block-uniform)
      // p = beta A22 v: thread (i, part) sums every NPART-th term of row i — read down the COLUMN i (A22 is symmetric): lanes along i
      for (int ib = 0; ib < m; ib += 128) {
        const int i = ib + (t & 127), part = t >> 7;
        if (i < m) {
          double acc = 0.0;
          for (int c = part; c < m; c += NPART) acc = __builtin_fma(A[(size_t)(k + 1 + c) * ld + (k + 1 + i)], vv[c], acc);
          pp[part * n + i] = acc;
        }
      }
      __syncthreads();
      if (t < 64) {
        double vtp = 0.0;
        for (int i = lane; i < m; i += 64) {
          double p = 0.0;
          for (int q = 0; q < NPART; q++) p += pp[q * n + i];
          p *= beta;
          pv[i] = p;
          vtp = __builtin_fma(vv[i], p, vtp);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vtp += __shfl_xor(vtp, o, 64);
        if (lane == 0) s_K = 0.5 * beta * vtp;
      }
      __syncthreads();
      const double K = s_K;
      // A22 -= v w^T + w v^T, w = p - K v (every entry: the block stays symmetric)
      for (int e = t; e < m * m; e += nt) {
        const int i = e / m, c = e - i * m;
        const double vi = vv[i], vc = vv[c];
        const double wi = pv[i] - K * vi, wc = pv[c] - K * vc;
        A[(size_t)(k + 1 + i) * ld + (k + 1 + c)] -= vi * wc + wi * vc;
      }
    }
    __syncthreads();
  }
  if (t == 0) {
    dd[n - 2] = A[(size_t)(n - 2) * ld + (n - 2)]; ee[n - 2] = A[(size_t)(n - 1) * ld + (n - 2)];
    dd[n - 1] = A[(size_t)(n - 1) * ld + (n - 1)]; ee[n - 1] = 0.0;
  }
  ESTAMP(9);
  // ---- phase 2: Z = Q (row-major), backward accumulation Q <- H_k Q on the rows / columns k + 1 ..
  for (int e = t; e < n * n; e += nt) { const int i = e / n, j = e - i * n; Z[(size_t)i * ld + j] = i == j ? 1.0 : 0.0; }
  __syncthreads();
  for (int k = n - 3; k >= 0; k--) {
    const int m = n - k - 1;
    const double beta = lam[k];
    if (beta == 0.0) continue;                               // (block-uniform: lam is not written in this phase)
    for (int i = t; i < m; i += nt) vv[i] = A[(size_t)(k + 1 + i) * ld + k];
    __syncthreads();
    for (int jb = 0; jb < m; jb += 128) {                     // s_j = sum_i v_i Q(i, j): lanes along j
      const int j = jb + (t & 127), part = t >> 7;
      if (j < m) {
        double acc = 0.0;
        for (int i = part; i < m; i += NPART) acc = __builtin_fma(vv[i], Z[(size_t)(k + 1 + i) * ld + (k + 1 + j)], acc);
        pp[part * n + j] = acc;
      }
    }
    __syncthreads();
    for (int j = t; j < m; j += nt) { double sj = 0.0; for (int q = 0; q < NPART; q++) sj += pp[q * n + j]; pv[j] = beta * sj; }
    __syncthreads();
    for (int e = t; e < m * m; e += nt) {
      const int i = e / m, j = e - i * m;
      Z[(size_t)(k + 1 + i) * ld + (k + 1 + j)] -= vv[i] * pv[j];
    }
    __syncthreads();
  }
  for (int e = t; e < n * n; e += nt) {                       // in-place transpose: row j of Z = column j of Q
    const int i = e / n, j = e - i * n;
    if (i < j) { const double a = Z[(size_t)i * ld + j], b = Z[(size_t)j * ld + i]; Z[(size_t)i * ld + j] = b; Z[(size_t)j * ld + i] = a; }
  }
  if (t == 0) { s_fail = 0; s_rng[0][2] = 0; s_rng[1][2] = 0; }
  __syncthreads();
  ESTAMP(10);
  if constexpr (std::is_same<PD, mlds_double *>::value) {
    if (park && n >= 3) {
      for (int e = t; e < n * n; e += nt) park[e] = Z[(size_t)(e / n) * ld + (e % n)];
      __syncthreads();
      tridiag_dc(A, Z, n, ld, dd, ee, vv, stamp);      // (A: the reflectors are consumed; vv .. : the 12 n doubles of scratch behind d and e)
      __shared__ int s_dcbad;
      if (t == 0) s_dcbad = 0;
      __syncthreads();
      for (int j = t; j < n; j += nt) if (!isfinite(vv[j])) s_dcbad = 1;
      for (int e = t; e < n * n; e += nt) Z[(size_t)(e / n) * ld + (e % n)] = park[e];
      __syncthreads();
      if (!s_dcbad) {
        // eigenvector j of A: sum_i Q_dc(i, j) (column i of the tridiagonalisation's Q = row i of Z)
        for (int e = t; e < n * n; e += nt) {
          const int j = e / n, c = e - j * n;
          double acc = 0.0;
          for (int i = 0; i < n; i++) acc = __builtin_fma(A[(size_t)i * ld + j], Z[(size_t)i * ld + c], acc);
          park[e] = acc;
        }
        for (int j = t; j < n; j += nt) lam[j] = vv[j];
        __syncthreads();
        ESTAMP(11);
        if (stamp && t == 0) stamp[12] = 0.0;
        return 2;
      }
      // (not finite: the QL below, on the untouched d / e and the restored Z)
    }
  }
  int nsweep = 0;
  // ---- phase 3. The lists: cs[(b * n + i) * 2 + {0, 1}] would need 4 n doubles; pv / pp are free now: list b lives in (b ? pp : cs)
  int l = 0, iter = 0;                                         // (thread 0's state of the QL iteration)
  // one sweep (or the end) into list b. e[i] couples d[i] and d[i + 1]
  // (wave 0 runs it: the search for the first negligible off-diagonal entry at or after l — up to n dependent LDS round trips on one
  //  lane, more than the sweep itself — is a ballot over the wave's lanes; the recurrence is lane 0's)
  auto scalar_sweep = [&](int b) {
    PD L = b ? pp : cs;
    for (;;) {
      if (l >= n) { if (lane == 0) { s_rng[b][0] = 1; s_rng[b][1] = 0; s_rng[b][2] = 1; } return; }
      int m = n - 1;
      for (int j0 = l; j0 < n - 1; j0 += 64) {
        const int j = j0 + lane;
        const bool negl = j < n - 1 && fabs(ee[j]) <= 2.220446049250313e-16 * (fabs(dd[j]) + fabs(dd[min(j + 1, n - 1)]));
        const unsigned long long bal = __ballot(negl);
        if (bal) { m = j0 + __builtin_ctzll(bal); break; }
      }
      if (m == l) { l++; iter = 0; continue; }
      if (iter++ >= 60) { if (lane == 0) s_fail = 1; l++; iter = 0; continue; }
      if (lane != 0) return;                                   // (the other lanes wait at the block barrier; l, iter stay in step: the sweep below changes neither)
      // The sweep itself is lane 0's. (Measured and dropped: the whole wave running it on wave-uniform values with d / e cached in lane
      // registers and fetched by v_readlane — 11.8 instead of 10.7 us per sweep: a lone wave issues one instruction every ~5.4 cycles
      // whatever it is, and the loop is bound by its ~55 instructions per rotation, not by the two LDS loads.)
      double g = (dd[l + 1] - dd[l]) / (2.0 * ee[l]);
      double r = sqrt(g * g + 1.0);
      g = dd[m] - dd[l] + ee[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
      double sn = 1.0, c = 1.0, p = 0.0;
      int i = m - 1;
      double e_i = ee[i], d_i = dd[i], d_ip1 = dd[m];          // (d[i + 1] of a step is the d[i] the step before it read: no load on the chain)
      int lo = l;
      for (; i >= l; i--) {
        const double e_nx = i > l ? ee[i - 1] : 0.0, d_nx = i > l ? dd[i - 1] : 0.0;      // (the next step's entries: in flight during this one)
        const double f = sn * e_i, bq = c * e_i;
        const double h = f * f + g * g;
        if (h == 0.0) { ee[i + 1] = 0.0; dd[i + 1] = d_ip1 - p; ee[m] = 0.0; lo = i + 1; break; }             // (recover from underflow: tqli)
        const double rinv = rsqrt_refined_m(h);
        r = h * rinv;
        ee[i + 1] = r;
        sn = f * rinv; c = g * rinv;
        g = d_ip1 - p;
        r = (d_i - g) * sn + 2.0 * c * bq;
        p = sn * r;
        dd[i + 1] = g + p;
        g = c * r - bq;
        L[2 * i] = c; L[2 * i + 1] = sn;
        d_ip1 = d_i; e_i = e_nx; d_i = d_nx;
      }
      if (lo == l) { dd[l] -= p; ee[l] = g; ee[m] = 0.0; }
      s_rng[b][0] = lo; s_rng[b][1] = m - 1; s_rng[b][2] = 0;
      return;
    }
  };
  if (t < 64) scalar_sweep(0);
  for (int b = 0;; b ^= 1) {
    __syncthreads();                                           // list b is complete; the rows have taken list b ^ 1
    const int lo = s_rng[b][0], hi = s_rng[b][1], fin = s_rng[b][2];
    if (fin) break;
    nsweep++;
    if (t < 64) scalar_sweep(b ^ 1);
    else if (!GFBE_EIG_NOAPPLY && t >= 64 && t - 64 < n) {
      // component k of the eigenvectors: rotations hi .. lo on the rows (i, i + 1) of Z, the rotated row i carried in a register
      const int k = t - 64;
      PD L = b ? pp : cs;
      double z1 = Z[(size_t)(hi + 1) * ld + k];
      if (hi >= lo) {
        double c = L[2 * hi], sn = L[2 * hi + 1], z0 = Z[(size_t)hi * ld + k];
        for (int i = hi; i >= lo; i--) {
          const int ip = i > lo ? i - 1 : i;                     // (the next rotation's operands: in flight during this one)
          const double c_n = L[2 * ip], sn_n = L[2 * ip + 1], z0_n = Z[(size_t)ip * ld + k];
          Z[(size_t)(i + 1) * ld + k] = sn * z0 + c * z1;
          z1 = c * z0 - sn * z1;
          c = c_n; sn = sn_n; z0 = z0_n;
        }
      }
      Z[(size_t)lo * ld + k] = z1;
    }
  }
  for (int j = t; j < n; j += nt) lam[j] = dd[j];
  __syncthreads();
  ESTAMP(11);
  if (stamp && t == 0) stamp[12] = (double)nsweep;
#undef ESTAMP
  return s_fail;
}

// Same one-sided Jacobi for n <= 16, run by ONE wave (8 column pairs x 8 lanes, wave-level
// synchronisation only): used for the dense pseudo-inverse of the dropped pose / speed-bias block.
__device__ void jacobi_eig_wave16(double *G, double *V, int n, double *lam, int lane) {
  const int grp = lane >> 3, gl = lane & 7;
  for (int e = lane; e < 16 * 16; e += 64) V[e] = ((e >> 4) == (e & 15)) ? 1.0 : 0.0;
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
  double scale = 0.0;
  for (int j = 0; j < n; j++) { double a = 0.0; for (int i = 0; i < n; i++) a += G[j * 16 + i] * G[j * 16 + i]; scale = fmax(scale, a); }
  const double tiny2 = fmax(scale * 1e-30, 1e-24);
  const int np = (n + 1) & ~1;
  for (int sweep = 0; sweep < 40; sweep++) {
    int rotated = 0;
    for (int r = 0; r < np - 1; r++) {
      const int k = grp;
      int p, q;
      if (k == 0) { p = np - 1; q = r; }
      else { p = (r + k) % (np - 1); q = (r - k + (np - 1)) % (np - 1); }
      const bool live = (k < np / 2) && p < n && q < n;
      if (p > q) { const int tt = p; p = q; q = tt; }
      double a = 0.0, b = 0.0, c = 0.0;
      if (live) for (int i = gl; i < n; i += 8) { const double x = G[p * 16 + i], y = G[q * 16 + i]; a += x * x; b += y * y; c += x * y; }
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) { a += __shfl_xor(a, o, 8); b += __shfl_xor(b, o, 8); c += __shfl_xor(c, o, 8); }
      const bool rot = live && !(fabs(c) <= 1e-14 * sqrt(a * b) || c == 0.0 || (a <= tiny2 && b <= tiny2));
      if (rot) {
        const double zeta = (b - a) / (2.0 * c);
        const double tn = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / sqrt(1.0 + tn * tn), sn = cs * tn;
        for (int i = gl; i < n; i += 8) {
          const double x = G[p * 16 + i], y = G[q * 16 + i];
          G[p * 16 + i] = cs * x - sn * y; G[q * 16 + i] = sn * x + cs * y;
          const double u = V[p * 16 + i], v = V[q * 16 + i];
          V[p * 16 + i] = cs * u - sn * v; V[q * 16 + i] = sn * u + cs * v;
        }
      }
      rotated |= __any(rot) ? 1 : 0;
      __threadfence_block();
      __builtin_amdgcn_wave_barrier();
    }
    if (!rotated) break;
  }
  if (lane < n) { double sum = 0.0; for (int i = 0; i < n; i++) sum += V[lane * 16 + i] * G[lane * 16 + i]; lam[lane] = sum; }
}

// ---- diagonally pivoted LDL^T of A' with pivots > eps:  A' ~= P L D+ L^T P^T,
//      J0 = D+^(1/2) L^T P^T,  r0 = D+^(-1/2) L^-1 P^T b'   (forward substitution folded in)
#define LDLT_THREADS 512
// Largest A' the register-resident LDL^T holds: an R = 8 tile per thread of a 22 x 22 thread grid (22^2 <= LDLT_THREADS), the
// published column in colbuf[2][192], the diagonal in three registers per lane of a wave. A new prior beyond it (reachable through
// the ABI only — the priors the reference builds stay near 100 dims, a GNSS window near 150) takes the eigen-decomposition path
// whatever gfbe_options.marg_sqrt says.
#define LDLT_MAX_N 176
static_assert(LDLT_MAX_N == 8 * 22 && 22 * 22 <= LDLT_THREADS && LDLT_MAX_N <= 192 && LDLT_MAX_N <= 3 * 64, "k_marg_ldlt<8>: thread grid / colbuf / diagonal registers");
// maximum / minimum of a 32-bit value over the wave, as a scalar: four DPP rotations inside the 16-lane rows, the rows' results
// through v_readlane
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define ROR_MAX(N) v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + (N), 0xf, 0xf, false))
  ROR_MAX(8); ROR_MAX(4); ROR_MAX(2); ROR_MAX(1);
#undef ROR_MAX
  const unsigned r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16);
  const unsigned r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
  return max(max(r0, r1), max(r2, r3));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#define ROR_MIN(N) v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x120 + (N), 0xf, 0xf, false))
  ROR_MIN(8); ROR_MIN(4); ROR_MIN(2); ROR_MIN(1);
#undef ROR_MIN
  const unsigned r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16);
  const unsigned r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
  return min(min(r0, r1), min(r2, r3));
}
#ifndef GFBE_LDLT_STAMP
#define GFBE_LDLT_STAMP 0      // diagnostics build: time stamps inside the first steps of the pivot loop (tools/diag_scripts/marg_stamps.py)
#endif
// loadA(i, j): entry (i, j) of A'; loadB(i): entry i of b' — from the compact arrays k_marg left (k_marg_ldlt), or formed on the fly
// from the Schur operands in LDS (the LDL^T at the tail of k_marg itself).
// NT: the threads that run this function (a multiple of 64, a power of two of waves)
template <int R, int NT = LDLT_THREADS, class LoadA, class LoadB>
__device__ __forceinline__ int ldlt_registers(LoadA loadA, LoadB loadB, double *__restrict__ J0,
                                              double *__restrict__ r0, const int n, const double eps, double *lstamp) {
  const int t = threadIdx.x;
#if GFBE_LDLT_STAMP
#define LSTAMP(i) do { if (t == 0 && k >= 8 && k < 12) lstamp[(k - 8) * 8 + (i)] = (double)wall_clock64(); } while (0)
#else
#define LSTAMP(i) do { } while (0)
#endif
    // The matrix lives in REGISTERS: thread (ti, tj) of a G x G grid (G = ceil(n / R)) owns the R x R tile
    // A'(R ti .. R ti + R - 1, R tj .. R tj + R - 1). Step k: every wave finds the largest remaining diagonal entry p_k
    // from its own register copy of the diagonal (no cross-wave reduction), the owners of row p_k publish it
    // through a double-buffered LDS vector, one block barrier, and every thread subtracts the rank-1 term
    // from its tile. Nothing is moved: eliminated rows / columns simply stay behind (masked on output), and
    // row k of J0 = sqrt(d_k) L(:,k)^T is streamed to HBM from the published vector as it is produced.
    __shared__ double colbuf[2][192];
    const int G = (n + R - 1) / R;
    const int ti = t / G, tj = t - ti * G;
    const bool owner_active = t < G * G;
    const int lane = t & 63, wv = t >> 6;
    double a[R][R];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
      for (int c = 0; c < R; c++) {
        const int i = ti * R + r, j = tj * R + c;
        a[r][c] = (owner_active && i < n && j < n) ? loadA(i, j) : 0.0;
      }
    double dg[3], bzr[3];
    bool alive[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int i = lane + 64 * q;
      alive[q] = i < n;
      dg[q] = alive[q] ? loadA(i, i) : 0.0;
      bzr[q] = alive[q] ? loadB(i) : 0.0;
    }
    int rank = n;
    for (int k = 0; k < n; k++) {
      LSTAMP(0);
      // arg-max of the remaining diagonal over the wave (largest value, smallest index among equals — every wave finds the same one
      // from its own copy). A positive double orders like its bit pattern: the high words' maximum, then the low words' among the
      // lanes that hold it, then the smallest index among those that hold both — three 32-bit reductions (DPP rotations inside the
      // 16-lane rows, the four row results through v_readlane into scalar max / min) instead of one on (double, index) pairs whose
      // compare-and-select chains were 0.6 us of a 1.3 us step. Entries that are not positive cannot be pivots: key 0.
      unsigned khi = 0, klo = 0;
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const unsigned h = (unsigned)__double2hiint(dg[q]), l = (unsigned)__double2loint(dg[q]);
        const bool cand = alive[q] && !(h >> 31);
        const bool gt = cand && (h > khi || (h == khi && l > klo));
        khi = gt ? h : khi; klo = gt ? l : klo;
      }
      const unsigned mh = wave_max_u32(khi);
      const unsigned ml = wave_max_u32(khi == mh ? klo : 0u);
      unsigned ci = 0xffffffffu;
#pragma unroll
      for (int q = 2; q >= 0; q--)
        if (alive[q] && (unsigned)__double2hiint(dg[q]) == mh && (unsigned)__double2loint(dg[q]) == ml) ci = (unsigned)(lane + 64 * q);
      const int bi = (int)wave_min_u32(ci);
      const double best = __hiloint2double((int)mh, (int)ml);
      LSTAMP(1);
      if (!(best > eps) || (mh | ml) == 0u) { rank = k; break; }     // identical in every wave
      const int pv = bi, pq = pv >> 6, pl = pv & 63;
      const double piv = best, inv = 1.0 / piv;
      const double zsel = pq == 0 ? bzr[0] : (pq == 1 ? bzr[1] : bzr[2]);
      const double zk = __shfl(zsel, pl, 64);
      double *cb = colbuf[k & 1];
      const int pr = pv / R, pc = pv - pr * R;
      if (owner_active && ti == pr) {                         // symmetric: row p_k == column p_k
#pragma unroll
        for (int c = 0; c < R; c++) {
          double v = a[0][c];
#pragma unroll
          for (int r = 1; r < R; r++) v = pc == r ? a[r][c] : v;
          cb[tj * R + c] = v;
        }
      }
      // (LDS traffic only: __syncthreads() also drains the vector-memory counter — the J0 row the previous step streamed out)
      LSTAMP(2);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      LSTAMP(3);
      double cx[3];
#pragma unroll
      for (int q = 0; q < 3; q++) cx[q] = cb[lane + 64 * q];
      if (wv == (k & (NT / 64 - 1))) {                                             // row k of J0, r0[k]
        const double rs = 1.0 / sqrt(piv);
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const int i = lane + 64 * q;
          if (i < n) J0[(size_t)k * n + i] = i == pv ? sqrt(piv) : (alive[q] ? cx[q] * rs : 0.0);
        }
        if (lane == 0) r0[k] = zk * rs;
      }
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const double li = cx[q] * inv;
        dg[q] -= li * cx[q];
        bzr[q] -= li * zk;
        if (lane + 64 * q == pv) alive[q] = false;
      }
      LSTAMP(4);
      if (owner_active) {
        double ci[R], cj[R];
#pragma unroll
        for (int r = 0; r < R; r++) { ci[r] = cb[ti * R + r] * inv; cj[r] = cb[tj * R + r]; }
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
          for (int c = 0; c < R; c++) a[r][c] -= ci[r] * cj[c];
      }
      LSTAMP(5);
    }
#undef LSTAMP
    for (int e = t + rank * n; e < n * n; e += NT) J0[e] = 0.0;      // (NT threads run this function, whatever the workgroup's size)
    for (int k = t + rank; k < n; k += NT) r0[k] = 0.0;
    return rank;
}

#ifndef MARG_THREADS
#define MARG_THREADS 1024
#endif
#ifndef GFBE_MARG_TP_THREADS
#define GFBE_MARG_TP_THREADS 512      // k_marg's workgroup in throughput batches
#endif
#define MARG_SQRT_PENDING (-1000000)
#define MARG_LDS_N 90   // A' up to this size is eigen-decomposed entirely inside LDS (2 n^2 doubles + the 14 n of tridiag_ql_eig's scratch)
#define MARG_LDS_DOUBLES (2 * MARG_LDS_N * MARG_LDS_N + 14 * MARG_LDS_N)

__global__ __launch_bounds__(MARG_THREADS) void k_marg(BatchDev d, int flag) {
  const int w = blockIdx.x;
  const WinDesc &ds = d.desc[w];
  const int t = threadIdx.x;
  __shared__ MargShared sh;
  __shared__ double Pm[16 * 16], Pv[16 * 16], Pw[16 * 16], Pl[16], Pinv16[16 * 16], bm[32];
  extern __shared__ __attribute__((aligned(16))) double marg_lds[];
  __shared__ int cflag;
  int *meta = d.mmeta + (size_t)w * (4 + 3 * GFBE_MAX_PRIOR_BLOCKS);
  const double *Xo = d.xout + (size_t)w * NA;
  double *A = d.mA + (size_t)w * ND * ND;      // working matrices (global / L2)
  double *bv = d.mb + (size_t)w * ND;
  double *J0 = d.mJ0 + (size_t)w * ND * ND;
  double *r0 = d.mr0 + (size_t)w * ND;
  const bool old = (flag == GFBE_MARGIN_OLD);
  // estimator.cpp:3391 `if (frame_count < WINDOW_SIZE) return;` — a window that is still filling up is not marginalised and its
  // caller's prior stays as it is (marg_ran = 0: gfbe_batch_download leaves prior_out untouched)
  if (ds.frame_count < GFBE_WINDOW_SIZE) return;
  if (t == 0) { d.ctl[w].marg_ran = 1; }
  double *stamp = d.timing + 24;
#define MSTAMP(i) do { if (w == 0 && t == 0) stamp[i] = (double)wall_clock64(); } while (0)
  MSTAMP(0);

  // which blocks do the factors of the marginalisation set touch? The look-ups (descriptor fields and cost slots in global memory:
  // a microsecond of latency each) by different threads side by side — on one lane they were 20 us of a single window's solve; every
  // flag is written with the same value by whoever finds it
  __shared__ int s_pmap[ND];       // prior column of a tangent dim (-1: not in the prior): read per entry of the assembly below
  for (int a = t; a < ND; a += blockDim.x) s_pmap[a] = ds.prior_map[a];
  if (t < GFBE_BLK_COUNT) sh.touched[t] = 0;
  if (t == 0) { sh.use_imu = sh.use_wheel = sh.use_plane = sh.use_gnss = 0; sh.passthrough = 0; }
  __syncthreads();
  if (t < ds.prior_nblk) sh.touched[ds.prior_blk_id[t]] = 1;
  if (old) {
    if (t >= 64 && t < 64 + ds.n_imu) {            // (one factor at most starts at frame 0)
      const int q = t - 64;
      if (d.imu_part[((size_t)w * MAX_IMU + q) * IMU_PART + IMU_PART - 2] >= 0.0 && ds.imu_frame[q] == 0) {
        sh.use_imu = 1 + q;
        sh.touched[0] = sh.touched[GFBE_BLK_SB0] = sh.touched[1] = sh.touched[GFBE_BLK_SB0 + 1] = 1;
      }
    } else if (t >= 128 && t < 128 + ds.n_wheel) {
      const int q = t - 128;
      if (d.wheel_part[((size_t)w * MAX_WHEEL + q) * WHEEL_PART + WHEEL_PART - 2] >= 0.0 && ds.wheel_frame[q] == 0) {
        sh.use_wheel = 1 + q;
        sh.touched[0] = sh.touched[1] = sh.touched[GFBE_BLK_EX_WHEEL] = sh.touched[GFBE_BLK_SX] = sh.touched[GFBE_BLK_SY] = sh.touched[GFBE_BLK_SW] = sh.touched[GFBE_BLK_TD_WHEEL] = 1;
      }
    } else if (t >= 192 && t < 192 + NF - 1) {
      const int j = t - 191;
      if (ds.pair_begin[j + 1] > ds.pair_begin[j]) sh.touched[0] = sh.touched[j] = sh.touched[GFBE_BLK_EX_CAM] = sh.touched[GFBE_BLK_TD] = 1;
    } else if (t == 224) {
      if (ds.n_plane > 0 && d.plane_part[(size_t)w * MAX_PLANE * PLANE_PART + PLANE_PART - 2] >= 0.0) {   // estimator.cpp:3441-3448
        sh.use_plane = 1;
        sh.touched[0] = sh.touched[GFBE_BLK_EX_WHEEL] = sh.touched[GFBE_BLK_PLANE_R] = sh.touched[GFBE_BLK_PLANE_Z] = 1;
      }
    } else if (t == 225) {
      if (ds.gnss_ready) {   // estimator.cpp:3459-3496: the GNSS factors of frame 0 (k_gnss mode 2), whether the window was slow or not
        sh.use_gnss = 1;
        if (ds.gnss_frame_begin[1] > 0)
          sh.touched[0] = sh.touched[GFBE_BLK_SB0] = sh.touched[1] = sh.touched[GFBE_BLK_SB0 + 1] = sh.touched[GFBE_BLK_YAW_ENU] = sh.touched[GFBE_BLK_ANC_ECEF] = 1;
        for (int k = 0; k < 4; k++) sh.touched[GFBE_BLK_RCV_DT0 + k] = sh.touched[GFBE_BLK_RCV_DT0 + 4 + k] = 1;
        sh.touched[GFBE_BLK_RCV_DDT0] = sh.touched[GFBE_BLK_RCV_DDT0 + 1] = 1;
      }
    }
  }
  __syncthreads();
  // the dropped dims by one lane, the kept blocks and their tangent dims by a wave next to it (two blocks per lane, positions by
  // prefix sums over the lanes: 88 blocks one after the other on one lane were ~4 us of a single window's marginalisation)
  if (t == 64) {
    int m = 0;
    if (old) {
      if (sh.touched[0]) for (int k = 0; k < 6; k++) sh.drop_dim[m++] = k;
      if (sh.touched[GFBE_BLK_SB0]) for (int k = 0; k < 9; k++) sh.drop_dim[m++] = T_SB(0) + k;
      if (sh.use_gnss) {   // drop sets {0, 1, 4, 5}, {0, 2}, {0}: rcv_dt[0][k], rcv_ddt[0]
        for (int k = 0; k < 4; k++) sh.drop_dim[m++] = T_DT(0, k);
        sh.drop_dim[m++] = T_DDT(0);
      }
    } else {
      const int pb = GFBE_BLK_POSE0 + GFBE_WINDOW_SIZE - 1;
      if (ds.prior_n > 0 && sh.touched[pb]) for (int k = 0; k < 6; k++) sh.drop_dim[m++] = T_POSE(GFBE_WINDOW_SIZE - 1) + k;
      else sh.passthrough = 1;                      // estimator.cpp:3600-3601: prior does not touch Pose[9]
    }
    sh.m = m;
  }
  if (t < 64) {
    static_assert(GFBE_BLK_COUNT <= 128, "two blocks per lane");
    int base_k = 0, base_n = 0;
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int q = t + 64 * u;
      bool kept = q < GFBE_BLK_COUNT && sh.touched[min(q, GFBE_BLK_COUNT - 1)];
      if (kept) {
        const bool dropped = old ? (q == 0 || q == GFBE_BLK_SB0 || (sh.use_gnss && ((q >= GFBE_BLK_RCV_DT0 && q < GFBE_BLK_RCV_DT0 + 4) || q == GFBE_BLK_RCV_DDT0)))
                                 : (q == GFBE_BLK_POSE0 + GFBE_WINDOW_SIZE - 1);
        kept = !dropped;
      }
      const int ls = kept ? blk_lsize(q) : 0;
      int ck = kept ? 1 : 0, cn = ls;          // inclusive prefix sums over the lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int ok = __shfl_up(ck, o, 64), on = __shfl_up(cn, o, 64);
        if (t >= o) { ck += ok; cn += on; }
      }
      if (kept) {
        sh.keep_id[base_k + ck - 1] = q;
        for (int k = 0; k < ls; k++) sh.keep_dim[base_n + cn - ls + k] = blk_tan(q) + k;
      }
      base_k += __shfl(ck, 63, 64); base_n += __shfl(cn, 63, 64);
    }
    if (t == 0) { sh.n_keep = base_k; sh.n = base_n; }
  }
  __syncthreads();
  if (sh.passthrough) {
    // nothing to marginalise: the prior is handed back unchanged
    const int n = ds.prior_n;
    for (int e = t; e < n * n; e += blockDim.x) J0[e] = d.prior_J0[(size_t)w * ND * ND + e];
    for (int e = t; e < n; e += blockDim.x) r0[e] = d.prior_r0[(size_t)w * ND + e];
    for (int e = t; e < PRIOR_X0; e += blockDim.x) d.mx0[(size_t)w * PRIOR_X0 + e] = d.prior_x0[(size_t)w * PRIOR_X0 + e];
    if (t == 0) {
      meta[0] = n > 0 ? 1 : 0; meta[1] = n; meta[2] = ds.prior_nblk; meta[3] = 1;
      for (int q = 0; q < ds.prior_nblk; q++) { meta[4 + q] = ds.prior_blk_id[q]; meta[4 + GFBE_MAX_PRIOR_BLOCKS + q] = ds.prior_blk_size[q]; meta[4 + 2 * GFBE_MAX_PRIOR_BLOCKS + q] = ds.prior_blk_idx[q]; }
      d.ctl[w].t_marg = (long long)wall_clock64();
    }
    return;
  }
  if (sh.m == 0) {   // marginalization_factor.cpp:205-210: "unstable tracking", valid = false
    if (t == 0) { meta[0] = 0; meta[1] = 0; meta[2] = 0; meta[3] = 0; }
    return;
  }
  const int n = sh.n, m = sh.m;
  MSTAMP(1);
  // ---- full A (ND x ND, landmark block already eliminated) and b over all tangent dims
  const double *pp = d.pair_part + (size_t)w * NF * VP_STRIDE;   // pairs (0, j): index j
  const double *sp = d.schur_part + (size_t)w * d.schur_groups * SCHUR_STRIDE;   // start frame 0 partial (slot 0: 15 dense 16x16 tiles)
  const double *ipart = sh.use_imu ? d.imu_part + ((size_t)w * MAX_IMU + sh.use_imu - 1) * IMU_PART : nullptr;
  const double *wpart = sh.use_wheel ? d.wheel_part + ((size_t)w * MAX_WHEEL + sh.use_wheel - 1) * WHEEL_PART : nullptr;
  const double *ppart = sh.use_plane ? d.plane_part + (size_t)w * MAX_PLANE * PLANE_PART : nullptr;
  const double *gpart = sh.use_gnss ? d.gnss_marg + (size_t)w * GN_MPART : nullptr;
  // only the dims of the marginalisation (dropped + kept, nn <= 101 of 182) are ever read back: pairs (ia >= ib) of that list
  const int nn = n + m, ntri = nn * (nn + 1) / 2;
  // entry e of the lower triangle over the marginalisation's dims: the sum of its terms in the fixed order pairs (0, 1..10), Schur,
  // IMU, wheel, plane, GNSS, prior. Two entries per thread and pass: the loads of both are in flight together (five dependent passes
  // of ~3 us each were the longest phase of the kernel for one window).
  auto entry_sum = [&](const int e, int &a, int &b) -> double {
    int ia = (int)((sqrtf(8.0f * e + 1.0f) - 1.0f) * 0.5f);
    while ((ia + 1) * (ia + 2) / 2 <= e) ia++;
    while (ia * (ia + 1) / 2 > e) ia--;
    const int ib = e - ia * (ia + 1) / 2;
    const int da = ia < m ? sh.drop_dim[ia] : sh.keep_dim[ia - m], db = ib < m ? sh.drop_dim[ib] : sh.keep_dim[ib - m];
    a = max(da, db); b = min(da, db);
    double s = 0.0;
    if (old && a < NV) {
      // pairs (0, j) whose factors reach both dims: every j when neither is a pose dim of a frame > 0 (pose 0, extrinsic, td: the
      // column inside the pair's block does not depend on j), the one j of that frame otherwise (none for two different frames)
      const int fa = a < 66 ? a / 6 : 0, fb = b < 66 ? b / 6 : 0;
      if (fa == 0 && fb == 0) {
        const int off = m_vp_off(m_vis_loc(a, 1), m_vis_loc(b, 1));
        for (int j = 1; j < NF; j++) s += pp[(size_t)j * VP_STRIDE + off];
      } else if (fa == 0 || fb == 0 || fa == fb) {
        const int j = max(fa, fb);
        s += pp[(size_t)j * VP_STRIDE + m_vp_off(m_vis_loc(a, j), m_vis_loc(b, j))];
      }
      s -= sp[m_schur_off(b, a)];                         // b <= a
    }
    if (ipart) { const int la = m_imu_loc(a), lb = m_imu_loc(b); if (la >= 0 && lb >= 0) s += ipart[la * 30 + lb]; }
    if (wpart) { const int la = m_wheel_loc(a), lb = m_wheel_loc(b); if (la >= 0 && lb >= 0) s += wpart[la * 22 + lb]; }
    if (ppart) { const int la = m_plane_loc(a), lb = m_plane_loc(b); if (la >= 0 && lb >= 0) s += ppart[la * 16 + lb]; }
    if (gpart) { const int la = m_gnss_loc(a), lb = m_gnss_loc(b); if (la >= 0 && lb >= 0) s += gpart[la * GN_M + lb]; }
    if (ds.prior_n > 0) { const int pa = s_pmap[a], pb = s_pmap[b]; if (pa >= 0 && pb >= 0) s += d.prior_H[(size_t)w * ND * ND + (size_t)pa * ds.prior_n + pb]; }
    return s;
  };
  for (int e = t; e < ntri; e += 2 * blockDim.x) {
    int a0, b0, a1 = 0, b1 = 0;
    const int e1 = e + blockDim.x;
    const double s0 = entry_sum(e, a0, b0);
    const double s1 = e1 < ntri ? entry_sum(e1, a1, b1) : 0.0;
    A[(size_t)a0 * ND + b0] = s0; A[(size_t)b0 * ND + a0] = s0;
    if (e1 < ntri) { A[(size_t)a1 * ND + b1] = s1; A[(size_t)b1 * ND + a1] = s1; }
  }
  for (int a = t; a < ND; a += blockDim.x) {
    double s = 0.0;
    if (old && a < NV) {
      for (int j = 1; j < NF; j++) {
        const int la = m_vis_loc(a, j);
        if (la >= 0) s += pp[(size_t)j * VP_STRIDE + m_vp_off(la, 19)];
      }
      s -= sp[m_schur_off(a, NV)];                        // column 73 = sum_l w_l h_l g_l
    }
    if (ipart) { const int la = m_imu_loc(a); if (la >= 0) s += ipart[900 + la]; }
    if (wpart) { const int la = m_wheel_loc(a); if (la >= 0) s += wpart[484 + la]; }
    if (ppart) { const int la = m_plane_loc(a); if (la >= 0) s += ppart[256 + la]; }
    if (gpart) { const int la = m_gnss_loc(a); if (la >= 0) s += gpart[GN_M * GN_M + la]; }
    if (ds.prior_n > 0 && s_pmap[a] >= 0) s += d.prior_g[(size_t)w * (ND + 2) + s_pmap[a]];
    bv[a] = s;
  }
  __syncthreads();
  MSTAMP(2);
  // ---- dense elimination of the m dropped dims: Amm = V diag(l) V^T, pinv with eps
  // (m <= 16: the one-wave path below; the 20 dims of a GNSS window: the workgroup-wide Jacobi on a 32-stride block of the dynamic
  //  LDS, which is free until A' is eigen-decomposed — the reference's construction without a shortcut)
  const int ms = m > 16 ? 32 : 16;
  double *Pinv = m > 16 ? marg_lds + 2 * 32 * 32 : Pinv16;
  if (t < m) bm[t] = bv[sh.drop_dim[t]];
  if (m > 16) {
    double *G32 = marg_lds, *V32 = marg_lds + 32 * 32, *lam32 = marg_lds + 3 * 32 * 32;
    for (int e = t; e < m * m; e += blockDim.x) {
      const int i = e / m, j = e % m;
      G32[j * 32 + i] = 0.5 * (A[(size_t)sh.drop_dim[i] * ND + sh.drop_dim[j]] + A[(size_t)sh.drop_dim[j] * ND + sh.drop_dim[i]]);
    }
    __syncthreads();
    // (round 6) the fast path of the 16-dim block below for up to 32 dims — the 20 dropped dims of a GNSS window: Cholesky factor by one
    // wave (lane = row, two 32-lane halves doing the same), its inverse column by column, Pinv = L^-T L^-1, certified by
    // lambda_min >= 1 / |Pinv|_F > 4 eps; the workgroup-wide Jacobi (~0.1 ms of a single window's marginalisation) only runs when the
    // block is rank-deficient or holds tiny eigenvalues, with the reference's thresholding as before.
    double *W32 = marg_lds + 3 * 32 * 32;      // L^-1 (lam32's block: the Jacobi, if it runs, runs afterwards)
    for (int e = t; e < 32 * 32; e += blockDim.x) { V32[e] = G32[e]; W32[e] = 0.0; }
    __syncthreads();
    if (t < 64) {
      // right-looking Cholesky in LDS (loops with run-time bounds: no register arrays in a kernel held to 128 registers), one wave
      bool ok = true;
      for (int k = 0; k < m; k++) {
        const double dkk = V32[k * 32 + k];
        if (!(dkk > 0.0) || !isfinite(dkk)) ok = false;
        const double lkk = sqrt(dkk);
        __builtin_amdgcn_wave_barrier();
        if (t == k) V32[k * 32 + k] = lkk;
        else if (t > k && t < m) V32[t * 32 + k] = V32[t * 32 + k] / lkk;
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        const int nr = m - k - 1;
        for (int idx = t; idx < nr * nr; idx += 64) {
          const int i = k + 1 + idx / nr, j = k + 1 + idx % nr;
          if (j <= i) V32[i * 32 + j] -= V32[i * 32 + k] * V32[j * 32 + k];
        }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
      }
      if (t < m) {                                                    // lane j: column j of L^-1 (its own earlier entries: same lane)
        const int j = t;
        for (int i = j; i < m; i++) {
          double acc = (i == j) ? 1.0 : 0.0;
          for (int k = j; k < i; k++) acc -= V32[i * 32 + k] * W32[k * 32 + j];
          W32[i * 32 + j] = acc / V32[i * 32 + i];
        }
      }
      if (t == 0) cflag = ok ? 1 : 0;
    }
    __syncthreads();
    double fro32 = 0.0;
    for (int e = t; e < 32 * 32; e += blockDim.x) {
      const int i = e >> 5, j = e & 31;
      double s = 0.0;
      for (int k = 0; k < 32; k++) s += W32[k * 32 + i] * W32[k * 32 + j];
      s = (i < m && j < m) ? s : 0.0;
      Pinv[e] = s;
      fro32 += s * s;
    }
    for (int o = 32; o > 0; o >>= 1) fro32 += __shfl_down(fro32, o, 64);
    if ((t & 63) == 0) Pl[t >> 6] = fro32;
    __syncthreads();
    double fsum = 0.0;
    for (int q = 0; q < (int)(blockDim.x >> 6); q++) fsum += Pl[q];
    const bool fast32 = cflag == 1 && isfinite(fsum) && 1.0 / sqrt(fsum) > 4.0 * d.opt.marg_eps;
    __syncthreads();
    if (!fast32) {
    jacobi_eig(G32, V32, m, 32, lam32, &cflag, nullptr);
    for (int e = t; e < m * m; e += blockDim.x) {
      const int i = e / m, j = e % m;
      double s = 0.0;
      for (int k = 0; k < m; k++) if (lam32[k] > d.opt.marg_eps) s += V32[k * 32 + i] * V32[k * 32 + j] / lam32[k];
      Pinv[i * 32 + j] = s;
    }
    __syncthreads();
    }
  } else {
  for (int e = t; e < m * m; e += blockDim.x) {
    const int i = e / m, j = e % m;
    Pm[j * 16 + i] = 0.5 * (A[(size_t)sh.drop_dim[i] * ND + sh.drop_dim[j]] + A[(size_t)sh.drop_dim[j] * ND + sh.drop_dim[i]]);
  }
  __syncthreads();
  // Fast path: when every eigenvalue of Amm is safely above eps the thresholded pseudo-inverse IS the inverse. One wave
  // takes the Cholesky factor L (lane = row), the lanes invert it column by column, Pinv = L^-T L^-1, and
  // lambda_min >= 1 / |Pinv|_F > 4 eps certifies it. Otherwise (rank-deficient / tiny eigenvalues) the eigen-decomposition
  // with the reference's thresholding runs as before.
  if (t < 64) {
    const int li = t & 15;
    double row[16];
#pragma unroll
    for (int q = 0; q < 16; q++) row[q] = (li < m && q < m) ? Pm[li * 16 + q] : (li == q ? 1.0 : 0.0);
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const double dkk = __shfl(row[k], k, 16);
      if (!(dkk > 0.0) || !isfinite(dkk)) ok = false;
      const double lkk = sqrt(dkk), lik = row[k] / lkk;
      row[k] = (li == k) ? lkk : lik;
#pragma unroll
      for (int j = k + 1; j < 16; j++) { const double ljk = __shfl(lik, j, 16); row[j] -= lik * ljk; }
    }
    if (t < 16) {
#pragma unroll
      for (int q = 0; q < 16; q++) Pv[li * 16 + q] = row[q];      // L (lower part meaningful)
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    if (t < 16) {                                                   // lane j: column j of L^-1 into Pw
      const int j = li;
      double x[16];
#pragma unroll
      for (int i = 0; i < 16; i++) {
        double acc = (i == j) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; k++) if (k >= j) acc -= Pv[i * 16 + k] * x[k];
        x[i] = i < j ? 0.0 : acc / Pv[i * 16 + i];
        Pw[i * 16 + j] = x[i];
      }
    }
    if (t == 0) cflag = ok ? 1 : 0;
  }
  __syncthreads();
  double fro = 0.0;
  for (int e = t; e < 256; e += blockDim.x) {
    const int i = e >> 4, j = e & 15;
    double s = 0.0;
    for (int k = 0; k < 16; k++) s += Pw[k * 16 + i] * Pw[k * 16 + j];
    s = (i < m && j < m) ? s : 0.0;
    Pinv[e] = s;
    fro += s * s;
  }
  for (int o = 32; o > 0; o >>= 1) fro += __shfl_down(fro, o, 64);
  if ((t & 63) == 0 && t < 256) Pl[t >> 6] = fro;
  __syncthreads();
  const bool fast = cflag == 1 && isfinite(Pl[0] + Pl[1] + Pl[2] + Pl[3]) && 1.0 / sqrt(Pl[0] + Pl[1] + Pl[2] + Pl[3]) > 4.0 * d.opt.marg_eps;
  __syncthreads();
  if (!fast) {
    if (t < 64) jacobi_eig_wave16(Pm, Pv, m, Pl, t);
    __syncthreads();
    for (int e = t; e < m * m; e += blockDim.x) {
      const int i = e / m, j = e % m;
      double s = 0.0;
      for (int k = 0; k < m; k++) if (Pl[k] > d.opt.marg_eps) s += Pv[k * 16 + i] * Pv[k * 16 + j] / Pl[k];
      Pinv[i * 16 + j] = s;
    }
    __syncthreads();
  }
  }
  MSTAMP(3);
  // T = A_rm * Pinv (n x m) kept in J0's storage; then A' and b' (compact, n x n) into r0/J0 staging
  // T and the dropped rows of A through LDS when they fit the block that is free until A' is square-rooted (the 86-dim prior:
  // 2.7k doubles): A' = A_rr - T A_mr then reads its 2 m operands per entry from LDS instead of L2 (33 -> ~12 us for one window;
  // the same products in the same order)
  const bool tl = m <= 16 && n <= 128;
  double *T = tl ? marg_lds : J0;   // n x ms
  double *Amr = marg_lds + 128 * 16;   // [m][n] (tl only)
  if (tl) for (int e = t; e < m * n; e += blockDim.x) { const int k = e / n, j = e - k * n; Amr[e] = A[(size_t)sh.drop_dim[k] * ND + sh.keep_dim[j]]; }
  for (int e = t; e < n * m; e += blockDim.x) {
    const int i = e / m, j = e % m;
    double s = 0.0;
    for (int k = 0; k < m; k++) s += A[(size_t)sh.keep_dim[i] * ND + sh.drop_dim[k]] * Pinv[k * ms + j];
    T[i * ms + j] = s;
  }
  __syncthreads();
  __shared__ int s_xo[GFBE_MAX_PRIOR_BLOCKS];
  auto write_blocks = [&](const int sweeps_code) {      // (every thread of the workgroup calls it: one block barrier inside)
  // ---- getParameterBlocks + addr_shift (estimator.cpp:3561-3590, 3644-3687)
  // (the kept blocks' values at the re-anchored state: every thread one double — a single lane's load -> store chain over the
  //  ~130 doubles of 16 blocks was a third of the kernel for one window)
  if (t == 0) {
    meta[0] = 1; meta[1] = n; meta[2] = sh.n_keep; meta[3] = sweeps_code;   // [3]: Jacobi sweeps (diagnostic)
    int idx = 0, xo = 0;
    for (int q = 0; q < sh.n_keep; q++) {
      const int id = sh.keep_id[q];
      int nid = id;
      const bool is_dt = id >= GFBE_BLK_RCV_DT0 && id < GFBE_BLK_RCV_DDT0, is_ddt = id >= GFBE_BLK_RCV_DDT0;
      if (old) { if (id < GFBE_BLK_EX_CAM || is_ddt) nid = id - 1; else if (is_dt) nid = id - 4; }   // slot i -> i - 1: pose, speed-bias, receiver clock
      else if (id == GFBE_BLK_POSE0 + GFBE_WINDOW_SIZE || id == GFBE_BLK_SB0 + GFBE_WINDOW_SIZE || id == GFBE_BLK_RCV_DDT0 + GFBE_WINDOW_SIZE) nid = id - 1;
      else if (is_dt && id >= GFBE_BLK_RCV_DT0 + 4 * GFBE_WINDOW_SIZE) nid = id - 4;
      meta[4 + q] = nid; meta[4 + GFBE_MAX_PRIOR_BLOCKS + q] = blk_gsize(id); meta[4 + 2 * GFBE_MAX_PRIOR_BLOCKS + q] = idx;
      s_xo[q] = xo;
      idx += blk_lsize(id); xo += blk_gsize(id);
    }
  }
  __syncthreads();
  for (int e = t; e < sh.n_keep * 16; e += blockDim.x) {
    const int q = e >> 4, k = e & 15, id = sh.keep_id[q];
    if (k < blk_gsize(id)) d.mx0[(size_t)w * PRIOR_X0 + s_xo[q] + k] = Xo[blk_amb(id) + k];
  }
    };
  double *Ap = J0 + (size_t)ND * ms;   // compact A' (n x n, ld = n) — fits: 32 ND + n^2 <= ND^2 for n <= 229
  for (int e = t; e < n * n; e += blockDim.x) {
    const int i = e / n, j = e % n;
    double s = A[(size_t)sh.keep_dim[i] * ND + sh.keep_dim[j]];
    if (tl) { for (int k = 0; k < m; k++) s -= T[i * ms + k] * Amr[k * n + j]; }
    else for (int k = 0; k < m; k++) s -= T[i * ms + k] * A[(size_t)sh.drop_dim[k] * ND + sh.keep_dim[j]];
    Ap[(size_t)i * n + j] = s;
  }
  for (int i = t; i < n; i += blockDim.x) {
    double s = bv[sh.keep_dim[i]];
    for (int k = 0; k < m; k++) s -= T[i * ms + k] * bm[k];
    r0[i] = s;           // b' staged in r0
  }
  __syncthreads();
  // keep A', b' for inspection (tests compare J0^T J0 with A'): copy into mA / mb compactly
  for (int e = t; e < n * n; e += blockDim.x) {
    const int i = e / n, j = e % n;
    A[e] = 0.5 * (Ap[(size_t)i * n + j] + Ap[(size_t)j * n + i]);
  }
  for (int i = t; i < n; i += blockDim.x) bv[i] = r0[i];
  __syncthreads();
  MSTAMP(4);
  __shared__ int order[ND];
  if (d.opt.marg_sqrt == 1 && n <= LDLT_MAX_N) {
    // the square root itself is taken by k_marg_ldlt (own kernel: the matrix is register-resident there)
    if (t == 0) sh.sweeps = MARG_SQRT_PENDING;
  } else {
  // ---- A' = V S V^T (the reference's construction). One-sided Jacobi; G (= A' V) and V live in LDS
  // when they fit (the common n = 86 prior), otherwise in the global scratch.
  const bool in_lds = (n <= MARG_LDS_N);
  double *G = in_lds ? marg_lds : J0;   // n x n; J0's storage (T / Ap) is free: both were consumed into A
  double *Vm = in_lds ? marg_lds + (size_t)n * n : d.mV + (size_t)w * ND * ND;
  for (int e = t; e < n * n; e += blockDim.x) G[e] = A[e];
  __syncthreads();
  double *lam = d.gts + (size_t)w * ND;           // solver scratch is dead by now
  // Householder + QL like Eigen's solver (round 5; the Jacobi of rounds 1-4 stays as the fallback of a sweep limit nobody has hit)
  double *wk = in_lds ? marg_lds + 2 * (size_t)n * n : (d.solve_big ? d.solveS + (size_t)w * d.solve_scratch_stride : d.solveY + (size_t)w * d.solve_scratch_stride);
  double *estamp = (GFBE_DIAG && w == 0) ? d.timing + (size_t)d.B * 32 : nullptr;      // (diagnostics build: phase stamps into the extra timing block)
  // (round 6: in LDS the tridiagonal eigenproblem is solved by divide & conquer — tridiag_dc — with the eigenvectors left in the window's
  //  global scratch d.mV, free in this case; GFBE_EIG_DC = 0 builds the QL iteration of round 5)
  double *park = (GFBE_EIG_DC && in_lds) ? d.mV + (size_t)w * ND * ND : nullptr;
  const int erc = in_lds ? tridiag_ql_eig((mlds_double *)G, (mlds_double *)Vm, n, n, lam, (mlds_double *)wk, estamp, park) : tridiag_ql_eig(G, Vm, n, n, lam, wk, estamp);
  if (erc == 1) {
    for (int e = t; e < n * n; e += blockDim.x) G[e] = A[e];
    __syncthreads();
    jacobi_eig(G, Vm, n, n, lam, &cflag, &sh.sweeps);
  } else if (t == 0) sh.sweeps = 1;
  if (erc == 2) Vm = park;
  __syncthreads();
  G = J0;                                         // J0 rows are written to global below
  // J0 = diag(sqrt(S)) V^T, r0 = diag(1/sqrt(S)) V^T b'   (marginalization_factor.cpp:294-302)
  // rows are ordered by ascending eigenvalue like Eigen's solver
  // (round 6: a rank sort by the whole workgroup — the insertion sort one lane ran over `lam` in global memory was ~n^2 / 4 dependent
  //  L2 round trips once the eigenvalues arrive unsorted, as the divide & conquer's do: ~0.4 ms of the call. Ties by index: the stable
  //  order of the insertion sort.)
  for (int j = t; j < n; j += blockDim.x) {
    const double lj = lam[j];
    int r = 0;
    for (int i = 0; i < n; i++) { const double li = lam[i]; r += (li < lj || (li == lj && i < j)) ? 1 : 0; }
    order[r] = j;
  }
  __syncthreads();
  for (int k = t; k < n; k += blockDim.x) {
    const int src = order[k];
    const double S = lam[src] > d.opt.marg_eps ? lam[src] : 0.0;
    const double Sinv = lam[src] > d.opt.marg_eps ? 1.0 / lam[src] : 0.0;
    double vb = 0.0;
    for (int i = 0; i < n; i++) vb += Vm[(size_t)src * n + i] * bv[i];
    r0[k] = sqrt(Sinv) * vb;
    d.Dp[(size_t)w * ND + k] = sqrt(S);
  }
  __syncthreads();
  for (int e = t; e < n * n; e += blockDim.x) {
    const int k = e / n, i = e % n;
    G[e] = d.Dp[(size_t)w * ND + k] * Vm[(size_t)order[k] * n + i];   // G aliases J0: row k of J0
  }
  }
  MSTAMP(5);
  write_blocks(sh.sweeps);
  if (t == 0) d.ctl[w].t_marg = (long long)wall_clock64();
}


// R = 4: priors of up to 88 dims (the 86 of the shipped configuration: 16 matrix registers per thread, two workgroups per CU);
// R = 6: up to 132 (a GNSS window's); R = 8: larger ones (up to 176). Both are launched when a batch may hold both kinds; a workgroup leaves the other kind alone.
template <int R>
__global__ __launch_bounds__(LDLT_THREADS) void k_marg_ldlt(BatchDev d) {
  const int w = blockIdx.x;
  int *meta = d.mmeta + (size_t)w * (4 + 3 * GFBE_MAX_PRIOR_BLOCKS);
  if (meta[0] != 1 || meta[3] != MARG_SQRT_PENDING) return;
  const int n = meta[1];
  if (R != (n <= 4 * 22 ? 4 : (n <= 6 * 22 ? 6 : 8))) return;      // (R = 6, round 6: the ~95-dim prior of a GNSS window on 36 instead of 64 entries per thread)
  const double *A = d.mA + (size_t)w * ND * ND;
  const double *bv = d.mb + (size_t)w * ND;
  double *J0 = d.mJ0 + (size_t)w * ND * ND;
  double *r0 = d.mr0 + (size_t)w * ND;
  double *stamp = d.timing + 24;
  const int rank = ldlt_registers<R>([&](int i, int j) { return A[(size_t)i * n + j]; }, [&](int i) { return bv[i]; }, J0, r0, n, d.opt.marg_eps,
                                     d.timing + (size_t)d.B * 32);
  if (threadIdx.x == 0) { meta[3] = -rank; d.ctl[w].t_marg = (long long)wall_clock64(); if (w == 0) stamp[6] = (double)wall_clock64(); }
}

// Throughput batches (round 6): ONE wave per window. The workgroup form above updates all n^2 entries at every pivot on eight waves (both
// triangles, 16 entries per thread): beside the other parts' kernels what counts is the waves it holds and the FP64 instructions it issues
// (it is bound by them: 512 windows x 86 pivots x 7744 entries x (mul + sub) at 8 cycles per instruction = the 145 us it took per launch).
// Here the LOWER triangle only, as 9 x 9 register tiles on the 55 tiles of a 10 x 10 grid — one lane each, n <= 90: 0.6 of the
// instructions, an eighth of the waves, no block barrier (the published row goes through LDS inside the wave). Row p_k of the symmetric
// matrix is row p_k of the tiles left of the diagonal tile and column p_k of the tiles below it. Every entry that is kept sees the
// operations ldlt_registers applies to it, in their order — but NOT the same bits at the end: the workgroup form updates entry (i, j)
// with (c_i / d) c_j and entry (j, i) with (c_j / d) c_i, its matrix is symmetric to rounding only, and it publishes row p_k from the
// upper triangle where this one reads the lower. Measured on 60 priors (tools/diag_scripts/ldlt_wave_check.py): J0^T J0 agrees to 1.1e-14
// of its largest entry, J0^T r0 to 1.1e-12 (entries of J0 along the weakest directions move by up to 1.5e-4; one prior keeps a pivot that
// sits on the eps threshold which the other drops). The two kernel sets are compared on tolerances everywhere (tests/test_gpu_parity.py::
// test_large_batch_throughput_path, tools/diag_soak_batch.py: 0 discrete differences in 300 windows); inside the throughput set a window's
// prior does not depend on the batch.
#ifndef GFBE_LDLT_TP
#define GFBE_LDLT_TP 1
#endif
enum { LW_R = 9, LW_G = 10, LW_TILES = LW_G * (LW_G + 1) / 2, LW_MAX_N = LW_R * LW_G, LW_CB = 128 };
static_assert(LW_TILES <= 64 && LW_MAX_N >= 4 * 22 && LW_MAX_N <= LW_CB, "k_marg_ldlt_tp: one lane per lower tile, every prior k_marg_ldlt<4> takes");
__global__ __launch_bounds__(64) void k_marg_ldlt_tp(BatchDev d) {
  const int w = blockIdx.x;
  int *meta = d.mmeta + (size_t)w * (4 + 3 * GFBE_MAX_PRIOR_BLOCKS);
  if (meta[0] != 1 || meta[3] != MARG_SQRT_PENDING) return;
  const int n = meta[1];
  if (n > 4 * 22) return;      // (k_marg_ldlt<8>, launched behind this kernel when the batch may hold such a prior)
  const double *A = d.mA + (size_t)w * ND * ND;
  const double *bv = d.mb + (size_t)w * ND;
  double *J0 = d.mJ0 + (size_t)w * ND * ND;
  double *r0 = d.mr0 + (size_t)w * ND;
  const double eps = d.opt.marg_eps;
  constexpr int R = LW_R;
  const int lane = threadIdx.x;
  __shared__ double colbuf[2][LW_CB];
  colbuf[0][lane] = 0.0; colbuf[0][lane + 64] = 0.0; colbuf[1][lane] = 0.0; colbuf[1][lane + 64] = 0.0;
  int ti = 0, tj = 0;
  const bool owner = lane < LW_TILES;
  if (owner) { while ((ti + 1) * (ti + 2) / 2 <= lane) ti++; tj = lane - ti * (ti + 1) / 2; }      // tile lane of the lower triangle, row-major: ti >= tj
  double a[R][R];
#pragma unroll
  for (int r = 0; r < R; r++)
#pragma unroll
    for (int c = 0; c < R; c++) {
      const int i = ti * R + r, j = tj * R + c;
      a[r][c] = (owner && i < n && j < n) ? A[(size_t)i * n + j] : 0.0;
    }
  double dg[2], bzr[2];
  bool alive[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int i = lane + 64 * q;
    alive[q] = i < n;
    dg[q] = alive[q] ? A[(size_t)i * n + i] : 0.0;
    bzr[q] = alive[q] ? bv[i] : 0.0;
  }
  int rank = n;
  for (int k = 0; k < n; k++) {
    // the largest remaining diagonal entry, smallest index among equals (ldlt_registers: three 32-bit reductions)
    unsigned khi = 0, klo = 0;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const unsigned h = (unsigned)__double2hiint(dg[q]), l = (unsigned)__double2loint(dg[q]);
      const bool cand = alive[q] && !(h >> 31);
      const bool gt = cand && (h > khi || (h == khi && l > klo));
      khi = gt ? h : khi; klo = gt ? l : klo;
    }
    const unsigned mh = wave_max_u32(khi);
    const unsigned ml = wave_max_u32(khi == mh ? klo : 0u);
    unsigned ci_ = 0xffffffffu;
#pragma unroll
    for (int q = 1; q >= 0; q--)
      if (alive[q] && (unsigned)__double2hiint(dg[q]) == mh && (unsigned)__double2loint(dg[q]) == ml) ci_ = (unsigned)(lane + 64 * q);
    const int pv = (int)wave_min_u32(ci_);
    const double piv = __hiloint2double((int)mh, (int)ml);
    if (!(piv > eps) || (mh | ml) == 0u) { rank = k; break; }
    const double inv = 1.0 / piv;
    const int pq = pv >> 6, pl = pv & 63;
    const double zk = __shfl(pq == 0 ? bzr[0] : bzr[1], pl, 64);
    double *cb = colbuf[k & 1];
    const int pr = pv / R, pc = pv - pr * R;
    if (owner && ti == pr) {                       // row p_k: its entries left of and inside the diagonal tile
#pragma unroll
      for (int c = 0; c < R; c++) {
        double v = a[0][c];
#pragma unroll
        for (int r = 1; r < R; r++) v = pc == r ? a[r][c] : v;
        cb[tj * R + c] = v;
      }
    }
    if (owner && tj == pr && ti > pr) {            // ... and below it: column p_k of the tiles under the diagonal tile
#pragma unroll
      for (int r = 0; r < R; r++) {
        double v = a[r][0];
#pragma unroll
        for (int c = 1; c < R; c++) v = pc == c ? a[r][c] : v;
        cb[ti * R + r] = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    double cx[2];
#pragma unroll
    for (int q = 0; q < 2; q++) cx[q] = cb[lane + 64 * q];
    {                                              // row k of J0, r0[k]
      const double rs = 1.0 / sqrt(piv);
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int i = lane + 64 * q;
        if (i < n) J0[(size_t)k * n + i] = i == pv ? sqrt(piv) : (alive[q] ? cx[q] * rs : 0.0);
      }
      if (lane == 0) r0[k] = zk * rs;
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const double li = cx[q] * inv;
      dg[q] -= li * cx[q];
      bzr[q] -= li * zk;
      if (lane + 64 * q == pv) alive[q] = false;
    }
    if (owner) {
      double ci[R], cj[R];
#pragma unroll
      for (int r = 0; r < R; r++) { ci[r] = cb[ti * R + r] * inv; cj[r] = cb[tj * R + r]; }
#pragma unroll
      for (int r = 0; r < R; r++)
#pragma unroll
        for (int c = 0; c < R; c++) a[r][c] -= ci[r] * cj[c];
    }
    __builtin_amdgcn_wave_barrier();
  }
  for (int e = lane + rank * n; e < n * n; e += 64) J0[e] = 0.0;
  for (int k = lane + rank; k < n; k += 64) r0[k] = 0.0;
  if (lane == 0) { meta[3] = -rank; d.ctl[w].t_marg = (long long)wall_clock64(); if (w == 0) d.timing[24 + 6] = (double)wall_clock64(); }
}

// MARGIN_OLD: the partials of the marginalisation set at the re-anchored state (visual factors of the landmarks
// starting in frame 0, the inertial / wheel factor of frame 0, their Schur partial)
// dense_elsewhere (throughput batches, end of round 6): the caller runs the frame-0 inertial / wheel / prior factors on its side stream, beside the
// visual kernels of the marginalisation set, and joins in front of launch_marginalize_finish (gfbe_host.cpp: enqueue_solve)
void launch_marginalize_partials(const BatchDev &d, hipStream_t s, bool dense_elsewhere) {
  if ((GFBE_FUSE_SMALL & 8) && d.B < DENSE_SPLIT_MIN_B && d.vis_Hs && !d.sharded && d.max_tiles > 0) {   // (small batches: two launches instead of four)
    launch_lin_small(d, 2, s);
    launch_pair_schur_marg(d, s);
    launch_gnss(d, 2, s);
    return;
  }
  launch_vis(d, 2, s);
  launch_pair(d, 1, s);
  if (!dense_elsewhere) launch_dense_factors(d, 2, 0, s);
  launch_gnss(d, 2, s);
  launch_schur(d, 1, s);
}
void launch_marginalize(const BatchDev &d, int flag, hipStream_t s) {
  if (flag == GFBE_MARGIN_OLD) launch_marginalize_partials(d, s);
  else launch_dense_factors(d, 3, 0, s);
  launch_marginalize_finish(d, flag, s);
}
hipError_t marg_init_device() {   // per device, from gfbe_create (see kernels_init_device)
  return hipFuncSetAttribute((const void *)k_marg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * MARG_LDS_DOUBLES));
}
void launch_marginalize_finish(const BatchDev &d, int flag, hipStream_t s) {
  // dynamic LDS: the eigen-decomposition of A' (marg_sqrt = 0) wants G and V resident; with the LDL^T square root only the
  // 20-dim dense elimination of a GNSS window uses it (four 32 x 32 blocks)
  const size_t lds = sizeof(double) * (d.opt.marg_sqrt == 1 ? 4 * 32 * 32 : MARG_LDS_DOUBLES);
  // (throughput batches: 512-thread workgroups, two per CU, overlap each other's barrier stalls — 1.04 -> 0.92 ms for the whole
  //  marginalisation of 1024 windows; a single window keeps the 1024 threads of its one workgroup)
  hipLaunchKernelGGL(k_marg, dim3(d.B), dim3(d.B >= DENSE_SPLIT_MIN_B ? GFBE_MARG_TP_THREADS : MARG_THREADS), lds, s, d, flag);
  if (d.opt.marg_sqrt == 1) {
    if (GFBE_LDLT_TP == 2 || (GFBE_LDLT_TP && d.B >= DENSE_SPLIT_MIN_B)) hipLaunchKernelGGL(k_marg_ldlt_tp, dim3(d.B), dim3(64), 0, s, d);      // (2: every batch — the bit-for-bit check against k_marg_ldlt<4>)
    else hipLaunchKernelGGL(k_marg_ldlt<4>, dim3(d.B), dim3(LDLT_THREADS), 0, s, d);
    if (d.marg_nmax > 4 * 22) hipLaunchKernelGGL(k_marg_ldlt<6>, dim3(d.B), dim3(LDLT_THREADS), 0, s, d);
    if (d.marg_nmax > 6 * 22) hipLaunchKernelGGL(k_marg_ldlt<8>, dim3(d.B), dim3(LDLT_THREADS), 0, s, d);
  }
}

}  // namespace gfd

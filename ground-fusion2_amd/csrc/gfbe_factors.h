// gfbe_factors.h — residual + analytic tangent-space Jacobian of the four factor families, as
// __host__ __device__ functions used by the HIP kernels (gfbe_kernels.hip). Reference semantics:
//   visual  Ground-Fusion++/vins_estimator/src/factor/projectionTwoFrameOneCamFactor.cpp:43-151
//   IMU     Ground-Fusion++/vins_estimator/src/factor/imu_factor.h:28-191, integration_base.h:169-195
//   wheel   Ground-Fusion++/vins_estimator/src/factor/wheel_factor.h:28-247, wheel_integration_base.h:180-219
//   prior   Ground-Fusion++/vins_estimator/src/factor/marginalization_factor.cpp:344-392
//   robust-loss corrector  marginalization_factor.cpp:46-77 (ceres::HuberLoss)
// Tangent-space Jacobian of a pose block = first 6 columns of the 7-wide global block
// (pose_local_parameterization.cpp:27-34).
#pragma once
#include "gfbe_math.h"
#include "../../include/gfbe.h"

namespace gfd {

// A pose staged for factor evaluation: translation + rotation matrix (computed once per
// workgroup into LDS instead of once per factor as the reference does, :83-85).
struct PoseRT {
  vec3 t;
  mat3 R;
};
GF_HD PoseRT make_pose(const double *p7) {
  PoseRT o;
  o.t = ld3(p7);
  o.R = qrot(ldq(p7 + 3));
  return o;
}

// ---------------------------------------------------------------------------------------------
// Visual factor. Outputs r[2]; if JAC: Ji/Jj/Je [2][6] row-major, Jl[2] (d/d inverse depth), Jt[2].
// ---------------------------------------------------------------------------------------------
template <bool JAC>
GF_HD void visual_eval(const PoseRT &Fi, const PoseRT &Fj, const PoseRT &Ex, double inv_dep, double td,
                       double pix, double piy, double piz, double pjx, double pjy,
                       double vix, double viy, double vjx, double vjy, double td_i, double td_j, double sqrt_info,
                       double *r, double *Ji, double *Jj, double *Je, double *Jl, double *Jt) {
  const double dti = td - td_i, dtj = td - td_j;
  const vec3 pi_td = mk3(pix - dti * vix, piy - dti * viy, piz);
  const double pjx_td = pjx - dtj * vjx, pjy_td = pjy - dtj * vjy;
  const double inv_l = 1.0 / inv_dep;
  const vec3 p_ci = scl(inv_l, pi_td);
  const vec3 p_bi = add(mv(Ex.R, p_ci), Ex.t);
  const vec3 p_w = add(mv(Fi.R, p_bi), Fi.t);
  const vec3 p_bj = tmv(Fj.R, sub(p_w, Fj.t));
  const vec3 p_cj = tmv(Ex.R, sub(p_bj, Ex.t));
  const double dep = p_cj[2];
  const double inv_z = 1.0 / dep;
  r[0] = sqrt_info * (p_cj[0] * inv_z - pjx_td);
  r[1] = sqrt_info * (p_cj[1] * inv_z - pjy_td);
  if (!JAC) return;
  // reduce = sqrt_info * d(pi(P))/dP   (:97-101)
  const double r00 = sqrt_info * inv_z, r02 = -sqrt_info * p_cj[0] * inv_z * inv_z;
  const double r11 = sqrt_info * inv_z, r12 = -sqrt_info * p_cj[1] * inv_z * inv_z;
  // A = ric^T Rj^T ; B = A Ri
  const mat3 A = tmul(Ex.R, transp(Fj.R));
  const mat3 B = mul(A, Fi.R);
  const mat3 Tm = mul(B, Ex.R);                       // ric^T Rj^T Ri ric  (:129)
  const mat3 ji_r = mneg(mul(B, hat(p_bi)));          // :107
  const mat3 jj_r = tmul(Ex.R, hat(p_bj));            // :119
  const mat3 je_p = tmul(Ex.R, msub(tmul(Fj.R, Fi.R), ident3()));   // :128
  const vec3 Tp = mv(Tm, p_ci);
  const vec3 lev = tmv(Ex.R, sub(tmv(Fj.R, sub(add(mv(Fi.R, Ex.t), Fi.t), Fj.t)), Ex.t));
  const mat3 je_r = madd(madd(mneg(mul(Tm, hat(p_ci))), hat(Tp)), hat(lev));   // :130-131
#pragma unroll
  for (int c = 0; c < 3; c++) {
    Ji[c] = r00 * A(0, c) + r02 * A(2, c);          Ji[6 + c] = r11 * A(1, c) + r12 * A(2, c);
    Ji[3 + c] = r00 * ji_r(0, c) + r02 * ji_r(2, c); Ji[9 + c] = r11 * ji_r(1, c) + r12 * ji_r(2, c);
    Jj[c] = -Ji[c];                                  Jj[6 + c] = -Ji[6 + c];
    Jj[3 + c] = r00 * jj_r(0, c) + r02 * jj_r(2, c); Jj[9 + c] = r11 * jj_r(1, c) + r12 * jj_r(2, c);
    Je[c] = r00 * je_p(0, c) + r02 * je_p(2, c);     Je[6 + c] = r11 * je_p(1, c) + r12 * je_p(2, c);
    Je[3 + c] = r00 * je_r(0, c) + r02 * je_r(2, c); Je[9 + c] = r11 * je_r(1, c) + r12 * je_r(2, c);
  }
  const vec3 tl = scl(-inv_l * inv_l, mv(Tm, pi_td));              // :139
  Jl[0] = r00 * tl[0] + r02 * tl[2];
  Jl[1] = r11 * tl[1] + r12 * tl[2];
  const vec3 tt = scl(-inv_l, mv(Tm, mk3(vix, viy, 0.0)));         // :144
  Jt[0] = r00 * tt[0] + r02 * tt[2] + sqrt_info * vjx;             // :145
  Jt[1] = r11 * tt[1] + r12 * tt[2] + sqrt_info * vjy;
}

// ---------------------------------------------------------------------------------------------
// Pose-pair constants of the visual factor. Every factor of a pose pair (i, j) shares the products of
// the three rotations; the kernels compute them once per workgroup and per pair instead of once per
// factor (the reference rebuilds them in every Evaluate, projectionTwoFrameOneCamFactor.cpp:83-131).
//   A = ric^T Rj^T, B = A Ri, Tm = B ric, u = A (ti - tj) + B tic - ric^T tic   =>   P_cj = Tm P_ci + u
// The Jacobian blocks use R [v]x = [R v]x R (R a rotation: parameter quaternions are unit):
//   d/dtheta_i : -B [P_bi]x        = -[Tm P_ci + B tic]x B
//   d/dtheta_j :  ric^T [P_bj]x    =  [P_cj + ric^T tic]x ric^T
//   d/dtheta_ex: -Tm [P_ci]x + [Tm P_ci]x + [u]x = [Tm P_ci]x (I - Tm) + [u]x
// ---------------------------------------------------------------------------------------------
// (the members the kernels need come in the order of their prefixes: the cost pass stages Tm and u, the linearisation of a window
//  with constant extrinsic / td — visual_lin_y below — the 33 doubles of PairConstY, the marginalisation's 13-column panel
//  PairConstR; only a free extrinsic or td needs the whole record)
struct PairConstY {
  mat3 Tm; vec3 u;                  // P_cj = Tm P_ci + u
  mat3 A;                           // ric^T Rj^T: the world-to-camera-j rotation (dr/dP_w = reduce A)
  mat3 Rj; vec3 dP;                 // rotation of pose j; ti - tj
};
struct PairConstR : PairConstY {
  mat3 B, ricT;
  vec3 Btic, c2;                    // c2 = ric^T tic
};
struct PairConst : PairConstR {
  mat3 ImTm, jep;                   // jep = B - ric^T  (translation block of the extrinsic)
};
// Per-frame constants of the start frame of a landmark tile (slot (i, i) of the pair table): a landmark's world-frame vectors
//   f = W p_ci (from the camera centre of frame i), e = f + wt (from the body origin of frame i), W = Ri ric, wt = Ri tic
//   dPc = P_i - P_0: the landmark seen from the window's origin (frame 0), P_w - P_0 = e + dPc (the panel's second half, below)
struct FrameConst {
  mat3 W; vec3 wt; mat3 R; vec3 dPc;
};
enum { PC_DOUBLES = sizeof(PairConst) / sizeof(double), PCR_DOUBLES = sizeof(PairConstR) / sizeof(double),
       PCY_DOUBLES = sizeof(PairConstY) / sizeof(double), FC_DOUBLES = sizeof(FrameConst) / sizeof(double) };
GF_HD FrameConst make_frame_const(const PoseRT &Fi, const PoseRT &Ex, const PoseRT &F0) {
  FrameConst f;
  f.W = mul(Fi.R, Ex.R);
  f.wt = mv(Fi.R, Ex.t);
  f.R = Fi.R;
  f.dPc = sub(Fi.t, F0.t);
  return f;
}
GF_HD PairConst make_pair_const(const PoseRT &Fi, const PoseRT &Fj, const PoseRT &Ex) {
  PairConst p;
  p.Rj = Fj.R;
  p.dP = sub(Fi.t, Fj.t);
  p.ricT = transp(Ex.R);
  p.A = tmul(Ex.R, transp(Fj.R));
  p.B = mul(p.A, Fi.R);
  p.Tm = mul(p.B, Ex.R);
  p.ImTm = msub(ident3(), p.Tm);
  p.jep = msub(p.B, p.ricT);
  p.Btic = mv(p.B, Ex.t);
  p.c2 = mv(p.ricT, Ex.t);
  p.u = sub(add(mv(p.A, sub(Fi.t, Fj.t)), p.Btic), p.c2);
  return p;
}
// [v]x * M
GF_HD mat3 hat_mul(const vec3 &v, const mat3 &M) {
  mat3 r;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    r(0, c) = v[1] * M(2, c) - v[2] * M(1, c);
    r(1, c) = v[2] * M(0, c) - v[0] * M(2, c);
    r(2, c) = v[0] * M(1, c) - v[1] * M(0, c);
  }
  return r;
}

template <bool JAC>
GF_HD void visual_eval_pc(const PairConst &pc, double inv_dep, double td, double pix, double piy, double piz, double pjx,
                          double pjy, double vix, double viy, double vjx, double vjy, double td_i, double td_j,
                          double sqrt_info, double *r, double *Ji, double *Jj, double *Je, double *Jl, double *Jt) {
  const double dti = td - td_i, dtj = td - td_j;
  const double inv_l = 1.0 / inv_dep;
  const vec3 p_ci = mk3((pix - dti * vix) * inv_l, (piy - dti * viy) * inv_l, piz * inv_l);
  const vec3 q = mv(pc.Tm, p_ci);
  const vec3 p_cj = add(q, pc.u);
  const double inv_z = 1.0 / p_cj[2];
  r[0] = sqrt_info * (p_cj[0] * inv_z - (pjx - dtj * vjx));
  r[1] = sqrt_info * (p_cj[1] * inv_z - (pjy - dtj * vjy));
  if (!JAC) return;
  const double r00 = sqrt_info * inv_z, r02 = -sqrt_info * p_cj[0] * inv_z * inv_z;
  const double r12 = -sqrt_info * p_cj[1] * inv_z * inv_z;   // r11 == r00
  const mat3 ji_r = mneg(hat_mul(add(q, pc.Btic), pc.B));
  const mat3 jj_r = hat_mul(add(p_cj, pc.c2), pc.ricT);
  const mat3 je_r = madd(hat_mul(q, pc.ImTm), hat(pc.u));
#pragma unroll
  for (int c = 0; c < 3; c++) {
    Ji[c] = r00 * pc.A(0, c) + r02 * pc.A(2, c);      Ji[6 + c] = r00 * pc.A(1, c) + r12 * pc.A(2, c);
    Ji[3 + c] = r00 * ji_r(0, c) + r02 * ji_r(2, c);  Ji[9 + c] = r00 * ji_r(1, c) + r12 * ji_r(2, c);
    Jj[c] = -Ji[c];                                   Jj[6 + c] = -Ji[6 + c];
    Jj[3 + c] = r00 * jj_r(0, c) + r02 * jj_r(2, c);  Jj[9 + c] = r00 * jj_r(1, c) + r12 * jj_r(2, c);
    Je[c] = r00 * pc.jep(0, c) + r02 * pc.jep(2, c);  Je[6 + c] = r00 * pc.jep(1, c) + r12 * pc.jep(2, c);
    Je[3 + c] = r00 * je_r(0, c) + r02 * je_r(2, c);  Je[9 + c] = r00 * je_r(1, c) + r12 * je_r(2, c);
  }
  // d/d lambda = reduce * Tm * p_i' * (-1/lambda^2) = -reduce * q / lambda   (:139)
  Jl[0] = -(r00 * q[0] + r02 * q[2]) * inv_l;
  Jl[1] = -(r00 * q[1] + r12 * q[2]) * inv_l;
  // d/d td = reduce * Tm * [v_i;0] * (-1/lambda) + sqrt_info * v_j   (:144-145)
  const double t0 = pc.Tm(0, 0) * vix + pc.Tm(0, 1) * viy, t1 = pc.Tm(1, 0) * vix + pc.Tm(1, 1) * viy;
  const double t2 = pc.Tm(2, 0) * vix + pc.Tm(2, 1) * viy;
  Jt[0] = -(r00 * t0 + r02 * t2) * inv_l + sqrt_info * vjx;
  Jt[1] = -(r00 * t1 + r12 * t2) * inv_l + sqrt_info * vjy;
}

// ceres::HuberLoss::Evaluate
GF_HD void huber_rho(double s, double delta, double *rho) {
  const double b = delta * delta;
  if (s > b) {
    const double rt = sqrt(s);
    rho[0] = 2.0 * delta * rt - b;
    const double r1 = delta / rt;
    rho[1] = r1 > 2.2250738585072014e-308 ? r1 : 2.2250738585072014e-308;
    rho[2] = -rho[1] / (2.0 * s);
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}
// Corrector coefficients (marginalization_factor.cpp:57-70): J <- sqrt_rho1 (J - alpha_sq_norm r r^T J),
// r <- residual_scaling r. Returns 0.5 rho(s).
GF_HD double corrector(double s, double delta, double *sqrt_rho1, double *residual_scaling, double *alpha_sq_norm) {
  double rho[3];
  huber_rho(s, delta, rho);
  *sqrt_rho1 = sqrt(rho[1]);
  if (s == 0.0 || rho[2] <= 0.0) {
    *residual_scaling = *sqrt_rho1;
    *alpha_sq_norm = 0.0;
  } else {
    const double D = 1.0 + 2.0 * s * rho[2] / rho[1];
    const double alpha = 1.0 - sqrt(D);
    *residual_scaling = *sqrt_rho1 / (1.0 - alpha);
    *alpha_sq_norm = alpha / s;
  }
  return 0.5 * rho[0];
}
// Apply to a 2-row block with columns given as separate row arrays.
GF_HD void correct_cols(double *row0, double *row1, int n, double r0, double r1, double sqrt_rho1, double asn) {
  if (asn == 0.0) {   // always the case for HuberLoss (rho'' <= 0): J <- sqrt(rho') J
    for (int c = 0; c < n; c++) { row0[c] *= sqrt_rho1; row1[c] *= sqrt_rho1; }
    return;
  }
  for (int c = 0; c < n; c++) {
    const double rtj = r0 * row0[c] + r1 * row1[c];
    row0[c] = sqrt_rho1 * (row0[c] - asn * r0 * rtj);
    row1[c] = sqrt_rho1 * (row1[c] - asn * r1 * rtj);
  }
}

// ---------------------------------------------------------------------------------------------
// visual_lin: what the linearisation kernels run per factor — residual, HuberLoss corrector and the corrected tangent
// Jacobian in one pass (projectionTwoFrameOneCamFactor.cpp:43-151 + marginalization_factor.cpp:46-77), written for the
// FP64 pipe of gfx950 (matrix-core and vector FP64 instructions do not overlap there, profiles/ubench/
// mfma_valu_overlap_mi355x.txt, so every vector instruction saved is time saved):
//   * explicit fused multiply-adds (the library is built with -ffp-contract=off: what is fused is fused HERE, identically
//     in every kernel that inlines this function, so the small-batch and throughput kernel sets stay bit-identical);
//   * the corrector's sqrt(rho') goes into the 2 x 3 projection derivative `reduce` BEFORE the Jacobian blocks are formed
//     (4 multiplications instead of 42 on the finished blocks; rho' = 1 for an inlier, where this changes no bit);
//     the rank-one term of a loss with rho'' > 0 (never HuberLoss: alpha = 0) is applied to the scaled blocks afterwards;
//   * FULL = false: the camera extrinsic and td are constant in this window (the shipped yamls: estimate_extrinsic 0,
//     estimate_td 0) — their Jacobian blocks are not formed, as Ceres passes jacobians[2] = jacobians[4] = nullptr
//     for constant blocks (projectionTwoFrameOneCamFactor.cpp:122,141).
// Returns 0.5 rho(|r|^2). JAC = false: cost only (r is the uncorrected residual).
// ---------------------------------------------------------------------------------------------
template <bool JAC, bool FULL, typename PC>
GF_HD double visual_lin(const PC &pc, double inv_dep, double td, double pix, double piy, double piz, double pjx,
                        double pjy, double vix, double viy, double vjx, double vjy, double td_i, double td_j,
                        double sqrt_info, double delta, double *r, double *Ji, double *Jj, double *Je, double *Jl, double *Jt) {
  const double dti = td - td_i, dtj = td - td_j;
  const double inv_l = 1.0 / inv_dep;
  const double cx = __builtin_fma(-dti, vix, pix) * inv_l, cy = __builtin_fma(-dti, viy, piy) * inv_l, cz = piz * inv_l;   // P_ci
  vec3 q;
#pragma unroll
  for (int a = 0; a < 3; a++) q[a] = __builtin_fma(pc.Tm(a, 0), cx, __builtin_fma(pc.Tm(a, 1), cy, pc.Tm(a, 2) * cz));
  const double X = q[0] + pc.u[0], Y = q[1] + pc.u[1], Z = q[2] + pc.u[2];                                               // P_cj
  const double inv_z = 1.0 / Z;
  const double r0 = sqrt_info * __builtin_fma(X, inv_z, -__builtin_fma(-dtj, vjx, pjx));
  const double r1 = sqrt_info * __builtin_fma(Y, inv_z, -__builtin_fma(-dtj, vjy, pjy));
  double s1, rs, asn;
  const double cost = corrector(__builtin_fma(r0, r0, r1 * r1), delta, &s1, &rs, &asn);
  if (!JAC) { r[0] = r0; r[1] = r1; return cost; }
  // reduce = sqrt(rho') sqrt_info d(pi(P))/dP
  const double si = s1 * sqrt_info;
  const double r00 = si * inv_z, r02 = -(r00 * X * inv_z), r12 = -(r00 * Y * inv_z);   // (r11 == r00)
  // rotation blocks through R [v]x = [R v]x R:  d/dtheta_i = -[q + B tic]x B,  d/dtheta_j = [P_cj + ric^T tic]x ric^T
  const double a0 = q[0] + pc.Btic[0], a1 = q[1] + pc.Btic[1], a2 = q[2] + pc.Btic[2];
  const double b0 = X + pc.c2[0], b1 = Y + pc.c2[1], b2 = Z + pc.c2[2];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    Ji[c] = __builtin_fma(r00, pc.A(0, c), r02 * pc.A(2, c));
    Ji[6 + c] = __builtin_fma(r00, pc.A(1, c), r12 * pc.A(2, c));
    Jj[c] = -Ji[c];
    Jj[6 + c] = -Ji[6 + c];
    // -[a]x B, rows 0..2 of column c
    const double i0 = __builtin_fma(a2, pc.B(1, c), -(a1 * pc.B(2, c))), i1 = __builtin_fma(a0, pc.B(2, c), -(a2 * pc.B(0, c))),
                 i2 = __builtin_fma(a1, pc.B(0, c), -(a0 * pc.B(1, c)));
    Ji[3 + c] = __builtin_fma(r00, i0, r02 * i2);
    Ji[9 + c] = __builtin_fma(r00, i1, r12 * i2);
    // [b]x ric^T
    const double j0 = __builtin_fma(b1, pc.ricT(2, c), -(b2 * pc.ricT(1, c))), j1 = __builtin_fma(b2, pc.ricT(0, c), -(b0 * pc.ricT(2, c))),
                 j2 = __builtin_fma(b0, pc.ricT(1, c), -(b1 * pc.ricT(0, c)));
    Jj[3 + c] = __builtin_fma(r00, j0, r02 * j2);
    Jj[9 + c] = __builtin_fma(r00, j1, r12 * j2);
    if constexpr (FULL) {
      Je[c] = __builtin_fma(r00, pc.jep(0, c), r02 * pc.jep(2, c));
      Je[6 + c] = __builtin_fma(r00, pc.jep(1, c), r12 * pc.jep(2, c));
      // [q]x (I - Tm) + [u]x
      const double e0 = __builtin_fma(q[1], pc.ImTm(2, c), -(q[2] * pc.ImTm(1, c))) + (c == 1 ? -pc.u[2] : (c == 2 ? pc.u[1] : 0.0));
      const double e1 = __builtin_fma(q[2], pc.ImTm(0, c), -(q[0] * pc.ImTm(2, c))) + (c == 0 ? pc.u[2] : (c == 2 ? -pc.u[0] : 0.0));
      const double e2 = __builtin_fma(q[0], pc.ImTm(1, c), -(q[1] * pc.ImTm(0, c))) + (c == 0 ? -pc.u[1] : (c == 1 ? pc.u[0] : 0.0));
      Je[3 + c] = __builtin_fma(r00, e0, r02 * e2);
      Je[9 + c] = __builtin_fma(r00, e1, r12 * e2);
    }
  }
  // d/d lambda = -reduce q / lambda
  Jl[0] = -(__builtin_fma(r00, q[0], r02 * q[2]) * inv_l);
  Jl[1] = -(__builtin_fma(r00, q[1], r12 * q[2]) * inv_l);
  if constexpr (FULL) {   // d/d td = -reduce Tm [v_i; 0] / lambda + sqrt(rho') sqrt_info v_j
    const double t0 = __builtin_fma(pc.Tm(0, 0), vix, pc.Tm(0, 1) * viy), t1 = __builtin_fma(pc.Tm(1, 0), vix, pc.Tm(1, 1) * viy);
    const double t2 = __builtin_fma(pc.Tm(2, 0), vix, pc.Tm(2, 1) * viy);
    Jt[0] = __builtin_fma(si, vjx, -(__builtin_fma(r00, t0, r02 * t2) * inv_l));
    Jt[1] = __builtin_fma(si, vjy, -(__builtin_fma(r00, t1, r12 * t2) * inv_l));
  }
  if (asn != 0.0) {   // a loss with rho'' > 0: J <- J - alpha/|r|^2 r r^T J on the sqrt(rho')-scaled blocks (r still uncorrected)
    auto fix = [&](double *row0, double *row1, int n) {
      for (int c = 0; c < n; c++) {
        const double rtj = __builtin_fma(r0, row0[c], r1 * row1[c]);
        row0[c] -= asn * r0 * rtj;
        row1[c] -= asn * r1 * rtj;
      }
    };
    fix(Ji, Ji + 6, 6); fix(Jj, Jj + 6, 6); fix(Jl, Jl + 1, 1);
    if constexpr (FULL) { fix(Je, Je + 6, 6); fix(Jt, Jt + 1, 1); }
  }
  r[0] = r0 * rs;
  r[1] = r1 * rs;
  return cost;
}

// ---------------------------------------------------------------------------------------------
// visual_lin_y: the linearisation of a window with constant extrinsic and td in the form the throughput kernels sum
// (round 4). Every Jacobian block of the factor is the 2 x 3 derivative of the residual with respect to the landmark's WORLD
// position, G = reduce ric^T Rj^T (reduce: projectionTwoFrameOneCamFactor.cpp:97-100, times the corrector's sqrt(rho')),
// times something that is constant over the factors of a pose pair or a 3-vector of the landmark — with e = P_w - P_i:
//   d/dP_i = G                       d/dtheta_i = -G [e]x Ri                      (:106-110;  Ri [P_bi]x = [e]x Ri)
//   d/dP_j = -G                      d/dtheta_j =  G [e + P_i - P_j]x Rj          (:118-122;  ric^T [P_bj]x = ric^T Rj^T [P_w - P_j]x Rj)
//   d/dlambda = -G f / lambda        f = Ri ric P_ci = e - Ri tic                 (:139)
// With the landmark taken from a point c common to the window (c = P_0; x = P_w - c = e + P_i - c) and
//   Y = [G | G [x]x]  (2 x 6 per factor; the rows of G [x]x are g x x),   T_f = [ I  [P_f - c]x R_f ;  0  -R_f ]  (6 x 6, per FRAME)
// this is  J_i = Y T_i,  J_j = -Y T_j:  the block of the pose pair in sum J^T J is -T_i^T (sum Y^T Y) T_j, a frame's diagonal
// block T_f^T (sum of Y^T Y over every factor that touches the frame) T_f.
// The kernels therefore sum the 7 x 7 matrix [Y r]^T [Y r] over the factors of a pair — ONE 16-wide matrix-core tile holds
// both rows of a factor — and apply the T_f once per frame (k_visasm), instead of summing the 13 x 13 one of [J_i J_j r]; and
// the per-factor work is G, two cross products and the landmark row below instead of four 3 x 3 products. (c inside the
// window keeps |x| at the size of the scene: G [x]x - G [P_i - c]x cancels nothing a landmark's own distance does not.)
// Landmark row (w = d/dlambda, d = G^T w): H_ll += w.w, g_l += w.r; the H_pl blocks of the two poses are
//   pose i: [ d ; Ri^T (e x d) ]  (summed over the landmark's factors: [ D ; Ri^T (e x D) ], D = sum d)
//   pose j: [ -d ; Rj^T (d x (e + P_i - P_j)) ].
// Out: r (corrected), g0 / g1 (rows of G), Jl. Returns 0.5 rho(|r|^2). cx, cy, cz = P_ci (the caller's, per landmark).
// ---------------------------------------------------------------------------------------------
template <typename PC>
GF_HD double visual_lin_y(const PC &pc, double cx, double cy, double cz, const vec3 &f, double inv_l, double td, double pjx, double pjy,
                          double vjx, double vjy, double td_j, double sqrt_info, double delta, double *r, double *g0, double *g1, double *Jl) {
  const double dtj = td - td_j;
  vec3 q;
#pragma unroll
  for (int a = 0; a < 3; a++) q[a] = __builtin_fma(pc.Tm(a, 0), cx, __builtin_fma(pc.Tm(a, 1), cy, pc.Tm(a, 2) * cz));
  const double X = q[0] + pc.u[0], Y = q[1] + pc.u[1], Z = q[2] + pc.u[2];                                               // P_cj
  const double inv_z = 1.0 / Z;
  const double r0 = sqrt_info * __builtin_fma(X, inv_z, -__builtin_fma(-dtj, vjx, pjx));
  const double r1 = sqrt_info * __builtin_fma(Y, inv_z, -__builtin_fma(-dtj, vjy, pjy));
  double s1, rs, asn;
  const double cost = corrector(__builtin_fma(r0, r0, r1 * r1), delta, &s1, &rs, &asn);
  const double si = s1 * sqrt_info;
  const double r00 = si * inv_z, r02 = -(r00 * X * inv_z), r12 = -(r00 * Y * inv_z);   // (r11 == r00)
#pragma unroll
  for (int c = 0; c < 3; c++) {
    g0[c] = __builtin_fma(r00, pc.A(0, c), r02 * pc.A(2, c));
    g1[c] = __builtin_fma(r00, pc.A(1, c), r12 * pc.A(2, c));
  }
  if (asn != 0.0) {   // a loss with rho'' > 0 (never HuberLoss): J <- J - alpha/|r|^2 r r^T J — on G, which every block is a multiple of
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double rtj = __builtin_fma(r0, g0[c], r1 * g1[c]);
      g0[c] -= asn * r0 * rtj;
      g1[c] -= asn * r1 * rtj;
    }
  }
  Jl[0] = -(__builtin_fma(g0[0], f[0], __builtin_fma(g0[1], f[1], g0[2] * f[2])) * inv_l);
  Jl[1] = -(__builtin_fma(g1[0], f[0], __builtin_fma(g1[1], f[1], g1[2] * f[2])) * inv_l);
  r[0] = r0 * rs;
  r[1] = r1 * rs;
  return cost;
}
// The candidate-cost pass: 0.5 rho(|r|^2) of one factor from the landmark's camera-frame point (cx, cy, cz — per landmark, not per
// factor: one division by lambda per landmark) — the same operations in the same order as visual_lin / visual_lin_y: the same bits.
template <typename PC>
GF_HD double visual_cost_y(const PC &pc, double cx, double cy, double cz, double td, double pjx, double pjy, double vjx, double vjy, double td_j,
                           double sqrt_info, double delta) {
  const double dtj = td - td_j;
  vec3 q;
#pragma unroll
  for (int a = 0; a < 3; a++) q[a] = __builtin_fma(pc.Tm(a, 0), cx, __builtin_fma(pc.Tm(a, 1), cy, pc.Tm(a, 2) * cz));
  const double X = q[0] + pc.u[0], Y = q[1] + pc.u[1], Z = q[2] + pc.u[2];
  const double inv_z = 1.0 / Z;
  const double r0 = sqrt_info * __builtin_fma(X, inv_z, -__builtin_fma(-dtj, vjx, pjx));
  const double r1 = sqrt_info * __builtin_fma(Y, inv_z, -__builtin_fma(-dtj, vjy, pjy));
  double s1, rs, asn;
  return corrector(__builtin_fma(r0, r0, r1 * r1), delta, &s1, &rs, &asn);
}
// a x b
GF_HD vec3 cross3(const vec3 &a, const vec3 &b) {
  return mk3(__builtin_fma(a[1], b[2], -(a[2] * b[1])), __builtin_fma(a[2], b[0], -(a[0] * b[2])), __builtin_fma(a[0], b[1], -(a[1] * b[0])));
}

// ---------------------------------------------------------------------------------------------
// sqrt_info = LLT(cov^-1).matrixL()^T (imu_factor.h:73, wheel_factor.h:85): partial-pivot LU inverse
// (Eigen's inverse() for n > 4) followed by a lower Cholesky. Single-thread, n <= 15.
// work: 2*n*n doubles. Returns 0 on success.
// ---------------------------------------------------------------------------------------------
GF_HD int sqrt_info_from_cov(const double *cov, int n, double *out, double *work) {
  double *lu = work, *inv = work + n * n;
  int perm[15];
  for (int i = 0; i < n * n; i++) lu[i] = cov[i];
  for (int i = 0; i < n; i++) perm[i] = i;
  for (int k = 0; k < n; k++) {
    int piv = k;
    double best = fabs(lu[k * n + k]);
    for (int i = k + 1; i < n; i++) { const double a = fabs(lu[i * n + k]); if (a > best) { best = a; piv = i; } }
    if (best == 0.0) return 1;
    if (piv != k) {
      for (int j = 0; j < n; j++) { const double t = lu[k * n + j]; lu[k * n + j] = lu[piv * n + j]; lu[piv * n + j] = t; }
      const int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    for (int i = k + 1; i < n; i++) {
      lu[i * n + k] /= lu[k * n + k];
      const double f = lu[i * n + k];
      for (int j = k + 1; j < n; j++) lu[i * n + j] -= f * lu[k * n + j];
    }
  }
  for (int c = 0; c < n; c++) {
    double y[15];
    for (int i = 0; i < n; i++) {
      double s = (perm[i] == c) ? 1.0 : 0.0;
      for (int j = 0; j < i; j++) s -= lu[i * n + j] * y[j];
      y[i] = s;
    }
    for (int i = n - 1; i >= 0; i--) {
      double s = y[i];
      for (int j = i + 1; j < n; j++) s -= lu[i * n + j] * inv[j * n + c];
      inv[i * n + c] = s / lu[i * n + i];
    }
  }
  // lower Cholesky of inv, written transposed (upper) into out
  for (int i = 0; i < n * n; i++) out[i] = 0.0;
  for (int j = 0; j < n; j++) {
    double d = inv[j * n + j];
    for (int k = 0; k < j; k++) d -= out[k * n + j] * out[k * n + j];
    if (!(d > 0.0)) return 2;
    d = sqrt(d);
    out[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = inv[i * n + j];
      for (int k = 0; k < j; k++) s -= out[k * n + i] * out[k * n + j];
      out[j * n + i] = s / d;      // L(i,j) stored at out(j,i)
    }
  }
  return 0;
}

// es = element stride of the destination (1: dense row-major; B: the window-minor layout of k_dense_raw)
GF_HD void put3(double *A, int lda, int r0, int c0, const mat3 &B, size_t es = 1) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[(size_t)((r0 + i) * lda + c0 + j) * es] = B(i, j);
}
GF_HD mat3 get3(const double *A, int lda, int r0, int c0) {
  mat3 B; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) B(i, j) = A[(r0 + i) * lda + c0 + j]; return B;
}

// ---------------------------------------------------------------------------------------------
// IMU factor, un-whitened: raw[15] and (if Jraw) Jraw[15][30] (caller zero-fills), columns
// pose_i(6) sb_i(9) pose_j(6) sb_j(9). Whitening by sqrt_info is done by the caller (in parallel).
// ---------------------------------------------------------------------------------------------
// part / nparts: the 18 non-zero 3 x 3 blocks of the Jacobian are dealt over `nparts` callers (block b belongs to part b % nparts;
// part 0 also writes the residual): a small batch lets the four waves of a workgroup take a quarter of the blocks each — same
// expressions per block, so the values do not depend on the split.
GF_HD void imu_raw(const gfbe_imu_preint *pre, double g_norm, const double *pose_i, const double *sb_i,
                   const double *pose_j, const double *sb_j, double *raw, double *Jraw, size_t es = 1, int part = 0, int nparts = 1) {
  const vec3 Pi = ld3(pose_i), Pj = ld3(pose_j);
  const quat Qi = ldq(pose_i + 3), Qj = ldq(pose_j + 3);
  const vec3 Vi = ld3(sb_i), Bai = ld3(sb_i + 3), Bgi = ld3(sb_i + 6);
  const vec3 Vj = ld3(sb_j), Baj = ld3(sb_j + 3), Bgj = ld3(sb_j + 6);
  const double dt = pre->sum_dt;
  const mat3 dp_dba = get3(pre->jacobian, 15, 0, 9), dp_dbg = get3(pre->jacobian, 15, 0, 12);
  const mat3 dq_dbg = get3(pre->jacobian, 15, 3, 12);
  const mat3 dv_dba = get3(pre->jacobian, 15, 6, 9), dv_dbg = get3(pre->jacobian, 15, 6, 12);
  const vec3 dba = sub(Bai, ld3(pre->linearized_ba)), dbg = sub(Bgi, ld3(pre->linearized_bg));
  const quat dq = ldq(pre->delta_q);
  const quat cq = qmul(dq, small_rot(mv(dq_dbg, dbg)));
  const vec3 cv = add(ld3(pre->delta_v), add(mv(dv_dba, dba), mv(dv_dbg, dbg)));
  const vec3 cp = add(ld3(pre->delta_p), add(mv(dp_dba, dba), mv(dp_dbg, dbg)));
  const quat Qi_inv = qinv(Qi);
  const mat3 RiT = qrot(Qi_inv);
  const vec3 G = mk3(0.0, 0.0, g_norm);
  const vec3 a_p = mv(RiT, sub(add(scl(0.5 * dt * dt, G), sub(Pj, Pi)), scl(dt, Vi)));
  const vec3 a_v = mv(RiT, add(scl(dt, G), sub(Vj, Vi)));
  const vec3 rp = sub(a_p, cp);
  const vec3 rq = scl(2.0, qvec(qmul(qinv(cq), qmul(Qi_inv, Qj))));
  const vec3 rv = sub(a_v, cv);
  if (part == 0)
    for (int k = 0; k < 3; k++) {
      raw[k * es] = rp[k]; raw[(3 + k) * es] = rq[k]; raw[(6 + k) * es] = rv[k];
      raw[(9 + k) * es] = Baj[k] - Bai[k]; raw[(12 + k) * es] = Bgj[k] - Bgi[k];
    }
  if (!Jraw) return;
  const mat3 I = ident3();
  if (0 % nparts == part) put3(Jraw, 30, 0, 0, mneg(RiT), es);                                                  // imu_factor.h:107
  if (1 % nparts == part) put3(Jraw, 30, 0, 3, hat(a_p), es);                                                   // :108
  if (2 % nparts == part) put3(Jraw, 30, 3, 3, mneg(qleft_qright3(qmul(qinv(Qj), Qi), cq)), es);                // :113-114
  if (3 % nparts == part) put3(Jraw, 30, 6, 3, hat(a_v), es);                                                   // :117
  if (4 % nparts == part) put3(Jraw, 30, 0, 6, mscl(-dt, RiT), es);                                             // :133
  if (5 % nparts == part) put3(Jraw, 30, 0, 9, mneg(dp_dba), es);
  if (6 % nparts == part) put3(Jraw, 30, 0, 12, mneg(dp_dbg), es);
  if (7 % nparts == part) put3(Jraw, 30, 3, 12, mneg(mul(qleft3(qmul(qmul(qinv(Qj), Qi), dq)), dq_dbg)), es);   // :142 (uncorrected delta_q)
  if (8 % nparts == part) put3(Jraw, 30, 6, 6, mneg(RiT), es);
  if (9 % nparts == part) put3(Jraw, 30, 6, 9, mneg(dv_dba), es);
  if (10 % nparts == part) put3(Jraw, 30, 6, 12, mneg(dv_dbg), es);
  if (11 % nparts == part) put3(Jraw, 30, 9, 9, mneg(I), es);
  if (12 % nparts == part) put3(Jraw, 30, 12, 12, mneg(I), es);
  if (13 % nparts == part) put3(Jraw, 30, 0, 15, RiT, es);                                                       // :162
  if (14 % nparts == part) put3(Jraw, 30, 3, 18, qleft3(qmul(qmul(qinv(cq), Qi_inv), Qj)), es);                  // :168
  if (15 % nparts == part) put3(Jraw, 30, 6, 21, RiT, es);                                                       // :179
  if (16 % nparts == part) put3(Jraw, 30, 9, 24, I, es);
  if (17 % nparts == part) put3(Jraw, 30, 12, 27, I, es);
}

// ---------------------------------------------------------------------------------------------
// Wheel factor, un-whitened: raw[6], Jraw[6][22] (caller zero-fills), columns pose_i(6) pose_j(6)
// ex_wheel(6) sx sy sw td_wheel.
// ---------------------------------------------------------------------------------------------
// part / nparts (as imu_raw): the Jacobian's work is eight items (numbered 0-4, 6-8) dealt over `nparts` callers (item j belongs to part j % nparts; part 0
// also writes the residual) — every caller forms the residual, each item its own intermediates: the same expressions whatever the split.
GF_HD void wheel_raw(const gfbe_wheel_preint *pre, const double *pose_i, const double *pose_j, const double *exw,
                     double sx, double sy, double sw, double td, double *raw, double *Jraw, size_t es = 1, int part = 0, int nparts = 1) {
  const vec3 Pi = ld3(pose_i), Pj = ld3(pose_j), tio = ld3(exw);
  const quat Qi = ldq(pose_i + 3), Qj = ldq(pose_j + 3), qio = ldq(exw + 3);
  const double *Jm = pre->jacobian;   // 6x3
  const vec3 dp_dsx = mk3(Jm[0], Jm[3], Jm[6]), dp_dsy = mk3(Jm[1], Jm[4], Jm[7]), dp_dsw = mk3(Jm[2], Jm[5], Jm[8]);
  const vec3 dq_dsw = mk3(Jm[11], Jm[14], Jm[17]);
  const double dsx = sx - pre->linearized_sx, dsy = sy - pre->linearized_sy, dsw = sw - pre->linearized_sw;
  const mat3 sv = diagm(sx, sy, 1.0);
  const mat3 Ri = qrot(Qi), Rj = qrot(Qj), rio = qrot(qio);
  const vec3 lin_vel = ld3(pre->linearized_vel), lin_gyr = ld3(pre->linearized_gyr);
  const vec3 vel_1 = ld3(pre->vel_1), gyr_1 = ld3(pre->gyr_1);
  const vec3 cp = add(ld3(pre->delta_p), add(add(scl(dsx, dp_dsx), scl(dsy, dp_dsy)), scl(dsw, dp_dsw)));   // :201
  const quat cq = qmul(qnormalize(ldq(pre->delta_q)), so3exp(scl(dsw, dq_dsw)));                             // :202
  const double dtd = td - pre->linearized_td;
  const quat e_fw = so3exp(scl(sw * dtd, lin_gyr));
  const quat q_time = qmul(qmul(e_fw, cq), so3exp(scl(-sw * dtd, gyr_1)));                                    // :205
  const mat3 Rcq = qrot(cq);
  const vec3 p_time = mv(qrot(e_fw), sub(add(mv(sv, scl(dtd, lin_vel)), cp), mv(Rcq, mv(sv, scl(dtd, vel_1)))));   // :206
  const mat3 RR = mul(Ri, rio);
  const vec3 world_d = sub(sub(add(mv(Rj, tio), Pj), mv(Ri, tio)), Pi);
  const vec3 rp = sub(tmv(RR, world_d), p_time);                                                              // :211
  const vec3 rq = so3log(qnormalize(qmul(qmul(qmul(qinv(q_time), qinv(qmul(Qi, qio))), Qj), qio)));           // :212
  if (part == 0)
    for (int k = 0; k < 3; k++) { raw[k * es] = rp[k]; raw[(3 + k) * es] = rq[k]; }
  if (!Jraw) return;
#define GF_WHEEL_ITEM(j) ((j) % nparts == part)
  if (GF_WHEEL_ITEM(0)) {      // position rows of the sw column
    const vec3 drdsw = scl(dsw, dq_dsw);
    const mat3 Jr_drdsw = jr_so3(drdsw);                                   // wheel_factor.h:110-112
    const vec3 fw = scl(sw * dtd, lin_gyr), fv = mv(sv, scl(dtd, lin_vel));
    const vec3 bv = mv(sv, scl(dtd, vel_1));
    const mat3 Jrtd = jr_so3(fw);
    const mat3 Efw = qrot(so3exp(fw));
    const vec3 inner = sub(add(fv, cp), mv(Rcq, bv));
    const vec3 t1 = mv(Rcq, mv(hat(mv(Jr_drdsw, dq_dsw)), mv(sv, scl(dtd, vel_1))));
    const vec3 t2 = mv(hat(mv(Jrtd, scl(dtd, lin_gyr))), inner);
    const vec3 c_sw_p = neg(mv(Efw, add(sub(dp_dsw, t1), t2)));                                                  // :223
    for (int k = 0; k < 3; k++) Jraw[(size_t)(k * 22 + 20) * es] = c_sw_p[k];
  }
  if (GF_WHEEL_ITEM(1)) {      // rotation rows of the sw column
    const mat3 Jr_inv = jr_inv_so3(rq);                                    // :106-108
    const vec3 drdsw = scl(dsw, dq_dsw);
    const mat3 Jr_drdsw = jr_so3(drdsw);
    const vec3 fw = scl(sw * dtd, lin_gyr), bw = scl(sw * dtd, gyr_1);
    const mat3 Jrtd = jr_so3(fw);
    const mat3 Emr = qrot(so3exp(neg(rq))), Ebw = qrot(so3exp(bw)), RcqT = qrot(qinv(cq));
    const vec3 u1 = add(mv(RcqT, mv(Jrtd, scl(dtd, lin_gyr))), mv(Jr_drdsw, dq_dsw));
    const vec3 c_sw_r = neg(mv(Jr_inv, mv(Emr, mv(Ebw, u1))));                                                   // :225
    for (int k = 0; k < 3; k++) Jraw[(size_t)((3 + k) * 22 + 20) * es] = c_sw_r[k];
  }
  if (GF_WHEEL_ITEM(2)) {      // position rows of the td_wheel column
    const vec3 fw = scl(sw * dtd, lin_gyr), fv = mv(sv, scl(dtd, lin_vel));
    const vec3 bv = mv(sv, scl(dtd, vel_1));
    const mat3 Jrtd = jr_so3(fw);
    const mat3 Efw = qrot(so3exp(fw));
    const vec3 inner = sub(add(fv, cp), mv(Rcq, bv));
    const vec3 t3 = mv(hat(mv(Jrtd, scl(sw, lin_gyr))), inner);
    const vec3 c_td_p = neg(mv(Efw, add(sub(mv(sv, lin_vel), mv(Rcq, mv(sv, vel_1))), t3)));                     // :236
    for (int k = 0; k < 3; k++) Jraw[(size_t)(k * 22 + 21) * es] = c_td_p[k];
  }
  if (GF_WHEEL_ITEM(3)) {      // rotation rows of the td_wheel column
    const mat3 Jr_inv = jr_inv_so3(rq);
    const vec3 fw = scl(sw * dtd, lin_gyr), bw = scl(sw * dtd, gyr_1);
    const mat3 Jrtd = jr_so3(fw), Jr_mtd = jr_so3(neg(fw));
    const mat3 Emr = qrot(so3exp(neg(rq))), Ebw = qrot(so3exp(bw)), RcqT = qrot(qinv(cq));
    const vec3 u2 = sub(mv(Ebw, mv(RcqT, mv(Jrtd, scl(sw, lin_gyr)))), mv(Jr_mtd, scl(sw, gyr_1)));
    const vec3 c_td_r = neg(mv(Jr_inv, mv(Emr, u2)));                                                            // :237
    for (int k = 0; k < 3; k++) Jraw[(size_t)((3 + k) * 22 + 21) * es] = c_td_r[k];
  }
  if (GF_WHEEL_ITEM(4)) {      // position rows of pose_i and of pose_j's translation
    const mat3 RRT = transp(RR);
    put3(Jraw, 22, 0, 0, mneg(RRT), es);                                                                            // :121
    put3(Jraw, 22, 0, 3, madd(mul(RRT, mul(Ri, hat(tio))), tmul(rio, hat(tmv(Ri, world_d)))), es);                  // :123
    put3(Jraw, 22, 0, 6, RRT, es);                                                                                  // :150
  }
  if (GF_WHEEL_ITEM(8)) {      // the sx, sy columns (item 8, not 5: with four callers it goes with the lightest share — measured, profiles/r5_lin_small_phases.txt)
    const vec3 fv = mv(sv, scl(dtd, lin_vel));
    const mat3 Efv = qrot(so3exp(fv));
    const mat3 I1 = diagm(1.0, 0.0, 0.0), I2 = diagm(0.0, 1.0, 0.0);
    const vec3 c_sx = neg(mv(Efv, sub(add(mv(I1, scl(dtd, lin_vel)), dp_dsx), mv(Rcq, mv(I1, scl(dtd, vel_1))))));   // :199
    const vec3 c_sy = neg(mv(Efv, sub(add(mv(I2, scl(dtd, lin_vel)), dp_dsy), mv(Rcq, mv(I2, scl(dtd, vel_1))))));   // :211
    for (int k = 0; k < 3; k++) { Jraw[(size_t)(k * 22 + 18) * es] = c_sx[k]; Jraw[(size_t)(k * 22 + 19) * es] = c_sy[k]; }
  }
  if (GF_WHEEL_ITEM(6)) {      // rotation rows of the pose and extrinsic blocks
    const mat3 Jr_inv = jr_inv_so3(rq);
    put3(Jraw, 22, 3, 3, mneg(mul(Jr_inv, qrot(qmul(qinv(qmul(Qj, qio)), Qi)))), es);                               // :131
    put3(Jraw, 22, 3, 9, mul(Jr_inv, qrot(qinv(qio))), es);                                                         // :157
    put3(Jraw, 22, 3, 15, mul(Jr_inv, msub(ident3(), qrot(qmul(qmul(qinv(qmul(Qj, qio)), Qi), qio)))), es);         // :174
  }
  if (GF_WHEEL_ITEM(7)) {      // position rows of pose_j's rotation and of the extrinsic
    const mat3 RRT = transp(RR);
    put3(Jraw, 22, 0, 9, mneg(mul(qrot(qmul(qinv(qmul(Qi, qio)), Qj)), hat(tio))), es);                             // :151
    put3(Jraw, 22, 0, 12, mul(RRT, msub(Rj, Ri)), es);                                                              // :170
    put3(Jraw, 22, 0, 15, hat(mv(RRT, world_d)), es);                                                               // :172
  }
#undef GF_WHEEL_ITEM
}

// Prior: tangent difference of one kept block w.r.t. its linearisation point
// (marginalization_factor.cpp:359-374). size 7 -> 6 outputs, otherwise `size` outputs.
GF_HD void prior_block_dx(const double *x, const double *x0, int size, double *dx) {
  if (size != 7) {
    for (int k = 0; k < size; k++) dx[k] = x[k] - x0[k];
    return;
  }
  for (int k = 0; k < 3; k++) dx[k] = x[k] - x0[k];
  const quat d = qmul(qinv(ldq(x0 + 3)), ldq(x + 3));
  const double sgn = (d.w >= 0.0) ? 2.0 : -2.0;
  dx[3] = sgn * d.x; dx[4] = sgn * d.y; dx[5] = sgn * d.z;
}

// PoseLocalParameterization::Plus with an optional PoseSubsetParameterization mask
// (pose_local_parameterization.cpp:12-27, pose_subset_parameterization.cpp:27-55).
GF_HD void pose_plus(const double *x, const double *d6, const unsigned char *mask6, double *y) {
  double d[6];
  for (int k = 0; k < 6; k++) d[k] = (mask6 && mask6[k]) ? 0.0 : d6[k];
  for (int k = 0; k < 3; k++) y[k] = x[k] + d[k];
  const quat q = qnormalize(qmul(ldq(x + 3), small_rot(mk3(d[3], d[4], d[5]))));
  y[3] = q.x; y[4] = q.y; y[5] = q.z; y[6] = q.w;
}

// OrientationSubsetParameterization::Plus (orientation_subset_parameterization.cpp:27-45): para_plane_R uses constant = {2}.
GF_HD void orientation_plus(const double *q4, const double *d3, const unsigned char *constant3, double *y4) {
  const vec3 dv = mk3(constant3[0] ? 0.0 : d3[0], constant3[1] ? 0.0 : d3[1], constant3[2] ? 0.0 : d3[2]);
  const quat q = qnormalize(qmul(ldq(q4), small_rot(dv)));
  y4[0] = q.x; y4[1] = q.y; y4[2] = q.z; y4[3] = q.w;
}

// PlaneFactor::Evaluate (plane_factor.h:25-122): r[3]; J[3][16] tangent columns pose_i (6) ex_wheel (6) plane_R (3) plane_Z (1),
// J may be null. (Roll / pitch of the ground normal seen from the wheel odometer frame, height of the odometer over the plane.)
GF_HD void plane_eval(const double *pose, const double *ex, const double *plane_q, double plane_z, const double *ninv, double *r, double *J) {
  const vec3 tio = ld3(ex), Pi = ld3(pose);
  const mat3 Rio = qrot(ldq(ex + 3)), Rpw = qrot(ldq(plane_q)), Ri = qrot(ldq(pose + 3));
  const vec3 up_p = tmv(Rpw, mk3(0, 0, 1)), up_b = tmv(Ri, up_p), up_o = tmv(Rio, up_b);
  const vec3 lever = add(Pi, mv(Ri, tio));
  r[0] = ninv[0] * up_o[0]; r[1] = ninv[1] * up_o[1]; r[2] = ninv[2] * (plane_z + mv(Rpw, lever)[2]);
  if (!J) return;
  for (int i = 0; i < 48; i++) J[i] = 0.0;
  const mat3 A = tmul(Rio, hat(up_b)), Bm = hat(up_o), Cq = tmul(Rio, tmul(Ri, hat(up_p)));
  const mat3 RpwRi = mul(Rpw, Ri), D = mul(RpwRi, hat(tio)), E = mul(Rpw, hat(lever));
  for (int row = 0; row < 2; row++)
    for (int j = 0; j < 3; j++) {
      J[row * 16 + 3 + j] = ninv[row] * A(row, j);
      J[row * 16 + 9 + j] = ninv[row] * Bm(row, j);
      J[row * 16 + 12 + j] = ninv[row] * Cq(row, j);
    }
  for (int j = 0; j < 3; j++) {
    J[32 + j] = ninv[2] * Rpw(2, j);
    J[32 + 3 + j] = -ninv[2] * D(2, j);
    J[32 + 6 + j] = ninv[2] * RpwRi(2, j);
    J[32 + 12 + j] = -ninv[2] * E(2, j);
  }
  J[32 + 15] = ninv[2];
}

// PoseAnchorFactor::Evaluate (pose_anchor_factor.cpp:8-32): r[6]; J[6][6] (may be null). The reference scales the WHOLE Jacobian by
// 2 sqrt_info (:29) — the position block is twice the derivative of its own residual — and builds the rotation block from the
// anchor alone: reproduced as it is.
GF_HD void anchor_eval(const double *x, const double *a, double sqrt_info, double *r, double *J) {
  const quat qa_inv = qinv(ldq(a + 3));
  const vec3 dv = qvec(qmul(ldq(x + 3), qa_inv));
  for (int i = 0; i < 3; i++) { r[i] = sqrt_info * (x[i] - a[i]); r[3 + i] = sqrt_info * 2.0 * dv[i]; }
  if (!J) return;
  for (int i = 0; i < 36; i++) J[i] = 0.0;
  const double s = 2.0 * sqrt_info;
  for (int i = 0; i < 3; i++) J[i * 6 + i] = s;
  const mat3 Jq = qright3(qa_inv);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) J[(3 + i) * 6 + 3 + j] = s * Jq(i, j);
}

}  // namespace gfd

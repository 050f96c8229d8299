// gfbe_preint.hip — mid-point pre-integration of IMU and wheel streams on the device.
// Reference (Ground-Fusion++/vins_estimator/src/factor):
//   IntegrationBase::{push_back, propagate, midPointIntegration}       integration_base.h:39-167
//   WheelIntegrationBase::{push_back, propagate, midPointIntegration}  wheel_integration_base.h:41-178
// One workgroup per interval (IMU: four waves — a thread per entry of the 15 x 15 products; one robot integrates ONE interval per
// camera frame, so the kernel's time is the latency of its sample recursion: 64 threads took ~4 us per sample; wheel: one wave): the
// sample recursion is sequential; inside a sample the lanes
// own entries of F*jacobian and F*cov*F^T + V*noise*V^T (the reference builds dynamic MatrixXd
// temporaries for these per sample, integration_base.h:99,117).
#include "gfbe_device.h"
#include "gfbe_factors.h"

namespace gfd {

// Which of the 17 distinct 3 x 3 blocks sits at block (row / 3, column / 3) of F (5 x 5) and V (5 x 6); -1: zero.
#define IMU_NSRC 17
#define PREINT_IMU_THREADS 256
__constant__ signed char IMU_FMAP[25] = {0, 1, 2, 3, 4,   -1, 5, -1, -1, 6,   -1, 7, 0, 8, 9,   -1, -1, -1, 0, -1,   -1, -1, -1, -1, 0};
__constant__ signed char IMU_VMAP[30] = {10, 11, 12, 11, -1, -1,   -1, 13, -1, 13, -1, -1,   14, 15, 16, 15, -1, -1,   -1, -1, -1, -1, 2, -1,   -1, -1, -1, -1, -1, 2};

__global__ __launch_bounds__(PREINT_IMU_THREADS) void k_preint_imu(int n, const int *off, const double *samples, const double *first,
                                                  const double *lin, const double *noise, gfbe_imu_preint *out) {
  const int iv = blockIdx.x, t = threadIdx.x;
  if (iv >= n) return;
  __shared__ double F[225], V[15 * 18], Jm[225], P[225], T1[225], T2[225], N[18], src[IMU_NSRC * 9];
  __shared__ signed char fmap[25], vmap[30];
  vec3 acc_0 = ld3(first + 6 * iv), gyr_0 = ld3(first + 6 * iv + 3);
  const vec3 ba = ld3(lin + 6 * iv), bg = ld3(lin + 6 * iv + 3);
  vec3 dp = mk3(0, 0, 0), dv = mk3(0, 0, 0);
  quat dq; dq.x = dq.y = dq.z = 0.0; dq.w = 1.0;
  double sum_dt = 0.0;
  for (int e = t; e < 225; e += PREINT_IMU_THREADS) { Jm[e] = (e / 15 == e % 15) ? 1.0 : 0.0; P[e] = 0.0; }
  if (t < 25) fmap[t] = IMU_FMAP[t];
  if (t < 30) vmap[t] = IMU_VMAP[t];
  if (t < 18) {
    const int g = t / 3;   // ACC_N GYR_N ACC_N GYR_N ACC_W GYR_W  (integration_base.h:30-36)
    const double sdev = (g == 0 || g == 2) ? noise[0] : (g == 1 || g == 3) ? noise[1] : (g == 4 ? noise[2] : noise[3]);
    N[t] = sdev * sdev;
  }
  __syncthreads();
  for (int s = off[iv]; s < off[iv + 1]; s++) {
    const double dt = samples[7 * (size_t)s];
    const vec3 acc_1 = ld3(samples + 7 * (size_t)s + 1), gyr_1 = ld3(samples + 7 * (size_t)s + 4);
    // state recursion (every lane redundantly; integration_base.h:72-80)
    const mat3 Rd = qrot(dq);
    const vec3 un_acc_0 = mv(Rd, sub(acc_0, ba));
    const vec3 un_gyr = sub(scl(0.5, add(gyr_0, gyr_1)), bg);
    quat hq; hq.x = un_gyr[0] * dt / 2; hq.y = un_gyr[1] * dt / 2; hq.z = un_gyr[2] * dt / 2; hq.w = 1.0;
    const quat rq = qmul(dq, hq);
    const mat3 Rr = qrot(rq);
    const vec3 un_acc_1 = mv(Rr, sub(acc_1, ba));
    const vec3 un_acc = scl(0.5, add(un_acc_0, un_acc_1));
    const vec3 rp = add(add(dp, scl(dt, dv)), scl(0.5 * dt * dt, un_acc));
    const vec3 rv = add(dv, scl(dt, un_acc));
    {   // F (15x15) and V (15x18), integration_base.h:83-129: the 17 distinct 3 x 3 blocks are formed by every lane (same
        // instructions, no extra time), lanes 0..8 publish one entry of each, then all lanes fill F and V through the block maps
      const mat3 Rw = hat(un_gyr), Ra0 = hat(sub(acc_0, ba)), Ra1 = hat(sub(acc_1, ba)), I = ident3();
      const mat3 ImW = msub(I, mscl(dt, Rw));
      const mat3 RrA1 = mul(Rr, Ra1);
      const mat3 v03 = mscl(0.25 * dt * dt * 0.5 * dt, mneg(RrA1)), v63 = mscl(0.5 * dt * 0.5 * dt, mneg(RrA1));
      const mat3 blk[IMU_NSRC] = {
          I,                                                                                                   //  0
          madd(mscl(-0.25 * dt * dt, mul(Rd, Ra0)), mscl(-0.25 * dt * dt, mul(RrA1, ImW))),                    //  1  F(0,3)
          mscl(dt, I),                                                                                         //  2
          mscl(-0.25 * dt * dt, madd(Rd, Rr)),                                                                 //  3  F(0,9)
          mscl(-0.25 * dt * dt * -dt, RrA1),                                                                   //  4  F(0,12)
          ImW,                                                                                                 //  5  F(3,3)
          mscl(-dt, I),                                                                                        //  6  F(3,12)
          madd(mscl(-0.5 * dt, mul(Rd, Ra0)), mscl(-0.5 * dt, mul(RrA1, ImW))),                                //  7  F(6,3)
          mscl(-0.5 * dt, madd(Rd, Rr)),                                                                       //  8  F(6,9)
          mscl(-0.5 * dt * -dt, RrA1),                                                                         //  9  F(6,12)
          mscl(0.25 * dt * dt, Rd), v03, mscl(0.25 * dt * dt, Rr),                                             // 10 11 12  V row 0
          mscl(0.5 * dt, I),                                                                                   // 13
          mscl(0.5 * dt, Rd), v63, mscl(0.5 * dt, Rr)};                                                        // 14 15 16  V row 6
      if (t < 9) {
#pragma unroll
        for (int m = 0; m < IMU_NSRC; m++) {
          double v = blk[m].m[0];
#pragma unroll
          for (int k = 1; k < 9; k++) v = (t == k) ? blk[m].m[k] : v;
          src[m * 9 + t] = v;
        }
      }
      __syncthreads();
      for (int e = t; e < 225 + 270; e += PREINT_IMU_THREADS) {
        const bool isF = e < 225;
        const int q = isF ? e : e - 225, ld = isF ? 15 : 18, r = q / ld, c = q - r * ld;
        const int m = isF ? fmap[(r / 3) * 5 + c / 3] : vmap[(r / 3) * 6 + c / 3];
        const double v = m >= 0 ? src[m * 9 + (r % 3) * 3 + c % 3] : 0.0;
        if (isF) F[q] = v; else V[q] = v;
      }
    }
    __syncthreads();
    for (int e = t; e < 225; e += PREINT_IMU_THREADS) {     // T1 = F * jacobian, T2 = F * cov
      const int i = e / 15, j = e % 15;
      double a = 0.0, b = 0.0;
      for (int k = 0; k < 15; k++) { a += F[i * 15 + k] * Jm[k * 15 + j]; b += F[i * 15 + k] * P[k * 15 + j]; }
      T1[e] = a; T2[e] = b;
    }
    __syncthreads();
    for (int e = t; e < 225; e += PREINT_IMU_THREADS) {     // cov = T2 * F^T + V N V^T
      const int i = e / 15, j = e % 15;
      double a = 0.0;
      for (int k = 0; k < 15; k++) a += T2[i * 15 + k] * F[j * 15 + k];
      double b = 0.0;
      for (int k = 0; k < 18; k++) b += (V[i * 18 + k] * N[k]) * V[j * 18 + k];
      P[e] = a + b;
      Jm[e] = T1[e];
    }
    __syncthreads();
    dp = rp; dv = rv; dq = qnormalize(rq);   // propagate(), :152-166
    sum_dt += dt;
    acc_0 = acc_1; gyr_0 = gyr_1;
  }
  gfbe_imu_preint *o = out + iv;
  if (t == 0) {
    o->sum_dt = sum_dt;
    for (int k = 0; k < 3; k++) { o->delta_p[k] = dp[k]; o->delta_v[k] = dv[k]; o->linearized_ba[k] = ba[k]; o->linearized_bg[k] = bg[k]; }
    o->delta_q[0] = dq.x; o->delta_q[1] = dq.y; o->delta_q[2] = dq.z; o->delta_q[3] = dq.w;
  }
  for (int e = t; e < 225; e += PREINT_IMU_THREADS) { o->jacobian[e] = Jm[e]; o->covariance[e] = P[e]; }
}

__global__ __launch_bounds__(64) void k_preint_wheel(int n, const int *off, const double *samples, const double *first,
                                                    const double *lin, const double *noise, gfbe_wheel_preint *out) {
  const int iv = blockIdx.x, t = threadIdx.x;
  if (iv >= n) return;
  __shared__ double F[36], V[72], P[36], T2[36], Jm[18], N[12];
  vec3 vel_0 = ld3(first + 6 * iv), gyr_0 = ld3(first + 6 * iv + 3);
  const vec3 lin_vel = vel_0, lin_gyr = gyr_0;
  const double lsx = lin[4 * iv], lsy = lin[4 * iv + 1], lsw = lin[4 * iv + 2], ltd = lin[4 * iv + 3];
  const mat3 sv = diagm(lsx, lsy, 1.0);
  vec3 dp = mk3(0, 0, 0), vel_1 = vel_0, gyr_1 = gyr_0;
  quat dq; dq.x = dq.y = dq.z = 0.0; dq.w = 1.0;
  double sum_dt = 0.0;
  if (t < 36) P[t] = 0.0;
  if (t < 18) Jm[t] = 0.0;
  if (t < 12) { const double sdev = ((t / 3) % 2 == 0) ? noise[0] : noise[1]; N[t] = sdev * sdev; }   // wheel_integration_base.h:32-36
  __syncthreads();
  for (int s = off[iv]; s < off[iv + 1]; s++) {
    const double dt = samples[7 * (size_t)s];
    vel_1 = ld3(samples + 7 * (size_t)s + 1); gyr_1 = ld3(samples + 7 * (size_t)s + 4);
    const mat3 Rd = qrot(dq);
    const vec3 un_vel_0 = mv(Rd, mv(sv, vel_0));
    const vec3 un_gyr = scl(0.5 * lsw, add(gyr_0, gyr_1));
    quat ddq; ddq.x = un_gyr[0] * dt / 2; ddq.y = un_gyr[1] * dt / 2; ddq.z = un_gyr[2] * dt / 2; ddq.w = 1.0;
    const quat rq = qmul(dq, ddq);
    const mat3 Rr = qrot(rq);
    const vec3 un_vel_1 = mv(Rr, mv(sv, vel_1));
    const vec3 rp = add(dp, scl(0.5 * dt, add(un_vel_0, un_vel_1)));
    if (t == 0) {   // wheel_integration_base.h:95-139
      for (int e = 0; e < 36; e++) F[e] = 0.0;
      for (int e = 0; e < 72; e++) V[e] = 0.0;
      const mat3 Rv0 = hat(mv(sv, vel_0)), Rv1 = hat(mv(sv, vel_1)), Rdd = qrot(ddq), I = ident3();
      put3(F, 6, 0, 0, I);
      put3(F, 6, 0, 3, mscl(-0.5 * dt, madd(mul(Rd, Rv0), mul(mul(Rr, Rv1), transp(Rdd)))));
      put3(F, 6, 3, 3, transp(Rdd));
      const mat3 Jr = jr_so3(scl(dt, un_gyr));
      const mat3 v03 = mscl(-0.25 * dt * dt, mul(mul(Rr, Rv1), Jr));
      put3(V, 12, 0, 0, mscl(0.5 * dt, mul(Rd, sv)));
      put3(V, 12, 0, 3, v03);
      put3(V, 12, 0, 6, mscl(0.5 * dt, mul(Rr, sv)));
      put3(V, 12, 0, 9, v03);
      put3(V, 12, 3, 3, mscl(0.5 * lsw * dt, Jr));
      put3(V, 12, 3, 9, mscl(0.5 * lsw * dt, Jr));
      const mat3 I1 = diagm(1, 0, 0), I2 = diagm(0, 1, 0);
      const vec3 a0 = scl(0.5 * dt, add(mv(Rd, mv(I1, vel_0)), mv(Rr, mv(I1, vel_1))));
      const vec3 a1 = scl(0.5 * dt, add(mv(Rd, mv(I2, vel_0)), mv(Rr, mv(I2, vel_1))));
      const vec3 last = mk3(Jm[3 * 3 + 2], Jm[4 * 3 + 2], Jm[5 * 3 + 2]);
      const vec3 cur = add(last, mv(Jr, scl(0.5 * dt, add(gyr_0, gyr_1))));
      const vec3 a2 = scl(0.5 * dt, add(mv(Rd, mv(hat(last), mv(sv, vel_0))), mv(Rr, mv(hat(cur), mv(sv, vel_1)))));
      for (int k = 0; k < 3; k++) { Jm[k * 3 + 0] += a0[k]; Jm[k * 3 + 1] += a1[k]; Jm[k * 3 + 2] += a2[k]; Jm[(3 + k) * 3 + 2] = cur[k]; }
    }
    __syncthreads();
    if (t < 36) { const int i = t / 6, j = t % 6; double a = 0.0; for (int k = 0; k < 6; k++) a += F[i * 6 + k] * P[k * 6 + j]; T2[t] = a; }
    __syncthreads();
    if (t < 36) {
      const int i = t / 6, j = t % 6;
      double a = 0.0, b = 0.0;
      for (int k = 0; k < 6; k++) a += T2[i * 6 + k] * F[j * 6 + k];
      for (int k = 0; k < 12; k++) b += (V[i * 12 + k] * N[k]) * V[j * 12 + k];
      P[t] = a + b;
    }
    __syncthreads();
    dp = rp; dq = qnormalize(rq);
    sum_dt += dt;
    vel_0 = vel_1; gyr_0 = gyr_1;
  }
  gfbe_wheel_preint *o = out + iv;
  if (t == 0) {
    o->sum_dt = sum_dt;
    for (int k = 0; k < 3; k++) { o->delta_p[k] = dp[k]; o->linearized_vel[k] = lin_vel[k]; o->linearized_gyr[k] = lin_gyr[k]; o->vel_1[k] = vel_1[k]; o->gyr_1[k] = gyr_1[k]; }
    o->delta_q[0] = dq.x; o->delta_q[1] = dq.y; o->delta_q[2] = dq.z; o->delta_q[3] = dq.w;
    o->linearized_sx = lsx; o->linearized_sy = lsy; o->linearized_sw = lsw; o->linearized_td = ltd;
  }
  if (t < 18) o->jacobian[t] = Jm[t];
  if (t < 36) o->covariance[t] = P[t];
}

void launch_preint_imu(int n, const int *off, const double *samples, const double *first, const double *lin,
                       const double *noise4, gfbe_imu_preint *out, hipStream_t s) {
  hipLaunchKernelGGL(k_preint_imu, dim3(n), dim3(PREINT_IMU_THREADS), 0, s, n, off, samples, first, lin, noise4, out);
}
void launch_preint_wheel(int n, const int *off, const double *samples, const double *first, const double *lin,
                         const double *noise2, gfbe_wheel_preint *out, hipStream_t s) {
  hipLaunchKernelGGL(k_preint_wheel, dim3(n), dim3(64), 0, s, n, off, samples, first, lin, noise2, out);
}

}  // namespace gfd

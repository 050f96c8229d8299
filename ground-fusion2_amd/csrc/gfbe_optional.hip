// gfbe_optional.hip — the optional in-window factors on the device, evaluation only:
//   PlaneFactor::Evaluate        factor/plane_factor.h:25-122         (3-D: roll / pitch of the ground normal in the wheel
//                                                                      odometer frame, height of the odometer above the plane)
//   PoseAnchorFactor::Evaluate   factor/pose_anchor_factor.cpp:8-32   (6-D, pins para_Pose[0] when GNSS starts)
//   OrientationSubsetParameterization::Plus   factor/orientation_subset_parameterization.cpp:27-45 (host)
// One thread per factor; tangent-space Jacobians (the leading 6 / 3 columns of every block, as the [I; 0] ComputeJacobian
// of the parameterizations gives). Stand-alone evaluation in batches of any size; the same device functions (plane_eval,
// anchor_eval in gfbe_factors.h) run inside the window solve when gfbe_window.use_plane / use_anchor are set.
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "gfbe_device.h"
#include "gfbe_factors.h"

using namespace gfd;

namespace {

struct PlaneConst { double ex[7], q[4], z, ninv[3]; };

__global__ __launch_bounds__(256) void k_plane(int n, const double *pose, PlaneConst pc, double *r, double *J, double *cost_part) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) {
    double res[3], Jl[48];
    plane_eval(pose + 7 * (size_t)k, pc.ex, pc.q, pc.z, pc.ninv, res, J ? Jl : nullptr);
    double c = 0.5 * res[0] * res[0];
    c += 0.5 * res[1] * res[1];
    c += 0.5 * res[2] * res[2];
    if (r) for (int i = 0; i < 3; i++) r[3 * (size_t)k + i] = res[i];
    if (J) for (int i = 0; i < 48; i++) J[(size_t)k * 48 + i] = Jl[i];
    cost_part[k] = c;       // summed in factor order on the host
  }
}

__global__ __launch_bounds__(256) void k_anchor(int n, const double *pose, const double *anchor, double sqrt_info, double *r, double *J,
                                                double *cost_part) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  double res[6], Jl[36];
  anchor_eval(pose + 7 * (size_t)k, anchor + 7 * (size_t)k, sqrt_info, res, J ? Jl : nullptr);
  double c = 0.0;
  for (int i = 0; i < 6; i++) c += 0.5 * res[i] * res[i];
  if (r) for (int i = 0; i < 6; i++) r[6 * (size_t)k + i] = res[i];
  if (J) for (int i = 0; i < 36; i++) J[(size_t)k * 36 + i] = Jl[i];
  cost_part[k] = c;
}

#define OP_CHECK(c, call)                                                                                      \
  do {                                                                                                         \
    hipError_t e_ = (call);                                                                                    \
    if (e_ != hipSuccess) { ctx_set_error(c, (std::string(#call) + ": " + hipGetErrorString(e_)).c_str()); st = GFBE_DEVICE_ERROR; goto done; } \
  } while (0)

}  // namespace

extern "C" gfbe_status gfbe_plane_eval(gfbe_ctx *c, int32_t n, const double *pose, const double *ex_wheel, const double *plane_R,
                                       double plane_Z, const double *noise_inv, double *r, double *J, double *cost) {
  if (!c || n < 0 || (n > 0 && !pose) || !ex_wheel || !plane_R || !noise_inv) return GFBE_BAD_INPUT;
  if (ctx_device(c) < 0) return GFBE_NO_DEVICE;
  if (cost) *cost = 0.0;
  if (n == 0) return GFBE_OK;
  hipStream_t s = ctx_stream(c);
  gfbe_status st = GFBE_OK;
  double *d = nullptr;
  const size_t np = (size_t)7 * n, nr = (size_t)3 * n, nj = (size_t)48 * n, nc = (size_t)n;
  std::vector<double> hc(nc);
  PlaneConst pc;
  for (int i = 0; i < 7; i++) pc.ex[i] = ex_wheel[i];
  for (int i = 0; i < 4; i++) pc.q[i] = plane_R[i];
  for (int i = 0; i < 3; i++) pc.ninv[i] = noise_inv[i];
  pc.z = plane_Z;
  OP_CHECK(c, hipMalloc((void **)&d, sizeof(double) * (np + nr + nj + nc)));
  OP_CHECK(c, hipMemcpyAsync(d, pose, sizeof(double) * np, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_plane, dim3((n + 255) / 256), dim3(256), 0, s, n, d, pc, r ? d + np : nullptr, J ? d + np + nr : nullptr, d + np + nr + nj);
  OP_CHECK(c, hipGetLastError());
  if (r) OP_CHECK(c, hipMemcpyAsync(r, d + np, sizeof(double) * nr, hipMemcpyDeviceToHost, s));
  if (J) OP_CHECK(c, hipMemcpyAsync(J, d + np + nr, sizeof(double) * nj, hipMemcpyDeviceToHost, s));
  OP_CHECK(c, hipMemcpyAsync(hc.data(), d + np + nr + nj, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  OP_CHECK(c, hipStreamSynchronize(s));
  if (cost) { double t = 0.0; for (int k = 0; k < n; k++) t += hc[k]; *cost = t; }
done:
  if (d) (void)hipFree(d);
  return st;
}

extern "C" gfbe_status gfbe_anchor_eval(gfbe_ctx *c, int32_t n, const double *pose, const double *anchor, double sqrt_info, double *r,
                                        double *J, double *cost) {
  if (!c || n < 0 || (n > 0 && (!pose || !anchor))) return GFBE_BAD_INPUT;
  if (ctx_device(c) < 0) return GFBE_NO_DEVICE;
  if (cost) *cost = 0.0;
  if (n == 0) return GFBE_OK;
  hipStream_t s = ctx_stream(c);
  gfbe_status st = GFBE_OK;
  double *d = nullptr;
  const size_t np = (size_t)7 * n, nr = (size_t)6 * n, nj = (size_t)36 * n;
  std::vector<double> hc(n);
  OP_CHECK(c, hipMalloc((void **)&d, sizeof(double) * (2 * np + nr + nj + n)));
  OP_CHECK(c, hipMemcpyAsync(d, pose, sizeof(double) * np, hipMemcpyHostToDevice, s));
  OP_CHECK(c, hipMemcpyAsync(d + np, anchor, sizeof(double) * np, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_anchor, dim3((n + 255) / 256), dim3(256), 0, s, n, d, d + np, sqrt_info, r ? d + 2 * np : nullptr,
                     J ? d + 2 * np + nr : nullptr, d + 2 * np + nr + nj);
  OP_CHECK(c, hipGetLastError());
  if (r) OP_CHECK(c, hipMemcpyAsync(r, d + 2 * np, sizeof(double) * nr, hipMemcpyDeviceToHost, s));
  if (J) OP_CHECK(c, hipMemcpyAsync(J, d + 2 * np + nr, sizeof(double) * nj, hipMemcpyDeviceToHost, s));
  OP_CHECK(c, hipMemcpyAsync(hc.data(), d + 2 * np + nr + nj, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  OP_CHECK(c, hipStreamSynchronize(s));
  if (cost) { double t = 0.0; for (int k = 0; k < n; k++) t += hc[k]; *cost = t; }
done:
  if (d) (void)hipFree(d);
  return st;
}

extern "C" void gfbe_orientation_subset_plus(const double *q, const double *delta, const uint8_t *constant, double *out) {
  const vec3 d = mk3(constant[0] ? 0.0 : delta[0], constant[1] ? 0.0 : delta[1], constant[2] ? 0.0 : delta[2]);
  const quat p = qnormalize(qmul(ldq(q), small_rot(d)));
  out[0] = p.x; out[1] = p.y; out[2] = p.z; out[3] = p.w;
}

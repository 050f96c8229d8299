// gfbe_gnss.hip — the GNSS factors of the window on the device, evaluation only:
//   GnssPsrDoppFactor / DtDdtFactor / DdtSmoothFactor as estimator.cpp:3239-3291 adds them. One thread per factor; the arithmetic
//   is in gfbe_gnss.h. Not part of the window solve: see include/gfbe.h (f2) for why.
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "gfbe_device.h"
#include "gfbe_gnss.h"

using namespace gfd;

namespace {

struct GnssArgs {
  int n_obs, has_iono;
  double iono[8];
  double dt[GFBE_WINDOW_SIZE];
  double ddt_weight;
};

// threads [0, n_obs): pseudo-range / Doppler factors; [n_obs, n_obs + 40): DtDdt (k-major); [n_obs + 40, n_obs + 50): DdtSmooth
__global__ __launch_bounds__(128) void k_gnss(GnssArgs a, const gfbe_gnss_obs *obs, const gfbe_state *st, const gfbe_gnss_state *g, double *r_obs,
                                              double *J_obs, double *r_clk, double *cost_part) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int W = GFBE_WINDOW_SIZE;
  if (k < a.n_obs) {
    const gfbe_gnss_obs o = obs[k];
    double r[2], J[36];
    gnss_psr_dopp_eval(o, a.has_iono ? a.iono : nullptr, st->para_Pose[o.lower_idx], st->para_SpeedBias[o.lower_idx], st->para_Pose[o.lower_idx + 1],
                       st->para_SpeedBias[o.lower_idx + 1], g->rcv_dt[o.frame][o.sys_idx], g->rcv_ddt[o.frame], g->yaw_enu_local, g->anc_ecef, r,
                       J_obs ? J : nullptr);
    if (r_obs) { r_obs[2 * (size_t)k] = r[0]; r_obs[2 * (size_t)k + 1] = r[1]; }
    if (J_obs) for (int q = 0; q < 36; q++) J_obs[36 * (size_t)k + q] = J[q];
    cost_part[k] = 0.5 * r[0] * r[0] + 0.5 * r[1] * r[1];
  } else if (k < a.n_obs + 4 * W) {
    const int q = k - a.n_obs, sys = q / W, i = q % W;
    const double r = gnss_dt_ddt_res(g->rcv_dt[i][sys], g->rcv_dt[i + 1][sys], g->rcv_ddt[i], g->rcv_ddt[i + 1], a.dt[i]);
    r_clk[q] = r;
    cost_part[k] = 0.5 * r * r;
  } else if (k < a.n_obs + 5 * W) {
    const int i = k - a.n_obs - 4 * W;
    const double r = gnss_ddt_smooth_res(g->rcv_ddt[i], g->rcv_ddt[i + 1], a.ddt_weight);
    r_clk[4 * W + i] = r;
    cost_part[k] = 0.5 * r * r;
  }
}

#define GN_CHECK(c, call)                                                                                      \
  do {                                                                                                         \
    hipError_t e_ = (call);                                                                                    \
    if (e_ != hipSuccess) { ctx_set_error(c, (std::string(#call) + ": " + hipGetErrorString(e_)).c_str()); st = GFBE_DEVICE_ERROR; goto done; } \
  } while (0)

}  // namespace

extern "C" gfbe_status gfbe_gnss_eval(gfbe_ctx *c, int32_t n_obs, const gfbe_gnss_obs *obs, const double *iono, const gfbe_state *state,
                                      const gfbe_gnss_state *gnss, const double *frame_dt, double ddt_weight, double *r_obs, double *J_obs,
                                      double *r_dt_ddt, double *r_smooth, double *cost) {
  const int W = GFBE_WINDOW_SIZE;
  if (!c || n_obs < 0 || (n_obs > 0 && !obs) || !state || !gnss || !frame_dt) return GFBE_BAD_INPUT;
  for (int k = 0; k < n_obs; k++) {
    const gfbe_gnss_obs &o = obs[k];
    if (o.frame < 0 || o.frame > W || o.lower_idx < 0 || o.lower_idx >= W || o.sys_idx < 0 || o.sys_idx > 3 || !(o.pr_uura > 0.0) || !(o.dp_uura > 0.0)) {
      ctx_set_error(c, ("gfbe_gnss_eval: observation " + std::to_string(k) + " has an index or a deviation out of range").c_str());
      return GFBE_BAD_INPUT;
    }
  }
  if (ctx_device(c) < 0) return GFBE_NO_DEVICE;
  hipStream_t s = ctx_stream(c);
  gfbe_status st = GFBE_OK;
  char *d = nullptr;
  const int nt = n_obs + 5 * W;
  const size_t o_obs = 0, o_st = (sizeof(gfbe_gnss_obs) * (size_t)n_obs + 255) & ~(size_t)255, o_g = o_st + ((sizeof(gfbe_state) + 255) & ~(size_t)255),
               o_r = o_g + ((sizeof(gfbe_gnss_state) + 255) & ~(size_t)255), o_J = o_r + sizeof(double) * 2 * (size_t)n_obs,
               o_clk = o_J + sizeof(double) * 36 * (size_t)n_obs, o_c = o_clk + sizeof(double) * 5 * W, total = o_c + sizeof(double) * nt;
  std::vector<double> hc(nt), hclk(5 * W);
  GnssArgs a;
  a.n_obs = n_obs; a.has_iono = iono != nullptr; a.ddt_weight = ddt_weight;
  for (int i = 0; i < 8; i++) a.iono[i] = iono ? iono[i] : 0.0;
  for (int i = 0; i < W; i++) a.dt[i] = frame_dt[i];
  GN_CHECK(c, hipMalloc((void **)&d, total));
  if (n_obs) GN_CHECK(c, hipMemcpyAsync(d + o_obs, obs, sizeof(gfbe_gnss_obs) * (size_t)n_obs, hipMemcpyHostToDevice, s));
  GN_CHECK(c, hipMemcpyAsync(d + o_st, state, sizeof(gfbe_state), hipMemcpyHostToDevice, s));
  GN_CHECK(c, hipMemcpyAsync(d + o_g, gnss, sizeof(gfbe_gnss_state), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_gnss, dim3((nt + 127) / 128), dim3(128), 0, s, a, (const gfbe_gnss_obs *)(d + o_obs), (const gfbe_state *)(d + o_st),
                     (const gfbe_gnss_state *)(d + o_g), r_obs ? (double *)(d + o_r) : nullptr, J_obs ? (double *)(d + o_J) : nullptr,
                     (double *)(d + o_clk), (double *)(d + o_c));
  GN_CHECK(c, hipGetLastError());
  if (r_obs && n_obs) GN_CHECK(c, hipMemcpyAsync(r_obs, d + o_r, sizeof(double) * 2 * (size_t)n_obs, hipMemcpyDeviceToHost, s));
  if (J_obs && n_obs) GN_CHECK(c, hipMemcpyAsync(J_obs, d + o_J, sizeof(double) * 36 * (size_t)n_obs, hipMemcpyDeviceToHost, s));
  GN_CHECK(c, hipMemcpyAsync(hclk.data(), d + o_clk, sizeof(double) * 5 * W, hipMemcpyDeviceToHost, s));
  GN_CHECK(c, hipMemcpyAsync(hc.data(), d + o_c, sizeof(double) * nt, hipMemcpyDeviceToHost, s));
  GN_CHECK(c, hipStreamSynchronize(s));
  if (r_dt_ddt) for (int q = 0; q < 4 * W; q++) r_dt_ddt[q] = hclk[q];
  if (r_smooth) for (int q = 0; q < W; q++) r_smooth[q] = hclk[4 * W + q];
  if (cost) { double t = 0.0; for (int k = 0; k < nt; k++) t += hc[k]; *cost = t; }
done:
  if (d) (void)hipFree(d);
  return st;
}

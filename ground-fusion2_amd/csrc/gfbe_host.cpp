// gfbe_host.cpp — the C ABI of include/gfbe.h: context, packing of windows into the HBM layout
// (gfbe_device.h), the fixed kernel sequence of one optimization() call, and the host-side
// landmark bookkeeping. Host code only orchestrates: all arithmetic of the hot path runs in the
// HIP kernels (gfbe_kernels.hip, gfbe_marg.hip, gfbe_preint.hip). There is no CPU fallback.
//
// Reference being replaced: Estimator::optimization()
//   Ground-Fusion++/vins_estimator/src/estimator/estimator.cpp:2951-3698
#include "gfbe_device.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <chrono>
#include <string>
#include <vector>

using namespace gfd;

namespace gfd {
void launch_preint_imu(int n, const int *d_off, const double *d_samples, const double *d_first, const double *d_lin,
                       const double *noise4, gfbe_imu_preint *d_out, hipStream_t s);
void launch_preint_wheel(int n, const int *d_off, const double *d_samples, const double *d_first, const double *d_lin,
                         const double *noise2, gfbe_wheel_preint *d_out, hipStream_t s);
}

struct ProfEntry {
  std::string name;
  int64_t launches = 0;
  double total_ms = 0.0;
  double bytes = 0.0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct gfbe_ctx {
  int device = -1;
  gfbe_options opt;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // second stream: the inertial / wheel / prior factors (few, latency-bound workgroups) run beside the visual kernels
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  std::string err;
  bool profiling = false;
  std::vector<ProfEntry> prof;
  std::vector<hipEvent_t> event_pool;
  gfbe_allreduce_fn allreduce = nullptr;
  void *allreduce_user = nullptr;
  int rank = 0, world = 1;
  // Device memory of freed batches, kept for the next upload: one window per camera frame is the reference's call
  // pattern, and ~70 hipMalloc / hipFree pairs per call cost more than its solve (3.6 ms of 4.1 ms measured).
  std::vector<std::pair<void *, size_t>> slab_cache;
  // grow-only device scratch of the short host-buffer calls (pre-integration): no hipMalloc / hipFree per call
  char *scratch = nullptr;
  size_t scratch_cap = 0;
};
enum : size_t { SLAB_CACHE_ENTRIES = 4, SLAB_CACHE_MAX_BYTES = (size_t)512 << 20 };

// The streams / events one (sub-)batch runs on: main stream, the aux stream of its dense factors, fork / join events.
struct Lane { hipStream_t s, aux; hipEvent_t fork, join; };

struct gfbe_batch {
  BatchDev d;
  // every device array of the batch is carved from ONE slab: a dry pass over the allocation sequence adds up the sizes,
  // the slab comes from the context's cache (or hipMalloc), the second pass hands out the pointers
  char *slab = nullptr;
  size_t slab_bytes = 0, slab_off = 0;
  bool dry = false;
  std::vector<std::vector<int>> slot_of;   // per window: ABI landmark -> global slot
  std::vector<int> L;
  double algo_bytes_lin = 0.0;             // algorithmic bytes of one visual linearisation of the batch
  size_t slab_n = 0;                       // doubles of the [H | g | E | eg | xa] slab
  // the launch sequence of one optimization() is fixed (no host decision inside): captured once per margin flag
  // into a hipGraph (second call) and replayed afterwards
  hipGraphExec_t graph[3] = {nullptr, nullptr, nullptr};
  int calls[3] = {0, 0, 0};
  // Large batches are uploaded as TWO halves; the second half (`second`) is solved on its own pair of streams beside
  // the first, so that kernels of different stages (e.g. the 1-workgroup-per-CU k_solve of one half and the visual
  // kernels of the other) share the GPU. Measured +13 % at 256 windows; four groups are host-launch-bound.
  gfbe_batch *second = nullptr;
  Lane lane2 = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_start2 = nullptr, ev_done2 = nullptr;
};

#define HIPCHK(ctx, call)                                                                        \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
      return GFBE_DEVICE_ERROR;                                                                  \
    }                                                                                            \
  } while (0)

namespace gfd {
hipStream_t ctx_stream(gfbe_ctx *c) { return c->stream; }
int ctx_device(const gfbe_ctx *c) { return c ? c->device : -1; }
void ctx_set_error(gfbe_ctx *c, const char *msg) { if (c) c->err = msg; }
// grow-only device scratch of the context (at least `bytes`; contents undefined); nullptr when the allocation fails
void *ctx_scratch(gfbe_ctx *c, size_t bytes) {
  if (bytes > c->scratch_cap) {
    (void)hipStreamSynchronize(c->stream);
    if (c->scratch) (void)hipFree(c->scratch);
    c->scratch = nullptr; c->scratch_cap = 0;
    const size_t cap = bytes + bytes / 2;
    if (hipMalloc((void **)&c->scratch, cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    c->scratch_cap = cap;
  }
  return c->scratch;
}
}  // namespace gfd

extern "C" {

void gfbe_default_options(gfbe_options *o) {
  o->max_num_iterations = 8;                 // m3dgr.yaml:109
  o->huber_delta = 1.0;                      // estimator.cpp:2959
  o->vis_sqrt_info = 600.0 / 1.5;            // estimator.cpp:193, parameters.h:23
  o->g_norm = 9.7944;                        // m3dgr.yaml:117
  o->initial_trust_region_radius = 1e4;      // Ceres 1.14 defaults (estimator.cpp:3364-3376 leaves them)
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->min_relative_decrease = 1e-3;
  o->jacobi_scaling = 1;
  o->marg_eps = 1e-8;                        // marginalization_factor.h:70
  o->marg_sqrt = 1;                          // pivoted LDL^T square root (0 = eigen-decomposition as in the reference)
  o->split_batch = 1;                        // batches of >= 128 windows run as two halves on two pairs of streams
  o->use_graph = 0;                          // 1: replay the fixed launch sequence of gfbe_batch_solve as a hipGraph (measured: no gain)
}

const char *gfbe_version(void) { return "gfbe 0.1.0 (gfx950, HIP)"; }
const char *gfbe_last_error(const gfbe_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

gfbe_status gfbe_create(gfbe_ctx **out, int device, const gfbe_options *opt) {
  if (!out) return GFBE_BAD_INPUT;
  gfbe_ctx *c = new gfbe_ctx();
  if (opt) c->opt = *opt; else gfbe_default_options(&c->opt);
  c->device = device;
  *out = c;
  if (device < 0) return GFBE_OK;   // host-only context: bookkeeping entry points only
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= device) {
    c->err = "no HIP device " + std::to_string(device) + " visible (the HIP back end has no CPU fallback)";
    return GFBE_NO_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) {
    c->err = "hipSetDevice/hipStreamCreate failed";
    return GFBE_DEVICE_ERROR;
  }
  c->own_stream = true;
  if (hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
    c->err = "hipStreamCreate/hipEventCreate (aux stream) failed";
    return GFBE_DEVICE_ERROR;
  }
  return GFBE_OK;
}

void gfbe_destroy(gfbe_ctx *c) {
  if (!c) return;
  for (auto e : c->event_pool) (void)hipEventDestroy(e);
  for (auto &p : c->prof) for (auto &ev : p.pending) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  if (c->aux) { (void)hipStreamSynchronize(c->aux); (void)hipStreamDestroy(c->aux); }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  for (auto &sl : c->slab_cache) (void)hipFree(sl.first);
  if (c->scratch) (void)hipFree(c->scratch);
  delete c;
}

gfbe_status gfbe_set_stream(gfbe_ctx *c, void *s) {
  if (!c || c->device < 0) return GFBE_NO_DEVICE;
  if (c->own_stream && c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
  if (s) { c->stream = (hipStream_t)s; c->own_stream = false; }
  else { if (hipStreamCreate(&c->stream) != hipSuccess) return GFBE_DEVICE_ERROR; c->own_stream = true; }
  return GFBE_OK;
}

gfbe_status gfbe_set_allreduce(gfbe_ctx *c, gfbe_allreduce_fn fn, void *user, int32_t rank, int32_t world) {
  if (!c) return GFBE_BAD_INPUT;
  if (fn && (world < 1 || world > 64 || rank < 0 || rank >= world)) { c->err = "gfbe_set_allreduce: rank / world_size out of range"; return GFBE_BAD_INPUT; }
  c->allreduce = fn; c->allreduce_user = user;
  c->rank = fn ? rank : 0; c->world = fn ? world : 1;   // a null hook switches the landmark sharding off
  return GFBE_OK;
}

// ---------------------------------------------------------------------------------------------
// a13 landmark bookkeeping (feature_manager.cpp:43-55, 249-267, 286-302; estimator.cpp:3326-3358, 3498-3531)
// ---------------------------------------------------------------------------------------------
int32_t gfbe_feature_count(const gfbe_feature_list *fl) {
  int32_t n = 0;
  for (int f = 0; f < fl->n; f++) n += (fl->n_obs[f] >= 4);     // used_num >= 4
  return n;
}
int32_t gfbe_visual_factor_count(const gfbe_feature_list *fl, int32_t only0) {
  int32_t k = 0;
  for (int f = 0; f < fl->n; f++)
    if (fl->n_obs[f] >= 4 && (!only0 || fl->start_frame[f] == 0)) k += fl->n_obs[f] - 1;
  return k;
}
int32_t gfbe_build_visual_factors(const gfbe_feature_list *fl, int32_t only0, int32_t *feature_index, int32_t *imu_i,
                                  int32_t *imu_j, double *pts_i, double *pts_j, double *vel_i, double *vel_j,
                                  double *td_i, double *td_j, double *para_Feature, uint8_t *feature_const) {
  int32_t out = 0, landmark = 0;
  for (int f = 0; f < fl->n; f++) {
    const int nobs = fl->n_obs[f];
    if (nobs < 4) continue;
    const int lm = landmark++;                                   // ++feature_index
    if (para_Feature) para_Feature[lm] = 1.0 / fl->estimated_depth[f];
    if (feature_const) feature_const[lm] = (fl->estimate_flag[f] == 1);
    const int start = fl->start_frame[f];
    if (only0 && start != 0) continue;
    const int base = fl->obs_offset[f];
    const double *o0 = fl->obs + 7 * (size_t)base;
    for (int k = 1; k < nobs; k++) {                             // imu_j = start + k, never == imu_i
      const double *ok = fl->obs + 7 * (size_t)(base + k);
      feature_index[out] = lm; imu_i[out] = start; imu_j[out] = start + k;
      pts_i[3 * out] = o0[0]; pts_i[3 * out + 1] = o0[1]; pts_i[3 * out + 2] = o0[2];
      pts_j[3 * out] = ok[0]; pts_j[3 * out + 1] = ok[1]; pts_j[3 * out + 2] = ok[2];
      vel_i[2 * out] = o0[5]; vel_i[2 * out + 1] = o0[6];
      vel_j[2 * out] = ok[5]; vel_j[2 * out + 1] = ok[6];
      td_i[out] = fl->obs_td[base]; td_j[out] = fl->obs_td[base + k];
      out++;
    }
  }
  return out;
}
void gfbe_set_depth(const gfbe_feature_list *fl, const double *para_Feature, double *estimated_depth, int32_t *solve_flag) {
  int lm = 0;
  for (int f = 0; f < fl->n; f++) {
    if (fl->n_obs[f] < 4) continue;
    const double dep = 1.0 / para_Feature[lm++];
    estimated_depth[f] = dep;
    solve_flag[f] = dep < 0 ? 2 : 1;
  }
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// profiling helper
// ---------------------------------------------------------------------------------------------
namespace {

struct Timed {
  gfbe_ctx *c;
  int idx = -1;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  Timed(gfbe_ctx *ctx, const char *name, double bytes) : c(ctx) {
    if (!c->profiling) return;
    for (size_t i = 0; i < c->prof.size(); i++) if (c->prof[i].name == name) idx = (int)i;
    if (idx < 0) { c->prof.emplace_back(); idx = (int)c->prof.size() - 1; c->prof[idx].name = name; }
    auto get = [&]() { hipEvent_t e; if (!c->event_pool.empty()) { e = c->event_pool.back(); c->event_pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
    e0 = get(); e1 = get();
    c->prof[idx].launches++;
    c->prof[idx].bytes += bytes;
    (void)hipEventRecord(e0, c->stream);
  }
  ~Timed() {
    if (idx < 0) return;
    (void)hipEventRecord(e1, c->stream);
    c->prof[idx].pending.emplace_back(e0, e1);
  }
};

void prof_collect(gfbe_ctx *c) {
  for (auto &p : c->prof) {
    for (auto &ev : p.pending) {
      (void)hipEventSynchronize(ev.second);
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) p.total_ms += ms;
      c->event_pool.push_back(ev.first); c->event_pool.push_back(ev.second);
    }
    p.pending.clear();
  }
}

// (the slab is zeroed once, before the second pass)
template <typename T>
gfbe_status dev_alloc(gfbe_ctx *c, gfbe_batch *b, T **p, size_t n) {
  const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
  if (b->dry) { b->slab_bytes += bytes; *p = nullptr; return GFBE_OK; }
  if (b->slab_off + bytes > b->slab_bytes) { c->err = "batch slab overrun"; return GFBE_DEVICE_ERROR; }
  *p = (T *)(b->slab + b->slab_off);
  b->slab_off += bytes;
  return GFBE_OK;
}
template <typename T>
gfbe_status dev_upload(gfbe_ctx *c, gfbe_batch *b, T **p, const std::vector<T> &h) {
  gfbe_status st = dev_alloc(c, b, p, h.size());
  if (st != GFBE_OK || b->dry) return st;
  if (!h.empty()) HIPCHK(c, hipMemcpyAsync(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
  return GFBE_OK;
}
// the smallest cached slab that fits (and is not more than twice too large), else a new one
gfbe_status slab_acquire(gfbe_ctx *c, gfbe_batch *b) {
  int best = -1;
  for (size_t i = 0; i < c->slab_cache.size(); i++) {
    const size_t cap = c->slab_cache[i].second;
    if (cap >= b->slab_bytes && cap <= 2 * b->slab_bytes && (best < 0 || cap < c->slab_cache[best].second)) best = (int)i;
  }
  if (best >= 0) {
    b->slab = (char *)c->slab_cache[best].first;
    b->slab_bytes = c->slab_cache[best].second;
    c->slab_cache.erase(c->slab_cache.begin() + best);
  } else {
    // sizes in steps of 1/4 of the leading power of two: the windows of consecutive frames (slightly different landmark
    // counts) land on the same cached slab
    size_t grain = (size_t)1 << 20;
    while (grain * 8 <= b->slab_bytes) grain *= 2;
    b->slab_bytes = (b->slab_bytes + grain - 1) / grain * grain;
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, b->slab_bytes);
    if (e != hipSuccess) {      // make room: drop the cache and retry once
      for (auto &sl : c->slab_cache) (void)hipFree(sl.first);
      c->slab_cache.clear();
      e = hipMalloc(&q, b->slab_bytes);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); c->err = std::string("hipMalloc(batch slab): ") + hipGetErrorString(e); return GFBE_DEVICE_ERROR; }
    b->slab = (char *)q;
  }
  HIPCHK(c, hipMemsetAsync(b->slab, 0, b->slab_bytes, c->stream));
  return GFBE_OK;
}
void slab_release(gfbe_ctx *c, gfbe_batch *b) {
  if (!b->slab) return;
  if (c && b->slab_bytes <= SLAB_CACHE_MAX_BYTES) {
    if (c->slab_cache.size() >= SLAB_CACHE_ENTRIES) { (void)hipFree(c->slab_cache.front().first); c->slab_cache.erase(c->slab_cache.begin()); }
    c->slab_cache.emplace_back(b->slab, b->slab_bytes);
  } else {
    (void)hipFree(b->slab);
  }
  b->slab = nullptr;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Upload: pack windows into the device layout.
// ---------------------------------------------------------------------------------------------
// tabs != nullptr: the visual factors of window w come from table tab0 + w of the device-resident feature tables
// (wins[w]->vis, n_feature, para_Feature, feature_const are ignored); the landmark arrays are then filled on the device.
static gfbe_status upload_one(gfbe_ctx *c, int32_t B, const gfbe_window *const *wins, gfbe_batch **out, gfbe_ftab *tabs = nullptr,
                              int tab0 = 0) {
  if (!c || !wins || !out || B <= 0) return GFBE_BAD_INPUT;
  if (c->device < 0 || !c->stream) { c->err = "HIP device context required (no CPU fallback)"; return GFBE_NO_DEVICE; }
  const bool dbg_t = getenv("GFBE_DEBUG_UPLOAD") != nullptr;   // phase times of the upload on stderr (tests/diag_e2e_latency.py)
  auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  HIPCHK(c, hipSetDevice(c->device));
  const double T0 = now();
  gfbe_batch *b = new gfbe_batch();
  *out = b;
  BatchDev &d = b->d;
  std::memset(&d, 0, sizeof d);
  d.B = B;
  d.opt = c->opt;
  std::vector<WinDesc> desc(B);
  std::vector<int> lm_info, lm_abi, lm_rec, tile_start;
  std::vector<double> lm_pts, lm_obs, lam0, x0((size_t)B * NA);
  std::vector<gfbe_imu_preint> imu;
  std::vector<gfbe_wheel_preint> wheel;
  std::vector<double> lio;
  std::vector<double> pr0((size_t)B * ND, 0.0), px0((size_t)B * PRIOR_X0, 0.0);
  // J0 of the priors travels compactly: host rows of nmax^2 doubles (nmax = the largest prior of the batch), copied into the
  // device slots of ND^2 doubles by one 2-D copy (a 2k-landmark window's prior is 86^2 of the 182^2 doubles of a slot)
  int pn_max = 0;
  for (int w = 0; w < B; w++) if (wins[w] && wins[w]->prior && wins[w]->prior->valid) pn_max = std::max(pn_max, std::min(wins[w]->prior->n, (int)ND));
  const size_t pj_row = (size_t)std::max(pn_max, 0) * std::max(pn_max, 0);
  std::vector<double> pJ0((size_t)B * pj_row, 0.0);
  b->slot_of.resize(B);
  b->L.resize(B);
  // first pass: sizes
  struct LmTmp { int start, m, abi; std::vector<int> fac; };
  std::vector<std::vector<LmTmp>> all_lms(B);
  int tot_lm = 0, tot_rec = 0, max_tiles = 0;
  std::vector<int> tcounts, tlayout;     // table source: per window [L, K, bins], layout table for the pack kernel
  if (tabs) {
    if (tab0 < 0 || tab0 + B > tabs->d.W) { c->err = "gfbe_batch_upload_tables: more windows than tables"; return GFBE_BAD_INPUT; }
    int *dcounts = tabs->d.hist + (size_t)tab0 * (FT_BINS + 2);
    launch_ftab_count(tabs->d, tabs->cur, tab0, B, dcounts, c->stream);
    tcounts.resize((size_t)B * (FT_BINS + 2));
    HIPCHK(c, hipMemcpyAsync(tcounts.data(), dcounts, sizeof(int) * tcounts.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    tlayout.assign((size_t)B * FT_LAY_STRIDE, 0);
  }
  for (int w = 0; w < B; w++) {
    const gfbe_window &win = *wins[w];
    WinDesc &ds = desc[w];
    std::memset(&ds, 0, sizeof ds);
    const int L = tabs ? tcounts[(size_t)w * (FT_BINS + 2)] : win.n_feature, K = tabs ? tcounts[(size_t)w * (FT_BINS + 2) + 1] : win.vis.n_factor;
    if (L < 0 || K < 0 || win.frame_count < 0 || win.frame_count > GFBE_WINDOW_SIZE || win.n_imu > MAX_IMU || win.n_wheel > MAX_WHEEL) {
      c->err = "window " + std::to_string(w) + ": bad sizes"; return GFBE_BAD_INPUT;
    }
    ds.L = L; ds.K = K; ds.frame_count = win.frame_count;
    if (tabs) {   // layout from the per-bin counts: groups by start frame (tile aligned), longer tracks first inside a group
      const int *cnt = &tcounts[(size_t)w * (FT_BINS + 2) + 2];
      int *lay = &tlayout[(size_t)w * FT_LAY_STRIDE];
      ds.lm_off = tot_lm;
      ds.tile_off = (int)tile_start.size();
      b->L[w] = L;
      lay[0] = tot_lm;
      int slots = 0;
      for (int s = 0; s < NF; s++) {
        ds.sf_tile_begin[s] = slots / LM_TILE;
        lay[FT_LAY_GRP + s] = slots;
        int in_group = 0;
        for (int m = MAXOBS; m >= 3; m--) { lay[FT_LAY_BIN + s * 8 + (m - 3)] = slots + in_group; in_group += cnt[s * 8 + (m - 3)]; }
        const int padded = (in_group + LM_TILE - 1) / LM_TILE * LM_TILE;
        for (int t = 0; t < padded / LM_TILE; t++) tile_start.push_back(s);
        slots += padded;
      }
      ds.sf_tile_begin[NF] = slots / LM_TILE;
      ds.lm_slots = slots;
      ds.n_tiles = slots / LM_TILE;
      max_tiles = std::max(max_tiles, ds.n_tiles);
      tot_lm += slots;
      ds.rec_off = tot_rec;
      tot_rec += K;
      continue;
    }
    std::vector<LmTmp> &lms = all_lms[w];
    lms.resize(L);
    for (int l = 0; l < L; l++) { lms[l].start = -1; lms[l].m = 0; lms[l].abi = l; }
    for (int k = 0; k < K; k++) {
      const int l = win.vis.feature_index[k], i = win.vis.imu_i[k], j = win.vis.imu_j[k];
      if (l < 0 || l >= L || i < 0 || j <= i || j > win.frame_count) { c->err = "window " + std::to_string(w) + ": bad visual factor " + std::to_string(k); return GFBE_BAD_INPUT; }
      if (lms[l].start < 0) lms[l].start = i;
      if (lms[l].start != i) { c->err = "visual factors of one landmark must share imu_i"; return GFBE_BAD_INPUT; }
      lms[l].fac.push_back(k);
    }
    for (int l = 0; l < L; l++) {
      LmTmp &lm = lms[l];
      if (lm.start < 0) lm.start = 0;
      std::sort(lm.fac.begin(), lm.fac.end(), [&](int a, int bb) { return win.vis.imu_j[a] < win.vis.imu_j[bb]; });
      lm.m = (int)lm.fac.size();
      if (lm.m > MAXOBS) { c->err = "landmark with more than 10 factors"; return GFBE_BAD_INPUT; }
      for (int k = 0; k < lm.m; k++)
        if (win.vis.imu_j[lm.fac[k]] != lm.start + 1 + k) { c->err = "landmark track must be contiguous from start_frame (feature_per_frame order)"; return GFBE_BAD_INPUT; }
    }
    // internal order: by start frame, then longer tracks first (uniform trip counts inside a wave)
    std::vector<int> order(L);
    for (int l = 0; l < L; l++) order[l] = l;
    std::stable_sort(order.begin(), order.end(), [&](int a, int bb) {
      if (lms[a].start != lms[bb].start) return lms[a].start < lms[bb].start;
      return lms[a].m > lms[bb].m;
    });
    ds.lm_off = tot_lm;
    ds.tile_off = (int)tile_start.size();
    b->slot_of[w].assign(L, -1);
    b->L[w] = L;
    int slots = 0, oi = 0;
    for (int s = 0; s < NF; s++) {
      ds.sf_tile_begin[s] = slots / LM_TILE;
      int cnt = 0;
      while (oi < L && lms[order[oi]].start == s) { b->slot_of[w][order[oi]] = tot_lm + slots + cnt; cnt++; oi++; }
      const int padded = (cnt + LM_TILE - 1) / LM_TILE * LM_TILE;
      for (int t = 0; t < padded / LM_TILE; t++) tile_start.push_back(s);
      slots += padded;
    }
    ds.sf_tile_begin[NF] = slots / LM_TILE;
    ds.lm_slots = slots;
    ds.n_tiles = slots / LM_TILE;
    max_tiles = std::max(max_tiles, ds.n_tiles);
    tot_lm += slots;
    ds.rec_off = tot_rec;
    tot_rec += K;
  }
  d.tot_lm = tot_lm; d.max_tiles = max_tiles; d.tot_rec = tot_rec;
  lm_info.assign(tot_lm, 0); lm_abi.assign(tot_lm, -1);
  lm_pts.assign((size_t)6 * tot_lm, 0.0); lm_obs.assign((size_t)MAXOBS * 5 * tot_lm, 0.0);
  lm_rec.assign((size_t)MAXOBS * tot_lm, 0); lam0.assign(tot_lm, 1.0);
  const size_t TL = tot_lm;
  double algo_bytes = 0.0;
  for (int w = 0; w < B; w++) {
    const gfbe_window &win = *wins[w];
    WinDesc &ds = desc[w];
    std::vector<LmTmp> &lms = all_lms[w];
    // pair-major record positions, assigned in slot order
    std::vector<int> pair_cnt(NPAIR + 1, 0);
    if (tabs) {   // factors of pair (s, s+1+k) = landmarks of start frame s with more than k factors
      const int *cnt = &tcounts[(size_t)w * (FT_BINS + 2) + 2];
      for (int s = 0; s < NF; s++)
        for (int k = 0; k < MAXOBS && s + 1 + k < NF; k++)
          for (int m = std::max(k + 1, 3); m <= MAXOBS; m++) pair_cnt[s * NF + s + 1 + k] += cnt[s * 8 + (m - 3)];
    } else {
      for (int k = 0; k < ds.K; k++) pair_cnt[win.vis.imu_i[k] * NF + win.vis.imu_j[k]]++;
    }
    int run = 0;
    for (int p = 0; p < NPAIR; p++) { ds.pair_begin[p] = run; run += pair_cnt[p]; }
    ds.pair_begin[NPAIR] = run;
    if (tabs) std::memcpy(&tlayout[(size_t)w * FT_LAY_STRIDE + FT_LAY_PAIR], ds.pair_begin, sizeof(int) * (NPAIR + 1));
    std::vector<int> fill(ds.pair_begin, ds.pair_begin + NPAIR);
    std::vector<std::pair<int, int>> by_slot;
    if (!tabs) for (int l = 0; l < ds.L; l++) by_slot.emplace_back(b->slot_of[w][l], l);
    std::sort(by_slot.begin(), by_slot.end());
    for (auto &sl : by_slot) {
      const int slot = sl.first, l = sl.second;
      const LmTmp &lm = lms[l];
      const bool is_const = win.feature_const && win.feature_const[l];
      lm_info[slot] = lm.start | (lm.m << 8) | ((is_const ? 1 : 0) << 16) | (1 << 24);
      lm_abi[slot] = l;
      lam0[slot] = win.para_Feature[l];
      if (lm.m > 0) {
        const int k0 = lm.fac[0];
        lm_pts[0 * TL + slot] = win.vis.pts_i[3 * k0]; lm_pts[1 * TL + slot] = win.vis.pts_i[3 * k0 + 1];
        lm_pts[2 * TL + slot] = win.vis.pts_i[3 * k0 + 2];
        lm_pts[3 * TL + slot] = win.vis.vel_i[2 * k0]; lm_pts[4 * TL + slot] = win.vis.vel_i[2 * k0 + 1];
        lm_pts[5 * TL + slot] = win.vis.td_i[k0];
      }
      for (int k = 0; k < lm.m; k++) {
        const int f = lm.fac[k];
        double *ob = &lm_obs[(size_t)k * 5 * TL + slot];
        ob[0] = win.vis.pts_j[3 * f]; ob[TL] = win.vis.pts_j[3 * f + 1];
        ob[2 * TL] = win.vis.vel_j[2 * f]; ob[3 * TL] = win.vis.vel_j[2 * f + 1]; ob[4 * TL] = win.vis.td_j[f];
        lm_rec[(size_t)k * TL + slot] = fill[lm.start * NF + lm.start + 1 + k]++;
      }
    }
    algo_bytes += 108.0 * ds.K;   // 12 f64 + 3 i32 per visual residual block, J never re-read by the host
    // dense state
    std::memcpy(&x0[(size_t)w * NA], &win.state, sizeof(double) * NA);
    // inertial factors
    for (int q = 0; q < NF; q++) ds.imu_of_frame[q] = ds.wheel_of_frame[q] = -1;
    ds.n_imu = win.n_imu; ds.imu_off = (int)imu.size();
    for (int k = 0; k < win.n_imu; k++) {
      if (win.imu_frame[k] < 0 || win.imu_frame[k] >= win.frame_count) { c->err = "bad imu_frame"; return GFBE_BAD_INPUT; }
      imu.push_back(win.imu[k]); ds.imu_frame[k] = win.imu_frame[k]; ds.imu_of_frame[win.imu_frame[k]] = k;
    }
    ds.n_wheel = win.n_wheel; ds.wheel_off = (int)wheel.size();
    for (int k = 0; k < win.n_wheel; k++) {
      if (win.wheel_frame[k] < 0 || win.wheel_frame[k] >= win.frame_count) { c->err = "bad wheel_frame"; return GFBE_BAD_INPUT; }
      wheel.push_back(win.wheel[k]); ds.wheel_frame[k] = win.wheel_frame[k]; ds.wheel_of_frame[win.wheel_frame[k]] = k;
    }
    // LiDAR factors on one pose
    ds.lio_n = win.lio.n > 0 ? win.lio.n : 0; ds.lio_off = (int)(lio.size() / 8); ds.lio_frame = win.lio.frame;
    ds.lio_sqrt_info = win.lio.sqrt_info; ds.lio_huber = win.lio.huber_delta;
    if (ds.lio_n > 0) {
      if (win.lio.frame < 0 || win.lio.frame > win.frame_count || !win.lio.pts || !win.lio.normals || !win.lio.offsets) { c->err = "bad lio block"; return GFBE_BAD_INPUT; }
      for (int k = 0; k < ds.lio_n; k++) {
        for (int q = 0; q < 3; q++) lio.push_back(win.lio.pts[3 * k + q]);
        for (int q = 0; q < 3; q++) lio.push_back(win.lio.normals[3 * k + q]);
        lio.push_back(win.lio.offsets[k]);
        lio.push_back(win.lio.weights ? win.lio.weights[k] : 1.0);
      }
    }
    // prior
    bool used[GFBE_BLK_COUNT];
    for (int q = 0; q < GFBE_BLK_COUNT; q++) used[q] = false;
    if (ds.lio_n > 0) used[ds.lio_frame] = true;
    for (int q = 0; q < ND; q++) ds.prior_map[q] = -1;
    if (win.prior && win.prior->valid && win.prior->n > 0) {
      const gfbe_prior &pr = *win.prior;
      if (pr.n > ND || pr.n_blocks > GFBE_MAX_PRIOR_BLOCKS) { c->err = "prior too large"; return GFBE_BAD_INPUT; }
      ds.prior_n = pr.n; ds.prior_nblk = pr.n_blocks;
      int xo = 0;
      for (int q = 0; q < pr.n_blocks; q++) {
        const int id = pr.block_id[q];
        if (id < 0 || id >= GFBE_BLK_COUNT || pr.block_size[q] != blk_gsize(id)) { c->err = "prior block table inconsistent"; return GFBE_BAD_INPUT; }
        ds.prior_blk_id[q] = id; ds.prior_blk_size[q] = pr.block_size[q]; ds.prior_blk_idx[q] = pr.block_idx[q];
        ds.prior_x0_off[q] = xo; xo += pr.block_size[q];
        used[id] = true;
        for (int k = 0; k < blk_lsize(id); k++) ds.prior_map[blk_tan(id) + k] = pr.block_idx[q] + k;
      }
      std::memcpy(&px0[(size_t)w * PRIOR_X0], pr.x0, sizeof(double) * xo);
      std::memcpy(&pJ0[(size_t)w * pj_row], pr.J0, sizeof(double) * pr.n * pr.n);
      std::memcpy(&pr0[(size_t)w * ND], pr.r0, sizeof(double) * pr.n);
    }
    // reduced program: blocks touched by a residual and not constant (Ceres drops the rest)
    for (int k = 0; k < win.n_imu; k++) { const int i = win.imu_frame[k]; used[i] = used[GFBE_BLK_SB0 + i] = used[i + 1] = used[GFBE_BLK_SB0 + i + 1] = true; }
    for (int k = 0; k < win.n_wheel; k++) {
      const int i = win.wheel_frame[k];
      used[i] = used[i + 1] = used[GFBE_BLK_EX_WHEEL] = used[GFBE_BLK_SX] = used[GFBE_BLK_SY] = used[GFBE_BLK_SW] = used[GFBE_BLK_TD_WHEEL] = true;
    }
    for (int p = 0; p < NPAIR; p++) if (pair_cnt[p] > 0) { used[p / NF] = used[p % NF] = used[GFBE_BLK_EX_CAM] = used[GFBE_BLK_TD] = true; }
    for (int q = 0; q < GFBE_BLK_COUNT; q++) {
      bool cst;
      if (q < GFBE_BLK_SB0) cst = win.pose_const[q] || q > win.frame_count;
      else if (q < GFBE_BLK_EX_CAM) cst = win.sb_const[q - GFBE_BLK_SB0] || (q - GFBE_BLK_SB0) > win.frame_count;
      else if (q == GFBE_BLK_EX_CAM) cst = win.ex_cam_const;
      else if (q == GFBE_BLK_EX_WHEEL) cst = win.ex_wheel_const;
      else if (q == GFBE_BLK_TD) cst = win.td_const;
      else if (q == GFBE_BLK_TD_WHEEL) cst = win.td_wheel_const;
      else cst = win.ix_wheel_const;
      ds.blk_free[q] = used[q] && !cst;
      if (ds.blk_free[q]) for (int k = 0; k < blk_lsize(q); k++) ds.act[blk_tan(q) + k] = 1;
    }
    std::memcpy(ds.ex_cam_mask, win.ex_cam_mask, 6);
    std::memcpy(ds.ex_wheel_mask, win.ex_wheel_mask, 6);
  }
  b->algo_bytes_lin = algo_bytes;
  gfbe_status st;
  const double T1 = now();
  d.rank = c->rank; d.world = c->world;
  for (int pass = 0; pass < 2; pass++) {
  b->dry = pass == 0;
  if (pass == 1 && (st = slab_acquire(c, b)) != GFBE_OK) return st;
#define UP(field, vec) if ((st = dev_upload(c, b, &d.field, vec)) != GFBE_OK) return st
#define AL(field, n) if ((st = dev_alloc(c, b, &d.field, (size_t)(n))) != GFBE_OK) return st
  UP(desc, desc);
  if (tabs) {
    AL(lm_info, TL); AL(lm_abi, TL); AL(lm_pts, (size_t)6 * TL); AL(lm_obs, (size_t)MAXOBS * 5 * TL); AL(lm_rec, (size_t)MAXOBS * TL); AL(lam0, TL);
  } else {
    UP(lm_info, lm_info); UP(lm_abi, lm_abi); UP(lm_pts, lm_pts); UP(lm_obs, lm_obs); UP(lm_rec, lm_rec); UP(lam0, lam0);
  }
  UP(x0, x0); UP(tile_start, tile_start); UP(imu, imu); UP(wheel, wheel);
  d.tot_lio = (int)(lio.size() / 8);
  UP(lio, lio); AL(lio_part, (size_t)B * LIOW_WGS * LIOW_PART);
  AL(prior_J0, (size_t)B * ND * ND); UP(prior_r0, pr0); UP(prior_x0, px0);
  if (!b->dry && pj_row > 0)
    HIPCHK(c, hipMemcpy2DAsync(d.prior_J0, sizeof(double) * ND * ND, pJ0.data(), sizeof(double) * pj_row, sizeof(double) * pj_row, B, hipMemcpyHostToDevice, c->stream));
  AL(raw_imu, (size_t)MAX_IMU * (15 + 450) * B); AL(raw_wheel, (size_t)MAX_WHEEL * (6 + 132) * B);
  AL(zero, 16); AL(vis_H, (size_t)B * NV * (NV + 1));
  if (B < DENSE_SPLIT_MIN_B && c->world == 1) { AL(vis_Hs, (size_t)B * VS_BLOCKS * NV * (NV + 1)); } else d.vis_Hs = nullptr; AL(asm_tab, (size_t)4 * (ND * (ND + 1) / 2)); AL(ctl, B); AL(x, (size_t)B * 2 * NA); AL(xout, (size_t)B * NA);
  AL(lam, 2 * TL); AL(lm_Hll, TL); AL(lm_gl, TL); AL(lm_hC, (size_t)HC * TL); AL(lm_hP, (size_t)MAXOBS * 6 * TL);
  AL(lm_sl, TL); AL(lm_yl, TL); AL(lm_vl, TL);
  AL(rec, (size_t)tot_rec * REC);
  AL(imu_sqrt, imu.size() * 225); AL(wheel_sqrt, wheel.size() * 36); AL(prior_H, (size_t)B * ND * ND);
  AL(pair_part, (size_t)B * NPAIR * VP_STRIDE); AL(vis_part, (size_t)B * std::max(max_tiles, 1) * MAXOBS * VP_STRIDE); AL(schur_part, (size_t)B * NF * SCHUR_STRIDE);
  AL(imu_part, (size_t)B * MAX_IMU * IMU_PART); AL(wheel_part, (size_t)B * MAX_WHEEL * WHEEL_PART);
  AL(prior_g, (size_t)B * (ND + 2));
  AL(tile_cost, (size_t)B * std::max(max_tiles, 1)); AL(tile_cand, (size_t)B * std::max(max_tiles, 1) * 4);
  AL(tile_gram, (size_t)B * std::max(max_tiles, 1) * 8); AL(dense_cand, (size_t)B * 4);
  // the partial reduced system [H | g | E | eg | xa] is one slab: a single all-reduce per linearisation when the
  // landmarks are sharded over ranks
  {
    const size_t nH = (size_t)B * ND * ND, ng = (size_t)B * ND, nE = (size_t)B * NV * NV, ne = (size_t)B * NV, nx = (size_t)B * d.world * XCHG;
    AL(H, nH + ng + nE + ne + nx);
    if (!b->dry) { d.g = d.H + nH; d.E = d.g + ng; d.eg = d.E + nE; d.xa = d.eg + ne; }
    b->slab_n = nH + ng + nE + ne + nx;
  }
  AL(xb, (size_t)B * d.world * XCHG); AL(xc, (size_t)B * d.world * XCHG);
  AL(sp, (size_t)B * ND); AL(Dp, (size_t)B * ND); AL(gts, (size_t)B * ND); AL(vp, (size_t)B * ND);
  AL(yp, (size_t)B * ND); AL(step, (size_t)B * ND);
  AL(dbg_imu, (size_t)B * MAX_IMU * 15 * 31); AL(dbg_wheel, (size_t)B * MAX_WHEEL * 6 * 23); AL(dbg_prior, (size_t)B * ND);
  AL(mA, (size_t)B * ND * ND); AL(mb, (size_t)B * ND); AL(mJ0, (size_t)B * ND * ND); AL(mr0, (size_t)B * ND); AL(mV, (size_t)B * ND * ND);
  AL(timing, (size_t)B * 32); AL(mmeta, (size_t)B * (4 + 3 * GFBE_MAX_PRIOR_BLOCKS)); AL(mx0, (size_t)B * PRIOR_X0);
#undef UP
#undef AL
  }
  if (tabs) {   // landmark arrays straight from the device-resident tables; the slot of every landmark comes back for the download
    // (layout table and slot map live in the tables' own scratch: no allocation on this path)
    int *dlay = tabs->d.layout + (size_t)tab0 * FT_LAY_STRIDE, *dslot = tabs->d.ids_scratch + (size_t)tab0 * tabs->d.F;
    HIPCHK(c, hipMemcpyAsync(dlay, tlayout.data(), sizeof(int) * tlayout.size(), hipMemcpyHostToDevice, c->stream));
    launch_ftab_pack(tabs->d, tabs->cur, tab0, B, d, dlay, dslot, c->stream);
    for (int w = 0; w < B; w++) {
      b->slot_of[w].resize(b->L[w]);
      if (b->L[w] > 0) HIPCHK(c, hipMemcpyAsync(b->slot_of[w].data(), dslot + (size_t)w * tabs->d.F, sizeof(int) * b->L[w], hipMemcpyDeviceToHost, c->stream));
    }
  }
  const double T2 = now();
  if (dbg_t) (void)hipStreamSynchronize(c->stream);
  const double T3 = now();
  { Timed t(c, "k_prep", 0); launch_prep(d, c->stream); launch_asm_table(d, c->stream); }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));   // host staging vectors die here
  if (dbg_t) fprintf(stderr, "upload: host pack %.3f ms, slab+copies enqueue %.3f ms, drain %.3f ms, k_prep+table %.3f ms (slab %.1f MB)\n", T1 - T0, T2 - T1, T3 - T2, now() - T3, b->slab_bytes / 1048576.0);
  return GFBE_OK;
}

extern "C" void gfbe_batch_free(gfbe_ctx *c, gfbe_batch *b);

// Batches of >= BATCH_SPLIT_MIN_B windows become `split_batch` parts (default two halves) solved side by side, each on
// its own pair of streams: a chain a -> a->second -> ...; part k + 1 runs on part k's `lane2`.
static gfbe_status make_lane(gfbe_ctx *c, gfbe_batch *a) {
  if (hipStreamCreateWithFlags(&a->lane2.s, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&a->lane2.aux, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&a->lane2.fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&a->lane2.join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&a->ev_start2, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&a->ev_done2, hipEventDisableTiming) != hipSuccess) {
    c->err = "hipStreamCreate/hipEventCreate (part of a split batch) failed";
    return GFBE_DEVICE_ERROR;
  }
  return GFBE_OK;
}
static gfbe_status upload_halves(gfbe_ctx *c, int32_t B, const gfbe_window *const *wins, gfbe_batch **out, gfbe_ftab *tabs) {
  if (!c || !wins || !out || B <= 0) return GFBE_BAD_INPUT;
  *out = nullptr;
  int parts = 1;
  if (B >= BATCH_SPLIT_MIN_B && c->world == 1 && c->opt.split_batch) parts = std::min(std::max(c->opt.split_batch, 2), std::max(B / DENSE_SPLIT_MIN_B, 1));
  gfbe_batch **link = out;
  gfbe_batch *prev = nullptr;
  int done = 0;
  gfbe_status st = GFBE_OK;
  for (int p = 0; p < parts && st == GFBE_OK; p++) {
    const int n = (B - done + (parts - p) - 1) / (parts - p);
    st = upload_one(c, n, wins + done, link, tabs, done);
    if (st == GFBE_OK && prev) st = make_lane(c, prev);
    if (st != GFBE_OK) break;
    prev = *link;
    link = &prev->second;
    done += n;
  }
  if (st != GFBE_OK) { gfbe_batch_free(c, *out); *out = nullptr; }
  return st;
}

// No C++ exception crosses the C ABI (host staging vectors can throw std::bad_alloc).
static gfbe_status upload_guarded(gfbe_ctx *c, int32_t B, const gfbe_window *const *wins, gfbe_batch **out, gfbe_ftab *tabs) {
  try {
    return upload_halves(c, B, wins, out, tabs);
  } catch (const std::exception &e) {
    if (c) c->err = std::string("gfbe_batch_upload: ") + e.what();
    if (out && *out) { gfbe_batch_free(c, *out); *out = nullptr; }
    return GFBE_BAD_INPUT;
  }
}

extern "C" gfbe_status gfbe_batch_upload(gfbe_ctx *c, int32_t B, const gfbe_window *const *wins, gfbe_batch **out) {
  return upload_guarded(c, B, wins, out, nullptr);
}

// Same as gfbe_batch_upload, with the visual factors of window w taken from table w of `t` on the device.
extern "C" gfbe_status gfbe_batch_upload_tables(gfbe_ctx *c, gfbe_ftab *t, int32_t B, const gfbe_window *const *wins, gfbe_batch **out) {
  if (!c || !t || !wins || !out || B <= 0) return GFBE_BAD_INPUT;
  if (c->world > 1) { c->err = "gfbe_batch_upload_tables: not available with landmark sharding"; return GFBE_BAD_INPUT; }
  return upload_guarded(c, B, wins, out, t);
}

extern "C" int32_t gfbe_batch_feature_count(const gfbe_batch *b, int32_t w) {
  if (w < 0) return -1;
  for (; b; b = b->second) {
    if (w < (int)b->L.size()) return b->L[w];
    w -= (int)b->L.size();
  }
  return -1;
}

extern "C" void gfbe_batch_free(gfbe_ctx *c, gfbe_batch *b) {
  if (!b) return;
  if (c && c->stream) (void)hipStreamSynchronize(c->stream);
  if (b->second) {
    if (b->lane2.s) (void)hipStreamSynchronize(b->lane2.s);
    gfbe_batch_free(c, b->second);
    if (b->lane2.s) (void)hipStreamDestroy(b->lane2.s);
    if (b->lane2.aux) (void)hipStreamDestroy(b->lane2.aux);
    for (hipEvent_t e : {b->lane2.fork, b->lane2.join, b->ev_start2, b->ev_done2}) if (e) (void)hipEventDestroy(e);
  }
  for (auto &g : b->graph) if (g) (void)hipGraphExecDestroy(g);
  slab_release(c, b);
  delete b;
}

// One linearisation of the whole batch at the current parameters (skipped on device for windows
// that only need a new radius).
static void enqueue_linearize(gfbe_ctx *c, gfbe_batch *b, const Lane &ln, bool first) {
  const BatchDev &d = b->d;
  // the first linearisation has every window active; later ones skip windows that only shrink the radius
  // fork: dense factors on the aux stream (serial and timed on the main stream when profiling)
  const bool overlap = !c->profiling && ln.aux && d.B >= DENSE_SPLIT_MIN_B;
  if (overlap) {
    (void)hipEventRecord(ln.fork, ln.s);
    (void)hipStreamWaitEvent(ln.aux, ln.fork, 0);
    launch_dense_factors(d, 0, 0, ln.aux);
    (void)hipEventRecord(ln.join, ln.aux);
  }
  const bool small = !c->profiling && d.B < DENSE_SPLIT_MIN_B;   // one launch for visual tiles + dense factors (k_lin_small)
  if (small) launch_lin_small(d, 0, ln.s);
  else {
    { Timed t(c, first ? "k_vis_lin_iter0" : "k_vis_lin", b->algo_bytes_lin); launch_vis(d, 0, ln.s); }
    if (!overlap) { Timed t(c, "k_dense", 0); launch_dense_factors(d, 0, 0, ln.s); }
  }
  if (d.tot_lio > 0) { Timed t(c, "k_lio_window", 0); launch_lio_window(d, 0, ln.s); }
  { Timed t(c, "k_schur", 0); launch_schur(d, 0, ln.s); }
  { Timed t(c, "k_visblock", 0); launch_visblock(d, ln.s); }
  if (overlap) (void)hipStreamWaitEvent(ln.s, ln.join, 0);   // join
  { Timed t(c, "k_assemble", 0); launch_assemble(d, ln.s); }
  if (d.world > 1) { Timed t(c, "allreduce_system", 0); c->allreduce(c->allreduce_user, d.H, (int64_t)b->slab_n, ln.s); }
  { Timed t(c, "k_solve", 0); launch_solve(d, ln.s); }
  { Timed t(c, "k_lm_step", 0); launch_lm_step(d, ln.s); }
  if (d.world > 1) {
    Timed t(c, "allreduce_scalars", 0);
    launch_xchg_gram(d, ln.s);
    c->allreduce(c->allreduce_user, d.xb, (int64_t)d.B * d.world * XCHG, ln.s);
  }
}

static gfbe_status enqueue_solve(gfbe_ctx *c, gfbe_batch *b, const Lane &ln, int32_t margin_flag) {
  const BatchDev &d = b->d;
  { Timed t(c, "k_reset", 0); launch_reset(d, ln.s); }
  const int iters = std::min(c->opt.max_num_iterations, 15);
  for (int it = 0; it < iters; it++) {
    enqueue_linearize(c, b, ln, it == 0);
    { Timed t(c, "k_step", 0); launch_step(d, ln.s); }
    { Timed t(c, "k_candidate", 0); launch_candidate(d, ln.s); }
    const bool overlap = !c->profiling && ln.aux && d.B >= DENSE_SPLIT_MIN_B;
    if (overlap) {
      (void)hipEventRecord(ln.fork, ln.s);
      (void)hipStreamWaitEvent(ln.aux, ln.fork, 0);
      launch_dense_factors(d, 1, 0, ln.aux);
      (void)hipEventRecord(ln.join, ln.aux);
    }
    const bool small = !c->profiling && d.B < DENSE_SPLIT_MIN_B;
    if (small) launch_lin_small(d, 1, ln.s);
    else { Timed t(c, "k_vis_cost", 0); launch_vis(d, 1, ln.s); }
    if (d.tot_lio > 0) { Timed t(c, "k_lio_window_cost", 0); launch_lio_window(d, 1, ln.s); }
    if (!small && !overlap) { Timed t(c, "k_dense_cost", 0); launch_dense_factors(d, 1, 0, ln.s); }
    else if (overlap) (void)hipStreamWaitEvent(ln.s, ln.join, 0);
    if (d.world > 1) {
      Timed t(c, "allreduce_scalars", 0);
      launch_xchg_cand(d, ln.s);
      c->allreduce(c->allreduce_user, d.xc, (int64_t)d.B * d.world * XCHG, ln.s);
    }
    { Timed t(c, "k_accept", 0); launch_accept(d, ln.s); }
  }
  { Timed t(c, "k_reanchor", 0); launch_reanchor(d, ln.s); }
  if (margin_flag != GFBE_MARGIN_NONE) {
    Timed t(c, "marginalize", 0);
    if (d.world > 1 && margin_flag == GFBE_MARGIN_OLD) {
      // the partials of the landmarks that start in frame 0 are summed over the ranks before k_marg reads them
      launch_marginalize_partials(d, ln.s);
      c->allreduce(c->allreduce_user, d.pair_part, (int64_t)d.B * NPAIR * VP_STRIDE, ln.s);
      c->allreduce(c->allreduce_user, d.schur_part, (int64_t)d.B * NF * SCHUR_STRIDE, ln.s);
      launch_marginalize_finish(d, margin_flag, ln.s);
    } else {
      launch_marginalize(d, margin_flag, ln.s);
    }
  }
  if (d.world > 1) {   // every rank ends with all inverse depths: owners contribute theirs, the others zeros
    launch_lam_mask(d, ln.s);
    c->allreduce(c->allreduce_user, d.lam, (int64_t)2 * d.tot_lm, ln.s);
  }
  return GFBE_OK;
}

extern "C" gfbe_status gfbe_batch_solve(gfbe_ctx *c, gfbe_batch *b, int32_t margin_flag) {
  if (!c || !b) return GFBE_BAD_INPUT;
  if (c->device < 0) return GFBE_NO_DEVICE;
  if (margin_flag < 0 || margin_flag > 2) return GFBE_BAD_INPUT;
  const BatchDev &d = b->d;
  if (d.world > 1 && !c->allreduce) { c->err = "batch was uploaded for landmark sharding but the all-reduce hook is gone"; return GFBE_BAD_INPUT; }
  // hipGraph replay: not while profiling (per-kernel events) and not with the all-reduce hook (host callback)
  // (not for split batches: capturing the cross-stream fork / join of the parts crashed the runtime on ROCm 7.2)
  const bool graphable = c->opt.use_graph && !c->profiling && d.world == 1 && !b->second;
  const Lane lane1 = {c->stream, c->aux, c->ev_fork, c->ev_join};
  // every part but the first on its own streams beside the first, joined back into the caller's stream
  auto enqueue_all = [&]() -> gfbe_status {
    gfbe_status st = GFBE_OK;
    if (b->second && !c->profiling) {
      (void)hipEventRecord(b->ev_start2, c->stream);
      for (gfbe_batch *p = b; p->second && st == GFBE_OK; p = p->second) {
        (void)hipStreamWaitEvent(p->lane2.s, b->ev_start2, 0);
        st = enqueue_solve(c, p->second, p->lane2, margin_flag);
        (void)hipEventRecord(p->ev_done2, p->lane2.s);
      }
      if (st == GFBE_OK) st = enqueue_solve(c, b, lane1, margin_flag);
      for (gfbe_batch *p = b; p->second; p = p->second) (void)hipStreamWaitEvent(c->stream, p->ev_done2, 0);
    } else {
      for (gfbe_batch *p = b; p && st == GFBE_OK; p = p->second) st = enqueue_solve(c, p, lane1, margin_flag);
    }
    return st;
  };
  if (graphable && b->graph[margin_flag]) {
    HIPCHK(c, hipGraphLaunch(b->graph[margin_flag], c->stream));
    return GFBE_OK;
  }
  if (graphable && b->calls[margin_flag]++ >= 1) {   // the first call ran eagerly (one-time attribute setup); capture now
    hipGraph_t g = nullptr;
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
      (void)enqueue_all();
      const hipError_t e = hipStreamEndCapture(c->stream, &g);
      if (e == hipSuccess && g && hipGraphInstantiate(&b->graph[margin_flag], g, nullptr, nullptr, 0) == hipSuccess) {
        (void)hipGraphDestroy(g);
        HIPCHK(c, hipGraphLaunch(b->graph[margin_flag], c->stream));
        return GFBE_OK;
      }
      if (g) (void)hipGraphDestroy(g);
      b->graph[margin_flag] = nullptr;
      (void)hipGetLastError();
    }
    c->opt.use_graph = 0;   // capture not available on this stream: stay eager
  }
  const gfbe_status st = enqueue_all();
  if (st != GFBE_OK) return st;
  HIPCHK(c, hipGetLastError());
  return GFBE_OK;
}

static gfbe_status download_one(gfbe_ctx *c, gfbe_batch *b, gfbe_state *out_state, double *const *out_feature,
                                gfbe_prior *const *prior_out, gfbe_summary *summary) {
  if (!c || !b) return GFBE_BAD_INPUT;
  const BatchDev &d = b->d;
  const int B = d.B;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<WinCtl> ctl(B);
  HIPCHK(c, hipMemcpy(ctl.data(), d.ctl, sizeof(WinCtl) * B, hipMemcpyDeviceToHost));
  if (out_state) HIPCHK(c, hipMemcpy(out_state, d.xout, sizeof(double) * NA * B, hipMemcpyDeviceToHost));
  if (out_feature) {
    std::vector<double> lam(2 * (size_t)d.tot_lm);
    HIPCHK(c, hipMemcpy(lam.data(), d.lam, sizeof(double) * lam.size(), hipMemcpyDeviceToHost));
    for (int w = 0; w < B; w++) {
      if (!out_feature[w]) continue;
      for (int l = 0; l < b->L[w]; l++) out_feature[w][l] = lam[(size_t)ctl[w].cur * d.tot_lm + b->slot_of[w][l]];
    }
  }
  if (prior_out) {
    std::vector<int> meta((size_t)B * (4 + 3 * GFBE_MAX_PRIOR_BLOCKS));
    HIPCHK(c, hipMemcpy(meta.data(), d.mmeta, sizeof(int) * meta.size(), hipMemcpyDeviceToHost));
    for (int w = 0; w < B; w++) {
      gfbe_prior *p = prior_out[w];
      if (!p) continue;
      const int *m = &meta[(size_t)w * (4 + 3 * GFBE_MAX_PRIOR_BLOCKS)];
      p->valid = m[0]; p->n = m[1]; p->n_blocks = m[2];
      if (!p->valid) continue;
      int xo = 0;
      for (int q = 0; q < p->n_blocks; q++) {
        p->block_id[q] = m[4 + q]; p->block_size[q] = m[4 + GFBE_MAX_PRIOR_BLOCKS + q]; p->block_idx[q] = m[4 + 2 * GFBE_MAX_PRIOR_BLOCKS + q];
        xo += p->block_size[q];
      }
      HIPCHK(c, hipMemcpy(p->x0, d.mx0 + (size_t)w * PRIOR_X0, sizeof(double) * xo, hipMemcpyDeviceToHost));
      HIPCHK(c, hipMemcpy(p->J0, d.mJ0 + (size_t)w * ND * ND, sizeof(double) * p->n * p->n, hipMemcpyDeviceToHost));
      HIPCHK(c, hipMemcpy(p->r0, d.mr0 + (size_t)w * ND, sizeof(double) * p->n, hipMemcpyDeviceToHost));
    }
  }
  gfbe_status worst = GFBE_OK;
  for (int w = 0; w < B; w++) {
    const WinCtl &k = ctl[w];
    if (summary) {
      gfbe_summary &s = summary[w];
      std::memset(&s, 0, sizeof s);
      s.status = k.status; s.iterations = k.iter; s.num_successful = k.num_successful; s.termination = k.termination;
      s.initial_cost = k.initial_cost; s.final_cost = k.cost; s.final_radius = k.radius;
      std::memcpy(s.cost_history, k.cost_history, sizeof s.cost_history);
      std::memcpy(s.accepted, k.accepted, sizeof s.accepted);
    }
    if (k.status == GFBE_NUMERICAL_FAILURE) worst = GFBE_NUMERICAL_FAILURE;
    else if (k.status == GFBE_NO_CONVERGENCE && worst == GFBE_OK) worst = GFBE_NO_CONVERGENCE;
  }
  if (worst == GFBE_NUMERICAL_FAILURE) c->err = "linear solve failed for every mu < 1 in at least one window";
  return worst;
}

extern "C" gfbe_status gfbe_batch_download(gfbe_ctx *c, gfbe_batch *b, gfbe_state *out_state, double *const *out_feature,
                                          gfbe_prior *const *prior_out, gfbe_summary *summary) {
  if (!c || !b) return GFBE_BAD_INPUT;
  gfbe_status worst = GFBE_OK;
  int done = 0;
  for (gfbe_batch *p = b; p; p = p->second) {
    const gfbe_status st = download_one(c, p, out_state ? out_state + done : nullptr, out_feature ? out_feature + done : nullptr,
                                        prior_out ? prior_out + done : nullptr, summary ? summary + done : nullptr);
    if (st > GFBE_NO_CONVERGENCE) return st;
    if (st > worst) worst = st;
    done += p->d.B;
  }
  return worst;
}

extern "C" gfbe_status gfbe_solve_batch(gfbe_ctx *c, int32_t n, const gfbe_window *const *win, int32_t margin_flag,
                                       gfbe_state *out_state, double *const *out_feature, gfbe_prior *const *prior_out,
                                       gfbe_summary *summary) {
  gfbe_batch *b = nullptr;
  gfbe_status st = gfbe_batch_upload(c, n, win, &b);
  if (st == GFBE_OK) st = gfbe_batch_solve(c, b, margin_flag);
  if (st == GFBE_OK) st = gfbe_batch_download(c, b, out_state, out_feature, prior_out, summary);
  gfbe_batch_free(c, b);
  return st;
}

extern "C" gfbe_status gfbe_solve_window(gfbe_ctx *c, const gfbe_window *win, int32_t margin_flag, gfbe_state *out_state,
                                        double *out_feature, gfbe_prior *prior_out, gfbe_summary *summary) {
  const gfbe_window *wins[1] = {win};
  double *feat[1] = {out_feature};
  gfbe_prior *pr[1] = {prior_out};
  if (margin_flag != GFBE_MARGIN_NONE && win && win->frame_count < GFBE_WINDOW_SIZE) margin_flag = GFBE_MARGIN_NONE;  // estimator.cpp:3391
  return gfbe_solve_batch(c, 1, wins, margin_flag, out_state, feat, prior_out ? pr : nullptr, summary);
}

// ---------------------------------------------------------------------------------------------
// Factor evaluation with block-CSR output (parity / inspection API).
// ---------------------------------------------------------------------------------------------
extern "C" gfbe_status gfbe_eval_factors(gfbe_ctx *c, const gfbe_window *win, int32_t robustify, double *vis_r, double *vis_J,
                                        double *imu_r, double *imu_J, double *wheel_r, double *wheel_J, double *prior_r,
                                        double *cost) {
  if (!c || !win) return GFBE_BAD_INPUT;
  gfbe_batch *b = nullptr;
  const gfbe_window *wins[1] = {win};
  const gfbe_options keep = c->opt;
  if (!robustify) c->opt.huber_delta = 1e150;   // rho(s) = s everywhere: the corrector becomes the identity
  gfbe_status st = gfbe_batch_upload(c, 1, wins, &b);
  c->opt = keep;
  if (st != GFBE_OK) { gfbe_batch_free(c, b); return st; }
  BatchDev &d = b->d;
  launch_reset(d, c->stream);
  launch_vis(d, 0, c->stream, /*write_records=*/1);
  launch_dense_factors(d, 0, 1, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int K = win->vis.n_factor;
  std::vector<WinDesc> ds(1);
  HIPCHK(c, hipMemcpy(ds.data(), d.desc, sizeof(WinDesc), hipMemcpyDeviceToHost));
  if (vis_r && K > 0) {
    std::vector<double> rec((size_t)K * REC);
    std::vector<int> lm_rec((size_t)MAXOBS * d.tot_lm);
    HIPCHK(c, hipMemcpy(rec.data(), d.rec, sizeof(double) * rec.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(lm_rec.data(), d.lm_rec, sizeof(int) * lm_rec.size(), hipMemcpyDeviceToHost));
    for (int k = 0; k < K; k++) {
      const int l = win->vis.feature_index[k], slot = b->slot_of[0][l];
      const int obs = win->vis.imu_j[k] - win->vis.imu_i[k] - 1;
      const double *r = &rec[(size_t)lm_rec[(size_t)obs * d.tot_lm + slot] * REC];
      vis_r[2 * k] = r[0]; vis_r[2 * k + 1] = r[1];
      if (vis_J) { std::memcpy(vis_J + 40 * (size_t)k, r + 2, sizeof(double) * 20); std::memcpy(vis_J + 40 * (size_t)k + 20, r + 22, sizeof(double) * 20); }
    }
  }
  if (imu_r && win->n_imu > 0) {
    std::vector<double> dbg((size_t)MAX_IMU * 15 * 31);
    HIPCHK(c, hipMemcpy(dbg.data(), d.dbg_imu, sizeof(double) * dbg.size(), hipMemcpyDeviceToHost));
    for (int k = 0; k < win->n_imu; k++) {
      std::memcpy(imu_r + 15 * k, &dbg[(size_t)k * 15 * 31], sizeof(double) * 15);
      if (imu_J) std::memcpy(imu_J + 450 * k, &dbg[(size_t)k * 15 * 31 + 15], sizeof(double) * 450);
    }
  }
  if (wheel_r && win->n_wheel > 0) {
    std::vector<double> dbg((size_t)MAX_WHEEL * 6 * 23);
    HIPCHK(c, hipMemcpy(dbg.data(), d.dbg_wheel, sizeof(double) * dbg.size(), hipMemcpyDeviceToHost));
    for (int k = 0; k < win->n_wheel; k++) {
      std::memcpy(wheel_r + 6 * k, &dbg[(size_t)k * 6 * 23], sizeof(double) * 6);
      if (wheel_J) std::memcpy(wheel_J + 132 * k, &dbg[(size_t)k * 6 * 23 + 6], sizeof(double) * 132);
    }
  }
  if (prior_r && ds[0].prior_n > 0) HIPCHK(c, hipMemcpy(prior_r, d.dbg_prior, sizeof(double) * ds[0].prior_n, hipMemcpyDeviceToHost));
  if (cost) {
    double total = 0.0;
    std::vector<double> tc(std::max(d.max_tiles, 1)), ip((size_t)MAX_IMU * IMU_PART), wp((size_t)MAX_WHEEL * WHEEL_PART), pg(ND + 2);
    HIPCHK(c, hipMemcpy(tc.data(), d.tile_cost, sizeof(double) * tc.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(ip.data(), d.imu_part, sizeof(double) * ip.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(wp.data(), d.wheel_part, sizeof(double) * wp.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(pg.data(), d.prior_g, sizeof(double) * pg.size(), hipMemcpyDeviceToHost));
    for (int q = 0; q < ds[0].n_tiles; q++) total += tc[q];
    for (int q = 0; q < win->n_imu; q++) total += ip[(size_t)q * IMU_PART + IMU_PART - 2];
    for (int q = 0; q < win->n_wheel; q++) total += wp[(size_t)q * WHEEL_PART + WHEEL_PART - 2];
    total += pg[ND];
    *cost = total;
  }
  gfbe_batch_free(c, b);
  return GFBE_OK;
}

// ---------------------------------------------------------------------------------------------
// Pre-integration (a6 / a8)
// ---------------------------------------------------------------------------------------------
template <typename REC_T>
static gfbe_status preint_common(gfbe_ctx *c, int n, const int32_t *offset, const double *samples, const double *first,
                                 const double *lin, int lin_w, const double *noise, int n_noise, REC_T *out, bool imu) {
  if (!c || n <= 0 || !offset || !samples || !out) return GFBE_BAD_INPUT;
  if (c->device < 0 || !c->stream) { c->err = "HIP device context required (no CPU fallback)"; return GFBE_NO_DEVICE; }
  const int tot = offset[n];
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t b_off = al(sizeof(int) * (n + 1)), b_s = al(sizeof(double) * 7 * std::max(tot, 1)), b_f = al(sizeof(double) * 6 * n),
               b_l = al(sizeof(double) * lin_w * n), b_n = al(sizeof(double) * 4), b_o = al(sizeof(REC_T) * n);
  const size_t need = b_off + b_s + b_f + b_l + b_n + b_o;
  if (!ctx_scratch(c, need)) { c->err = "hipMalloc(pre-integration scratch) failed"; return GFBE_DEVICE_ERROR; }
  char *base = c->scratch;
  int *d_off = (int *)base; base += b_off;
  double *d_s = (double *)base; base += b_s;
  double *d_f = (double *)base; base += b_f;
  double *d_l = (double *)base; base += b_l;
  double *d_n = (double *)base; base += b_n;
  REC_T *d_o = (REC_T *)base;
  HIPCHK(c, hipMemcpyAsync(d_off, offset, sizeof(int) * (n + 1), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_s, samples, sizeof(double) * 7 * tot, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_f, first, sizeof(double) * 6 * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_l, lin, sizeof(double) * lin_w * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_n, noise, sizeof(double) * n_noise, hipMemcpyHostToDevice, c->stream));
  if (imu) launch_preint_imu(n, d_off, d_s, d_f, d_l, d_n, (gfbe_imu_preint *)d_o, c->stream);
  else launch_preint_wheel(n, d_off, d_s, d_f, d_l, d_n, (gfbe_wheel_preint *)d_o, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(out, d_o, sizeof(REC_T) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GFBE_OK;
}

extern "C" gfbe_status gfbe_preintegrate_imu(gfbe_ctx *c, int32_t n, const int32_t *offset, const double *samples,
                                            const double *first, const double *lin, const double noise[4], gfbe_imu_preint *out) {
  return preint_common(c, n, offset, samples, first, lin, 6, noise, 4, out, true);
}
extern "C" gfbe_status gfbe_preintegrate_wheel(gfbe_ctx *c, int32_t n, const int32_t *offset, const double *samples,
                                              const double *first, const double *lin, const double noise[2], gfbe_wheel_preint *out) {
  return preint_common(c, n, offset, samples, first, lin, 4, noise, 2, out, false);
}

// diagnostics: copy the k_solve phase stamps of window w (32 doubles, 10 ns ticks)
extern "C" gfbe_status gfbe_debug_timing(gfbe_ctx *c, gfbe_batch *b, int32_t w, double *out32) {
  if (b && b->second && w >= b->d.B) return gfbe_debug_timing(c, b->second, w - b->d.B, out32);
  if (!c || !b || !out32 || w < 0 || w >= b->d.B) return GFBE_BAD_INPUT;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(out32, b->d.timing + (size_t)w * 32, sizeof(double) * 32, hipMemcpyDeviceToHost));
  return GFBE_OK;
}

// ---------------------------------------------------------------------------------------------
// profiling hooks
// ---------------------------------------------------------------------------------------------
extern "C" gfbe_status gfbe_profile_enable(gfbe_ctx *c, int32_t on) { if (!c) return GFBE_BAD_INPUT; c->profiling = on != 0; return GFBE_OK; }
extern "C" int32_t gfbe_profile_count(const gfbe_ctx *c) { return c ? (int32_t)c->prof.size() : 0; }
extern "C" gfbe_status gfbe_profile_get(const gfbe_ctx *cc, int32_t i, const char **name, int64_t *launches, double *total_ms, double *bytes) {
  gfbe_ctx *c = const_cast<gfbe_ctx *>(cc);
  if (!c || i < 0 || i >= (int)c->prof.size()) return GFBE_BAD_INPUT;
  prof_collect(c);
  if (name) *name = c->prof[i].name.c_str();
  if (launches) *launches = c->prof[i].launches;
  if (total_ms) *total_ms = c->prof[i].total_ms;
  if (bytes) *bytes = c->prof[i].bytes;
  return GFBE_OK;
}
extern "C" void gfbe_profile_reset(gfbe_ctx *c) {
  if (!c) return;
  prof_collect(c);
  for (auto &p : c->prof) { p.launches = 0; p.total_ms = 0; p.bytes = 0; }
}

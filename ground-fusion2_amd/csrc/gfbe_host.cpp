// gfbe_host.cpp — the C ABI of include/gfbe.h: context, packing of windows into the HBM layout
// (gfbe_device.h), the fixed kernel sequence of one optimization() call, and the host-side
// landmark bookkeeping. Host code only orchestrates: all arithmetic of the hot path runs in the
// HIP kernels (gfbe_kernels.hip, gfbe_marg.hip, gfbe_preint.hip). There is no CPU fallback.
//
// Reference being replaced: Estimator::optimization()
//   Ground-Fusion++/vins_estimator/src/estimator/estimator.cpp:2951-3698
#include "gfbe_device.h"
#ifndef GFBE_CLEAR_LM
#define GFBE_CLEAR_LM 0
#endif
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <chrono>
#include <string>
#include <vector>

using namespace gfd;
static_assert(sizeof(gfbe_state) == sizeof(double) * NA, "gfbe_state layout: NA doubles");


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

// Host threads of the upload / download packing: a fixed set of workers, one parallel-for at a time, the caller takes part.
class HostPool {
 public:
  explicit HostPool(int n) { for (int i = 0; i < n; i++) th_.emplace_back([this] { loop(); }); }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; gen_++; }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  int size() const { return (int)th_.size(); }
  // fn(i) for i in [0, n) on up to `helpers` workers + the calling thread; returns when all are done. Only as many workers are woken as the
  // job wants (round 6: a 47-thread pool sized for the packing passes woke every thread for the 15-helper unpacking of a download — 1.2 ms of
  // work became 3.4 - 4.3): a worker JOINS a job under the lock while places are left, the caller closes the job (no places) before it waits
  // for those who joined.
  void run(int n, int helpers, const std::function<void(int)> &fn) {
    helpers = std::min(helpers, (int)th_.size());
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn; n_ = n; next_.store(0); places_ = helpers; running_ = 0; gen_++;
    }
    for (int i = 0; i < helpers; i++) cv_.notify_one();
    for (int i; (i = next_.fetch_add(1)) < n;) fn(i);
    std::unique_lock<std::mutex> lk(m_);
    places_ = 0;
    done_.wait(lk, [this] { return running_ == 0; });
    fn_ = nullptr;
  }

 private:
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      const std::function<void(int)> *fn;
      int n;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        if (places_ <= 0) continue;     // the job is full, or already closed
        places_--; running_++;
        fn = fn_; n = n_;
      }
      for (int i; (i = next_.fetch_add(1)) < n;) (*fn)(i);
      bool last;
      { std::lock_guard<std::mutex> lk(m_); last = --running_ == 0; }
      if (last) done_.notify_all();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)> *fn_ = nullptr;
  int n_ = 0, places_ = 0, running_ = 0;
  std::atomic<int> next_{0};
  unsigned long long gen_ = 0;
  bool stop_ = false;
};

// Streams are taken from a process-wide pool per device and handed back at gfbe_destroy, never destroyed (round 6). The HIP runtime deals
// streams to its hardware queues in creation order over the life of the process: the first context's eight streams land on eight queues,
// a later context's — created after the first one's were destroyed — do not (measured: the same workload at 113.7k solves/s on the first
// context of a process and at 105 - 106k on every later one, whatever library it came from). With the pool a later context runs on the
// streams — and queues — the first one had.
namespace {
struct StreamPool {
  struct Slot { hipStream_t s; bool busy; };
  std::mutex m;
  std::vector<Slot> blocking[16], nonblocking[16];
};
StreamPool &stream_pool() { static StreamPool p; return p; }
// The free stream created EARLIEST is handed out first, so a context that asks in the same order as its predecessor (stream, aux, copy,
// dl, then the lanes as split batches appear) gets the same stream in the same role.
hipError_t stream_acquire(int device, bool nonblock, hipStream_t *out) {
  const bool pooled = device >= 0 && device < 16;
  StreamPool &p = stream_pool();
  std::lock_guard<std::mutex> lk(p.m);
  if (pooled) {
    for (auto &sl : nonblock ? p.nonblocking[device] : p.blocking[device])
      if (!sl.busy) { sl.busy = true; *out = sl.s; return hipSuccess; }
  }
  const hipError_t e = nonblock ? hipStreamCreateWithFlags(out, hipStreamNonBlocking) : hipStreamCreate(out);
  if (e == hipSuccess && pooled) (nonblock ? p.nonblocking[device] : p.blocking[device]).push_back({*out, true});
  return e;
}
void stream_release(int device, bool nonblock, hipStream_t s) {
  if (!s) return;
  (void)hipStreamSynchronize(s);
  if (device >= 0 && device < 16) {
    StreamPool &p = stream_pool();
    std::lock_guard<std::mutex> lk(p.m);
    for (auto &sl : nonblock ? p.nonblocking[device] : p.blocking[device])
      if (sl.s == s) { sl.busy = false; return; }
  }
  (void)hipStreamDestroy(s);
}
}  // namespace

struct gfbe_ctx {
  int device = -1;
  gfbe_options opt;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // second stream: the inertial / wheel / prior factors (few, latency-bound workgroups) run beside the visual kernels
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  std::string err;
  std::string note;                                   // set by gfbe_create (never an error): gfbe_create_note
  double host_ms[4] = {0.0, 0.0, 0.0, 0.0};           // gfbe_host_times: upload pack | upload rest | download wait | download unpack
  bool profiling = false;
  std::vector<ProfEntry> prof;
  std::vector<hipEvent_t> event_pool;
  int *asm_full = nullptr, *asm_compact = nullptr;    // the window-independent assembly tables (asm_tables_build, once per context)
  int asm_compact_n = 0;
  std::vector<hipEvent_t> sync_event_pool;            // hipEventDisableTiming events of freed batches (three per batch: created and destroyed per
                                                      // gfbe_solve_window call they cost ~50 us of its 1.45 ms)
  gfbe_allreduce_fn allreduce = nullptr;
  void *allreduce_user = nullptr;
  int allreduce_rc = 0;                               // first non-zero return of the hook during the current gfbe_batch_solve
  int rank = 0, world = 1;
  // Device memory of freed batches, kept for the next upload: one window per camera frame is the reference's call
  // pattern, and ~70 hipMalloc / hipFree pairs per call cost more than its solve (3.6 ms of 4.1 ms measured).
  std::vector<std::pair<void *, size_t>> slab_cache;
  // grow-only device scratch of the short host-buffer calls (pre-integration): no hipMalloc / hipFree per call
  char *scratch = nullptr;
  size_t scratch_cap = 0;
  char *scratch_pin = nullptr;                        // pinned host mirror for the pre-integration calls (their own, smaller, size)
  size_t scratch_pin_cap = 0;
  // host <-> device hand-over beside the solves: uploads (one H2D copy + preparation kernels) run on `copy`, downloads
  // (gather kernel + one D2H copy) on `dl`, so that batch k + 1 is uploaded and batch k - 1 downloaded while batch k solves
  hipStream_t copy = nullptr, dl = nullptr;
  std::vector<std::pair<void *, size_t>> pin_cache;   // pinned host staging of freed batches
  std::unique_ptr<HostPool> pool;                     // packing threads (gfbe_options.host_threads)
  bool want_records = false;                          // gfbe_eval_factors: allocate the block-CSR record array
  // streams / events of the extra parts of split batches, recycled (creating and destroying two streams and four events
  // per uploaded batch cost ~2 ms of host time and synchronised with the device)
  struct LaneSet { hipStream_t s, aux; hipEvent_t fork, join, start2, done2; };
  std::vector<LaneSet> lane_pool;
};
// (a 1024-window batch of 2k-landmark windows is a 7 GB slab; 288 GB of HBM leave room to keep a few)
enum : size_t { SLAB_CACHE_ENTRIES = 6, SLAB_CACHE_MAX_BYTES = (size_t)48 << 30, PIN_CACHE_ENTRIES = 12, PIN_CACHE_MAX_BYTES = (size_t)8 << 30 };

// The streams / events one (sub-)batch runs on: main stream, the aux stream of its dense factors, fork / join events.
struct Lane { hipStream_t s, aux; hipEvent_t fork, join; };

struct gfbe_batch {
  BatchDev d;
  // every device array of the batch is carved from ONE slab: a dry pass over the allocation sequence adds up the sizes,
  // the slab comes from the context's cache (or hipMalloc), the second pass hands out the pointers
  char *slab = nullptr;
  size_t slab_bytes = 0, slab_off = 0;
  bool dry = false;
  std::vector<std::vector<int>> slot_of;   // per window: ABI landmark -> global slot (host-fed batches)
  std::vector<int> L;
  std::vector<unsigned char> anchor_only;      // window: MARGIN_SECOND_NEW meets an invalid prior that lists Pose[WINDOW_SIZE-1] (estimator.cpp:3622-3632)
  // upload region [0, up_end) of the slab = the pinned mirror up_h; [up_end, zero_end) is cleared; the rest is written before read
  char *up_h = nullptr;
  size_t up_cap = 0, up_bytes = 0, up_end = 0, zero_end = 0;
  struct PoisonEntry { const char *name; size_t off, bytes; };
  std::vector<PoisonEntry> poison_list;     // (GFBE_POISON_UNCLEARED test hook: the slab's arrays by name)
  // results: [dl_fix | dl_feat | dl_J0] at the end of the slab -> dl_h (pinned) in one copy
  char *dl_h = nullptr;
  char *lay_h = nullptr;                   // table-fed batches: the pinned source of the layout table's copy
  size_t lay_cap = 0;
  size_t dl_cap = 0, dl_bytes = 0;
  std::vector<int> feat_off;
  std::vector<long long> j0_off;
  std::vector<double> up_win_bytes;
  hipEvent_t ev_up = nullptr, ev_done = nullptr, ev_dl = nullptr;   // upload complete / last solve complete / results on the host
  int last_flag = GFBE_MARGIN_NONE;
  bool fetched = false;                    // dl_h holds the results of the last solve
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
  // (head of a split batch only) the parts hold the windows sorted by size, see upload_halves: order[k] = the caller's index of the
  // window at position k of the chain of parts, place[i] = the position of the caller's window i; empty: identity
  std::vector<int> order, place;
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
// grow-only PINNED host mirror of that scratch (the staging buffer of the calls that move host arrays through it in one copy each way)
void *ctx_scratch_pinned(gfbe_ctx *c, size_t bytes) {
  if (bytes > c->scratch_pin_cap) {
    (void)hipStreamSynchronize(c->stream);
    if (c->scratch_pin) (void)hipHostFree(c->scratch_pin);
    c->scratch_pin = nullptr; c->scratch_pin_cap = 0;
    const size_t cap = bytes + bytes / 2;
    if (hipHostMalloc((void **)&c->scratch_pin, cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    c->scratch_pin_cap = cap;
  }
  return c->scratch_pin;
}
}  // namespace gfd

extern "C" {

int32_t gfbe_options_size(void) { return (int32_t)sizeof(gfbe_options); }

void gfbe_default_options(gfbe_options *o) {
  o->struct_size = (int32_t)sizeof(gfbe_options);
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
  o->split_batch = 1;                        // batches of >= 128 windows run as 2..4 parts side by side, each on its own pair of streams
  o->use_graph = 0;                          // 1: replay the fixed launch sequence of gfbe_batch_solve as a hipGraph (measured: no gain)
  o->max_solver_time_in_seconds = 0.0;       // no cap (the reference: SOLVER_TIME = 0.04, estimator.cpp:3369-3376)
  o->host_threads = 0;                       // packing threads: min(hardware threads, 24)
  o->solve_kernel = 0;                       // chain-eliminated factorisation of the reduced system (k_solve_chain_tw below 32 windows, k_solve_chain from there)
  o->test_fail_chol_iter = 0; o->test_fail_chol_count = 1;
  o->speculative_linearization = 1;
  o->merge_lin_schur = 0;                    // 1: throughput batches run k_linschur (evaluation + landmark elimination in one launch) — measured slower, include/gfbe.h
  o->sharded_mu_retries = 1;                 // (8 = DoglegStrategy's whole mu ladder; every retry is three more launches per linearisation)
}

const char *gfbe_version(void) { return "gfbe 0.1.0 (gfx950, HIP)"; }
const char *gfbe_last_error(const gfbe_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }
const char *gfbe_create_note(const gfbe_ctx *ctx) { return ctx ? ctx->note.c_str() : ""; }
void gfbe_host_times(const gfbe_ctx *ctx, double *out4) { if (out4) for (int q = 0; q < 4; q++) out4[q] = ctx ? ctx->host_ms[q] : 0.0; }

gfbe_status gfbe_create(gfbe_ctx **out, int device, const gfbe_options *opt) {
  if (!out) return GFBE_BAD_INPUT;
  gfbe_ctx *c = new gfbe_ctx();
  *out = c;
  // (only the first member is read before the size is known to be the library's own)
  if (opt && opt->struct_size != (int32_t)sizeof(gfbe_options)) {
    c->device = -1;
    gfbe_default_options(&c->opt);
    c->err = "options ABI mismatch: gfbe_options.struct_size = " + std::to_string(opt->struct_size) + ", the library's is " +
             std::to_string(sizeof(gfbe_options)) + " (start from gfbe_default_options of the header this library was built with)";
    return GFBE_BAD_INPUT;
  }
  if (opt) c->opt = *opt; else gfbe_default_options(&c->opt);
  c->device = device;
  *out = c;
  if (device < 0) return GFBE_OK;   // host-only context: bookkeeping entry points only
  // A context drives up to eight streams (two parts x (main + dense-factor stream), upload, download, the caller's): with the
  // HIP runtime's default of four hardware queues they share queues and an upload queues up behind a whole solve. The queue
  // count is a process-wide setting read when the runtime initialises, so it belongs to the caller (the Python
  // host layer and bench.py set GPU_MAX_HW_QUEUES=8 before they load HIP): the library only says so when it finds less.
  {
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    if (!q || atoi(q) < 8) c->note = "GPU_MAX_HW_QUEUES is unset or below 8 — uploads and downloads will share hardware queues with the solver streams (set it before HIP initialises)";
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= device) {
    c->err = "no HIP device " + std::to_string(device) + " visible (the HIP back end has no CPU fallback)";
    return GFBE_NO_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess || stream_acquire(device, false, &c->stream) != hipSuccess) {
    c->err = "hipSetDevice/hipStreamCreate failed";
    return GFBE_DEVICE_ERROR;
  }
  c->own_stream = true;
  if (stream_acquire(device, true, &c->aux) != hipSuccess ||
      stream_acquire(device, true, &c->copy) != hipSuccess ||
      stream_acquire(device, true, &c->dl) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
    c->err = "hipStreamCreate/hipEventCreate (aux stream) failed";
    return GFBE_DEVICE_ERROR;
  }
  // kernel attributes are per device: every context sets them for its own GPU
  hipError_t ea = kernels_init_device();
  if (ea == hipSuccess) ea = marg_init_device();
  if (ea == hipSuccess) ea = gnss_init_device();
  if (ea == hipSuccess) ea = dense_init_device();
  if (ea == hipSuccess) ea = lin_small_init_device();
  if (ea != hipSuccess) { c->err = std::string("hipFuncSetAttribute(dynamic LDS): ") + hipGetErrorString(ea); return GFBE_DEVICE_ERROR; }
  ea = asm_tables_build(&c->asm_full, &c->asm_compact, &c->asm_compact_n, c->stream);
  if (ea != hipSuccess) { c->err = std::string("assembly tables (hipMalloc / k_asm_table / k_asm_compact): ") + hipGetErrorString(ea); return GFBE_DEVICE_ERROR; }
  return GFBE_OK;
}

void gfbe_destroy(gfbe_ctx *c) {
  if (!c) return;
  for (auto e : c->event_pool) (void)hipEventDestroy(e);
  for (auto e : c->sync_event_pool) (void)hipEventDestroy(e);
  if (c->asm_full) (void)hipFree(c->asm_full);
  if (c->asm_compact) (void)hipFree(c->asm_compact);
  for (auto &p : c->prof) for (auto &ev : p.pending) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  if (c->own_stream && c->stream) stream_release(c->device, false, c->stream);
  stream_release(c->device, true, c->aux);
  stream_release(c->device, true, c->copy);
  stream_release(c->device, true, c->dl);
  for (auto &sl : c->pin_cache) (void)hipHostFree(sl.first);
  for (auto &l : c->lane_pool) {
    stream_release(c->device, true, l.s); stream_release(c->device, true, l.aux);
    for (hipEvent_t e : {l.fork, l.join, l.start2, l.done2}) (void)hipEventDestroy(e);
  }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  for (auto &sl : c->slab_cache) (void)hipFree(sl.first);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->scratch_pin) (void)hipHostFree(c->scratch_pin);
  delete c;
}

gfbe_status gfbe_set_stream(gfbe_ctx *c, void *s) {
  if (!c || c->device < 0) return GFBE_NO_DEVICE;
  if (c->own_stream && c->stream) stream_release(c->device, false, c->stream);
  if (s) { c->stream = (hipStream_t)s; c->own_stream = false; }
  else { if (stream_acquire(c->device, false, &c->stream) != hipSuccess) return GFBE_DEVICE_ERROR; c->own_stream = true; }
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

// Two passes over one allocation sequence: the dry pass adds up the sizes, the slab comes from the context's cache (or
// hipMalloc), the second pass hands out the pointers. The arrays the host fills come FIRST ("upload region"): they have a
// mirror at the same offsets in one pinned host buffer, the packing threads write straight into that mirror, and the whole
// region crosses PCIe as ONE hipMemcpyAsync. Only the arrays the kernels expect zeroed are cleared (one hipMemsetAsync).
template <typename T>
gfbe_status dev_alloc(gfbe_ctx *c, gfbe_batch *b, T **p, size_t n) {
  const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
  if (b->dry) { b->slab_bytes += bytes; *p = nullptr; return GFBE_OK; }
  if (b->slab_off + bytes > b->slab_bytes) { c->err = "batch slab overrun"; return GFBE_DEVICE_ERROR; }
  *p = (T *)(b->slab + b->slab_off);
  b->slab_off += bytes;
  return GFBE_OK;
}
// an array of the upload region: device pointer + its pinned host mirror (valid in the second pass)
template <typename T>
gfbe_status up_alloc(gfbe_ctx *c, gfbe_batch *b, T **p, T **h, size_t n) {
  const size_t off = b->dry ? b->slab_bytes : b->slab_off;
  gfbe_status st = dev_alloc(c, b, p, n);
  if (st != GFBE_OK) return st;
  if (b->dry) { *h = nullptr; b->up_bytes = b->slab_bytes; }
  else *h = (T *)(b->up_h + off);
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
  return GFBE_OK;
}
void slab_release(gfbe_ctx *c, gfbe_batch *b) {
  if (!b->slab) return;
  size_t cached = 0;
  if (c) for (auto &sl : c->slab_cache) cached += sl.second;
  if (c && cached + b->slab_bytes <= SLAB_CACHE_MAX_BYTES) {
    if (c->slab_cache.size() >= SLAB_CACHE_ENTRIES) { (void)hipFree(c->slab_cache.front().first); c->slab_cache.erase(c->slab_cache.begin()); }
    c->slab_cache.emplace_back(b->slab, b->slab_bytes);
  } else {
    (void)hipFree(b->slab);
  }
  b->slab = nullptr;
}
// pinned host staging (upload mirror / download buffer): hipHostMalloc costs ~1 ms per 10 MB, so the buffers are cached too
char *pin_acquire(gfbe_ctx *c, size_t bytes, size_t *cap_out) {
  int best = -1;
  for (size_t i = 0; i < c->pin_cache.size(); i++) {
    const size_t cap = c->pin_cache[i].second;
    if (cap >= bytes && cap <= 2 * bytes + ((size_t)1 << 20) && (best < 0 || cap < c->pin_cache[best].second)) best = (int)i;
  }
  if (best >= 0) {
    char *q = (char *)c->pin_cache[best].first;
    *cap_out = c->pin_cache[best].second;
    c->pin_cache.erase(c->pin_cache.begin() + best);
    return q;
  }
  size_t grain = (size_t)1 << 16;
  while (grain * 8 <= bytes) grain *= 2;
  const size_t cap = (bytes + grain - 1) / grain * grain;
  void *q = nullptr;
  if (hipHostMalloc(&q, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  *cap_out = cap;
  return (char *)q;
}
void pin_release(gfbe_ctx *c, char *p, size_t cap) {
  if (!p) return;
  size_t cached = 0;
  if (c) for (auto &sl : c->pin_cache) cached += sl.second;
  if (c && cached + cap <= PIN_CACHE_MAX_BYTES) {
    if (c->pin_cache.size() >= PIN_CACHE_ENTRIES) { (void)hipHostFree(c->pin_cache.front().first); c->pin_cache.erase(c->pin_cache.begin()); }
    c->pin_cache.emplace_back(p, cap);
  } else {
    (void)hipHostFree(p);
  }
}

// One window per task on the context's host threads (the caller takes part); n == 1 or host_threads == 1 runs inline.
// heavy: the packing passes of an upload (~200 us of scan + fill per 2k-landmark window) — the default thread count is higher there than
// for the unpacking of a download (~5 us per window: more than 16 threads only add wake-ups).
void host_parallel(gfbe_ctx *c, int n, const std::function<void(int)> &fn, bool heavy = false) {
  const int hc = (int)std::max(1u, std::thread::hardware_concurrency());
  // (measured, 1024 windows per batch on a 2 x 64-core host, ONE count for both: 8 threads 40.8k, 16: 58.6k, 24: 69.5k, 32: 62-68k, 64: 50.4k,
  //  128: 35.5k solves/s end to end. Round 6, the two jobs apart, ms per batch at 8 / 16 / 24 / 32 / 48 / 64 threads: packing 15.4 / 7.9 /
  //  6.3 / 5.7 / 5.0 / 9.6, unpacking 1.3 / 1.2 / 1.4 / 2.4 / 3.6 / 1.0: 48 for the first when the host has the cores, 16 for the second)
  int want = c->opt.host_threads > 0 ? c->opt.host_threads : (heavy ? (hc >= 128 ? 48 : std::min(24, hc)) : std::min(16, hc));
  want = std::min(want, n);
  if (want <= 1) { for (int i = 0; i < n; i++) fn(i); return; }
  if (!c->pool || c->pool->size() < want - 1) c->pool.reset(new HostPool(want - 1));
  c->pool->run(n, want - 1, fn);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Upload: pack windows into the device layout.
// ---------------------------------------------------------------------------------------------
namespace {
// what the scan of one window's factor list leaves for the fill pass
struct WinScan {
  int L = 0, K = 0, slots = 0, n_tiles = 0;
  int sf_tile_begin[NF + 1];
  int pair_begin[NPAIR + 1];
  std::vector<int> slot_rel;      // ABI landmark -> slot relative to the window's first slot
  std::vector<unsigned char> lstart, lm;
  std::string err;
};

// Validates the visual factor list of one window and lays its landmarks out: groups by start frame (tile aligned), inside a
// group longer tracks first, ties in ABI order (a stable counting sort over (start, m) bins).
bool scan_window(const gfbe_window &win, int w, WinScan &sc) {
  const int L = win.n_feature, K = win.vis.n_factor;
  sc.L = L; sc.K = K;
  auto fail = [&](const std::string &m) { sc.err = "window " + std::to_string(w) + ": " + m; return false; };
  if (L > 0 && !win.para_Feature) return fail("para_Feature is null");
  if (K > 0 && (!win.vis.feature_index || !win.vis.imu_i || !win.vis.imu_j || !win.vis.pts_i || !win.vis.pts_j || !win.vis.vel_i ||
                !win.vis.vel_j || !win.vis.td_i || !win.vis.td_j)) return fail("null visual factor array");
  sc.lstart.assign(L, 255); sc.lm.assign(L, 0);
  std::vector<unsigned short> mask(L, 0);
  int pair_cnt[NPAIR];
  for (int p = 0; p < NPAIR; p++) pair_cnt[p] = 0;
  for (int k = 0; k < K; k++) {
    const int l = win.vis.feature_index[k], i = win.vis.imu_i[k], j = win.vis.imu_j[k];
    if (l < 0 || l >= L || i < 0 || j <= i || j > win.frame_count) return fail("bad visual factor " + std::to_string(k));
    if (sc.lstart[l] == 255) sc.lstart[l] = (unsigned char)i;
    if (sc.lstart[l] != i) return fail("visual factors of one landmark must share imu_i");
    const unsigned short bit = (unsigned short)(1u << (j - i - 1));
    if (mask[l] & bit) return fail("two visual factors of one landmark on the same frame");
    mask[l] |= bit; sc.lm[l]++;
    pair_cnt[i * NF + j]++;
  }
  int bin_cnt[NF][MAXOBS + 1];
  for (int s = 0; s < NF; s++) for (int m = 0; m <= MAXOBS; m++) bin_cnt[s][m] = 0;
  for (int l = 0; l < L; l++) {
    if (sc.lstart[l] == 255) sc.lstart[l] = 0;
    const int m = sc.lm[l];
    if (m > MAXOBS) return fail("landmark with more than 10 factors");
    if (mask[l] != (unsigned short)((1u << m) - 1)) return fail("landmark track must be contiguous from start_frame (feature_per_frame order)");
    bin_cnt[sc.lstart[l]][m]++;
  }
  int bin_base[NF][MAXOBS + 1];
  int slots = 0;
  for (int s = 0; s < NF; s++) {
    sc.sf_tile_begin[s] = slots / LM_TILE;
    int in_group = 0;
    for (int m = MAXOBS; m >= 0; m--) { bin_base[s][m] = slots + in_group; in_group += bin_cnt[s][m]; }
    slots += (in_group + LM_TILE - 1) / LM_TILE * LM_TILE;
  }
  sc.sf_tile_begin[NF] = slots / LM_TILE;
  sc.slots = slots; sc.n_tiles = slots / LM_TILE;
  sc.slot_rel.resize(L);
  for (int l = 0; l < L; l++) sc.slot_rel[l] = bin_base[sc.lstart[l]][sc.lm[l]]++;
  int run = 0;
  for (int p = 0; p < NPAIR; p++) { sc.pair_begin[p] = run; run += pair_cnt[p]; }
  sc.pair_begin[NPAIR] = run;
  return true;
}

// Upper bound of the tangent size of the prior a marginalisation of this window can return (MarginalizationInfo::n): the
// blocks the marginalisation set touches (k_marg builds the same table on the device) minus the dropped ones.
int prior_out_bound(const gfbe_window &win, const int *pair_begin, bool old) {
  bool touched[GFBE_BLK_COUNT];
  for (int q = 0; q < GFBE_BLK_COUNT; q++) touched[q] = false;
  const gfbe_prior *pr = (win.prior && win.prior->valid && win.prior->n > 0) ? win.prior : nullptr;
  if (pr) for (int q = 0; q < pr->n_blocks; q++) touched[pr->block_id[q]] = true;
  if (!old) return pr ? pr->n : 0;
  for (int k = 0; k < win.n_imu; k++) if (win.imu_frame[k] == 0) touched[0] = touched[GFBE_BLK_SB0] = touched[1] = touched[GFBE_BLK_SB0 + 1] = true;
  for (int k = 0; k < win.n_wheel; k++)
    if (win.wheel_frame[k] == 0) touched[0] = touched[1] = touched[GFBE_BLK_EX_WHEEL] = touched[GFBE_BLK_SX] = touched[GFBE_BLK_SY] = touched[GFBE_BLK_SW] = touched[GFBE_BLK_TD_WHEEL] = true;
  if (win.use_plane && win.frame_count > 0) touched[0] = touched[GFBE_BLK_EX_WHEEL] = touched[GFBE_BLK_PLANE_R] = touched[GFBE_BLK_PLANE_Z] = true;
  if (win.gnss_ready) {
    touched[0] = touched[GFBE_BLK_SB0] = touched[1] = touched[GFBE_BLK_SB0 + 1] = touched[GFBE_BLK_YAW_ENU] = touched[GFBE_BLK_ANC_ECEF] = true;
    for (int k = 0; k < 4; k++) touched[GFBE_BLK_RCV_DT0 + 4 + k] = true;
    touched[GFBE_BLK_RCV_DDT0 + 1] = true;
  }
  if (pair_begin) { for (int j = 1; j < NF; j++) if (pair_begin[j + 1] > pair_begin[j]) touched[0] = touched[j] = touched[GFBE_BLK_EX_CAM] = touched[GFBE_BLK_TD] = true; }
  else for (int q = 0; q < NF; q++) touched[q] = touched[GFBE_BLK_EX_CAM] = touched[GFBE_BLK_TD] = true;   // (table-fed: pair counts live on the device)
  int n = 0;
  for (int q = 0; q < GFBE_BLK_COUNT; q++) if (touched[q] && q != 0 && q != GFBE_BLK_SB0) n += blk_lsize(q);
  return std::min(n, (int)ND);
}
}  // namespace

// tabs != nullptr: the visual factors of window w come from table tab0 + w of the device-resident feature tables
// (wins[w]->vis, n_feature, para_Feature, feature_const are ignored); the landmark arrays are then filled on the device.
// The call returns when the windows are packed into pinned memory and the copy + preparation kernels are enqueued on the
// context's copy stream: the caller's buffers are free again, gfbe_batch_solve waits for the batch's `ev_up`.
static gfbe_status upload_one(gfbe_ctx *c, int32_t B, const gfbe_window *const *wins, gfbe_batch **out, gfbe_ftab *tabs = nullptr,
                              int tab0 = 0) {
  if (!c || !wins || !out || B <= 0) return GFBE_BAD_INPUT;
  if (c->device < 0 || !c->stream) { c->err = "HIP device context required (no CPU fallback)"; return GFBE_NO_DEVICE; }
  const bool dbg_t = diag_getenv("GFBE_DEBUG_UPLOAD") != nullptr && diag_getenv("GFBE_DEBUG_UPLOAD")[0] != 0;   // phase times of the upload on stderr (tools/diag_e2e_latency.py)
  auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  HIPCHK(c, hipSetDevice(c->device));
  const double T0 = now();
  for (int w = 0; w < B; w++) {
    const gfbe_window *win = wins[w];
    if (!win) { c->err = "window " + std::to_string(w) + ": null pointer"; return GFBE_BAD_INPUT; }
    if (win->frame_count < 0 || win->frame_count > GFBE_WINDOW_SIZE || win->n_imu < 0 || win->n_imu > MAX_IMU || win->n_wheel < 0 || win->n_wheel > MAX_WHEEL ||
        (!tabs && (win->n_feature < 0 || win->vis.n_factor < 0)) || (win->n_imu > 0 && (!win->imu || !win->imu_frame)) ||
        (win->n_wheel > 0 && (!win->wheel || !win->wheel_frame))) {
      c->err = "window " + std::to_string(w) + ": bad sizes"; return GFBE_BAD_INPUT;
    }
  }
  if (c->allreduce && c->opt.max_solver_time_in_seconds > 0.0) {
    // (every rank would stop on its own clock: the replicated dense state would diverge between the ranks)
    c->err = "max_solver_time_in_seconds is not available with landmark sharding (gfbe_set_allreduce)"; return GFBE_BAD_INPUT;
  }
  gfbe_batch *b = new gfbe_batch();
  *out = b;
  BatchDev &d = b->d;
  std::memset(&d, 0, sizeof d);
  d.B = B;
  d.opt = c->opt;
  for (hipEvent_t *e : {&b->ev_up, &b->ev_done, &b->ev_dl}) {
    if (!c->sync_event_pool.empty()) { *e = c->sync_event_pool.back(); c->sync_event_pool.pop_back(); }
    else if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) { c->err = "hipEventCreate (batch) failed"; return GFBE_DEVICE_ERROR; }
  }
  // the stream the upload runs on: the copy stream, beside whatever the main stream is solving. The table-fed path reads the
  // device-resident tables there too, behind the table operations enqueued so far (gfbe_ftab::ev_ops) — round 5: it used to run on the
  // main stream and WAITED there for its landmark counts, i.e. for every solve queued before it: 12.9 ms of host time per 1024
  // windows, and table-fed batches slower than host-fed ones
  // (a small table-fed batch — the one-robot frame loop — stays on the main stream behind its table operations: nothing is queued there
  //  to wait for, and a hop across streams costs ~15 us of the frame)
  // (round 5: every batch below 32 windows — its upload is two kernels now (k_ingest_small reads the pinned staging buffer itself), ~40 us
  //  that would hide behind nothing, and the hop from the copy stream to the main stream cost 11 us of a single window's call)
  hipStream_t us = B < DENSE_SPLIT_MIN_B ? c->stream : c->copy;
  if (tabs && us != c->stream && tabs->ev_ops) HIPCHK(c, hipStreamWaitEvent(us, tabs->ev_ops, 0));
  // (a SMALL table-fed upload right behind a large one from the same tables — no table operation in between, so nobody has
  //  waited for ev_read — rewrites the tables' shared histogram / layout / id scratch on the main stream while the large batch's pack
  //  kernel may still read them on the copy stream: the main stream waits for the last reader first. The flag stays set: the next table
  //  operation still orders itself behind that reader through ft_ready.)
  if (tabs && us == c->stream && tabs->read_pending && tabs->ev_read) HIPCHK(c, hipStreamWaitEvent(us, tabs->ev_read, 0));
  b->slot_of.resize(B);
  b->L.resize(B);
  b->anchor_only.assign(B, 0);
  b->up_win_bytes.assign(B, 0.0);
  std::vector<WinScan> scan(B);
  std::vector<int> tcounts, tlayout;     // table source: per window [L, K, bins], layout table for the pack kernel
  // ---- pass 1 (parallel over windows): validate the factor lists, lay the landmarks out
  if (tabs) {
    if (tab0 < 0 || tab0 + B > tabs->d.W) { c->err = "gfbe_batch_upload_tables: more windows than tables"; return GFBE_BAD_INPUT; }
    int *dcounts = tabs->d.hist + (size_t)tab0 * (FT_BINS + 2);
    launch_ftab_count(tabs->d, tabs->cur, tab0, B, dcounts, us);
    tcounts.resize((size_t)B * (FT_BINS + 2));
    // (through the tables' pinned staging mirror when it is large enough: a pageable device-to-host copy is staged by the runtime)
    const bool pinned = tabs->stage_h && sizeof(int) * tcounts.size() <= tabs->stage_cap;
    HIPCHK(c, hipMemcpyAsync(pinned ? (void *)tabs->stage_h : (void *)tcounts.data(), dcounts, sizeof(int) * tcounts.size(), hipMemcpyDeviceToHost, us));
    HIPCHK(c, hipStreamSynchronize(us));
    if (pinned) std::memcpy(tcounts.data(), tabs->stage_h, sizeof(int) * tcounts.size());
    tlayout.assign((size_t)B * FT_LAY_STRIDE, 0);
    for (int w = 0; w < B; w++) {   // layout from the per-bin counts: groups by start frame (tile aligned), longer tracks first inside a group
      WinScan &sc = scan[w];
      const int *cnt = &tcounts[(size_t)w * (FT_BINS + 2) + 2];
      int *lay = &tlayout[(size_t)w * FT_LAY_STRIDE];
      sc.L = tcounts[(size_t)w * (FT_BINS + 2)]; sc.K = tcounts[(size_t)w * (FT_BINS + 2) + 1];
      if (sc.L < 0 || sc.K < 0) { c->err = "window " + std::to_string(w) + ": bad sizes"; return GFBE_BAD_INPUT; }
      int slots = 0;
      for (int s = 0; s < NF; s++) {
        sc.sf_tile_begin[s] = slots / LM_TILE;
        lay[FT_LAY_GRP + s] = slots;
        int in_group = 0;
        for (int m = MAXOBS; m >= 3; m--) { lay[FT_LAY_BIN + s * 8 + (m - 3)] = slots + in_group; in_group += cnt[s * 8 + (m - 3)]; }
        slots += (in_group + LM_TILE - 1) / LM_TILE * LM_TILE;
      }
      sc.sf_tile_begin[NF] = slots / LM_TILE;
      sc.slots = slots; sc.n_tiles = slots / LM_TILE;
      // factors of pair (s, s+1+k) = landmarks of start frame s with more than k factors
      int pair_cnt[NPAIR];
      for (int p = 0; p < NPAIR; p++) pair_cnt[p] = 0;
      for (int s = 0; s < NF; s++)
        for (int k = 0; k < MAXOBS && s + 1 + k < NF; k++)
          for (int m = std::max(k + 1, 3); m <= MAXOBS; m++) pair_cnt[s * NF + s + 1 + k] += cnt[s * 8 + (m - 3)];
      int run = 0;
      for (int p = 0; p < NPAIR; p++) { sc.pair_begin[p] = run; run += pair_cnt[p]; }
      sc.pair_begin[NPAIR] = run;
      std::memcpy(&lay[FT_LAY_PAIR], sc.pair_begin, sizeof(int) * (NPAIR + 1));
    }
  } else {
    std::atomic<int> bad(-1);
    const auto t_scan0 = std::chrono::steady_clock::now();
    host_parallel(c, B, [&](int w) { if (!scan_window(*wins[w], w, scan[w])) { int e = -1; bad.compare_exchange_strong(e, w); } }, true);
    c->host_ms[0] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_scan0).count();
    if (bad.load() >= 0) { c->err = scan[bad.load()].err; return GFBE_BAD_INPUT; }
  }
  // ---- serial: offsets of every window in the batch-wide arrays
  std::vector<WinDesc> desc_tmp(B);   // (only the offset fields are set here; the fill pass completes the descriptors)
  std::vector<int> tile_start;
  std::vector<int> feat_off(B + 1, 0);
  std::vector<long long> j0_off(B + 1, 0);
  int tot_n0 = 0;
  int max_sf_tiles = 0;
  int tot_lm = 0, tot_rec = 0, max_tiles = 0, n_imu_tot = 0, n_wheel_tot = 0, tot_lio = 0, pn_max = 0, tot_gnss = 0, any_gnss = 0, gnss_dims = 0, gnss_max = 0, marg_nmax = 0;
  double algo_bytes = 0.0;
  for (int w = 0; w < B; w++) {
    const gfbe_window &win = *wins[w];
    const WinScan &sc = scan[w];
    WinDesc &ds = desc_tmp[w];
    std::memset(&ds, 0, sizeof ds);
    ds.L = sc.L; ds.K = sc.K; ds.frame_count = win.frame_count;
    ds.lm_off = tot_lm; ds.lm_slots = sc.slots; ds.n_tiles = sc.n_tiles; ds.tile_off = (int)tile_start.size();
    ds.rec_off = tot_rec;
    ds.vel_off = tot_n0;
    tot_n0 += tabs ? 0 : sc.pair_begin[NF];      // (records of the pairs (0, j): pair index i * NF + j, i-major)
    for (int s = 0; s < NF; s++) for (int t = sc.sf_tile_begin[s]; t < sc.sf_tile_begin[s + 1]; t++) tile_start.push_back(s);
    if (tabs) tlayout[(size_t)w * FT_LAY_STRIDE] = tot_lm;
    tot_lm += sc.slots; tot_rec += sc.K; max_tiles = std::max(max_tiles, sc.n_tiles);
    for (int s = 0; s < NF; s++) max_sf_tiles = std::max(max_sf_tiles, sc.sf_tile_begin[s + 1] - sc.sf_tile_begin[s]);
    ds.imu_off = n_imu_tot; ds.wheel_off = n_wheel_tot; ds.lio_off = tot_lio;
    n_imu_tot += win.n_imu; n_wheel_tot += win.n_wheel; tot_lio += win.lio.n > 0 ? win.lio.n : 0;
    if (win.gnss_ready) {
      if (win.n_gnss < 0 || (win.n_gnss > 0 && !win.gnss_obs)) { c->err = "window " + std::to_string(w) + ": gnss_ready without observations array"; return GFBE_BAD_INPUT; }
      for (int k = 0; k < win.n_gnss; k++) {
        const gfbe_gnss_obs &o = win.gnss_obs[k];
        if (o.frame < 0 || o.frame > GFBE_WINDOW_SIZE || o.lower_idx < 0 || o.lower_idx >= GFBE_WINDOW_SIZE || (o.lower_idx != o.frame && o.lower_idx != o.frame - 1) ||
            o.sys_idx < 0 || o.sys_idx > 3 || !(o.pr_uura > 0.0) || !(o.dp_uura > 0.0)) {
          c->err = "window " + std::to_string(w) + ": GNSS observation " + std::to_string(k) + " has an index or a deviation out of range"; return GFBE_BAD_INPUT;
        }
      }
      ds.gnss_ready = 1; ds.n_gnss = win.n_gnss; ds.gnss_off = tot_gnss;
      tot_gnss += win.n_gnss; any_gnss = 1; gnss_max = std::max(gnss_max, win.n_gnss);
    }
    b->L[w] = sc.L;
    feat_off[w + 1] = feat_off[w] + sc.L;
    // MARGIN_SECOND_NEW with an INVALID last_marginalization_info that still lists Pose[WINDOW_SIZE-1] (estimator.cpp:3600, 3622-3632):
    // the reference marginalises a PoseAnchorFactor on Pose[0] with drop set {Pose[0]} — six dims dropped, nothing kept — and ends
    // with a valid, empty MarginalizationInfo. Nothing to compute: gfbe_batch_download hands back exactly that.
    if (win.prior && !win.prior->valid && win.frame_count == GFBE_WINDOW_SIZE && win.prior->n_blocks > 0 && win.prior->n_blocks <= GFBE_MAX_PRIOR_BLOCKS)
      for (int q = 0; q < win.prior->n_blocks; q++) if (win.prior->block_id[q] == GFBE_BLK_POSE0 + GFBE_WINDOW_SIZE - 1) b->anchor_only[w] = 1;
    if (win.prior && win.prior->valid && win.prior->n > 0) {
      const gfbe_prior &pr = *win.prior;
      if (pr.n > ND || pr.n_blocks < 0 || pr.n_blocks > GFBE_MAX_PRIOR_BLOCKS || !pr.J0 || !pr.r0) { c->err = "window " + std::to_string(w) + ": prior too large or without J0 / r0"; return GFBE_BAD_INPUT; }
      bool seen[GFBE_BLK_COUNT];
      for (int q = 0; q < GFBE_BLK_COUNT; q++) seen[q] = false;
      int xo = 0;
      for (int q = 0; q < pr.n_blocks; q++) {
        const int id = pr.block_id[q];
        if (id < 0 || id >= GFBE_BLK_COUNT || pr.block_size[q] != blk_gsize(id) || seen[id] || pr.block_idx[q] < 0 || pr.block_idx[q] + blk_lsize(id) > pr.n) {
          c->err = "window " + std::to_string(w) + ": prior block table inconsistent (id, size, duplicate or offset out of range)"; return GFBE_BAD_INPUT;
        }
        seen[id] = true; xo += pr.block_size[q];
        if (id >= GFBE_BLK_ANC_ECEF) gnss_dims = 1;
      }
      if (xo > (int)PRIOR_X0) { c->err = "prior x0 too large"; return GFBE_BAD_INPUT; }
      pn_max = std::max(pn_max, pr.n);
    }
    const int nb = std::max(prior_out_bound(win, tabs ? nullptr : sc.pair_begin, true), prior_out_bound(win, nullptr, false));
    j0_off[w + 1] = j0_off[w] + (long long)nb * nb;
    marg_nmax = std::max(marg_nmax, nb);
    algo_bytes += 108.0 * sc.K;   // 12 f64 + 3 i32 per visual residual block, J never re-read by the host
  }
  d.tot_lm = tot_lm; d.max_tiles = max_tiles; d.tot_rec = tot_rec; d.tot_lio = tot_lio; d.max_sf_tiles = max_sf_tiles;
  d.rank = c->rank; d.world = c->world; d.sharded = c->allreduce ? 1 : 0;
  d.schur_groups = B >= DENSE_SPLIT_MIN_B ? SCHUR_GROUPS : (c->allreduce ? NF : 2 * NF);
  d.test_fail_chol_iter = c->opt.test_fail_chol_iter;   // (test hook of the mu-retry path, 0 in production: gfbe_options)
  b->algo_bytes_lin = algo_bytes;
  for (int w = 0; w < B; w++) if (!wins[w]->ex_cam_const || !wins[w]->td_const) d.vis_full = 1;
  // Host-fed batches that hold td and the camera extrinsic constant everywhere: the observations cross PCIe already shifted to the
  // window's td (two doubles per factor instead of five: k_expand would apply the same shift, projectionTwoFrameOneCamFactor.cpp:
  // 60-61, before anything reads them); velocity and td of an observation travel only for the landmarks that start in frame 0 — the
  // marginalisation's td / extrinsic columns are the only readers (estimator.cpp:3498-3531 takes the factors with imu_i == 0).
  // 378 -> ~200 KB of the 630 KB a 2000-landmark window uploads. (Not for gfbe_eval_factors: its records carry every td column.)
  d.obs_compact = (!tabs && !d.vis_full && !c->want_records) ? 1 : 0;
  for (int w = 0; w < B; w++) if (wins[w]->use_plane || wins[w]->use_anchor) d.any_plane = 1;
  for (int w = 0; w < B; w++) if (wins[w]->prior && wins[w]->prior->valid) d.prior_n_max = std::max(d.prior_n_max, (int)wins[w]->prior->n);
  d.any_gnss = any_gnss; d.tot_gnss = tot_gnss; d.gnss_max_obs = gnss_max; d.marg_nmax = marg_nmax;
  d.nu = (any_gnss || gnss_dims) ? (int)ND : (int)NC;       // a batch without GNSS blocks never touches the last 59 tangent dims
  d.solve_big = d.nu > NC;                                  // (decided per batch: k_solve / k_solve_chain hold the 187 core dims only)
  if (diag_getenv("GFBE_VIS_FULL")) d.vis_full = 1;   // (diagnostics build only: force the 20-column panel)
  // speculative linearisation (gfbe_options.speculative_linearization): batches whose candidate costs are all formed by the visual /
  // dense-factor / LiDAR / GNSS launches (no all-reduce hook) get a second set of the linearisation's outputs
  // (round 6: also with an all-reduce hook — the landmark-sharded solve: the candidate's pass linearises its own tiles, the ranks' candidate
  //  costs travel as before; one evaluation pass less per iteration there too. The number of collectives per iteration stays: DESIGN.md section 7)
  d.spec = (c->opt.speculative_linearization && max_tiles > 0 &&
            (B >= DENSE_SPLIT_MIN_B || (GFBE_FUSE_SMALL & 2) || c->allreduce)) ? 1 : 0;
  // k_linschur (gfbe_options.merge_lin_schur): throughput batches on the 7 x 7 panel, every tile on this rank
  d.linschur = (c->opt.merge_lin_schur && B >= DENSE_SPLIT_MIN_B && !d.vis_full && !c->allreduce && max_tiles > 0) ? 1 : 0;
  const size_t TL = tot_lm;
  const size_t pj_row = (size_t)pn_max * pn_max;   // J0 of the priors travels compactly: rows of pn_max^2 doubles, spread into the ND^2 slots on the device
  const double T1 = now();
  // ---- allocation sequence (dry pass, then the real one)
  gfbe_status st;
  WinDesc *h_desc = nullptr; int *h_lm_info = nullptr, *h_lm_abi = nullptr, *h_tile_start = nullptr, *h_feat_off = nullptr;
  long long *h_j0_off = nullptr;
  double *h_fvel = nullptr;
  double *h_lm_pts = nullptr, *h_lam0 = nullptr, *h_fobs = nullptr, *h_x0 = nullptr, *h_lio = nullptr, *h_pr0 = nullptr, *h_px0 = nullptr, *h_pJ0 = nullptr;
  gfbe_imu_preint *h_imu = nullptr; gfbe_wheel_preint *h_wheel = nullptr;
  gfbe_gnss_obs *h_gnss = nullptr;
  double *d_pJ0c = nullptr;
  const bool want_rec = c->want_records;
  const char *poison_env = diag_getenv("GFBE_POISON_UNCLEARED");   // (test hook of the diagnostics build, see the enqueue below; read per upload)
  for (int pass = 0; pass < 2; pass++) {
    b->dry = pass == 0;
    if (pass == 1) {
      st = slab_acquire(c, b);
      if (st != GFBE_OK && d.spec) {
        // (the second set of the linearisation's outputs is ~1 MB per 2k-landmark window — a batch that fitted the device
        //  before the option existed must still fit: size the slab again without it; the note travels with the batch's create note)
        d.spec = 0;
        c->err.clear();
        c->note = "speculative_linearization switched off for a batch: its second set of outputs did not fit the device";
        b->slab_bytes = 0; b->up_bytes = 0; b->slab_off = 0;
        pass = -1;
        continue;
      }
      if (st != GFBE_OK) return st;
      b->up_h = pin_acquire(c, b->up_bytes, &b->up_cap);
      if (!b->up_h) { c->err = "hipHostMalloc(upload staging) failed"; return GFBE_DEVICE_ERROR; }
    }
#define UP(field, host, n) if ((st = up_alloc(c, b, &d.field, &host, (size_t)(n))) != GFBE_OK) return st
#define AL(field, n) do { const size_t o_ = b->slab_off; if ((st = dev_alloc(c, b, &d.field, (size_t)(n))) != GFBE_OK) return st; \
                          if (!b->dry && poison_env) b->poison_list.push_back({#field, o_, b->slab_off - o_}); } while (0)
    // -- upload region
    UP(desc, h_desc, B); UP(tile_start, h_tile_start, tile_start.size()); UP(x0, h_x0, (size_t)B * NA);
    UP(imu, h_imu, n_imu_tot); UP(wheel, h_wheel, n_wheel_tot); UP(lio, h_lio, (size_t)tot_lio * 8);
    UP(prior_r0, h_pr0, (size_t)B * ND); UP(prior_x0, h_px0, (size_t)B * PRIOR_X0);
    UP(dl_feat_off, h_feat_off, B + 1); UP(dl_j0_off, h_j0_off, B + 1);
    UP(gnss_obs, h_gnss, std::max(tot_gnss, 1));
    if ((st = up_alloc(c, b, &d_pJ0c, &h_pJ0, (size_t)B * pj_row)) != GFBE_OK) return st;
    if (!tabs) { UP(lm_info, h_lm_info, TL); UP(lm_abi, h_lm_abi, TL); UP(lm_pts, h_lm_pts, (size_t)6 * TL); UP(lam0, h_lam0, TL); UP(fobs, h_fobs, (size_t)tot_rec * (d.obs_compact ? 2 : 5)); UP(fvel, h_fvel, d.obs_compact ? (size_t)std::max(tot_n0, 1) * 3 : 1); }
    const size_t up_end = b->dry ? b->slab_bytes : b->slab_off;
    // -- arrays the kernels expect zeroed at the start (rows past a track's length, partials of absent factors, ...)
    if (tabs) { AL(lm_info, TL); AL(lm_abi, TL); AL(lm_pts, (size_t)6 * TL); AL(lam0, TL); d.fobs = nullptr; d.fvel = nullptr; }
#if GFBE_CLEAR_LM      // (diagnostics: the landmark rows back in the cleared region)
    AL(lm_obs, (size_t)MAXOBS * 5 * TL); AL(lm_rec, (size_t)MAXOBS * TL); AL(lm_hP, (size_t)MAXOBS * 6 * TL);
#endif
    AL(lio_part, (size_t)B * LIOW_WGS * LIOW_PART);
    AL(raw_imu, (size_t)MAX_IMU * (15 + 450) * 4 * ((B + 3) / 4)); AL(raw_wheel, (size_t)MAX_WHEEL * (6 + 132) * 4 * ((B + 3) / 4));   // [factor][window / 4][value][window % 4]
    AL(zero, 16); AL(vis_H, (size_t)B * NV * (NV + 1));
    d.vs_blocks = d.vis_full ? (int)VS_BLOCKS : 1;
    if (B < DENSE_SPLIT_MIN_B && !c->allreduce) { AL(vis_Hs, (size_t)B * d.vs_blocks * NV * (NV + 1)); } else d.vis_Hs = nullptr;
    AL(ctl, B);
    AL(lam, 2 * TL); AL(lm_Hll, TL); AL(lm_gl, TL); AL(lm_hC, (size_t)HC * TL);
    AL(lm_sl, TL); AL(lm_yl, TL); AL(lm_vl, TL); AL(lm_sw, TL);
    AL(imu_sqrt, (size_t)n_imu_tot * 225); AL(wheel_sqrt, (size_t)n_wheel_tot * 36);
    AL(pair_part, (size_t)B * NF * VP_STRIDE); AL(schur_part, (size_t)B * d.schur_groups * SCHUR_STRIDE);
    AL(imu_part, (size_t)B * MAX_IMU * IMU_PART); AL(wheel_part, (size_t)B * MAX_WHEEL * WHEEL_PART);
    AL(plane_part, d.any_plane ? (size_t)B * MAX_PLANE * PLANE_PART : 1); AL(anchor_part, d.any_plane ? (size_t)B * ANCHOR_PART : 1);
    AL(prior_g, (size_t)B * (ND + 2));
    if (d.spec) {   // the second set (BatchDev::spec), cleared like the first
      AL(lm_Hll2, TL); AL(lm_gl2, TL); AL(lm_hC2, (size_t)HC * TL); AL(lm_sw2, TL);
      AL(imu_part2, (size_t)B * MAX_IMU * IMU_PART); AL(wheel_part2, (size_t)B * MAX_WHEEL * WHEEL_PART);
      AL(plane_part2, d.any_plane ? (size_t)B * MAX_PLANE * PLANE_PART : 1); AL(anchor_part2, d.any_plane ? (size_t)B * ANCHOR_PART : 1);
      AL(prior_g2, (size_t)B * (ND + 2)); AL(lio_part2, (size_t)B * LIOW_WGS * LIOW_PART);
      if (d.linschur) { AL(schur_part2, (size_t)B * d.schur_groups * SCHUR_STRIDE); } else d.schur_part2 = nullptr;
    } else {
      d.schur_part2 = nullptr;
      d.lm_Hll2 = d.lm_gl2 = d.lm_hC2 = d.lm_sw2 = d.imu_part2 = d.wheel_part2 = d.plane_part2 = d.anchor_part2 = d.prior_g2 = d.lio_part2 = nullptr;
    }
    AL(tile_cost, (size_t)B * std::max(max_tiles, 1)); AL(tile_cand, (size_t)B * std::max(max_tiles, 1) * 4);
    AL(tile_cnt, B < DENSE_SPLIT_MIN_B ? (size_t)B * std::max(max_tiles, 1) : 1);
    AL(win_cnt, B < DENSE_SPLIT_MIN_B ? (size_t)B * 2 : 1);
    AL(tile_gram, (size_t)B * std::max(max_tiles, 1) * 8); AL(dense_cand, (size_t)B * 4);
    AL(xb, (size_t)B * d.world * XCHG); AL(xc, (size_t)B * d.world * XCHG);
    if (d.sharded) { AL(Er, (size_t)B * (NV * NV + NV)); AL(sys_pack, (size_t)B * sys_pack_doubles_host(d.nu, d.world)); } else { d.Er = nullptr; d.sys_pack = nullptr; }
    AL(sp, (size_t)B * ND); AL(Dp, (size_t)B * ND); AL(gts, (size_t)B * ND); AL(vp, (size_t)B * ND);
    AL(yp, (size_t)B * ND); AL(step, (size_t)B * ND);
    AL(timing, (size_t)(B + 1) * 32);   // (+ one block for the phase stamps of a diagnostics build)
    AL(mmeta, (size_t)B * (4 + 3 * GFBE_MAX_PRIOR_BLOCKS)); AL(mx0, (size_t)B * PRIOR_X0);
    const size_t zero_end = b->dry ? b->slab_bytes : b->slab_off;
    // -- written before they are read: no clearing (block-CSR records only exist for the inspection API)
    AL(prior_J0, (size_t)B * ND * ND);     // (the n x n prior block arrives by copy; nothing reads past it)
#if !GFBE_CLEAR_LM
    AL(lm_obs, (size_t)MAXOBS * 5 * TL); AL(lm_rec, (size_t)MAXOBS * TL);   // (k_expand / k_ftab_pack write the rows of a track; the evaluation uses a row only below the track's length: 0.9 of the 2.8 MB per window that used to be cleared)
    AL(lm_hP, (size_t)MAXOBS * 6 * TL);    // (k_vis writes the rows below a track's length, k_schur masks the others per landmark: 1.0 MB per window)
#endif
    AL(vis_part, (size_t)B * std::max(max_tiles, 1) * MAXOBS * VP_STRIDE);   // (a tile's steps below its longest track are written by k_vis, the others never read)
    // (the second set: the solve's linearisation only — its 7 x 7 partials take VPY_STRIDE doubles per step when no window frees the extrinsic / td)
    if (d.spec) { AL(lm_hP2, (size_t)MAXOBS * 6 * TL); AL(vis_part2, (size_t)B * std::max(max_tiles, 1) * MAXOBS * (d.vis_full ? (size_t)VP_STRIDE : (size_t)VPY_STRIDE)); }
    else { d.lm_hP2 = nullptr; d.vis_part2 = nullptr; }
    AL(mA, (size_t)B * ND * ND); AL(mb, (size_t)B * ND); AL(mJ0, (size_t)B * ND * ND); AL(mr0, (size_t)B * ND);   // (k_marg / k_marg_ldlt write what they and k_gather read)
    AL(x, (size_t)B * 2 * NA); AL(xout, (size_t)B * NA);                 // (k_reset / k_reanchor write them before anything reads)
    AL(pc, (size_t)B * 3 * NPAIR * PAIR_CONST_DOUBLES);                   // (written by the kernels that produce a state)
    AL(prior_H, (size_t)B * ND * ND);      // (k_prep writes the n x n block k_assemble reads)
    // (k_assemble writes every entry of H (lower triangle) its table lists, g, E, eg it owns, k_visblock the exchange row: no clearing
    //  here — a batch on the compact table clears H once at upload, below)
    // the partial reduced system [H | g | E | eg | xa] is one slab: a single all-reduce per linearisation when the
    // landmarks are sharded over ranks
    {
      const size_t nH = (size_t)B * ND * ND, ng = (size_t)B * ND, nE = (size_t)B * NV * NV, ne = (size_t)B * NV, nx = (size_t)B * d.world * XCHG;
      AL(H, nH + ng + nE + ne + nx);
      if (!b->dry) { d.g = d.H + nH; d.E = d.g + ng; d.eg = d.E + nE; d.xa = d.eg + ne; }
      b->slab_n = nH + ng + nE + ne + nx;
    }
    AL(dbg_imu, (size_t)B * MAX_IMU * 15 * 31); AL(dbg_wheel, (size_t)B * MAX_WHEEL * 6 * 23); AL(dbg_prior, (size_t)B * ND);
    AL(rec, want_rec ? (size_t)tot_rec * REC : 1); AL(mV, (size_t)B * ND * ND);
    AL(vis_contrib, B < DENSE_SPLIT_MIN_B ? (size_t)B * std::max(max_tiles, 1) * MAXOBS * 16 * LM_TILE : 1);
    d.solve_scratch_stride = d.solve_big ? (size_t)BIG_LD * BIG_LD : solve_chain_scratch_doubles();
    AL(solveY, d.solve_big ? 1 : (size_t)B * solve_chain_scratch_doubles());
    AL(solveS, d.solve_big ? (size_t)B * BIG_LD * BIG_LD : 1);
    AL(gnss_J, (size_t)std::max(tot_gnss, 1) * 36); AL(gnss_r, (size_t)std::max(tot_gnss, 1) * 2);
    if (d.spec) { AL(gnss_J2, (size_t)std::max(tot_gnss, 1) * 36); AL(gnss_r2, (size_t)std::max(tot_gnss, 1) * 2); AL(gnss_cost2, (size_t)B * 2); }
    else { d.gnss_J2 = d.gnss_r2 = d.gnss_cost2 = nullptr; }
    AL(gnss_cost, (size_t)B * 2); AL(gnss_marg, any_gnss ? (size_t)B * GN_MPART : 1);
    AL(dl_fix, (size_t)B * DL_FIX); AL(dl_feat, feat_off[B]); AL(dl_J0, (size_t)j0_off[B]);
    if (!b->dry) { b->up_end = up_end; b->zero_end = zero_end; }
#undef UP
#undef AL
  }
  // download staging: [dl_fix | dl_feat | dl_J0] are contiguous at the end of the slab -> one device-to-host copy
  b->dl_bytes = (size_t)((char *)(d.dl_J0 + j0_off[B]) - (char *)d.dl_fix);
  b->feat_off = feat_off; b->j0_off = j0_off;
  const double T2 = now();
  // ---- pass 2 (parallel over windows): fill the pinned mirror of the upload region
  std::memcpy(h_tile_start, tile_start.data(), sizeof(int) * tile_start.size());
  std::memcpy(h_feat_off, feat_off.data(), sizeof(int) * (B + 1));
  std::memcpy(h_j0_off, j0_off.data(), sizeof(long long) * (B + 1));
  std::atomic<int> bad(-1);
  std::vector<std::string> errs(B);
  const auto t_fill0 = std::chrono::steady_clock::now();
  host_parallel(c, B, [&](int w) {
    const gfbe_window &win = *wins[w];
    const WinScan &sc = scan[w];
    WinDesc &ds = h_desc[w];
    ds = desc_tmp[w];
    auto fail = [&](const char *m) { errs[w] = "window " + std::to_string(w) + ": " + m; int e = -1; bad.compare_exchange_strong(e, w); };
    std::memcpy(ds.sf_tile_begin, sc.sf_tile_begin, sizeof ds.sf_tile_begin);
    std::memcpy(ds.pair_begin, sc.pair_begin, sizeof ds.pair_begin);
    double bytes = sizeof(WinDesc) + sizeof(double) * NA;
    if (!tabs) {
      // landmark scalars: padding slots first (valid = 0, abi -1, lambda 1), then the landmarks
      const int o = ds.lm_off;
      for (int q = 0; q < sc.slots; q++) { h_lm_info[o + q] = 0; h_lm_abi[o + q] = -1; h_lam0[o + q] = 1.0; }
      for (int r = 0; r < 6; r++) std::memset(h_lm_pts + r * TL + o, 0, sizeof(double) * sc.slots);
      for (int l = 0; l < sc.L; l++) {
        const int slot = o + sc.slot_rel[l];
        const bool is_const = win.feature_const && win.feature_const[l];
        h_lm_info[slot] = sc.lstart[l] | (sc.lm[l] << 8) | ((is_const ? 1 : 0) << 16) | (1 << 24);
        h_lm_abi[slot] = l;
        h_lam0[slot] = win.para_Feature[l];
      }
      const bool compact = d.obs_compact != 0;
      double *fo = h_fobs + (size_t)ds.rec_off * (compact ? 2 : 5), *fv = compact ? h_fvel + (size_t)ds.vel_off * 3 : nullptr;
      const double tdw = win.state.para_Td;
      for (int k = 0; k < sc.K; k++) {
        const int l = win.vis.feature_index[k], i = win.vis.imu_i[k], j = win.vis.imu_j[k];
        const int rel = sc.slot_rel[l];
        const int rec = sc.pair_begin[i * NF + j] + (rel - sc.sf_tile_begin[i] * LM_TILE);
        if (compact) {
          // p' = p - (td - td_obs) v: k_expand's fused multiply-add (a correctly rounded std::fma is the same number; an observation
          // stamped with the window's td — the usual case — is not touched)
          const double dtj = tdw - win.vis.td_j[k];
          double *f = fo + (size_t)rec * 2;
          f[0] = dtj == 0.0 ? win.vis.pts_j[3 * k] : std::fma(-dtj, win.vis.vel_j[2 * k], win.vis.pts_j[3 * k]);
          f[1] = dtj == 0.0 ? win.vis.pts_j[3 * k + 1] : std::fma(-dtj, win.vis.vel_j[2 * k + 1], win.vis.pts_j[3 * k + 1]);
          if (i == 0) { double *v = fv + (size_t)rec * 3; v[0] = win.vis.vel_j[2 * k]; v[1] = win.vis.vel_j[2 * k + 1]; v[2] = win.vis.td_j[k]; }
        } else {
          double *f = fo + (size_t)rec * 5;
          f[0] = win.vis.pts_j[3 * k]; f[1] = win.vis.pts_j[3 * k + 1]; f[2] = win.vis.vel_j[2 * k]; f[3] = win.vis.vel_j[2 * k + 1]; f[4] = win.vis.td_j[k];
        }
        if (j == i + 1) {   // the landmark's first observation travels with its first factor
          const size_t slot = (size_t)o + rel;
          h_lm_pts[0 * TL + slot] = win.vis.pts_i[3 * k]; h_lm_pts[1 * TL + slot] = win.vis.pts_i[3 * k + 1]; h_lm_pts[2 * TL + slot] = win.vis.pts_i[3 * k + 2];
          h_lm_pts[3 * TL + slot] = win.vis.vel_i[2 * k]; h_lm_pts[4 * TL + slot] = win.vis.vel_i[2 * k + 1]; h_lm_pts[5 * TL + slot] = win.vis.td_i[k];
        }
      }
      b->slot_of[w].resize(sc.L);
      for (int l = 0; l < sc.L; l++) b->slot_of[w][l] = o + sc.slot_rel[l];
      bytes += (double)sc.slots * (4 + 4 + 8 + 48) + (compact ? 16.0 * sc.K + 24.0 * sc.pair_begin[NF] : 40.0 * sc.K);
    }
    // dense state
    std::memcpy(h_x0 + (size_t)w * NA, &win.state, sizeof(double) * NA);
    // inertial factors
    for (int q = 0; q < NF; q++) ds.imu_of_frame[q] = ds.wheel_of_frame[q] = -1;
    ds.n_imu = win.n_imu;
    for (int k = 0; k < win.n_imu; k++) {
      if (win.imu_frame[k] < 0 || win.imu_frame[k] >= win.frame_count) return fail("bad imu_frame");
      h_imu[ds.imu_off + k] = win.imu[k]; ds.imu_frame[k] = win.imu_frame[k]; ds.imu_of_frame[win.imu_frame[k]] = k;
    }
    ds.n_wheel = win.n_wheel;
    for (int k = 0; k < win.n_wheel; k++) {
      if (win.wheel_frame[k] < 0 || win.wheel_frame[k] >= win.frame_count) return fail("bad wheel_frame");
      h_wheel[ds.wheel_off + k] = win.wheel[k]; ds.wheel_frame[k] = win.wheel_frame[k]; ds.wheel_of_frame[win.wheel_frame[k]] = k;
    }
    bytes += sizeof(gfbe_imu_preint) * win.n_imu + sizeof(gfbe_wheel_preint) * win.n_wheel;
    // LiDAR factors on one pose
    ds.lio_n = win.lio.n > 0 ? win.lio.n : 0; ds.lio_frame = win.lio.frame;
    ds.lio_sqrt_info = win.lio.sqrt_info; ds.lio_huber = win.lio.huber_delta;
    if (ds.lio_n > 0) {
      if (win.lio.frame < 0 || win.lio.frame > win.frame_count || !win.lio.pts || !win.lio.normals || !win.lio.offsets) return fail("bad lio block");
      double *lo = h_lio + (size_t)ds.lio_off * 8;
      for (int k = 0; k < ds.lio_n; k++) {
        for (int q = 0; q < 3; q++) { lo[8 * k + q] = win.lio.pts[3 * k + q]; lo[8 * k + 3 + q] = win.lio.normals[3 * k + q]; }
        lo[8 * k + 6] = win.lio.offsets[k];
        lo[8 * k + 7] = win.lio.weights ? win.lio.weights[k] : 1.0;
      }
      bytes += 64.0 * ds.lio_n;
    }
    // prior
    bool used[GFBE_BLK_COUNT];
    for (int q = 0; q < GFBE_BLK_COUNT; q++) used[q] = false;
    if (ds.lio_n > 0) used[ds.lio_frame] = true;
    for (int q = 0; q < ND; q++) ds.prior_map[q] = -1;
    std::memset(h_pr0 + (size_t)w * ND, 0, sizeof(double) * ND);
    std::memset(h_px0 + (size_t)w * PRIOR_X0, 0, sizeof(double) * PRIOR_X0);
    if (pj_row) std::memset(h_pJ0 + (size_t)w * pj_row, 0, sizeof(double) * pj_row);
    if (win.prior && win.prior->valid && win.prior->n > 0) {
      const gfbe_prior &pr = *win.prior;   // (validated in the serial pass)
      ds.prior_n = pr.n; ds.prior_nblk = pr.n_blocks;
      int xo = 0;
      for (int q = 0; q < pr.n_blocks; q++) {
        const int id = pr.block_id[q];
        ds.prior_blk_id[q] = id; ds.prior_blk_size[q] = pr.block_size[q]; ds.prior_blk_idx[q] = pr.block_idx[q];
        ds.prior_x0_off[q] = xo; xo += pr.block_size[q];
        used[id] = true;
        for (int k = 0; k < blk_lsize(id); k++) ds.prior_map[blk_tan(id) + k] = pr.block_idx[q] + k;
      }
      std::memcpy(h_px0 + (size_t)w * PRIOR_X0, pr.x0, sizeof(double) * xo);
      std::memcpy(h_pJ0 + (size_t)w * pj_row, pr.J0, sizeof(double) * pr.n * pr.n);
      std::memcpy(h_pr0 + (size_t)w * ND, pr.r0, sizeof(double) * pr.n);
      bytes += 8.0 * ((double)pr.n * pr.n + pr.n + xo);
    }
    // reduced program: blocks touched by a residual and not constant (Ceres drops the rest)
    for (int k = 0; k < win.n_imu; k++) { const int i = win.imu_frame[k]; used[i] = used[GFBE_BLK_SB0 + i] = used[i + 1] = used[GFBE_BLK_SB0 + i + 1] = true; }
    for (int k = 0; k < win.n_wheel; k++) {
      const int i = win.wheel_frame[k];
      used[i] = used[i + 1] = used[GFBE_BLK_EX_WHEEL] = used[GFBE_BLK_SX] = used[GFBE_BLK_SY] = used[GFBE_BLK_SW] = used[GFBE_BLK_TD_WHEEL] = true;
    }
    for (int p = 0; p < NPAIR; p++) if (sc.pair_begin[p + 1] > sc.pair_begin[p]) { used[p / NF] = used[p % NF] = used[GFBE_BLK_EX_CAM] = used[GFBE_BLK_TD] = true; }
    // optional in-window factors: PlaneFactor on every pose i < frame_count (estimator.cpp:3214-3220), PoseAnchorFactor on Pose[0]
    ds.n_plane = win.use_plane ? std::min(win.frame_count, (int)MAX_PLANE) : 0;
    ds.use_anchor = win.use_anchor ? 1 : 0;
    for (int q = 0; q < 3; q++) ds.plane_noise_inv[q] = win.plane_noise_inv[q];
    for (int q = 0; q < 7; q++) ds.anchor_pose[q] = win.anchor_pose[q];
    ds.anchor_sqrt_info = win.anchor_sqrt_info;
    for (int i = 0; i < ds.n_plane; i++) used[i] = true;
    if (ds.n_plane > 0) used[GFBE_BLK_EX_WHEEL] = used[GFBE_BLK_PLANE_R] = used[GFBE_BLK_PLANE_Z] = true;
    if (ds.use_anchor) used[0] = true;
    // GNSS (estimator.cpp:2965-3002, 3239-3291): the observations sorted by frame (stable: the reference's insertion order is
    // frame-major already), the lowspeed gate from the window's velocities, the blocks the factors touch
    if (ds.gnss_ready) {
      int cnt[NF + 1];
      for (int q = 0; q <= NF; q++) cnt[q] = 0;
      for (int k = 0; k < win.n_gnss; k++) cnt[win.gnss_obs[k].frame + 1]++;
      for (int q = 0; q < NF; q++) cnt[q + 1] += cnt[q];
      for (int q = 0; q <= NF; q++) ds.gnss_frame_begin[q] = cnt[q];
      for (int k = 0; k < win.n_gnss; k++) h_gnss[ds.gnss_off + cnt[win.gnss_obs[k].frame]++] = win.gnss_obs[k];
      ds.gnss_has_iono = win.gnss_iono ? 1 : 0;
      for (int q = 0; q < 8; q++) ds.gnss_iono[q] = win.gnss_iono ? win.gnss_iono[q] : 0.0;
      for (int q = 0; q < GFBE_WINDOW_SIZE; q++) ds.gnss_frame_dt[q] = win.gnss_frame_dt[q];
      ds.gnss_ddt_weight = win.gnss_ddt_weight;
      double ax = 0.0, ay = 0.0;
      for (int i = 0; i <= GFBE_WINDOW_SIZE; i++) { ax += std::fabs(win.state.para_SpeedBias[i][0]); ay += std::fabs(win.state.para_SpeedBias[i][1]); }
      ax /= GFBE_WINDOW_SIZE + 1; ay /= GFBE_WINDOW_SIZE + 1;
      ds.gnss_factors = !(std::sqrt(ax * ax + ay * ay) < 0.3);
      if (ds.gnss_factors) {
        for (int k = 0; k < win.n_gnss; k++) {
          const gfbe_gnss_obs &o = win.gnss_obs[k];
          used[o.lower_idx] = used[GFBE_BLK_SB0 + o.lower_idx] = used[o.lower_idx + 1] = used[GFBE_BLK_SB0 + o.lower_idx + 1] = true;
          used[GFBE_BLK_YAW_ENU] = used[GFBE_BLK_ANC_ECEF] = true;
        }
        for (int q = GFBE_BLK_RCV_DT0; q < GFBE_BLK_COUNT; q++) used[q] = true;     // DtDdtFactor / DdtSmoothFactor chains
      }
      bytes += sizeof(gfbe_gnss_obs) * win.n_gnss;
    }
    for (int q = 0; q < GFBE_BLK_COUNT; q++) {
      bool cst;
      if (q < GFBE_BLK_SB0) cst = win.pose_const[q] || q > win.frame_count;
      else if (q < GFBE_BLK_EX_CAM) cst = win.sb_const[q - GFBE_BLK_SB0] || (q - GFBE_BLK_SB0) > win.frame_count;
      else if (q == GFBE_BLK_EX_CAM) cst = win.ex_cam_const;
      else if (q == GFBE_BLK_EX_WHEEL) cst = win.ex_wheel_const;
      else if (q == GFBE_BLK_TD) cst = win.td_const;
      else if (q == GFBE_BLK_TD_WHEEL) cst = win.td_wheel_const;
      else if (q == GFBE_BLK_PLANE_R || q == GFBE_BLK_PLANE_Z) cst = win.plane_const;
      else if (q == GFBE_BLK_YAW_ENU) cst = win.gnss_ready != 0;         // estimator.cpp:2991
      else if (q == GFBE_BLK_ANC_ECEF || q >= GFBE_BLK_RCV_DT0) cst = false;
      else cst = win.ix_wheel_const;
      ds.blk_free[q] = used[q] && !cst;
      if (ds.blk_free[q]) for (int k = 0; k < blk_lsize(q); k++) ds.act[blk_tan(q) + k] = 1;
    }
    ds.act[T_PLR + 3] = 0;   // the plane quaternion's 4th slot only exists in the prior (three tangent dims in the solve)
    std::memcpy(ds.ex_cam_mask, win.ex_cam_mask, 6);
    std::memcpy(ds.ex_wheel_mask, win.ex_wheel_mask, 6);
    b->up_win_bytes[w] = bytes;
  }, true);
  c->host_ms[0] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_fill0).count();
  if (bad.load() >= 0) { c->err = errs[bad.load()]; return GFBE_BAD_INPUT; }
  // k_solve_chain eliminates the speed-bias blocks as a chain: IMUFactor couples SpeedBias[k] with its neighbours only, and the
  // priors the reference builds keep SpeedBias[0] alone (estimator.cpp:3400-3433, 3600-3632). A prior with any other speed-bias
  // block breaks that structure: such a batch takes the monolithic factorisation (gfbe_options.solve_kernel = 1 forces it).
  {
    int ntile = 1, mono = c->opt.solve_kernel == 1;
    for (int w = 0; w < B; w++) {
      const WinDesc &ds = h_desc[w];
      ntile = std::max(ntile, solve_chain_tiles(ds.act));
      for (int q = 0; q < ds.prior_nblk; q++) if (ds.prior_blk_id[q] > GFBE_BLK_SB0 && ds.prior_blk_id[q] < GFBE_BLK_EX_CAM) mono = 1;
    }
    d.solve_ntile = ntile; d.solve_mono = mono;
    {   // (round 6) a batch with GNSS dims: the chain kernel with nine tile columns when every window's dense part fits and the priors keep the chain's structure
      int n_dense_max = 0;
      for (int w = 0; w < B; w++) {
        int nd = 0;
        for (int a = 0; a < ND; a++) if (h_desc[w].act[a] && !(a >= T_SB(0) && a < T_SB(0) + 9 * NF)) nd++;
        n_dense_max = std::max(n_dense_max, nd);
      }
      d.solve_wide = d.solve_big && !mono && c->opt.solve_kernel != 4 && solve_chain_wide_fits(n_dense_max);
    }
    // the chain kernel of the batch: the twisted one (both ends at once, eight waves, one workgroup per CU) where a window's latency
    // counts — below DENSE_SPLIT_MIN_B windows, like the rest of the small-batch kernel set —, the classic one for throughput
    // (a dense part of six tile columns — a free camera extrinsic — does not fit its LDS layout: the classic kernel whatever was asked)
    d.solve_tw = (c->opt.solve_kernel == 3 || (c->opt.solve_kernel != 2 && B < DENSE_SPLIT_MIN_B)) && solve_chain_tw_fits(ntile);
    // the assembly table of the context: the compact one unless a prior couples a speed-bias block other than SpeedBias[0] (entries
    // outside the set it lists) or the batch carries GNSS dims
    bool prior_sb = false;
    for (int w = 0; w < B && !prior_sb; w++)
      for (int q = 0; q < h_desc[w].prior_nblk; q++) if (h_desc[w].prior_blk_id[q] > GFBE_BLK_SB0 && h_desc[w].prior_blk_id[q] < GFBE_BLK_EX_CAM) prior_sb = true;
    // (GFBE_ASM_FULL, diagnostics build only: the full table for every batch — tests/test_gpu_solve_kernels.py compares the two entry by entry)
    const bool compact = !prior_sb && d.nu == NC && c->asm_compact_n > 0 && !diag_getenv("GFBE_ASM_FULL");
    d.asm_tab = compact ? c->asm_compact : c->asm_full;
    d.asm_legacy = diag_getenv("GFBE_ASM_LEGACY") ? 1 : 0;      // (diagnostics build only: diag_getenv is nullptr in the product)
    d.asm_n = compact ? c->asm_compact_n : d.nu * (d.nu + 1) / 2;
  }
  const double T3 = now();
  // ---- enqueue: clear what must start as zero, ONE host-to-device copy, then the preparation kernels
  // (a small batch: all of it in one kernel that reads the pinned staging buffer across PCIe itself — k_ingest_small)
  const bool ingest = B < DENSE_SPLIT_MIN_B && (b->up_end % 16) == 0 && ((b->zero_end - b->up_end) % 16) == 0;
  // (test hook, tests/test_gpu_uncleared.py: the part of the slab that is NOT cleared filled with NaN bit patterns — a kernel that
  //  uses what nobody wrote poisons its results instead of finding the previous batch's numbers there)
  //  GFBE_POISON_UNCLEARED=1: all of it; =<array name>: that array alone. Ahead of BOTH upload paths (the hook used to
  //  switch k_ingest_small off, so the shipped single-window upload's cleared ranges were never tested under poison).
  if (poison_env)
    for (const auto &pl : b->poison_list)
      if (pl.off >= b->zero_end && (!strcmp(poison_env, "1") || !strcmp(poison_env, "clean") || !strcmp(poison_env, pl.name)))
        HIPCHK(c, hipMemsetAsync(b->slab + pl.off, strcmp(poison_env, "clean") ? 0xFF : 0, pl.bytes, us));   // ("clean": zeros, for a scan that poisons one array at a time)
  if (ingest) {
    const bool clearH = d.asm_tab == c->asm_compact;
    static_assert((sizeof(double) * ND * ND) % 16 == 0, "k_ingest_small clears H in 16-byte units");
    launch_ingest_small(b->up_h, b->slab, b->up_end, b->slab + b->up_end, b->zero_end > b->up_end ? b->zero_end - b->up_end : 0,
                        d.H, clearH ? sizeof(double) * (size_t)B * ND * ND : 0,
                        pj_row > 0 ? (const double *)h_pJ0 : nullptr, d.prior_J0, pj_row > 0 ? B : 0, pj_row,
                        (size_t)ND * ND, us);
  } else {
  if (b->zero_end > b->up_end) HIPCHK(c, hipMemsetAsync(b->slab + b->up_end, 0, b->zero_end - b->up_end, us));
  // the compact assembly table writes only the entries of H some factor can reach: the others are read (as the zeros they are) by
  // the solve kernels and must start as zeros — H is not part of the cleared region (the full table writes every entry)
  if (d.asm_tab == c->asm_compact) HIPCHK(c, hipMemsetAsync(d.H, 0, sizeof(double) * (size_t)B * ND * ND, us));
  HIPCHK(c, hipMemcpyAsync(b->slab, b->up_h, b->up_end, hipMemcpyHostToDevice, us));
  if (pj_row > 0)
    HIPCHK(c, hipMemcpy2DAsync(d.prior_J0, sizeof(double) * ND * ND, d_pJ0c, sizeof(double) * pj_row, sizeof(double) * pj_row, B, hipMemcpyDeviceToDevice, us));
  }
  if (tabs) {   // landmark arrays straight from the device-resident tables (layout table and slot map live in the tables' own scratch)
    int *dlay = tabs->d.layout + (size_t)tab0 * FT_LAY_STRIDE, *dslot = tabs->d.ids_scratch + (size_t)tab0 * tabs->d.F;
    // (the layout table travels from a pinned buffer the batch keeps: nobody waits for the copy)
    b->lay_h = pin_acquire(c, sizeof(int) * tlayout.size(), &b->lay_cap);
    if (!b->lay_h) { c->err = "hipHostMalloc(layout staging) failed"; return GFBE_DEVICE_ERROR; }
    std::memcpy(b->lay_h, tlayout.data(), sizeof(int) * tlayout.size());
    HIPCHK(c, hipMemcpyAsync(dlay, b->lay_h, sizeof(int) * tlayout.size(), hipMemcpyHostToDevice, us));
    launch_ftab_pack(tabs->d, tabs->cur, tab0, B, d, dlay, dslot, us);
    if (tabs->ev_read && us != c->stream) { HIPCHK(c, hipEventRecord(tabs->ev_read, us)); tabs->read_pending = true; }
    // (and the other way round: a small upload reads the shared scratch on the MAIN stream; a large one that follows it waits for ev_ops
    //  on the copy stream before it rewrites them — recorded again here, behind this reader)
    else if (tabs->ev_ops && us == c->stream) HIPCHK(c, hipEventRecord(tabs->ev_ops, us));
  } else if (B >= DENSE_SPLIT_MIN_B) {
    launch_expand(d, us);
  }
  if (B < DENSE_SPLIT_MIN_B) launch_upload_small(d, tabs ? 0 : 1, us);    // (one launch: gfbe_kernels.hip)
  else launch_prep(d, us);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(b->ev_up, us));
  if (dbg_t) {
    (void)hipStreamSynchronize(us);
    fprintf(stderr, "upload: scan %.3f ms, layout + staging %.3f ms, fill %.3f ms, copy + prep %.3f ms (slab %.1f MB, upload %.2f MB, cleared %.1f MB)\n",
            T1 - T0, T2 - T1, T3 - T2, now() - T3, b->slab_bytes / 1048576.0, b->up_end / 1048576.0, (b->zero_end - b->up_end) / 1048576.0);
  }
  return GFBE_OK;
}

extern "C" void gfbe_batch_free(gfbe_ctx *c, gfbe_batch *b);

// Batches of >= BATCH_SPLIT_MIN_B windows become `split_batch` parts (default two halves) solved side by side, each on
// its own pair of streams: a chain a -> a->second -> ...; part k + 1 runs on part k's `lane2`.
static gfbe_status make_lane(gfbe_ctx *c, gfbe_batch *a) {
  if (!c->lane_pool.empty()) {
    const gfbe_ctx::LaneSet l = c->lane_pool.back();
    c->lane_pool.pop_back();
    a->lane2 = {l.s, l.aux, l.fork, l.join}; a->ev_start2 = l.start2; a->ev_done2 = l.done2;
    return GFBE_OK;
  }
  if (stream_acquire(c->device, true, &a->lane2.s) != hipSuccess ||
      stream_acquire(c->device, true, &a->lane2.aux) != hipSuccess ||
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
  // parts of a throughput batch, solved side by side (gfbe_options.split_batch): 1 = the measured default — four parts for
  // batches of >= 2048 windows, none below (with k_visasm and two k_solve_chain workgroups per CU a single part fills the GPU:
  // 1024 windows 62.7k solves/s whole against 61.2k as 4 x 256 and 56.7k as two halves; 512 windows 59.2k against 50.7k / 52.0k;
  // 2048: 65.3k as four parts against 64.3k whole; 4096: 67.2k against 65.0k, 64.1k as two and 62.5k as eight parts) — an
  // explicit count >= 2 is taken as it is
  int parts = 1;
  if (B >= BATCH_SPLIT_MIN_B && !c->allreduce && c->opt.split_batch) {
    if (c->opt.split_batch < 0) { c->err = "gfbe_options.split_batch must be >= 0"; return GFBE_BAD_INPUT; }
    parts = c->opt.split_batch == 1 ? (B >= 2048 ? 4 : 1) : c->opt.split_batch;
    // (measured: a heterogeneous batch of 256 windows as four size classes of 64: 30.1k solves/s against 60.9k whole — parts that small
    //  do not fill the GPU; from 2048 windows on the parts exist anyway and are made size classes below)
    parts = std::max(1, std::min(std::min(parts, (int)MAX_BATCH_PARTS), std::max(B / DENSE_SPLIT_MIN_B, 1)));   // (every part beyond the first owns a pair of streams)
  }
  // A part's kernels are launched for its LARGEST window (landmark tiles per window: the grids of k_vis / k_lm_step / k_schur), and
  // the parts run side by side: with the windows of a heterogeneous batch sorted by their number of visual factors, every part's
  // grid fits its own windows instead of the batch's largest one. A window's result does not depend on its place or its
  // neighbours (tests/test_gpu_parity.py::test_large_batch_throughput_path), so the order is the library's to choose; results go
  // back to the caller's places (download_one). Host-fed batches only (a table-fed window is tied to its table's index).
  std::vector<int> order;
  std::vector<const gfbe_window *> sorted;
  if (parts > 1 && !tabs) {
    int kmin = INT32_MAX, kmax = 0;
    for (int w = 0; w < B; w++) { if (!wins[w]) { kmin = kmax = 0; break; } kmin = std::min(kmin, (int)wins[w]->vis.n_factor); kmax = std::max(kmax, (int)wins[w]->vis.n_factor); }
    if (kmax > kmin + kmin / 8) {       // (a homogeneous batch stays as it is)
      order.resize(B);
      for (int w = 0; w < B; w++) order[w] = w;
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return wins[a]->vis.n_factor > wins[b]->vis.n_factor; });
      sorted.resize(B);
      for (int k = 0; k < B; k++) sorted[k] = wins[order[k]];
      wins = sorted.data();
    }
  }
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
  if (st != GFBE_OK) { gfbe_batch_free(c, *out); *out = nullptr; return st; }
  if (!order.empty()) {
    (*out)->place.resize(B);
    for (int k = 0; k < B; k++) (*out)->place[order[k]] = k;
    (*out)->order = std::move(order);
  }
  return st;
}

// No C++ exception crosses the C ABI (host staging vectors can throw std::bad_alloc).
static gfbe_status upload_guarded(gfbe_ctx *c, int32_t B, const gfbe_window *const *wins, gfbe_batch **out, gfbe_ftab *tabs) {
  try {
    const auto t0 = std::chrono::steady_clock::now();
    if (c) c->host_ms[0] = c->host_ms[1] = 0.0;      // (gfbe_host_times: the packing passes add themselves up over the parts)
    const gfbe_status st = upload_halves(c, B, wins, out, tabs);
    if (c) c->host_ms[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() - c->host_ms[0];
    return st;
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
  if (c->allreduce) { c->err = "gfbe_batch_upload_tables: not available with landmark sharding"; return GFBE_BAD_INPUT; }
  return upload_guarded(c, B, wins, out, t);
}

extern "C" int32_t gfbe_batch_feature_count(const gfbe_batch *b, int32_t w) {
  if (w < 0) return -1;
  if (b && !b->place.empty()) { if (w >= (int)b->place.size()) return -1; w = b->place[w]; }
  for (; b; b = b->second) {
    if (w < (int)b->L.size()) return b->L[w];
    w -= (int)b->L.size();
  }
  return -1;
}

extern "C" void gfbe_batch_free(gfbe_ctx *c, gfbe_batch *b) {
  if (!b) return;
  // wait for THIS batch's work only (upload, last solve, download): other batches of the context keep running
  for (hipEvent_t e : {b->ev_up, b->ev_done, b->ev_dl}) if (e) (void)hipEventSynchronize(e);
  if (b->second) {
    // (the part's own events are waited for in the recursive call; its streams are pooled and may already serve a newer batch)
    gfbe_batch_free(c, b->second);
    if (c && b->lane2.s && b->lane2.aux && b->lane2.fork && b->lane2.join && b->ev_start2 && b->ev_done2 && c->lane_pool.size() < 16) {
      c->lane_pool.push_back({b->lane2.s, b->lane2.aux, b->lane2.fork, b->lane2.join, b->ev_start2, b->ev_done2});   // (idle: the part's events were waited for)
    } else {
      stream_release(c ? c->device : -1, true, b->lane2.s);
      stream_release(c ? c->device : -1, true, b->lane2.aux);
      for (hipEvent_t e : {b->lane2.fork, b->lane2.join, b->ev_start2, b->ev_done2}) if (e) (void)hipEventDestroy(e);
    }
  }
  for (auto &g : b->graph) if (g) (void)hipGraphExecDestroy(g);
  for (hipEvent_t e : {b->ev_up, b->ev_done, b->ev_dl}) if (e) { if (c && c->sync_event_pool.size() < 64) c->sync_event_pool.push_back(e); else (void)hipEventDestroy(e); }
  slab_release(c, b);
  if (c) { pin_release(c, b->up_h, b->up_cap); pin_release(c, b->dl_h, b->dl_cap); pin_release(c, b->lay_h, b->lay_cap); }
  else { if (b->up_h) (void)hipHostFree(b->up_h); if (b->dl_h) (void)hipHostFree(b->dl_h); if (b->lay_h) (void)hipHostFree(b->lay_h); }
  delete b;
}

// The all-reduce hook of the landmark-sharded solve; the first failure is kept for gfbe_batch_solve's status.
static void run_allreduce(gfbe_ctx *c, double *ptr, int64_t n, hipStream_t s) {
  const int32_t rc = c->allreduce(c->allreduce_user, ptr, n, s);
  if (rc != 0 && c->allreduce_rc == 0) c->allreduce_rc = rc;
}

// GFBE_FUSE_SMALL (gfbe_device.h): which launches of an iteration are merged for this batch. Small batches on the latency path only:
// not while profiling (per-kernel events), not with an all-reduce hook, not without landmarks (k_lm_step is not launched then).
#ifndef GFBE_MARG_DENSE_ASIDE
#define GFBE_MARG_DENSE_ASIDE 1      // the marginalisation's dense factors of a throughput batch on the side stream (0: in line, rounds 1-6)
#endif
static int small_fuse(const gfbe_ctx *c, const BatchDev &d) {
  if (c->profiling || d.B >= DENSE_SPLIT_MIN_B || !d.vis_Hs || d.sharded || d.max_tiles == 0) return 0;
  int f = GFBE_FUSE_SMALL;
  // (a LiDAR window's candidate cost — and a GNSS window's beyond the size k_lin_small takes itself — is a launch of its own between
  //  k_lin_small<1 / 3> and k_accept)
  if (d.tot_lio > 0 || (d.any_gnss && !lin_small_takes_gnss(d, 1))) f &= ~4;
  return f;
}

// One linearisation of the whole batch at the current parameters (skipped on device for windows
// that only need a new radius).
static void enqueue_linearize(gfbe_ctx *c, gfbe_batch *b, const Lane &ln, bool first, bool have_lin = false) {
  const BatchDev &d = b->d;
  // the first linearisation has every window active; later ones skip windows that only shrink the radius
  // fork: dense factors on the aux stream (serial and timed on the main stream when profiling)
  // (have_lin: the last iteration's candidate pass linearised, BatchDev::spec)
  const bool overlap = !c->profiling && ln.aux && d.B >= DENSE_SPLIT_MIN_B && !have_lin;
  if (overlap) {
    (void)hipEventRecord(ln.fork, ln.s);
    (void)hipStreamWaitEvent(ln.aux, ln.fork, 0);
    launch_dense_factors(d, 0, 0, ln.aux);
    (void)hipEventRecord(ln.join, ln.aux);
  }
  const bool small = !c->profiling && d.B < DENSE_SPLIT_MIN_B;   // one launch for visual tiles + dense factors (k_lin_small)
  if (have_lin) { }
  else if (small) launch_lin_small(d, 0, ln.s);
  else {
    if (d.linschur) { Timed t(c, first ? "k_linschur_iter0" : "k_linschur", b->algo_bytes_lin); launch_linschur(d, 0, 0, ln.s); }
    else { Timed t(c, first ? "k_vis_lin_iter0" : "k_vis_lin", b->algo_bytes_lin); launch_vis(d, 0, ln.s); }
    if (!overlap) { Timed t(c, "k_dense", 0); launch_dense_factors(d, 0, 0, ln.s); }
  }
  if (d.tot_lio > 0 && !have_lin) { Timed t(c, "k_lio_window", 0); launch_lio_window(d, 0, ln.s); }
  const int fuse = small_fuse(c, d);
  if (d.linschur) {
    // (the landmark elimination ran with the evaluation. have_lin: the candidate's pass did both; what is left is the window whose set was
    //  formed with another mu — after an invalid step — and is evaluated again: the gated launch)
    if (have_lin) { Timed t(c, "k_linschur_gate", 0); launch_linschur(d, 0, 1, ln.s); }
  } else { Timed t(c, first ? "k_schur_iter0" : "k_schur", 0); launch_schur(d, 0, ln.s, fuse & 1); }
  if (d.vis_Hs && !(fuse & 1)) { Timed t(c, first ? "k_visblock_iter0" : "k_visblock", 0); launch_visblock(d, ln.s); }   // (throughput batches: inside k_visasm, below)
  if (overlap) (void)hipStreamWaitEvent(ln.s, ln.join, 0);   // join
  { Timed t(c, first ? "k_assemble_iter0" : "k_assemble", 0); launch_assemble(d, ln.s); }
  if (d.any_gnss) { Timed t(c, "k_gnss", 0); launch_gnss(d, 0, ln.s, have_lin ? 2 : 0); }      // (have_lin: the sums of the candidate pass's evaluation)
  if (d.sharded) {   // one all-reduce per linearisation, on the packed triangle of the system (164 KB per window instead of 528 KB)
    Timed t(c, "allreduce_system", 0);
    launch_sys_pack(d, 0, ln.s);
    run_allreduce(c, d.sys_pack, (int64_t)d.B * (int64_t)sys_pack_doubles_host(d.nu, d.world), ln.s);
    launch_sys_pack(d, 1, ln.s);
  }
  { Timed t(c, first ? "k_solve_iter0" : "k_solve", 0); launch_solve(d, ln.s); }
  if (d.sharded) {
    // the mu retry of DoglegStrategy when the landmarks are sharded: a window whose factorisation failed gets E rebuilt for the
    // larger mu from every rank's own tiles, one more all-reduce (E | eg only), and another factorisation — up to
    // gfbe_options.sharded_mu_retries times (pass k of the solve kernel hands a window that fails again on to pass k + 1 with
    // mu x 10; the last pass gives up like DoglegStrategy at max_mu)
    Timed t(c, "mu_retry_sharded", 0);
    const int passes = std::min(std::max(c->opt.sharded_mu_retries, 0), 8);
    for (int k = 1; k <= passes; k++) {
      launch_rebuild_E_shard(d, ln.s);
      run_allreduce(c, d.Er, (int64_t)d.B * (NV * NV + NV), ln.s);
      launch_solve(d, ln.s, k);
    }
  }
  { Timed t(c, first ? "k_lm_step_iter0" : "k_lm_step", 0); launch_lm_step(d, ln.s, fuse & 2); }
  if (d.sharded) {
    Timed t(c, "allreduce_scalars", 0);
    launch_xchg_gram(d, ln.s);
    run_allreduce(c, d.xb, (int64_t)d.B * d.world * XCHG, ln.s);
  }
}

static gfbe_status enqueue_solve(gfbe_ctx *c, gfbe_batch *b, const Lane &ln, int32_t margin_flag) {
  const BatchDev &d = b->d;
  { Timed t(c, "k_reset", 0); launch_reset(d, ln.s); }
  const int iters = std::min(c->opt.max_num_iterations, 15);
  // speculative linearisation (BatchDev::spec): every candidate pass but the last linearises at the candidate, into the second set of
  // outputs; an accepted step makes that set the current one and a rejected one keeps the old linearisation (DoglegStrategy's reuse)
  // — either way the next iteration needs no linearisation launch.
  const bool spec = d.spec && (d.B >= DENSE_SPLIT_MIN_B || (small_fuse(c, d) & 2) || d.sharded);      // (small batches: on the fused launch sequence, or sharded — its launches are never fused)
  for (int it = 0; it < iters; it++) {
    enqueue_linearize(c, b, ln, it == 0, spec && it > 0);
    const int fuse = small_fuse(c, d);
    if (!(fuse & 2)) {
      { Timed t(c, "k_step", 0); launch_step(d, ln.s); }
      { Timed t(c, "k_candidate", 0); launch_candidate(d, ln.s); }
    }
    const bool overlap = !c->profiling && ln.aux && d.B >= DENSE_SPLIT_MIN_B;
    const bool lin_cand = spec && it + 1 < iters;      // this candidate pass linearises (the last one of a solve only needs the costs)
    if (overlap) {
      (void)hipEventRecord(ln.fork, ln.s);
      (void)hipStreamWaitEvent(ln.aux, ln.fork, 0);
      launch_dense_factors(d, lin_cand ? 0 : 1, 0, ln.aux, lin_cand);
      (void)hipEventRecord(ln.join, ln.aux);
    }
    const bool small = !c->profiling && d.B < DENSE_SPLIT_MIN_B;
    if (small) launch_lin_small(d, lin_cand ? 3 : 1, ln.s, fuse);
    else if (lin_cand && d.linschur) { Timed t(c, "k_linschur", b->algo_bytes_lin); launch_linschur(d, 1, 0, ln.s); }
    else if (lin_cand) { Timed t(c, "k_vis_lin", b->algo_bytes_lin); launch_vis(d, 0, ln.s, 0, 1); }
    else { Timed t(c, "k_vis_cost", 0); launch_vis(d, 1, ln.s); }
    if (d.tot_lio > 0) { Timed t(c, lin_cand ? "k_lio_window" : "k_lio_window_cost", 0); launch_lio_window(d, lin_cand ? 0 : 1, ln.s, lin_cand); }
    if (d.any_gnss && !(small && lin_small_takes_gnss(d, lin_cand ? 3 : 1))) {      // (small batches: a workgroup of k_lin_small did it)
      Timed t(c, lin_cand ? "k_gnss_eval" : "k_gnss_cost", 0); launch_gnss(d, lin_cand ? 0 : 1, ln.s, lin_cand ? 1 : 0);
    }
    if (!small && !overlap) { Timed t(c, lin_cand ? "k_dense" : "k_dense_cost", 0); launch_dense_factors(d, lin_cand ? 0 : 1, 0, ln.s, lin_cand); }
    else if (overlap) (void)hipStreamWaitEvent(ln.s, ln.join, 0);
    if (d.sharded) {
      Timed t(c, "allreduce_scalars", 0);
      launch_xchg_cand(d, ln.s);
      run_allreduce(c, d.xc, (int64_t)d.B * d.world * XCHG, ln.s);
    }
    if (!(fuse & 4)) { Timed t(c, "k_accept", 0); launch_accept(d, ln.s, lin_cand ? 1 : 0); }
  }
  { Timed t(c, "k_reanchor", 0); launch_reanchor(d, ln.s); }
  if (margin_flag != GFBE_MARGIN_NONE) {
    Timed t(c, "marginalize", 0);
    if (d.sharded && margin_flag == GFBE_MARGIN_OLD) {
      // the partials of the landmarks that start in frame 0 are summed over the ranks before k_marg reads them
      launch_marginalize_partials(d, ln.s);
      run_allreduce(c, d.pair_part, (int64_t)d.B * NF * VP_STRIDE, ln.s);
      run_allreduce(c, d.schur_part, (int64_t)d.B * d.schur_groups * SCHUR_STRIDE, ln.s);
      launch_marginalize_finish(d, margin_flag, ln.s);
    } else if (GFBE_MARG_DENSE_ASIDE && margin_flag == GFBE_MARGIN_OLD && !c->profiling && ln.aux && d.B >= DENSE_SPLIT_MIN_B) {
      // (end of round 6) throughput batches: the frame-0 inertial / wheel / prior factors of the marginalisation set (k_dense<false>: a few
      // latency-bound workgroups per window, ~40 us per launch over 512 windows) on the side stream, beside k_vis<2> / k_pairsum / k_schur —
      // they read the re-anchored state and nothing of the visual kernels', k_marg reads them all: fork behind k_reanchor, join in front of k_marg
      (void)hipEventRecord(ln.fork, ln.s);
      (void)hipStreamWaitEvent(ln.aux, ln.fork, 0);
      launch_dense_factors(d, 2, 0, ln.aux);
      (void)hipEventRecord(ln.join, ln.aux);
      launch_marginalize_partials(d, ln.s, true);
      (void)hipStreamWaitEvent(ln.s, ln.join, 0);
      launch_marginalize_finish(d, margin_flag, ln.s);
    } else {
      launch_marginalize(d, margin_flag, ln.s);
    }
  }
  if (d.sharded) {   // every rank ends with all inverse depths: owners contribute theirs, the others zeros
    launch_lam_mask(d, ln.s);
    run_allreduce(c, d.lam, (int64_t)2 * d.tot_lm, ln.s);
  }
  return GFBE_OK;
}

extern "C" gfbe_status gfbe_batch_solve(gfbe_ctx *c, gfbe_batch *b, int32_t margin_flag) {
  if (!c || !b) return GFBE_BAD_INPUT;
  if (c->device < 0) return GFBE_NO_DEVICE;
  if (margin_flag < 0 || margin_flag > 2) { c->err = "gfbe_batch_solve: margin_flag out of range"; return GFBE_BAD_INPUT; }
  const BatchDev &d = b->d;
  if (d.sharded && !c->allreduce) { c->err = "batch was uploaded for landmark sharding but the all-reduce hook is gone"; return GFBE_BAD_INPUT; }
  // hipGraph replay: not while profiling (per-kernel events) and not with the all-reduce hook (host callback)
  // (not for split batches: capturing the cross-stream fork / join of the parts crashed the runtime on ROCm 7.2)
  const bool graphable = c->opt.use_graph && !c->profiling && !d.sharded && !b->second;
  const Lane lane1 = {c->stream, c->aux, c->ev_fork, c->ev_join};
  // the uploads ran on the copy stream: the solve starts when they have landed
  for (gfbe_batch *p = b; p; p = p->second) { HIPCHK(c, hipStreamWaitEvent(c->stream, p->ev_up, 0)); p->last_flag = margin_flag; p->fetched = false; }
  // (ev_done of every part is recorded on the caller's stream after the parts have joined it)
  auto mark_done = [&]() -> gfbe_status { for (gfbe_batch *p = b; p; p = p->second) HIPCHK(c, hipEventRecord(p->ev_done, c->stream)); return GFBE_OK; };
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
    return mark_done();
  }
  if (graphable && b->calls[margin_flag]++ >= 1) {   // the first call ran eagerly (one-time attribute setup); capture now
    hipGraph_t g = nullptr;
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
      (void)enqueue_all();
      const hipError_t e = hipStreamEndCapture(c->stream, &g);
      if (e == hipSuccess && g && hipGraphInstantiate(&b->graph[margin_flag], g, nullptr, nullptr, 0) == hipSuccess) {
        (void)hipGraphDestroy(g);
        HIPCHK(c, hipGraphLaunch(b->graph[margin_flag], c->stream));
        return mark_done();
      }
      if (g) (void)hipGraphDestroy(g);
      b->graph[margin_flag] = nullptr;
      (void)hipGetLastError();
    }
    c->opt.use_graph = 0;   // capture not available on this stream: stay eager
  }
  c->allreduce_rc = 0;
  const gfbe_status st = enqueue_all();
  if (st != GFBE_OK) return st;
  HIPCHK(c, hipGetLastError());
  if (c->allreduce_rc != 0) {   // (the partial sums of this solve were not all reduced: no result is handed back)
    c->err = "all-reduce hook failed with code " + std::to_string(c->allreduce_rc) + " during gfbe_batch_solve";
    (void)mark_done();
    return GFBE_DEVICE_ERROR;
  }
  return mark_done();
}

// Results of one part: the gather kernel and ONE device-to-host copy on the download stream (beside whatever the main stream
// is solving), then the host scatter into the caller's structures on the packing threads.
static gfbe_status fetch_one(gfbe_ctx *c, gfbe_batch *b) {
  if (b->fetched) return GFBE_OK;
  const BatchDev &d = b->d;
  if (!b->dl_h) {
    b->dl_h = pin_acquire(c, b->dl_bytes, &b->dl_cap);
    if (!b->dl_h) { c->err = "hipHostMalloc(download staging) failed"; return GFBE_DEVICE_ERROR; }
  }
  // A small batch (a single window's latency path) gathers and copies on the solver's own stream, behind the solve: no event to wait
  // for across streams (~15 us of a 1.45 ms call); throughput batches use the download stream, beside whatever the solver runs next.
  // (The two waits stay unconditional: on the stream that recorded the events they cost nothing, and they keep the gather behind
  //  the upload and the solve when the caller changed streams in between — gfbe_set_stream — or downloads without a solve.)
  hipStream_t ds = d.B < DENSE_SPLIT_MIN_B ? c->stream : c->dl;
  HIPCHK(c, hipStreamWaitEvent(ds, b->ev_up, 0));
  HIPCHK(c, hipStreamWaitEvent(ds, b->ev_done, 0));
  if (d.B < DENSE_SPLIT_MIN_B) {
    // (a small batch: k_gather writes the pinned staging buffer itself — the three result arrays keep their offsets —, no copy command
    //  and no idle stream between the kernel and the copy: ~12 us of a single window's call)
    BatchDev dg = d;
    dg.dl_fix = (double *)b->dl_h;
    dg.dl_feat = (double *)(b->dl_h + ((const char *)d.dl_feat - (const char *)d.dl_fix));
    dg.dl_J0 = (double *)(b->dl_h + ((const char *)d.dl_J0 - (const char *)d.dl_fix));
    launch_gather(dg, b->last_flag, ds);
    HIPCHK(c, hipGetLastError());
  } else {
    launch_gather(d, b->last_flag, ds);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(b->dl_h, d.dl_fix, b->dl_bytes, hipMemcpyDeviceToHost, ds));
  }
  HIPCHK(c, hipEventRecord(b->ev_dl, ds));
  return GFBE_OK;
}
// out_* are the caller's arrays; window w of this part is the caller's window order[first + w] (or first + w)
static gfbe_status download_one(gfbe_ctx *c, gfbe_batch *b, gfbe_state *out_state, double *const *out_feature,
                                gfbe_prior *const *prior_out, gfbe_summary *summary, const int first, const int *order) {
  const BatchDev &d = b->d;
  const int B = d.B;
  const auto t_wait0 = std::chrono::steady_clock::now();
  HIPCHK(c, hipEventSynchronize(b->ev_dl));
  const auto t_wait1 = std::chrono::steady_clock::now();
  c->host_ms[2] += std::chrono::duration<double, std::milli>(t_wait1 - t_wait0).count();
  b->fetched = true;
  // (the three result arrays are separate 256-byte aligned allocations at the end of the slab: offsets from the device pointers)
  const double *fix = (const double *)b->dl_h;
  const double *feat = (const double *)(b->dl_h + ((const char *)d.dl_feat - (const char *)d.dl_fix));
  const double *J0s = (const double *)(b->dl_h + ((const char *)d.dl_J0 - (const char *)d.dl_fix));
  std::vector<int> status(B);
  host_parallel(c, B, [&](int w) {
    const double *f = fix + (size_t)w * DL_FIX;
    WinCtl k;
    std::memcpy(&k, f, sizeof k);
    status[w] = k.status;
    const int *m = (const int *)(f + DL_OFF_META);
    double dl_bytes = 8.0 * DL_FIX + 8.0 * b->L[w];
    const int cw = order ? order[first + w] : first + w;      // the caller's index
    if (out_state) std::memcpy(out_state + cw, f + DL_OFF_X, sizeof(double) * NA);
    if (out_feature && out_feature[cw] && b->L[w] > 0) std::memcpy(out_feature[cw], feat + b->feat_off[w], sizeof(double) * b->L[w]);
    // the prior is only touched when a marginalisation ran for this window (estimator.cpp:3391: a window that is still filling
    // up, or MARGIN_NONE, leaves last_marginalization_info as it is)
    if (prior_out && prior_out[cw] && b->last_flag == GFBE_MARGIN_SECOND_NEW && b->anchor_only[w]) {
      // estimator.cpp:3622-3632 (see upload_one): the invalid prior is replaced by a valid, empty one
      prior_out[cw]->valid = 1; prior_out[cw]->n = 0; prior_out[cw]->n_blocks = 0;
    } else if (prior_out && prior_out[cw] && b->last_flag != GFBE_MARGIN_NONE && k.marg_ran) {
      gfbe_prior *p = prior_out[cw];
      p->valid = m[0] == 1 ? 1 : 0; p->n = m[1]; p->n_blocks = m[2];
      if (p->valid) {
        int xo = 0;
        for (int q = 0; q < p->n_blocks; q++) {
          p->block_id[q] = m[4 + q]; p->block_size[q] = m[4 + GFBE_MAX_PRIOR_BLOCKS + q]; p->block_idx[q] = m[4 + 2 * GFBE_MAX_PRIOR_BLOCKS + q];
          xo += p->block_size[q];
        }
        std::memcpy(p->x0, f + DL_OFF_X0, sizeof(double) * xo);
        std::memcpy(p->J0, J0s + b->j0_off[w], sizeof(double) * p->n * p->n);
        std::memcpy(p->r0, f + DL_OFF_R0, sizeof(double) * p->n);
        dl_bytes += 8.0 * p->n * p->n;
      }
    }
    if (summary) {
      gfbe_summary &s = summary[cw];
      std::memset(&s, 0, sizeof s);
      s.status = k.status; s.iterations = k.iter; s.num_successful = k.num_successful; s.termination = k.termination;
      s.initial_cost = k.initial_cost; s.final_cost = k.cost; s.final_radius = k.radius;
      std::memcpy(s.cost_history, k.cost_history, sizeof s.cost_history);
      std::memcpy(s.accepted, k.accepted, sizeof s.accepted);
      s.ms_solve = k.t_solved > k.t_start ? (double)(k.t_solved - k.t_start) * 1e-5 : 0.0;     // 100 MHz ticks
      s.ms_marginalize = k.t_marg > k.t_solved ? (double)(k.t_marg - k.t_solved) * 1e-5 : 0.0;
      s.bytes_uploaded = b->up_win_bytes[w]; s.bytes_downloaded = dl_bytes;
    }
  });
  c->host_ms[3] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_wait1).count();
  gfbe_status worst = GFBE_OK;
  for (int w = 0; w < B; w++) {
    const int m0 = ((const int *)(fix + (size_t)w * DL_FIX + DL_OFF_META))[0];
    if (m0 == -2) { c->err = "new prior larger than its download slot"; return GFBE_DEVICE_ERROR; }
    if (status[w] == GFBE_NUMERICAL_FAILURE) worst = GFBE_NUMERICAL_FAILURE;
    else if (status[w] == GFBE_NO_CONVERGENCE && worst == GFBE_OK) worst = GFBE_NO_CONVERGENCE;
  }
  if (worst == GFBE_NUMERICAL_FAILURE) c->err = "linear solve failed for every mu < 1 in at least one window";
  return worst;
}

extern "C" gfbe_status gfbe_batch_download(gfbe_ctx *c, gfbe_batch *b, gfbe_state *out_state, double *const *out_feature,
                                          gfbe_prior *const *prior_out, gfbe_summary *summary) {
  if (!c || !b) return GFBE_BAD_INPUT;
  if (c->device < 0) return GFBE_NO_DEVICE;
  // all parts' gathers and copies are enqueued first, then unpacked part by part
  c->host_ms[2] = c->host_ms[3] = 0.0;      // (summed over the parts of this call)
  for (gfbe_batch *p = b; p; p = p->second) { const gfbe_status st = fetch_one(c, p); if (st != GFBE_OK) return st; }
  gfbe_status worst = GFBE_OK;
  int done = 0;
  for (gfbe_batch *p = b; p; p = p->second) {
    const gfbe_status st = download_one(c, p, out_state, out_feature, prior_out, summary, done, b->order.empty() ? nullptr : b->order.data());
    // (a window whose linear solves all failed is ITS failure — summary[w].status —: the other parts are still unpacked)
    if (st > GFBE_NUMERICAL_FAILURE) return st;
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
  c->want_records = true;
  gfbe_status st = gfbe_batch_upload(c, 1, wins, &b);
  c->want_records = false;
  c->opt = keep;
  if (st != GFBE_OK) { gfbe_batch_free(c, b); return st; }
  BatchDev &d = b->d;
  HIPCHK(c, hipStreamWaitEvent(c->stream, b->ev_up, 0));
  launch_reset(d, c->stream);
  launch_vis(d, 0, c->stream, /*write_records=*/1);
  launch_dense_factors(d, 0, 1, c->stream);
  launch_gnss(d, 0, c->stream);       // (its cost; the normal-equation entries it adds to are not read here)
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
    if (d.any_plane) {   // PlaneFactors and the PoseAnchorFactor (use_plane / use_anchor)
      std::vector<double> pp((size_t)MAX_PLANE * PLANE_PART), ap(ANCHOR_PART);
      HIPCHK(c, hipMemcpy(pp.data(), d.plane_part, sizeof(double) * pp.size(), hipMemcpyDeviceToHost));
      HIPCHK(c, hipMemcpy(ap.data(), d.anchor_part, sizeof(double) * ap.size(), hipMemcpyDeviceToHost));
      for (int q = 0; q < ds[0].n_plane; q++) total += pp[(size_t)q * PLANE_PART + PLANE_PART - 2];
      if (ds[0].use_anchor) total += ap[ANCHOR_PART - 2];
    }
    if (ds[0].gnss_factors) {           // GNSS factors inside the window (gnss_ready and not lowspeed)
      double gc[2];
      HIPCHK(c, hipMemcpy(gc, d.gnss_cost, sizeof gc, hipMemcpyDeviceToHost));
      total += gc[0];
    }
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
  if (need > c->scratch_pin_cap) {
    if (c->scratch_pin) (void)hipHostFree(c->scratch_pin);
    c->scratch_pin = nullptr; c->scratch_pin_cap = 0;
    const size_t cap = std::max<size_t>(need + need / 2, (size_t)64 << 10);
    if (hipHostMalloc((void **)&c->scratch_pin, cap) != hipSuccess) { (void)hipGetLastError(); c->err = "hipHostMalloc(pre-integration staging) failed"; return GFBE_DEVICE_ERROR; }
    c->scratch_pin_cap = cap;
  }
  // arguments packed into the pinned mirror of the scratch: ONE host-to-device copy, the kernel, ONE copy back (five pageable
  // copies up and one down cost more than the integration of one interval)
  char *base = c->scratch, *pin = c->scratch_pin;
  const size_t o_off = 0, o_s = o_off + b_off, o_f = o_s + b_s, o_l = o_f + b_f, o_n = o_l + b_l, o_o = o_n + b_n;
  std::memcpy(pin + o_off, offset, sizeof(int) * (n + 1));
  std::memcpy(pin + o_s, samples, sizeof(double) * 7 * tot);
  if (first) std::memcpy(pin + o_f, first, sizeof(double) * 6 * n);
  if (lin) std::memcpy(pin + o_l, lin, sizeof(double) * lin_w * n);
  if (noise) std::memcpy(pin + o_n, noise, sizeof(double) * n_noise);
  int *d_off = (int *)(base + o_off);
  double *d_s = (double *)(base + o_s), *d_f = (double *)(base + o_f), *d_l = (double *)(base + o_l), *d_n = (double *)(base + o_n);
  REC_T *d_o = (REC_T *)(base + o_o);
  HIPCHK(c, hipMemcpyAsync(base, pin, o_o, hipMemcpyHostToDevice, c->stream));
  if (imu) launch_preint_imu(n, d_off, d_s, d_f, d_l, d_n, (gfbe_imu_preint *)d_o, c->stream);
  else launch_preint_wheel(n, d_off, d_s, d_f, d_l, d_n, (gfbe_wheel_preint *)d_o, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(pin + o_o, d_o, sizeof(REC_T) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memcpy(out, pin + o_o, sizeof(REC_T) * n);
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
  if (b && !b->place.empty() && w >= 0 && w < (int)b->place.size()) w = b->place[w];      // (a split batch holds its windows sorted by size)
  if (b && b->second && w > b->d.B) return gfbe_debug_timing(c, b->second, w - b->d.B, out32);
  if (!c || !b || !out32 || w < 0 || w > b->d.B) return GFBE_BAD_INPUT;   // (w == B: the extra block of the first part)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(out32, b->d.timing + (size_t)w * 32, sizeof(double) * 32, hipMemcpyDeviceToHost));
  return GFBE_OK;
}
extern "C" gfbe_status gfbe_debug_vector(gfbe_ctx *c, gfbe_batch *b, int32_t w, int32_t which, double *out) {
  if (!c || !b || !out || which < 0 || (which > 3 && !(which >= 1000 && which < 1000 + ND) && !(which >= 2000 && which <= 2000 + NV))) return GFBE_BAD_INPUT;
  int off = (!b->place.empty() && w >= 0 && w < (int)b->place.size()) ? b->place[w] : w;
  gfbe_batch *p = b;
  while (p && off >= p->d.B) { off -= p->d.B; p = p->second; }
  if (!p || off < 0) return GFBE_BAD_INPUT;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (which >= 1000 && which < 1000 + ND) {           // row (which - 1000) of the assembled H (lower triangle valid)
    HIPCHK(c, hipMemcpy(out, p->d.H + (size_t)off * ND * ND + (size_t)(which - 1000) * ND, sizeof(double) * ND, hipMemcpyDeviceToHost));
    return GFBE_OK;
  }
  if (which >= 2000 && which < 2000 + NV + 1) {       // row (which - 2000) of E (NV entries; row NV: eg)
    const double *srcE = which == 2000 + NV ? p->d.eg + (size_t)off * NV : p->d.E + (size_t)off * NV * NV + (size_t)(which - 2000) * NV;
    HIPCHK(c, hipMemcpy(out, srcE, sizeof(double) * NV, hipMemcpyDeviceToHost));
    return GFBE_OK;
  }
  const double *src = which == 0 ? p->d.yp : which == 1 ? p->d.vp : which == 2 ? p->d.sp : p->d.g;
  HIPCHK(c, hipMemcpy(out, src + (size_t)off * ND, sizeof(double) * ND, hipMemcpyDeviceToHost));
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

// One robot, frame by frame, from a compiled caller: the steady-state branch of Estimator::processImage
// (vins_estimator/src/estimator/estimator.cpp:1133-1215) on top of libgfbe.so with the FeatureManager lists resident on the
// device —
//   f_manager.addFeatureCheckParallax     -> gfbe_ftab_add_frame            (keyframe decision = the marginalisation flag)
//   f_manager.triangulate                 -> gfbe_ftab_triangulate
//   processIMU's pre_integrations[j]      -> gfbe_preintegrate_imu / _wheel (only the interval that changed)
//   optimization()                        -> gfbe_batch_upload_tables + gfbe_batch_solve + gfbe_batch_download
//   f_manager.setDepth                    -> gfbe_ftab_set_depth
//   movingConsistencyCheckW + removeOutlier -> gfbe_ftab_check_outliers + gfbe_ftab_remove_outlier
//   slideWindow                           -> gfbe_slide_window_state + gfbe_ftab_remove_back_shift_depth / _remove_front
//   f_manager.removeFailures              -> gfbe_ftab_remove_failures
// — the loop ground-fusion2_amd/stream.py::run_stream drives from Python (tests/test_gpu_stream.py), without the interpreter:
// what a frame costs a C++ estimator. The stream (tracker output, raw inertial / wheel samples, initial state) is read from a
// file written by tools/dump_stream.py; the newest pose after every solve goes to a second file.
//   g++ -O2 -std=c++17 -I include examples/stream_loop.cpp -L ground-fusion2_amd/csrc -lgfbe -Wl,-rpath,$PWD/ground-fusion2_amd/csrc -o stream_loop
//   ./stream_loop stream.bin traj.bin
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gfbe.h"

namespace {
constexpr int W = GFBE_WINDOW_SIZE;
struct Interval { std::vector<double> samples; double first[6]; int n() const { return (int)(samples.size() / 7); } };
struct Frame { std::vector<int32_t> ids; std::vector<double> obs; };
struct Stream {
  int n_kf = 0, use_wheel = 0;
  double min_parallax = 0, depth_threshold = 0, tic[3], ric[9], ba[3], bg[3], imu_noise[4], wheel_noise[2];
  gfbe_state st{};
  std::vector<Frame> frames;
  std::vector<Interval> imu, wheel;
};

bool read_stream(const char *path, Stream &S) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return false;
  bool ok = true;
  auto I = [&](int32_t *p, size_t n) { ok = ok && std::fread(p, 4, n, f) == n; };
  auto D = [&](double *p, size_t n) { ok = ok && std::fread(p, 8, n, f) == n; };
  int32_t head[4];
  I(head, 4);
  if (!ok || head[0] != 0x47465354 || head[2] != W) { std::fclose(f); return false; }
  S.n_kf = head[1]; S.use_wheel = head[3];
  double two[2];
  D(two, 2); S.min_parallax = two[0]; S.depth_threshold = two[1];
  D(S.tic, 3); D(S.ric, 9); D(S.ba, 3); D(S.bg, 3); D(S.imu_noise, 4); D(S.wheel_noise, 2);
  D(&S.st.para_Pose[0][0], 11 * 7); D(&S.st.para_SpeedBias[0][0], 11 * 9); D(S.st.para_Ex_Pose, 7); D(S.st.para_Ex_Pose_wheel, 7);
  D(S.st.para_Ix_wheel, 3); D(two, 2); S.st.para_Td = two[0]; S.st.para_Td_wheel = two[1];
  S.frames.resize(S.n_kf);
  for (Frame &fr : S.frames) {
    int32_t n; I(&n, 1);
    if (!ok) break;
    fr.ids.resize(n); fr.obs.resize((size_t)n * 8);
    I(fr.ids.data(), n); D(fr.obs.data(), (size_t)n * 8);
  }
  for (std::vector<Interval> *set : {&S.imu, &S.wheel}) {
    if (set == &S.wheel && !S.use_wheel) break;
    int32_t cnt; I(&cnt, 1);
    if (!ok) break;
    set->resize(cnt);
    for (Interval &iv : *set) {
      int32_t n; I(&n, 1);
      if (!ok) break;
      iv.samples.resize((size_t)n * 7);
      D(iv.samples.data(), (size_t)n * 7); D(iv.first, 6);
    }
  }
  std::fclose(f);
  return ok;
}

// ---- small SO(3) helpers (quaternions x y z w, Hamilton product: utility.h / Eigen's convention)
void qmul(const double a[4], const double b[4], double o[4]) {
  const double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[0] = aw * bx + ax * bw + ay * bz - az * by; o[1] = aw * by + ay * bw + az * bx - ax * bz;
  o[2] = aw * bz + az * bw + ax * by - ay * bx; o[3] = aw * bw - ax * bx - ay * by - az * bz;
}
void qnormalize(double q[4]) { const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]); for (int k = 0; k < 4; k++) q[k] /= n; }
void qrot(const double qi[4], double R[9]) {
  double q[4] = {qi[0], qi[1], qi[2], qi[3]};
  qnormalize(q);
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
void mv(const double R[9], const double v[3], double o[3]) { for (int i = 0; i < 3; i++) o[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2]; }
void pose_rows(const double pose[][7], int n, double *out) {   // [P | R row-major] per frame: the pose argument of the table calls
  for (int k = 0; k < n; k++) { std::memcpy(out + 12 * k, pose[k], 24); qrot(pose[k] + 3, out + 12 * k + 3); }
}
// processIMU's dead reckoning of the newest frame through the incoming interval (estimator.cpp:898-927: mid-point rule)
void propagate(double pose[7], double sb[9], const Interval &iv, double g_norm) {
  double P[3] = {pose[0], pose[1], pose[2]}, V[3] = {sb[0], sb[1], sb[2]}, q[4] = {pose[3], pose[4], pose[5], pose[6]};
  qnormalize(q);
  const double *ba = sb + 3, *bg = sb + 6, g[3] = {0, 0, g_norm};
  double acc0[3] = {iv.first[0], iv.first[1], iv.first[2]}, gyr0[3] = {iv.first[3], iv.first[4], iv.first[5]};
  for (int s = 0; s < iv.n(); s++) {
    const double *row = &iv.samples[(size_t)7 * s], dt = row[0], *acc1 = row + 1, *gyr1 = row + 4;
    double R[9], a[3], un0[3], un1[3], dq[4], qn[4];
    qrot(q, R);
    for (int k = 0; k < 3; k++) a[k] = acc0[k] - ba[k];
    mv(R, a, un0);
    for (int k = 0; k < 3; k++) { un0[k] -= g[k]; dq[k] = 0.5 * (0.5 * (gyr0[k] + gyr1[k]) - bg[k]) * dt; }
    dq[3] = 1.0;
    qmul(q, dq, qn);
    qnormalize(qn);
    std::memcpy(q, qn, sizeof q);
    qrot(q, R);
    for (int k = 0; k < 3; k++) a[k] = acc1[k] - ba[k];
    mv(R, a, un1);
    for (int k = 0; k < 3; k++) {
      const double un = 0.5 * (un0[k] + (un1[k] - g[k]));
      P[k] += dt * V[k] + 0.5 * dt * dt * un;
      V[k] += dt * un;
      acc0[k] = acc1[k]; gyr0[k] = gyr1[k];
    }
  }
  for (int k = 0; k < 3; k++) { pose[k] = P[k]; sb[k] = V[k]; }
  std::memcpy(pose + 3, q, sizeof q);
}

// one pre-integration slot = the raw intervals it covers (MARGIN_SECOND_NEW merges the two newest, estimator.cpp:3790-3806)
struct Slot { std::vector<int> parts; bool dirty = true; };
template <typename REC, typename FN>
bool integrate_dirty(std::vector<Slot> &slots, std::vector<REC> &rec, const std::vector<Interval> &raw, const double *lin, int lin_w, FN call) {
  std::vector<int32_t> off(1, 0);
  std::vector<double> samples, first, linv;
  std::vector<int> which;
  for (int i = 0; i < (int)slots.size(); i++) {
    if (!slots[i].dirty) continue;
    for (int p : slots[i].parts) samples.insert(samples.end(), raw[p].samples.begin(), raw[p].samples.end());
    off.push_back((int32_t)(samples.size() / 7));
    first.insert(first.end(), raw[slots[i].parts[0]].first, raw[slots[i].parts[0]].first + 6);
    linv.insert(linv.end(), lin, lin + lin_w);
    which.push_back(i);
  }
  if (which.empty()) return true;
  std::vector<REC> out(which.size());
  if (!call((int32_t)which.size(), off.data(), samples.data(), first.data(), linv.data(), out.data())) return false;
  for (size_t k = 0; k < which.size(); k++) { rec[which[k]] = out[k]; slots[which[k]].dirty = false; }
  return true;
}
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

int main(int argc, char **argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s stream.bin traj_out.bin\n", argv[0]); return 2; }
  Stream S;
  if (!read_stream(argv[1], S)) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  setenv("GPU_MAX_HW_QUEUES", "8", 0);   // before HIP initialises
  gfbe_options opt;
  gfbe_default_options(&opt);
  gfbe_ctx *ctx = nullptr;
  if (gfbe_create(&ctx, 0, &opt) != GFBE_OK) { std::fprintf(stderr, "gfbe_create: %s\n", gfbe_last_error(ctx)); return 1; }   // (no GPU: fails loudly)
#define CHECK(call) do { if ((call) != GFBE_OK) { std::fprintf(stderr, "%s: %s\n", #call, gfbe_last_error(ctx)); return 1; } } while (0)
  gfbe_ftab_options fo;
  gfbe_ftab_default_options(&fo);
  fo.min_parallax = S.min_parallax; fo.depth_threshold = S.depth_threshold;
  const int32_t CAP = 16384;
  gfbe_ftab *tab = nullptr;
  CHECK(gfbe_ftab_create(ctx, 1, CAP, &fo, &tab));

  gfbe_state st = S.st;
  double tic_ric[12];
  std::memcpy(tic_ric, S.tic, 24); std::memcpy(tic_ric + 3, S.ric, 72);
  const double td0 = 0.0;
  int32_t kf = 0, counters[3];
  double parallax = 0.0;
  for (int fc = 0; fc < W; fc++) {      // frames 0 .. 9 fill the window (the initialisation phase is out of scope)
    const Frame &fr = S.frames[fc];
    const int32_t off[2] = {0, (int32_t)fr.ids.size()};
    CHECK(gfbe_ftab_add_frame(ctx, tab, &fc, off, fr.ids.data(), fr.obs.data(), &td0, &kf, counters, &parallax));
  }
  std::vector<Slot> imu_slots(W), wheel_slots(W);
  for (int i = 0; i < W; i++) { imu_slots[i].parts = {i}; wheel_slots[i].parts = {i}; }
  std::vector<gfbe_imu_preint> imu_rec(W);
  std::vector<gfbe_wheel_preint> wheel_rec(W);
  int32_t frame_idx[W];
  for (int i = 0; i < W; i++) frame_idx[i] = i;
  const double lin_imu[6] = {S.ba[0], S.ba[1], S.ba[2], S.bg[0], S.bg[1], S.bg[2]}, lin_wheel[4] = {1.0, 1.0, 1.0, 0.0};

  gfbe_prior prior{};
  std::vector<double> J0((size_t)GFBE_DENSE_DIM * GFBE_DENSE_DIM), r0(GFBE_DENSE_DIM);
  prior.J0 = J0.data(); prior.r0 = r0.data(); prior.valid = 0;
  std::vector<double> feature(CAP), poses(12 * (W + 1)), traj, frame_ms, costs;
  std::vector<int32_t> rm_ids(CAP), iters, flags, n_lm, n_out;

  for (int k = W; k < S.n_kf; k++) {
    const double t0 = now_ms();
    const Frame &fr = S.frames[k];
    const int32_t fcW = W, off[2] = {0, (int32_t)fr.ids.size()};
    CHECK(gfbe_ftab_add_frame(ctx, tab, &fcW, off, fr.ids.data(), fr.obs.data(), &td0, &kf, counters, &parallax));
    const int32_t flag = kf ? GFBE_MARGIN_OLD : GFBE_MARGIN_SECOND_NEW;      // estimator.cpp:1005-1013
    pose_rows(st.para_Pose, W + 1, poses.data());
    CHECK(gfbe_ftab_triangulate(ctx, tab, poses.data(), tic_ric, 0));
    if (!integrate_dirty(imu_slots, imu_rec, S.imu, lin_imu, 6, [&](int32_t n, const int32_t *o, const double *s, const double *f, const double *l, gfbe_imu_preint *out) {
          return gfbe_preintegrate_imu(ctx, n, o, s, f, l, S.imu_noise, out) == GFBE_OK; })) { std::fprintf(stderr, "preintegrate_imu: %s\n", gfbe_last_error(ctx)); return 1; }
    if (S.use_wheel && !integrate_dirty(wheel_slots, wheel_rec, S.wheel, lin_wheel, 4, [&](int32_t n, const int32_t *o, const double *s, const double *f, const double *l, gfbe_wheel_preint *out) {
          return gfbe_preintegrate_wheel(ctx, n, o, s, f, l, S.wheel_noise, out) == GFBE_OK; })) { std::fprintf(stderr, "preintegrate_wheel: %s\n", gfbe_last_error(ctx)); return 1; }

    gfbe_window win{};
    win.frame_count = W;
    win.state = st;
    win.ex_cam_const = win.ex_wheel_const = win.ix_wheel_const = win.td_const = win.td_wheel_const = 1;      // ESTIMATE_EXTRINSIC / _TD = 0
    win.n_imu = W; win.imu_frame = frame_idx; win.imu = imu_rec.data();
    if (S.use_wheel) { win.n_wheel = W; win.wheel_frame = frame_idx; win.wheel = wheel_rec.data(); }
    win.prior = prior.valid ? &prior : nullptr;
    const gfbe_window *wins[1] = {&win};
    gfbe_batch *batch = nullptr;
    CHECK(gfbe_batch_upload_tables(ctx, tab, 1, wins, &batch));      // the landmarks go from the tables to the solver on the device
    CHECK(gfbe_batch_solve(ctx, batch, flag));
    const int32_t L = gfbe_batch_feature_count(batch, 0);
    gfbe_summary sum{};
    double *featp[1] = {feature.data()};
    gfbe_prior *priorp[1] = {&prior};
    const gfbe_status rc = gfbe_batch_download(ctx, batch, &st, featp, priorp, &sum);
    if (rc != GFBE_OK && rc != GFBE_NO_CONVERGENCE) { std::fprintf(stderr, "gfbe_batch_download: status %d %s\n", (int)rc, gfbe_last_error(ctx)); return 1; }   // (NO_CONVERGENCE: the 8-iteration budget ran out, the normal case)
    gfbe_batch_free(ctx, batch);
    const int32_t offL[2] = {0, L};
    CHECK(gfbe_ftab_set_depth(ctx, tab, offL, feature.data()));
    traj.insert(traj.end(), st.para_Pose[W], st.para_Pose[W] + 7);
    iters.push_back(sum.iterations); flags.push_back(flag); costs.push_back(sum.final_cost); n_lm.push_back(L);
    // movingConsistencyCheckW + removeOutlier (estimator.cpp:1171-1176)
    pose_rows(st.para_Pose, W + 1, poses.data());
    const int32_t offC[2] = {0, CAP};
    int32_t n_rm = 0;
    CHECK(gfbe_ftab_check_outliers(ctx, tab, poses.data(), tic_ric, 1, offC, rm_ids.data(), &n_rm));
    n_out.push_back(n_rm);
    const int32_t offR[2] = {0, n_rm};
    CHECK(gfbe_ftab_remove_outlier(ctx, tab, offR, rm_ids.data()));
    // slideWindow (estimator.cpp:3700-3899)
    double back[12], new0[12], marg[12], nw[12], tmp[3];
    pose_rows(st.para_Pose, 1, back);
    gfbe_slide_window_state(&st, flag);
    const bool more = k < S.n_kf - 1;
    if (flag == GFBE_MARGIN_OLD) {
      imu_slots.erase(imu_slots.begin()); imu_rec.erase(imu_rec.begin());
      imu_slots.push_back(Slot{{k}, true}); imu_rec.emplace_back();
      if (S.use_wheel) { wheel_slots.erase(wheel_slots.begin()); wheel_rec.erase(wheel_rec.begin()); wheel_slots.push_back(Slot{{k}, true}); wheel_rec.emplace_back(); }
      pose_rows(st.para_Pose, 1, new0);
      // marg_P = back_P0 + back_R0 tic, marg_R = back_R0 ric; likewise for the new frame 0 (estimator.cpp:3872-3885)
      for (const auto &pr : {std::make_pair(back, marg), std::make_pair(new0, nw)}) {
        mv(pr.first + 3, S.tic, tmp);
        for (int a = 0; a < 3; a++) pr.second[a] = pr.first[a] + tmp[a];
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) { double s = 0; for (int c = 0; c < 3; c++) s += pr.first[3 + 3 * a + c] * S.ric[3 * c + b]; pr.second[3 + 3 * a + b] = s; }
      }
      CHECK(gfbe_ftab_remove_back_shift_depth(ctx, tab, marg, nw));
    } else {
      imu_slots[W - 2].parts.insert(imu_slots[W - 2].parts.end(), imu_slots[W - 1].parts.begin(), imu_slots[W - 1].parts.end());
      imu_slots[W - 2].dirty = true; imu_slots[W - 1] = Slot{{k}, true};
      if (S.use_wheel) {
        wheel_slots[W - 2].parts.insert(wheel_slots[W - 2].parts.end(), wheel_slots[W - 1].parts.begin(), wheel_slots[W - 1].parts.end());
        wheel_slots[W - 2].dirty = true; wheel_slots[W - 1] = Slot{{k}, true};
      }
      CHECK(gfbe_ftab_remove_front(ctx, tab, &fcW));
    }
    CHECK(gfbe_ftab_remove_failures(ctx, tab));
    frame_ms.push_back(now_ms() - t0);
    if (more) propagate(st.para_Pose[W], st.para_SpeedBias[W], S.imu[k], opt.g_norm);      // the next image's processIMU
  }
  int32_t n_feat = 0;
  CHECK(gfbe_ftab_size(ctx, tab, &n_feat));
  gfbe_ftab_destroy(ctx, tab);
  gfbe_destroy(ctx);

  FILE *f = std::fopen(argv[2], "wb");
  if (!f) { std::fprintf(stderr, "cannot write %s\n", argv[2]); return 2; }
  const int32_t n = (int32_t)iters.size();
  std::fwrite(&n, 4, 1, f);
  std::fwrite(traj.data(), 8, traj.size(), f); std::fwrite(costs.data(), 8, costs.size(), f);
  std::fwrite(iters.data(), 4, iters.size(), f); std::fwrite(flags.data(), 4, flags.size(), f);
  std::fwrite(n_lm.data(), 4, n_lm.size(), f); std::fwrite(n_out.data(), 4, n_out.size(), f);
  std::fclose(f);
  std::vector<double> sorted = frame_ms;
  std::sort(sorted.begin(), sorted.end());
  double mean3 = 0;
  for (size_t i = 3; i < frame_ms.size(); i++) mean3 += frame_ms[i];
  std::printf("compiled loop: %d frames, %d features in the table at the end; per frame: median %.3f ms, mean after 3 frames %.3f ms, max %.3f ms\n",
              n, n_feat, sorted[sorted.size() / 2], frame_ms.size() > 3 ? mean3 / (frame_ms.size() - 3) : 0.0, sorted.back());
  return 0;
}

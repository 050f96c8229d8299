// oracle/gfo_ftab.cpp — TEST INFRASTRUCTURE ONLY (CPU oracle).
// FeatureManager / slideWindow operations restated with the reference's own container semantics
// (a std::list of features in insertion order, a std::vector of observations per feature):
//   addFeatureCheckParallax   VE/estimator/feature_manager.cpp:57-116, compensatedParallax2 :978-1011
//   setDepth :249-267, removeFailures :269-278, clearDepth :280-284, getDepthVector :286-302
//   triangulate :669-724 (JacobiSVD -> here a one-sided Jacobi SVD), triangulateWithDepth :726-799
//   removeOutlier :801-816, removeBackShiftDepth :818-856, removeBack :858-874, removeFront :914-934
//   Estimator::outliersRejection estimator.cpp:3971-4028, movingConsistencyCheckW :4030-4074
//   Estimator::slideWindow (state shift) estimator.cpp:3700-3858
// Same C signatures as the product's gfbe_ftab_* (ctx ignored).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <list>
#include <set>
#include <vector>

#include "gfo_api.h"

namespace {
struct Obs { double v[8]; double td; };
struct Feat {
  int id, start, eflag = 0, sflag = 0, used = 0;
  double depth = -1.0;
  std::vector<Obs> obs;
  int end_frame() const { return start + (int)obs.size() - 1; }
};
}  // namespace
struct gfo_ftab {
  std::vector<std::list<Feat>> tab;
  gfbe_ftab_options opt;
  int cap;
};

namespace {
struct V3 { double x, y, z; };
inline V3 mv(const double *R, V3 a) { return {R[0] * a.x + R[1] * a.y + R[2] * a.z, R[3] * a.x + R[4] * a.y + R[5] * a.z, R[6] * a.x + R[7] * a.y + R[8] * a.z}; }
inline V3 tmv(const double *R, V3 a) { return {R[0] * a.x + R[3] * a.y + R[6] * a.z, R[1] * a.x + R[4] * a.y + R[7] * a.z, R[2] * a.x + R[5] * a.y + R[8] * a.z}; }
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 scl(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline void mm(const double *A, const double *B, double *C) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j]; C[3 * i + j] = s; } }
inline void tmm(const double *A, const double *B, double *C) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += A[3 * k + i] * B[3 * k + j]; C[3 * i + j] = s; } }
inline V3 pt(const Obs &o) { return {o.v[0], o.v[1], o.v[2]}; }

// smallest right singular vector of an m x 4 matrix: one-sided (Hestenes) Jacobi on the columns
void smallest_right_sv(std::vector<double> A, int m, double v[4]) {
  double V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rot = false;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 4; q++) {
        double a = 0, b = 0, c = 0;
        for (int i = 0; i < m; i++) { a += A[4 * i + p] * A[4 * i + p]; b += A[4 * i + q] * A[4 * i + q]; c += A[4 * i + p] * A[4 * i + q]; }
        if (c == 0.0 || std::fabs(c) <= 1e-16 * std::sqrt(a * b)) continue;
        rot = true;
        const double zeta = (b - a) / (2.0 * c);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
        for (int i = 0; i < m; i++) { const double x = A[4 * i + p], y = A[4 * i + q]; A[4 * i + p] = cs * x - sn * y; A[4 * i + q] = sn * x + cs * y; }
        for (int i = 0; i < 4; i++) { const double x = V[4 * i + p], y = V[4 * i + q]; V[4 * i + p] = cs * x - sn * y; V[4 * i + q] = sn * x + cs * y; }
      }
    if (!rot) break;
  }
  int best = 0; double bn = 1e300;
  for (int j = 0; j < 4; j++) { double n = 0; for (int i = 0; i < m; i++) n += A[4 * i + j] * A[4 * i + j]; if (n < bn) { bn = n; best = j; } }
  for (int i = 0; i < 4; i++) v[i] = V[4 * i + best];
}

double reproj_err(const double *Ri, V3 Pi, const double *ric, V3 tic, const double *Rj, V3 Pj, double depth, V3 uvi, V3 uvj) {
  const V3 pw = add(mv(Ri, add(mv(ric, scl(depth, uvi)), tic)), Pi);
  const V3 pc = tmv(ric, sub(tmv(Rj, sub(pw, Pj)), tic));
  const double rx = pc.x / pc.z - uvj.x, ry = pc.y / pc.z - uvj.y;
  return std::sqrt(rx * rx + ry * ry);
}
double reproj_err3d(const double *Ri, V3 Pi, const double *ric, V3 tic, const double *Rj, V3 Pj, double depth, V3 uvi, V3 uvj) {
  const V3 pw = add(mv(Ri, add(mv(ric, scl(depth, uvi)), tic)), Pi);
  const V3 pc = tmv(ric, sub(tmv(Rj, sub(pw, Pj)), tic));
  const V3 d = sub(pc, uvj);
  return std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z) / depth;
}
double parallax2(const Feat &f, int frame_count) {   // compensatedParallax2
  const Obs &fi = f.obs[frame_count - 2 - f.start], &fj = f.obs[frame_count - 1 - f.start];
  const double uj = fj.v[0], vj = fj.v[1];
  const double dep = fi.v[2], ui = fi.v[0] / dep, vi = fi.v[1] / dep;
  const double du = ui - uj, dv = vi - vj;
  return std::max(0.0, std::sqrt(std::min(du * du + dv * dv, du * du + dv * dv)));
}
}  // namespace

extern "C" {

void gfo_ftab_default_options(gfbe_ftab_options *o) { o->init_depth = 5.0; o->focal_length = 600.0; o->min_parallax = 10.0 / 600.0; o->depth_threshold = 3.0; }

int32_t gfo_ftab_create(void *, int32_t n, int32_t cap, const gfbe_ftab_options *opt, gfo_ftab **out) {
  gfo_ftab *t = new gfo_ftab();
  t->tab.resize(n); t->cap = cap;
  if (opt) t->opt = *opt; else gfo_ftab_default_options(&t->opt);
  *out = t;
  return GFBE_OK;
}
void gfo_ftab_destroy(void *, gfo_ftab *t) { delete t; }

int32_t gfo_ftab_add_frame(void *, gfo_ftab *t, const int32_t *frame_count, const int32_t *offset, const int32_t *feature_id,
                           const double *obs8, const double *td, int32_t *keyframe, int32_t *counters, double *avg_parallax) {
  for (size_t w = 0; w < t->tab.size(); w++) {
    auto &L = t->tab[w];
    const int fc = frame_count[w];
    double psum = 0.0; int pnum = 0, last_track = 0, fresh = 0, longt = 0;
    for (int k = offset[w]; k < offset[w + 1]; k++) {
      Obs o; std::memcpy(o.v, obs8 + 8 * (size_t)k, sizeof o.v); o.td = td[w];
      const int id = feature_id[k];
      auto it = std::find_if(L.begin(), L.end(), [id](const Feat &f) { return f.id == id; });
      if (it == L.end()) { Feat f; f.id = id; f.start = fc; f.obs.push_back(o); L.push_back(f); fresh++; }
      else { it->obs.push_back(o); last_track++; if (it->obs.size() >= 4) longt++; }
    }
    if ((int)L.size() > t->cap) return GFBE_BAD_INPUT;
    if (counters) { counters[3 * w] = last_track; counters[3 * w + 1] = fresh; counters[3 * w + 2] = longt; }
    double avg = 0.0; int kf;
    if (fc < 2 || last_track < 20 || longt < 40 || fresh > 0.5 * last_track) kf = 1;
    else {
      for (const Feat &f : L)
        if (f.start <= fc - 2 && f.start + (int)f.obs.size() - 1 >= fc - 1) { psum += parallax2(f, fc); pnum++; }
      if (pnum == 0) kf = 1;
      else { avg = psum / pnum * t->opt.focal_length; kf = (psum / pnum >= t->opt.min_parallax) ? 1 : 0; }
    }
    if (keyframe) keyframe[w] = kf;
    if (avg_parallax) avg_parallax[w] = avg;
  }
  return GFBE_OK;
}

int32_t gfo_ftab_remove_back_shift_depth(void *, gfo_ftab *t, const double *margPR, const double *newPR) {
  for (size_t w = 0; w < t->tab.size(); w++) {
    auto &L = t->tab[w];
    const double *mP = margPR + 12 * w, *mR = mP + 3, *nP = newPR + 12 * w, *nR = nP + 3;
    for (auto it = L.begin(), nx = L.begin(); it != L.end(); it = nx) {
      nx++;
      if (it->start != 0) { it->start--; continue; }
      const V3 uv = pt(it->obs[0]);
      it->obs.erase(it->obs.begin());
      if (it->obs.size() < 2) { L.erase(it); continue; }
      const V3 pi = scl(it->depth, uv);
      const V3 wp = add(mv(mR, pi), {mP[0], mP[1], mP[2]});
      const V3 pj = tmv(nR, sub(wp, {nP[0], nP[1], nP[2]}));
      it->depth = pj.z > 0 ? pj.z : t->opt.init_depth;
    }
  }
  return GFBE_OK;
}
int32_t gfo_ftab_remove_back(void *, gfo_ftab *t) {
  for (auto &L : t->tab)
    for (auto it = L.begin(), nx = L.begin(); it != L.end(); it = nx) {
      nx++;
      if (it->start != 0) it->start--;
      else { it->obs.erase(it->obs.begin()); if (it->obs.empty()) L.erase(it); }
    }
  return GFBE_OK;
}
int32_t gfo_ftab_remove_front(void *, gfo_ftab *t, const int32_t *frame_count) {
  for (size_t w = 0; w < t->tab.size(); w++) {
    auto &L = t->tab[w];
    const int fc = frame_count[w];
    for (auto it = L.begin(), nx = L.begin(); it != L.end(); it = nx) {
      nx++;
      if (it->start == fc) { it->start--; continue; }
      const int j = GFBE_WINDOW_SIZE - 1 - it->start;
      if (it->end_frame() < fc - 1) continue;
      it->obs.erase(it->obs.begin() + j);
      if (it->obs.empty()) L.erase(it);
    }
  }
  return GFBE_OK;
}
int32_t gfo_ftab_remove_outlier(void *, gfo_ftab *t, const int32_t *offset, const int32_t *ids) {
  for (size_t w = 0; w < t->tab.size(); w++) {
    std::set<int> S(ids + offset[w], ids + offset[w + 1]);
    auto &L = t->tab[w];
    for (auto it = L.begin(), nx = L.begin(); it != L.end(); it = nx) { nx++; if (S.count(it->id)) L.erase(it); }
  }
  return GFBE_OK;
}
int32_t gfo_ftab_remove_failures(void *, gfo_ftab *t) {
  for (auto &L : t->tab)
    for (auto it = L.begin(), nx = L.begin(); it != L.end(); it = nx) { nx++; if (it->sflag == 2) L.erase(it); }
  return GFBE_OK;
}
int32_t gfo_ftab_clear_depth(void *, gfo_ftab *t) {
  for (auto &L : t->tab) for (auto &f : L) f.depth = -1.0;
  return GFBE_OK;
}
int32_t gfo_ftab_set_depth(void *, gfo_ftab *t, const int32_t *offset, const double *x) {
  for (size_t w = 0; w < t->tab.size(); w++) {
    int idx = -1;
    for (auto &f : t->tab[w]) {
      f.used = (int)f.obs.size();
      if (f.used < 4) continue;
      f.depth = 1.0 / x[offset[w] + ++idx];
      f.sflag = f.depth < 0 ? 2 : 1;
    }
  }
  return GFBE_OK;
}
int32_t gfo_ftab_get_depth_vector(void *, gfo_ftab *t, const int32_t *offset, double *x, int32_t *count) {
  for (size_t w = 0; w < t->tab.size(); w++) {
    int idx = -1;
    for (auto &f : t->tab[w]) {
      f.used = (int)f.obs.size();
      if (f.used < 4) continue;
      ++idx;
      if (offset[w] + idx < offset[w + 1]) x[offset[w] + idx] = 1.0 / f.depth;
    }
    if (count) count[w] = idx + 1;
  }
  return GFBE_OK;
}

int32_t gfo_ftab_triangulate(void *, gfo_ftab *t, const double *poses, const double *tic_ric, int32_t with_depth) {
  for (size_t w = 0; w < t->tab.size(); w++) {
    const double *PR = poses + 132 * w, *tic_ = tic_ric + 12 * w, *ric = tic_ + 3;
    const V3 tic = {tic_[0], tic_[1], tic_[2]};
    auto camT = [&](int f) { const double *P = PR + 12 * f; return add({P[0], P[1], P[2]}, mv(P + 3, tic)); };
    auto camR = [&](int f, double *R) { mm(PR + 12 * f + 3, ric, R); };
    for (auto &f : t->tab[w]) {
      if (!with_depth) {
        if (f.depth > 0) continue;
        f.used = (int)f.obs.size();
        if (f.used < 4) continue;
        const int i0 = f.start;
        const V3 t0 = camT(i0);
        double R0[9]; camR(i0, R0);
        std::vector<double> A(8 * f.obs.size());
        int row = 0, j = i0 - 1;
        for (const Obs &o : f.obs) {
          j++;
          const V3 t1 = camT(j);
          double R1[9], R[9]; camR(j, R1);
          const V3 tt = tmv(R0, sub(t1, t0));
          tmm(R0, R1, R);
          // P = [R^T | -R^T t]
          double P[12];
          const V3 mt = tmv(R, tt);
          for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) P[4 * r + c] = R[3 * c + r]; }
          P[3] = -mt.x; P[7] = -mt.y; P[11] = -mt.z;
          const double n = std::sqrt(o.v[0] * o.v[0] + o.v[1] * o.v[1] + o.v[2] * o.v[2]);
          const double fx = o.v[0] / n, fy = o.v[1] / n, fz = o.v[2] / n;
          for (int c = 0; c < 4; c++) { A[4 * row + c] = fx * P[8 + c] - fz * P[c]; A[4 * (row + 1) + c] = fy * P[8 + c] - fz * P[4 + c]; }
          row += 2;
        }
        double v[4];
        smallest_right_sv(A, row, v);
        f.depth = v[2] / v[3];
        f.eflag = 2;
        if (f.depth < 0.1) { f.depth = t->opt.init_depth; f.eflag = 0; }
      } else {
        f.used = (int)f.obs.size();
        if (f.used < 4) continue;
        if (f.depth > 0) continue;
        const int s = f.start;
        const V3 tr = camT(s);
        double Rr[9]; camR(s, Rr);
        double sum = 0.0; int cnt = 0;
        for (int i = 0; i < (int)f.obs.size(); i++) {
          const V3 t0 = camT(s + i);
          double R0[9]; camR(s + i, R0);
          const double dep = f.obs[i].v[7];
          if (dep < 0.1 || dep > t->opt.depth_threshold) continue;
          const V3 p0 = scl(dep, pt(f.obs[i]));
          const V3 t2r = tmv(Rr, sub(t0, tr));
          double R2r[9]; tmm(Rr, R0, R2r);
          for (int j = 0; j < (int)f.obs.size(); j++) {
            if (i == j) continue;
            const V3 t1 = camT(s + j);
            double R1[9], R20[9]; camR(s + j, R1);
            const V3 t20 = tmv(R0, sub(t1, t0));
            tmm(R0, R1, R20);
            const V3 pp = sub(tmv(R20, p0), tmv(R20, t20));
            const double rx = f.obs[j].v[0] - pp.x / pp.z, ry = f.obs[j].v[1] - pp.y / pp.z;
            if (std::sqrt(rx * rx + ry * ry) < 10.0 / 460) { const V3 pr = add(mv(R2r, p0), t2r); sum += pr.z; cnt++; }
          }
        }
        if (cnt == 0) continue;
        f.depth = sum / cnt;
        f.eflag = 1;
        if (f.depth < 0.1) { f.depth = t->opt.init_depth; f.eflag = 0; }
      }
    }
  }
  return GFBE_OK;
}

int32_t gfo_ftab_check_outliers(void *, gfo_ftab *t, const double *poses, const double *tic_ric, int32_t mode,
                                const int32_t *offset, int32_t *ids_out, int32_t *count_out) {
  for (size_t w = 0; w < t->tab.size(); w++) {
    const double *PR = poses + 132 * w, *tic_ = tic_ric + 12 * w, *ric = tic_ + 3;
    const V3 tic = {tic_[0], tic_[1], tic_[2]};
    std::set<int> rm;
    for (auto &f : t->tab[w]) {
      f.used = (int)f.obs.size();
      if (mode == 0) { if (f.used < 4) continue; }
      else { if (!(f.used >= 2 && f.start < GFBE_WINDOW_SIZE - 2)) continue; if (f.depth < 0) continue; }
      double err = 0, err3 = 0; int cnt = 0;
      const int i = f.start;
      int j = i - 1;
      const V3 uvi = pt(f.obs[0]);
      for (const Obs &o : f.obs) {
        j++;
        if (i == j) continue;
        const double *Pi = PR + 12 * i, *Pj = PR + 12 * j;
        err += reproj_err(Pi + 3, {Pi[0], Pi[1], Pi[2]}, ric, tic, Pj + 3, {Pj[0], Pj[1], Pj[2]}, f.depth, uvi, pt(o));
        if (mode == 1) err3 += reproj_err3d(Pi + 3, {Pi[0], Pi[1], Pi[2]}, ric, tic, Pj + 3, {Pj[0], Pj[1], Pj[2]}, f.depth, uvi, pt(o));
        cnt++;
      }
      if (mode == 0) { if (err / cnt * t->opt.focal_length > 3) rm.insert(f.id); }
      else if (cnt > 0 && (t->opt.focal_length * err / cnt > 10 || err3 / cnt > 2.0)) rm.insert(f.id);
    }
    int k = 0;
    for (int id : rm) { if (offset[w] + k < offset[w + 1]) ids_out[offset[w] + k] = id; k++; }
    count_out[w] = k;
  }
  return GFBE_OK;
}

int32_t gfo_ftab_size(void *, gfo_ftab *t, int32_t *n) { for (size_t w = 0; w < t->tab.size(); w++) n[w] = (int)t->tab[w].size(); return GFBE_OK; }

int32_t gfo_ftab_download(void *, gfo_ftab *t, int32_t w, int32_t *id, int32_t *start, int32_t *nobs, double *obs8, double *obs_td,
                          double *depth, int32_t *eflag, int32_t *sflag) {
  int k = 0;
  for (const Feat &f : t->tab[w]) {
    if (id) id[k] = f.id;
    if (start) start[k] = f.start;
    if (nobs) nobs[k] = (int)f.obs.size();
    if (depth) depth[k] = f.depth;
    if (eflag) eflag[k] = f.eflag;
    if (sflag) sflag[k] = f.sflag;
    if (obs8) { std::memset(obs8 + 88 * (size_t)k, 0, 88 * sizeof(double)); for (size_t o = 0; o < f.obs.size() && o < 11; o++) std::memcpy(obs8 + 88 * (size_t)k + 8 * o, f.obs[o].v, 64); }
    if (obs_td) { for (int o = 0; o < 11; o++) obs_td[11 * (size_t)k + o] = o < (int)f.obs.size() ? f.obs[o].td : 0.0; }
    k++;
  }
  return GFBE_OK;
}

void gfo_slide_window_state(gfbe_state *s, int32_t flag) {
  const int W = GFBE_WINDOW_SIZE;
  if (flag == GFBE_MARGIN_OLD) {
    for (int i = 0; i < W; i++) {   // Rs[i].swap(Rs[i+1]) ... then [W] = [W-1]
      std::memcpy(s->para_Pose[i], s->para_Pose[i + 1], sizeof s->para_Pose[0]);
      std::memcpy(s->para_SpeedBias[i], s->para_SpeedBias[i + 1], sizeof s->para_SpeedBias[0]);
    }
    std::memcpy(s->para_Pose[W], s->para_Pose[W - 1], sizeof s->para_Pose[0]);
    std::memcpy(s->para_SpeedBias[W], s->para_SpeedBias[W - 1], sizeof s->para_SpeedBias[0]);
  } else if (flag == GFBE_MARGIN_SECOND_NEW) {
    std::memcpy(s->para_Pose[W - 1], s->para_Pose[W], sizeof s->para_Pose[0]);
    std::memcpy(s->para_SpeedBias[W - 1], s->para_SpeedBias[W], sizeof s->para_SpeedBias[0]);
  }
}

}  // extern "C"

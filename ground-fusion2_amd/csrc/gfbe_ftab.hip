// gfbe_ftab.hip — device-resident feature tables: the FeatureManager / slideWindow operations either side of
// the solve, one workgroup per table, W tables per launch.
//
//   addFeatureCheckParallax   VE/estimator/feature_manager.cpp:57-116, compensatedParallax2 :978-1011
//   setDepth :249-267, removeFailures :269-278, clearDepth :280-284, getDepthVector :286-302
//   triangulate :669-724, triangulateWithDepth :726-799
//   removeOutlier :801-816, removeBackShiftDepth :818-856, removeBack :858-874, removeFront :914-934
//   Estimator::outliersRejection estimator.cpp:3971-4028, movingConsistencyCheckW :4030-4074
//
// A table is the reference's std::list<FeaturePerId> in insertion order, stored SoA with a fixed ELL of
// WINDOW_SIZE + 1 observation slots per feature. Erasing keeps the order: every thread owns a CONTIGUOUS chunk of
// the list, a block-wide exclusive scan of the per-chunk survivor counts gives each chunk its destination, and the
// survivors are copied (with their edits) into the other half of a ping-pong buffer. Purely integer / byte work
// plus tiny per-feature geometry: HBM- and latency-bound, no MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "gfbe_device.h"
#include "gfbe_math.h"

using namespace gfd;

namespace {

constexpr int NOBS = FT_NOBS;   // observation slots per feature
constexpr int OW = FT_OW;       // x y z u v vx vy depth
constexpr int FT_THREADS = 1024;

enum { OP_BACK_SHIFT = 0, OP_BACK = 1, OP_FRONT = 2, OP_OUTLIER = 3, OP_FAILURES = 4 };

__device__ __forceinline__ int block_exclusive_scan(int v, int *total, int *lds /* >= 17 ints */) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
  if (lane == 63) lds[wave] = x;
  __syncthreads();
  if (t == 0) { int run = 0; for (int q = 0; q < FT_THREADS / 64; q++) { const int c = lds[q]; lds[q] = run; run += c; } lds[16] = run; }
  __syncthreads();
  const int excl = lds[wave] + x - v;
  *total = lds[16];
  __syncthreads();
  return excl;
}

__device__ __forceinline__ vec3 mulR(const double *R, const vec3 &a) {   // row-major 3x3
  return mk3(R[0] * a[0] + R[1] * a[1] + R[2] * a[2], R[3] * a[0] + R[4] * a[1] + R[5] * a[2], R[6] * a[0] + R[7] * a[1] + R[8] * a[2]);
}
__device__ __forceinline__ vec3 mulRT(const double *R, const vec3 &a) {
  return mk3(R[0] * a[0] + R[3] * a[1] + R[6] * a[2], R[1] * a[0] + R[4] * a[1] + R[7] * a[2], R[2] * a[0] + R[5] * a[1] + R[8] * a[2]);
}
__device__ __forceinline__ void mm3(const double *A, const double *B, double *C) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0.0; for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j]; C[3 * i + j] = s; }
}
__device__ __forceinline__ void tmm3(const double *A, const double *B, double *C) {   // A^T B
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0.0; for (int k = 0; k < 3; k++) s += A[3 * k + i] * B[3 * k + j]; C[3 * i + j] = s; }
}

// ---- erasing operations: decide per feature, scan (k_ftab_erase: one workgroup per table), copy the survivors into the
//      other buffer (k_ftab_erase_copy: one thread per (feature, observation slot), any number of workgroups — a single
//      table keeps the whole GPU busy instead of one CU) -------------------------------------------------------------
__global__ __launch_bounds__(FT_THREADS) void k_ftab_erase(FtabDev T, int cur, int op, const double *argA, const double *argB,
                                                           const int *iarg, const int *ids_off, const int *ids) {
  const int w = blockIdx.x, t = threadIdx.x;
  __shared__ int lds[20];
  const int n = T.count[w];
  const size_t base = (size_t)w * T.F;
  const int *id = T.id[cur] + base, *start = T.start[cur] + base, *nobs = T.nobs[cur] + base;
  const int *sflag = T.sflag[cur] + base;
  const double *depth = T.depth[cur] + base, *obs = T.obs[cur] + base * NOBS * OW;
  int *keep = T.keep + base;
  double *nd = T.ndepth + base;
  const int chunk = (n + FT_THREADS - 1) / FT_THREADS, f0 = t * chunk, f1 = min(n, f0 + chunk);
  int survivors = 0;
  for (int f = f0; f < f1; f++) {
    // keep[f]: 0 erased; 1 kept, start unchanged; 2 kept, start - 1; 3 + j kept, observation j erased
    int k = 1;
    double dnew = depth[f];
    if (op == OP_BACK_SHIFT || op == OP_BACK) {
      if (start[f] != 0) k = 2;
      else {
        const int left = nobs[f] - 1;
        if (op == OP_BACK) k = left == 0 ? 0 : 3;
        else if (left < 2) k = 0;
        else {
          k = 3;
          const double *mP = argA + 12 * w, *nP = argB + 12 * w;
          const vec3 uv = ld3(obs + (size_t)f * NOBS * OW);
          const vec3 wp = add(mulR(mP + 3, scl(dnew, uv)), ld3(mP));
          const vec3 pj = mulRT(nP + 3, sub(wp, ld3(nP)));
          dnew = pj[2] > 0 ? pj[2] : T.opt.init_depth;
        }
      }
    } else if (op == OP_FRONT) {
      const int fc = iarg[w];
      if (start[f] == fc) k = 2;
      else if (start[f] + nobs[f] - 1 < fc - 1) k = 1;
      else k = (nobs[f] - 1 == 0) ? 0 : 3 + (GFBE_WINDOW_SIZE - 1 - start[f]);
    } else if (op == OP_OUTLIER) {
      for (int q = ids_off[w]; q < ids_off[w + 1]; q++) if (ids[q] == id[f]) { k = 0; break; }
    } else if (op == OP_FAILURES) {
      if (sflag[f] == 2) k = 0;
    }
    keep[f] = k; nd[f] = dnew;
    survivors += k != 0;
  }
  int total;
  int dst = block_exclusive_scan(survivors, &total, lds);
  int *dsti = T.ids_scratch + base;          // destination of feature f in the other buffer
  for (int f = f0; f < f1; f++) dsti[f] = keep[f] ? dst++ : -1;
  if (t == 0) { T.cnt_scratch[w] = n; T.count[w] = total; }
}
__global__ __launch_bounds__(256) void k_ftab_erase_copy(FtabDev T, int cur) {
  const int w = blockIdx.y;
  const int n = T.cnt_scratch[w];            // features before the operation
  const int g = blockIdx.x * 256 + threadIdx.x, f = g / NOBS, q = g - f * NOBS;
  if (f >= n) return;
  const size_t base = (size_t)w * T.F;
  const int k = T.keep[base + f], dst = T.ids_scratch[base + f];
  if (k == 0) return;
  const int o = 1 - cur;
  const int drop = k >= 3 ? k - 3 : -1, m = T.nobs[cur][base + f];
  if (q == 0) {
    T.id[o][base + dst] = T.id[cur][base + f]; T.eflag[o][base + dst] = T.eflag[cur][base + f]; T.sflag[o][base + dst] = T.sflag[cur][base + f];
    T.depth[o][base + dst] = T.ndepth[base + f];
    T.start[o][base + dst] = k == 2 ? T.start[cur][base + f] - 1 : T.start[cur][base + f];
    T.nobs[o][base + dst] = drop >= 0 ? m - 1 : m;
  }
  // output slot q <- source slot q (+ 1 behind the erased observation); slots past the track are zero
  const int sq = (drop >= 0 && q >= drop) ? q + 1 : q;
  const bool live = sq < m && sq < NOBS;
  const double *src = T.obs[cur] + ((base + f) * NOBS + (live ? sq : 0)) * OW;
  double *d8 = T.obs[o] + ((base + dst) * NOBS + q) * OW;
#pragma unroll
  for (int c = 0; c < OW; c++) d8[c] = live ? src[c] : 0.0;
  T.td[o][(base + dst) * NOBS + q] = live ? T.td[cur][(base + f) * NOBS + sq] : 0.0;
}

// ---- addFeatureCheckParallax ------------------------------------------------------------------------------
// find_if over the list for every incoming feature (feature_manager.cpp:70-74). Ids are unique inside a table (a feature is
// appended only when the search fails), so at most one list entry matches: workgroup (x, y) compares 64 incoming ids with a
// tile of FT_MATCH_TILE list ids staged in LDS and the one hit, if any, writes the entry (match[] starts at -1). One thread
// scanning the whole list (3 500 dependent global loads) took ~0.3 ms per frame of one robot.
#define FT_MATCH_TILE 256     // (1024: 53 us per call for a 3 500-feature list — a thread walked 1024 LDS entries; 256: four times the workgroups, a quarter of the walk)
__global__ __launch_bounds__(64) void k_ftab_match(FtabDev T, int cur, const int *offset, const int *fid, int *match) {
  const int w = blockIdx.z, j0 = offset[w], m = offset[w + 1] - j0, a = blockIdx.x * 64 + threadIdx.x;
  const int n = T.count[w], f0 = blockIdx.y * FT_MATCH_TILE, nt = min(n - f0, FT_MATCH_TILE);
  if ((int)blockIdx.x * 64 >= m || nt <= 0) return;       // (workgroup-uniform)
  __shared__ int s_id[FT_MATCH_TILE];
  const int *id = T.id[cur] + (size_t)w * T.F + f0;
  for (int f = threadIdx.x; f < nt; f += 64) s_id[f] = id[f];
  __syncthreads();
  if (a >= m) return;
  const int want = fid[j0 + a];
  int hit = -1;
  for (int f = nt - 1; f >= 0; f--) hit = s_id[f] == want ? f : hit;     // (the first entry of the tile that matches)
  if (hit >= 0) match[j0 + a] = f0 + hit;
}
__global__ __launch_bounds__(FT_THREADS) void k_ftab_add(FtabDev T, int cur, const int *frame_count, const int *offset, const int *fid,
                                                         const double *obs8, const double *tdv, int *match, int *keyframe,
                                                         int *counters, double *avg_parallax) {
  const int w = blockIdx.x, t = threadIdx.x;
  __shared__ int lds[20];
  __shared__ int s_cnt[3];
  __shared__ double s_red[FT_THREADS / 64];
  const int n = T.count[w], fc = frame_count[w];
  const size_t base = (size_t)w * T.F;
  int *id = T.id[cur] + base, *start = T.start[cur] + base, *nobs = T.nobs[cur] + base, *eflag = T.eflag[cur] + base, *sflag = T.sflag[cur] + base;
  double *depth = T.depth[cur] + base, *obs = T.obs[cur] + base * NOBS * OW, *td = T.td[cur] + base * NOBS;
  const int j0 = offset[w], m = offset[w + 1] - j0;
  if (t < 3) s_cnt[t] = 0;
  __syncthreads();
  // match[] = find_if over the list for every incoming feature (k_ftab_match)
  const int chunk = (m + FT_THREADS - 1) / FT_THREADS, a0 = t * chunk, a1 = min(m, a0 + chunk);
  int fresh = 0;
  for (int a = a0; a < a1; a++) fresh += match[j0 + a] < 0;
  int total_new;
  int dst = n + block_exclusive_scan(fresh, &total_new, lds);
  if (n + total_new > T.F) { if (t == 0) T.err[w] |= 1; total_new = 0; }
  int tracked = 0, longt = 0;
  for (int a = a0; a < a1; a++) {
    const int hit = match[j0 + a];
    const double *src = obs8 + (size_t)(j0 + a) * OW;
    if (hit >= 0) {
      const int k = nobs[hit];
      tracked++;
      if (k >= NOBS) { T.err[w] |= 2; continue; }
      for (int c = 0; c < OW; c++) obs[((size_t)hit * NOBS + k) * OW + c] = src[c];
      td[(size_t)hit * NOBS + k] = tdv[w];
      nobs[hit] = k + 1;
      if (k + 1 >= 4) longt++;
    } else if (total_new > 0) {
      id[dst] = fid[j0 + a]; start[dst] = fc; nobs[dst] = 1; eflag[dst] = 0; sflag[dst] = 0; depth[dst] = -1.0;
      for (int q = 0; q < NOBS; q++) { for (int c = 0; c < OW; c++) obs[((size_t)dst * NOBS + q) * OW + c] = q == 0 ? src[c] : 0.0; td[(size_t)dst * NOBS + q] = q == 0 ? tdv[w] : 0.0; }
      dst++;
    }
  }
  atomicAdd(&s_cnt[0], tracked); atomicAdd(&s_cnt[1], fresh); atomicAdd(&s_cnt[2], longt);
  __threadfence_block();
  __syncthreads();
  const int n2 = n + total_new, last_track = s_cnt[0], n_new = s_cnt[1], n_long = s_cnt[2];
  if (t == 0) { T.count[w] = n2; counters[3 * w] = last_track; counters[3 * w + 1] = n_new; counters[3 * w + 2] = n_long; }
  if (fc < 2 || last_track < 20 || n_long < 40 || n_new > 0.5 * last_track) {
    if (t == 0) { keyframe[w] = 1; avg_parallax[w] = 0.0; }
    return;
  }
  // compensatedParallax2 summed in chunk order (fixed association)
  double ps = 0.0; int pn = 0;
  const int ch2 = (n2 + FT_THREADS - 1) / FT_THREADS;
  for (int f = t * ch2; f < min(n2, (t + 1) * ch2); f++) {
    if (start[f] <= fc - 2 && start[f] + nobs[f] - 1 >= fc - 1) {
      const double *oi = obs + ((size_t)f * NOBS + (fc - 2 - start[f])) * OW, *oj = obs + ((size_t)f * NOBS + (fc - 1 - start[f])) * OW;
      const double ui = oi[0] / oi[2], vi = oi[1] / oi[2];
      const double du = ui - oj[0], dv = vi - oj[1];
      ps += sqrt(du * du + dv * dv);
      pn++;
    }
  }
  for (int o = 32; o > 0; o >>= 1) { ps += __shfl_down(ps, o, 64); pn += __shfl_down(pn, o, 64); }
  if ((t & 63) == 0) { s_red[t >> 6] = ps; lds[t >> 6] = pn; }
  __syncthreads();
  if (t == 0) {
    double sum = 0.0; int num = 0;
    for (int q = 0; q < FT_THREADS / 64; q++) { sum += s_red[q]; num += lds[q]; }
    if (num == 0) { keyframe[w] = 1; avg_parallax[w] = 0.0; }
    else { avg_parallax[w] = sum / num * T.opt.focal_length; keyframe[w] = (sum / num >= T.opt.min_parallax) ? 1 : 0; }
  }
}

// ---- in-place per-feature operations ----------------------------------------------------------------------
// mode 0: clearDepth; 1: setDepth(x); 2: getDepthVector -> x
__global__ __launch_bounds__(FT_THREADS) void k_ftab_depth(FtabDev T, int cur, int mode, const int *offset, double *x, int *count_out) {
  const int w = blockIdx.x, t = threadIdx.x;
  __shared__ int lds[20];
  const int n = T.count[w];
  const size_t base = (size_t)w * T.F;
  const int *nobs = T.nobs[cur] + base;
  double *depth = T.depth[cur] + base;
  int *sflag = T.sflag[cur] + base;
  const int chunk = (n + FT_THREADS - 1) / FT_THREADS, f0 = t * chunk, f1 = min(n, f0 + chunk);
  if (mode == 0) { for (int f = f0; f < f1; f++) depth[f] = -1.0; return; }
  int used = 0;
  for (int f = f0; f < f1; f++) used += nobs[f] >= 4;
  int total;
  int idx = block_exclusive_scan(used, &total, lds);
  for (int f = f0; f < f1; f++) {
    if (nobs[f] < 4) continue;
    if (mode == 1) { const double dd = 1.0 / x[offset[w] + idx]; depth[f] = dd; sflag[f] = dd < 0 ? 2 : 1; }
    else if (offset[w] + idx < offset[w + 1]) x[offset[w] + idx] = 1.0 / depth[f];
    idx++;
  }
  if (t == 0 && count_out) count_out[w] = total;
}

// smallest eigenvector of a symmetric 4 x 4 matrix (cyclic Jacobi), the smallest right singular vector of A
__device__ void smallest_eigvec4(double G[16], double v[4]) {
  double V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 30; sweep++) {
    double off = 0.0, dia = 0.0;
    for (int p = 0; p < 4; p++) for (int q = 0; q < 4; q++) { if (p == q) dia += G[4 * p + p] * G[4 * p + p]; else off += G[4 * p + q] * G[4 * p + q]; }
    if (off <= 1e-34 * dia) break;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 4; q++) {
        const double apq = G[4 * p + q];
        if (apq == 0.0) continue;
        const double theta = (G[4 * q + q] - G[4 * p + p]) / (2.0 * apq);
        const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(1.0 + theta * theta));
        const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
        for (int k = 0; k < 4; k++) { const double x = G[4 * k + p], y = G[4 * k + q]; G[4 * k + p] = c * x - s * y; G[4 * k + q] = s * x + c * y; }
        for (int k = 0; k < 4; k++) { const double x = G[4 * p + k], y = G[4 * q + k]; G[4 * p + k] = c * x - s * y; G[4 * q + k] = s * x + c * y; }
        for (int k = 0; k < 4; k++) { const double x = V[4 * k + p], y = V[4 * k + q]; V[4 * k + p] = c * x - s * y; V[4 * k + q] = s * x + c * y; }
      }
  }
  int best = 0;
  for (int j = 1; j < 4; j++) if (G[4 * j + j] < G[4 * best + best]) best = j;
  for (int i = 0; i < 4; i++) v[i] = V[4 * i + best];
}

__global__ __launch_bounds__(256) void k_ftab_triangulate(FtabDev T, int cur, const double *poses, const double *tic_ric, int with_depth) {
  const int w = blockIdx.y;
  const int n = T.count[w];
  const size_t base = (size_t)w * T.F;
  const double *PR = poses + 132 * (size_t)w, *tic = tic_ric + 12 * (size_t)w, *ric = tic + 3;
  // the camera pose of every window frame (R_f ric, P_f + R_f tic), once per workgroup: every (feature, observation) formed it again
  // from global memory — a dependent load and a 3 x 3 product per observation on the single thread that owns the feature (one robot:
  // 99 us per call). The same products of the same operands: the same depths.
  __shared__ double s_Rc[NOBS][9], s_tc[NOBS][3];
  if (threadIdx.x < NOBS) {
    const int fr = threadIdx.x;
    double Rc[9];
    mm3(PR + 12 * fr + 3, ric, Rc);
    const vec3 tc = add(ld3(PR + 12 * fr), mulR(PR + 12 * fr + 3, ld3(tic)));
    for (int q = 0; q < 9; q++) s_Rc[fr][q] = Rc[q];
    for (int q = 0; q < 3; q++) s_tc[fr][q] = tc[q];
  }
  __syncthreads();
  for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < n; f += gridDim.x * blockDim.x) {
    const int m = T.nobs[cur][base + f], s = T.start[cur][base + f];
    if (m < 4 || T.depth[cur][base + f] > 0) continue;
    const double *ob = T.obs[cur] + (base + f) * NOBS * OW;
    auto camT = [&](int fr) { return mk3(s_tc[fr][0], s_tc[fr][1], s_tc[fr][2]); };
    double R0[9];
    for (int q = 0; q < 9; q++) R0[q] = s_Rc[s][q];
    const vec3 t0 = camT(s);
    double dnew; int fl;
    if (!with_depth) {
      double G[16];
      for (int q = 0; q < 16; q++) G[q] = 0.0;
      for (int o = 0; o < m; o++) {
        double R1[9], R[9];
        for (int q = 0; q < 9; q++) R1[q] = s_Rc[s + o][q];
        const vec3 tt = mulRT(R0, sub(camT(s + o), t0));
        tmm3(R0, R1, R);
        const vec3 mt = mulRT(R, tt);
        double P[12];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) P[4 * r + c] = R[3 * c + r];
        P[3] = -mt[0]; P[7] = -mt[1]; P[11] = -mt[2];
        const double nn = sqrt(ob[o * OW] * ob[o * OW] + ob[o * OW + 1] * ob[o * OW + 1] + ob[o * OW + 2] * ob[o * OW + 2]);
        const double fx = ob[o * OW] / nn, fy = ob[o * OW + 1] / nn, fz = ob[o * OW + 2] / nn;
        double r0[4], r1[4];
        for (int c = 0; c < 4; c++) { r0[c] = fx * P[8 + c] - fz * P[c]; r1[c] = fy * P[8 + c] - fz * P[4 + c]; }
        for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) G[4 * a + b] += r0[a] * r0[b] + r1[a] * r1[b];
      }
      double v[4];
      smallest_eigvec4(G, v);
      dnew = v[2] / v[3]; fl = 2;
    } else {
      double sum = 0.0; int cnt = 0;
      for (int i = 0; i < m; i++) {
        const double dep = ob[i * OW + 7];
        if (dep < 0.1 || dep > T.opt.depth_threshold) continue;
        double Ri[9], R2r[9];
        for (int q = 0; q < 9; q++) Ri[q] = s_Rc[s + i][q];
        const vec3 ti = camT(s + i);
        const vec3 p0 = scl(dep, ld3(ob + i * OW));
        const vec3 t2r = mulRT(R0, sub(ti, t0));
        tmm3(R0, Ri, R2r);
        for (int j = 0; j < m; j++) {
          if (i == j) continue;
          double Rj[9], R20[9];
          for (int q = 0; q < 9; q++) Rj[q] = s_Rc[s + j][q];
          const vec3 t20 = mulRT(Ri, sub(camT(s + j), ti));
          tmm3(Ri, Rj, R20);
          const vec3 pp = sub(mulRT(R20, p0), mulRT(R20, t20));
          const double rx = ob[j * OW] - pp[0] / pp[2], ry = ob[j * OW + 1] - pp[1] / pp[2];
          if (sqrt(rx * rx + ry * ry) < 10.0 / 460) { sum += add(mulR(R2r, p0), t2r)[2]; cnt++; }
        }
      }
      if (cnt == 0) continue;
      dnew = sum / cnt; fl = 1;
    }
    if (dnew < 0.1) { dnew = T.opt.init_depth; fl = 0; }
    T.depth[cur][base + f] = dnew;
    T.eflag[cur][base + f] = fl;
  }
}

// outliersRejection (mode 0) / movingConsistencyCheckW (mode 1): flags in keep[], then the flagged ids in
// ascending order (the iteration order of the reference's std::set)
__global__ __launch_bounds__(FT_THREADS) void k_ftab_outliers(FtabDev T, int cur, const double *poses, const double *tic_ric, int mode,
                                                              int *ids_out, int *count_out) {
  const int w = blockIdx.x, t = threadIdx.x;
  __shared__ int lds[20];
  const int n = T.count[w];
  const size_t base = (size_t)w * T.F;
  const double *PR = poses + 132 * (size_t)w, *tic = tic_ric + 12 * (size_t)w, *ric = tic + 3;
  int *keep = T.keep + base;
  const int *id = T.id[cur] + base;
  const int chunk = (n + FT_THREADS - 1) / FT_THREADS, f0 = t * chunk, f1 = min(n, f0 + chunk);
  int mine = 0;
  for (int f = f0; f < f1; f++) {
    const int m = T.nobs[cur][base + f], s = T.start[cur][base + f];
    const double dep = T.depth[cur][base + f];
    int bad = 0;
    const bool consider = mode == 0 ? (m >= 4) : (m >= 2 && s < GFBE_WINDOW_SIZE - 2 && !(dep < 0));
    if (consider) {
      const double *ob = T.obs[cur] + (base + f) * NOBS * OW;
      const vec3 uvi = ld3(ob);
      const vec3 pw = add(mulR(PR + 12 * s + 3, add(mulR(ric, scl(dep, uvi)), ld3(tic))), ld3(PR + 12 * s));
      double err = 0.0, err3 = 0.0; int cnt = 0;
      for (int o = 1; o < m; o++) {
        const int j = s + o;
        const vec3 pc = mulRT(ric, sub(mulRT(PR + 12 * j + 3, sub(pw, ld3(PR + 12 * j))), ld3(tic)));
        const double rx = pc[0] / pc[2] - ob[o * OW], ry = pc[1] / pc[2] - ob[o * OW + 1];
        err += sqrt(rx * rx + ry * ry);
        if (mode == 1) { const vec3 dd = sub(pc, ld3(ob + o * OW)); err3 += sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]) / dep; }
        cnt++;
      }
      if (mode == 0) bad = (err / cnt * T.opt.focal_length > 3) ? 1 : 0;
      else bad = (cnt > 0 && (T.opt.focal_length * err / cnt > 10 || err3 / cnt > 2.0)) ? 1 : 0;
    }
    keep[f] = bad;
    mine += bad;
  }
  int total;
  int dst = block_exclusive_scan(mine, &total, lds);
  // the flagged ids, compacted into LDS; the rank of an id among them is its position in the ascending output
  extern __shared__ int flagged[];
  for (int f = f0; f < f1; f++) if (keep[f]) flagged[dst++] = id[f];
  __syncthreads();
  for (int f = f0; f < f1; f++) {
    if (!keep[f]) continue;
    const int me = id[f];
    int rank = 0;
    for (int g = 0; g < total; g++) rank += flagged[g] < me;
    ids_out[base + rank] = me;
  }
  if (t == 0) count_out[w] = total;
}

// ---- landmarks of a table -> the solver's landmark arrays (gfbe_batch_upload_tables) ------------------------
// Landmarks = features with >= 4 observations, in list order (getFeatureCount / the factor loop of
// estimator.cpp:3330-3358). Bin of a landmark: (start frame s, m = observations - 1 in 3..10) -> s * 8 + (m - 3).
__global__ __launch_bounds__(FT_THREADS) void k_ftab_count(FtabDev T, int cur, int w0, int *counts) {
  const int w = w0 + blockIdx.x, t = threadIdx.x;
  __shared__ int hist[FT_BINS + 2];
  for (int q = t; q < FT_BINS + 2; q += FT_THREADS) hist[q] = 0;
  __syncthreads();
  const int n = T.count[w];
  const size_t base = (size_t)w * T.F;
  for (int f = t; f < n; f += FT_THREADS) {
    const int m = T.nobs[cur][base + f] - 1, s = T.start[cur][base + f];
    if (m < 3) continue;
    atomicAdd(&hist[0], 1);
    atomicAdd(&hist[1], m);
    atomicAdd(&hist[2 + s * 8 + (m - 3)], 1);
  }
  __syncthreads();
  for (int q = t; q < FT_BINS + 2; q += FT_THREADS) counts[(size_t)blockIdx.x * (FT_BINS + 2) + q] = hist[q];
}

// layout per window (ints): [0] lm_off, [1 .. 1+FT_BINS) first slot of every bin relative to lm_off (groups sorted by start
// frame, longer tracks first), [.. + NF) first slot of every start-frame group, [.. + NPAIR+1) pair_begin.
enum { LAY_BIN = FT_LAY_BIN, LAY_GRP = FT_LAY_GRP, LAY_PAIR = FT_LAY_PAIR, LAY_STRIDE = FT_LAY_STRIDE };
__global__ __launch_bounds__(FT_THREADS) void k_ftab_landmarks(FtabDev T, int cur, int w0, BatchDev d, const int *layout, int *slot_of) {
  const int wl = blockIdx.x, w = w0 + wl, t = threadIdx.x;
  __shared__ int lds[20];
  const int *lay = layout + (size_t)wl * LAY_STRIDE;
  const int n = T.count[w];
  const size_t base = (size_t)w * T.F;
  const int *start = T.start[cur] + base, *nobs = T.nobs[cur] + base, *eflag = T.eflag[cur] + base;
  const double *depth = T.depth[cur] + base;
  int *keep = T.keep + base;       // scratch: slot of feature f (or -1)
  const int chunk = (n + FT_THREADS - 1) / FT_THREADS, f0 = t * chunk, f1 = min(n, f0 + chunk);
  // landmark index in list order
  int mine = 0;
  for (int f = f0; f < f1; f++) mine += nobs[f] >= 4;
  int total;
  int lidx = block_exclusive_scan(mine, &total, lds);
  // rank inside its bin in list order: RANKERS threads take contiguous pieces of the list and count their landmarks per bin
  // in LDS (bin-major), thread b scans bin b over the rankers, then every ranker numbers its own landmarks — three barriers
  // instead of one block scan per bin
  enum { RANKERS = 256 };
  __shared__ unsigned short hb[FT_BINS][RANKERS + 2];     // (+2: rows one bank apart)
  for (int q = t; q < FT_BINS * (RANKERS + 2); q += FT_THREADS) (&hb[0][0])[q] = 0;
  __syncthreads();
  const int rchunk = (n + RANKERS - 1) / RANKERS, r0 = t * rchunk, r1 = min(n, r0 + rchunk);
  if (t < RANKERS)
    for (int f = r0; f < r1; f++) {
      keep[f] = -1;
      if (nobs[f] >= 4) hb[start[f] * 8 + (nobs[f] - 4)][t]++;
    }
  __syncthreads();
  if (t < FT_BINS) {
    int run = 0;
    for (int k = 0; k < RANKERS; k++) { const int cth = hb[t][k]; hb[t][k] = (unsigned short)run; run += cth; }
  }
  __syncthreads();
  if (t < RANKERS)
    for (int f = r0; f < r1; f++)
      if (nobs[f] >= 4) { const int bin = start[f] * 8 + (nobs[f] - 4); keep[f] = lay[LAY_BIN + bin] + hb[bin][t]++; }
  __threadfence_block();
  __syncthreads();
  const int lm_off = lay[0];
  for (int f = f0; f < f1; f++) {      // per-landmark scalars and the list-order -> slot map (the bulk goes to k_ftab_landmarks_write)
    if (nobs[f] < 4) continue;
    const int s = start[f], m = nobs[f] - 1, slot = lm_off + keep[f];
    d.lm_info[slot] = s | (m << 8) | ((eflag[f] == 1 ? 1 : 0) << 16) | (1 << 24);
    d.lm_abi[slot] = lidx;
    d.lam0[slot] = 1.0 / depth[f];
    slot_of[(size_t)wl * T.F + lidx] = slot;
    lidx++;
  }
}
// observations and record indices of every landmark: one thread per (feature, observation slot), any number of workgroups
__global__ __launch_bounds__(256) void k_ftab_landmarks_write(FtabDev T, int cur, int w0, BatchDev d, const int *layout) {
  const int wl = blockIdx.y, w = w0 + wl;
  const int g = blockIdx.x * 256 + threadIdx.x, f = g / NOBS, q = g - f * NOBS;
  if (f >= T.count[w]) return;
  const size_t base = (size_t)w * T.F, TL = d.tot_lm;
  const int n_o = T.nobs[cur][base + f];
  if (n_o < 4 || q >= n_o) return;
  const int *lay = layout + (size_t)wl * LAY_STRIDE;
  const int s = T.start[cur][base + f], rel = T.keep[base + f], slot = lay[0] + rel;
  const double *o = T.obs[cur] + ((base + f) * NOBS + q) * OW;
  double tdq = T.td[cur][(base + f) * NOBS + q];
  // a batch whose windows all hold td constant stores the observation already shifted to the window's td (what every factor evaluation
  // would compute, projectionTwoFrameOneCamFactor.cpp:60-61) and td in place of the observation's own: see expand_body (gfbe_kernels.hip)
  double ox = o[0], oy = o[1];
  if (!d.vis_full) {
    const double tdw = d.x0[(size_t)wl * NA + A_TD], dt = tdw - tdq;
    ox = __builtin_fma(-dt, o[5], ox); oy = __builtin_fma(-dt, o[6], oy); tdq = tdw;
  }
  if (q == 0) {
    d.lm_pts[0 * TL + slot] = ox; d.lm_pts[1 * TL + slot] = oy; d.lm_pts[2 * TL + slot] = o[2];
    d.lm_pts[3 * TL + slot] = o[5]; d.lm_pts[4 * TL + slot] = o[6]; d.lm_pts[5 * TL + slot] = tdq;
  } else {
    const int k = q - 1;
    double *ob = d.lm_obs + (size_t)k * 5 * TL + slot;
    ob[0] = ox; ob[TL] = oy; ob[2 * TL] = o[5]; ob[3 * TL] = o[6]; ob[4 * TL] = tdq;
    d.lm_rec[(size_t)k * TL + slot] = lay[LAY_PAIR + s * NF + s + 1 + k] + (rel - lay[LAY_GRP + s]);   // slot order inside a pair = group order
  }
}
__global__ void k_fill(double *p, size_t n, double v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// rows [W][F] with cnt[w] used entries -> one contiguous list (so that the host copies exactly sum(cnt) values)
__global__ __launch_bounds__(FT_THREADS) void k_ftab_pack(int W, int F, const int *cnt, const int *rows, int *packed) {
  __shared__ int s_off[2];
  for (int w = 0, run = 0; w < W; w++) {
    if (threadIdx.x == 0) { s_off[0] = run; s_off[1] = cnt[w]; }
    __syncthreads();
    const int o = s_off[0], n = s_off[1];
    for (int k = threadIdx.x; k < n; k += FT_THREADS) packed[o + k] = rows[(size_t)w * F + k];
    run = o + n;
    __syncthreads();
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
namespace {
#define FT_CHECK(c, call)                                                                                      \
  do {                                                                                                         \
    hipError_t e_ = (call);                                                                                    \
    if (e_ != hipSuccess) { ctx_set_error(c, (std::string(#call) + ": " + hipGetErrorString(e_)).c_str()); return GFBE_DEVICE_ERROR; } \
  } while (0)

template <typename T>
gfbe_status ft_alloc(gfbe_ctx *c, gfbe_ftab *t, T **p, size_t n) {
  void *q = nullptr;
  FT_CHECK(c, hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
  FT_CHECK(c, hipMemsetAsync(q, 0, std::max<size_t>(n, 1) * sizeof(T), ctx_stream(c)));
  t->allocs.push_back(q);
  *p = (T *)q;
  return GFBE_OK;
}
// Host arguments of one table operation -> the table's staging chunk (one pinned host mirror, ONE host-to-device copy before
// the launch, ONE device-to-host copy of the output range after it, one wait). `need` = upper bound of the staged bytes.
// `defer`: an operation without outputs — its arguments go through a slot of the table's ring and nobody waits; every
// operation runs on the context's stream, so whoever reads a result later (add_frame, check_outliers, size, the solver's
// hand-over) sees the tables after it.
struct Staged {
  gfbe_ctx *c;
  gfbe_ftab *t;
  char *bh = nullptr, *bd = nullptr;
  size_t cap = 0, off = 0, ulo = SIZE_MAX, uhi = 0, dlo = SIZE_MAX, dhi = 0;
  int slot = -1;
  bool ok = true;
  struct Out { void *h; size_t off, bytes; };
  std::vector<Out> outs;
  Staged(gfbe_ctx *ctx, gfbe_ftab *tab, size_t need, bool defer = false) : c(ctx), t(tab) {
    need += 4096;
    if (defer && t->ring_d && need <= (size_t)gfbe_ftab::RING_SLOT) {
      slot = t->ring_next;
      t->ring_next = (slot + 1) % gfbe_ftab::RING;
      if (t->ring_used[slot]) (void)hipEventSynchronize(t->ring_ev[slot]);
      bh = t->ring_h + (size_t)slot * gfbe_ftab::RING_SLOT; bd = t->ring_d + (size_t)slot * gfbe_ftab::RING_SLOT; cap = gfbe_ftab::RING_SLOT;
      return;
    }
    if (need > t->stage_cap) {
      (void)hipStreamSynchronize(ctx_stream(c));
      if (t->stage_d) (void)hipFree(t->stage_d);
      if (t->stage_h) (void)hipHostFree(t->stage_h);
      t->stage_d = t->stage_h = nullptr; t->stage_cap = 0;
      const size_t ncap = std::max<size_t>(2 * need, (size_t)1 << 20);
      if (hipMalloc((void **)&t->stage_d, ncap) != hipSuccess || hipHostMalloc((void **)&t->stage_h, ncap) != hipSuccess) { ok = false; return; }
      t->stage_cap = ncap;
    }
    bh = t->stage_h; bd = t->stage_d; cap = t->stage_cap;
  }
  ~Staged() { finish(); }
  template <typename T>
  T *up(const T *h, size_t n) {
    const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
    if (!ok || off + bytes > cap) { ok = false; return nullptr; }
    if (h && n) { std::memcpy(bh + off, h, n * sizeof(T)); ulo = std::min(ulo, off); uhi = std::max(uhi, off + n * sizeof(T)); }
    T *p = (T *)(bd + off);
    off += bytes;
    return p;
  }
  void flush() {   // before the launch
    if (ok && uhi > ulo) (void)hipMemcpyAsync(bd + ulo, bh + ulo, uhi - ulo, hipMemcpyHostToDevice, ctx_stream(c));
  }
  template <typename T>
  void down(T *h, const T *dptr, size_t n) {     // (operations with outputs are never deferred)
    if (!h || !n || !ok || slot >= 0) return;
    const char *p = (const char *)dptr;
    if (p >= bd && p < bd + cap) {
      const size_t o = (size_t)(p - bd);
      outs.push_back({h, o, n * sizeof(T)});
      dlo = std::min(dlo, o); dhi = std::max(dhi, o + n * sizeof(T));
    } else {
      (void)hipMemcpyAsync(h, dptr, n * sizeof(T), hipMemcpyDeviceToHost, ctx_stream(c));
    }
  }
  void finish() {
    if (slot >= 0) {     // deferred: mark the slot busy until the stream has passed this point
      (void)hipEventRecord(t->ring_ev[slot], ctx_stream(c));
      t->ring_used[slot] = true;
      slot = -2;
      return;
    }
    if (slot == -2 || !bh) return;
    if (dhi > dlo) (void)hipMemcpyAsync(bh + dlo, bd + dlo, dhi - dlo, hipMemcpyDeviceToHost, ctx_stream(c));
    (void)hipStreamSynchronize(ctx_stream(c));
    for (const Out &o : outs) std::memcpy(o.h, bh + o.off, o.bytes);
    outs.clear(); dlo = SIZE_MAX; dhi = 0;
  }
};
gfbe_status ft_ready(gfbe_ctx *c, gfbe_ftab *t) {
  if (!c || !t) return GFBE_BAD_INPUT;
  if (ctx_device(c) < 0) return GFBE_NO_DEVICE;
  // (a batch upload may still be reading the tables on the copy stream: this operation runs behind it)
  if (t->read_pending && t->ev_read) { FT_CHECK(c, hipStreamWaitEvent(ctx_stream(c), t->ev_read, 0)); t->read_pending = false; }
  return GFBE_OK;
}
// launch errors of the operation (the sticky table errors — capacity, observation count — are raised by k_ftab_add alone and
// come back with gfbe_ftab_add_frame's own results)
gfbe_status ft_finish(gfbe_ctx *c, gfbe_ftab *t) {
  FT_CHECK(c, hipGetLastError());
  if (t && t->ev_ops) FT_CHECK(c, hipEventRecord(t->ev_ops, ctx_stream(c)));      // (what gfbe_batch_upload_tables waits for)
  return GFBE_OK;
}
}  // namespace

namespace gfd {
void launch_ftab_count(const FtabDev &T, int cur, int w0, int n, int *counts, hipStream_t s) {
  hipLaunchKernelGGL(k_ftab_count, dim3(n), dim3(FT_THREADS), 0, s, T, cur, w0, counts);
}
void launch_ftab_pack(const FtabDev &T, int cur, int w0, int n, const BatchDev &d, const int *layout, int *slot_of, hipStream_t s) {
  if (d.tot_lm > 0) hipLaunchKernelGGL(k_fill, dim3((unsigned)((d.tot_lm + 255) / 256)), dim3(256), 0, s, d.lam0, (size_t)d.tot_lm, 1.0);
  (void)hipMemsetAsync(d.lm_abi, 0xFF, sizeof(int) * (size_t)d.tot_lm, s);     // padding slots: -1
  hipLaunchKernelGGL(k_ftab_landmarks, dim3(n), dim3(FT_THREADS), 0, s, T, cur, w0, d, layout, slot_of);
  hipLaunchKernelGGL(k_ftab_landmarks_write, dim3((unsigned)(((size_t)T.F * NOBS + 255) / 256), n), dim3(256), 0, s, T, cur, w0, d, layout);
}
}  // namespace gfd

extern "C" {

void gfbe_ftab_default_options(gfbe_ftab_options *o) {
  o->init_depth = 5.0;               // parameters.cpp:484
  o->focal_length = 600.0;           // parameters.h:23
  o->min_parallax = 10.0 / 600.0;    // parameters.cpp:351-352 with keyframe_parallax: 10.0 (m3dgr.yaml:110)
  o->depth_threshold = 3.0;          // m3dgr.yaml:17
}

gfbe_status gfbe_ftab_create(gfbe_ctx *c, int32_t n_tables, int32_t cap, const gfbe_ftab_options *opt, gfbe_ftab **out) {
  if (!c || !out || n_tables < 1 || cap < 1 || cap > 16384) return GFBE_BAD_INPUT;   // 16384: the LDS id list of check_outliers (64 KB)
  if (ctx_device(c) < 0) return GFBE_NO_DEVICE;
  gfbe_ftab *t = new gfbe_ftab();
  *out = nullptr;
  // (a failed allocation leaves nothing behind: the table built so far is destroyed and *out stays null)
  struct Guard { gfbe_ctx *c; gfbe_ftab *t; bool armed = true; ~Guard() { if (armed) gfbe_ftab_destroy(c, t); } } guard{c, t};
  FtabDev &d = t->d;
  d.W = n_tables; d.F = cap;
  if (opt) d.opt = *opt; else gfbe_ftab_default_options(&d.opt);
  const size_t N = (size_t)n_tables * cap;
  gfbe_status st;
#define FA(p, n) if ((st = ft_alloc(c, t, &p, n)) != GFBE_OK) return st
  FA(d.count, n_tables); FA(d.err, n_tables); FA(d.keep, N); FA(d.ndepth, N); FA(d.ids_scratch, N); FA(d.cnt_scratch, n_tables);
  FA(d.hist, (size_t)n_tables * (FT_BINS + 2)); FA(d.layout, (size_t)n_tables * FT_LAY_STRIDE);
  for (int b = 0; b < 2; b++) {
    FA(d.id[b], N); FA(d.start[b], N); FA(d.nobs[b], N); FA(d.eflag[b], N); FA(d.sflag[b], N);
    FA(d.depth[b], N); FA(d.obs[b], N * NOBS * OW); FA(d.td[b], N * NOBS);
  }
#undef FA
  { Staged warm(c, t, (size_t)n_tables * cap * 8); }   // staging sized for the largest result (every id flagged) up front
  FT_CHECK(c, hipMalloc((void **)&t->ring_d, (size_t)gfbe_ftab::RING * gfbe_ftab::RING_SLOT));
  FT_CHECK(c, hipHostMalloc((void **)&t->ring_h, (size_t)gfbe_ftab::RING * gfbe_ftab::RING_SLOT));
  for (int k = 0; k < gfbe_ftab::RING; k++) FT_CHECK(c, hipEventCreateWithFlags(&t->ring_ev[k], hipEventDisableTiming));
  FT_CHECK(c, hipEventCreateWithFlags(&t->ev_ops, hipEventDisableTiming));
  FT_CHECK(c, hipEventCreateWithFlags(&t->ev_read, hipEventDisableTiming));
  FT_CHECK(c, hipEventRecord(t->ev_ops, ctx_stream(c)));
  FT_CHECK(c, hipStreamSynchronize(ctx_stream(c)));
  guard.armed = false;
  *out = t;
  return GFBE_OK;
}

void gfbe_ftab_destroy(gfbe_ctx *c, gfbe_ftab *t) {
  if (!t) return;
  if (c && ctx_device(c) >= 0) (void)hipStreamSynchronize(ctx_stream(c));
  for (void *p : t->allocs) (void)hipFree(p);
  if (t->stage_d) (void)hipFree(t->stage_d);
  if (t->stage_h) (void)hipHostFree(t->stage_h);
  if (t->ring_d) (void)hipFree(t->ring_d);
  if (t->ring_h) (void)hipHostFree(t->ring_h);
  for (hipEvent_t e : t->ring_ev) if (e) (void)hipEventDestroy(e);
  if (t->ev_read) { (void)hipEventSynchronize(t->ev_read); (void)hipEventDestroy(t->ev_read); }
  if (t->ev_ops) (void)hipEventDestroy(t->ev_ops);
  delete t;
}

gfbe_status gfbe_ftab_add_frame(gfbe_ctx *c, gfbe_ftab *t, const int32_t *frame_count, const int32_t *offset, const int32_t *feature_id,
                                const double *obs8, const double *td, int32_t *keyframe, int32_t *counters, double *avg_parallax) {
  gfbe_status st = ft_ready(c, t);
  if (st != GFBE_OK) return st;
  if (!frame_count || !offset || !td) return GFBE_BAD_INPUT;
  const int W = t->d.W, M = offset[W];
  for (int w = 0; w < W; w++)
    for (int k = offset[w] + 1; k < offset[w + 1]; k++)
      if (feature_id[k] <= feature_id[k - 1]) { ctx_set_error(c, "gfbe_ftab_add_frame: feature ids of a table must be strictly ascending"); return GFBE_BAD_INPUT; }
  std::vector<int> err(W, 0);
  {
    Staged s(c, t, (size_t)M * (OW * 8 + 8) + (size_t)W * 64 + 16 * 256);
    int *dfc = s.up(frame_count, W), *doff = s.up(offset, W + 1), *dfid = s.up(feature_id, M);
    double *dobs = s.up(obs8, (size_t)M * OW), *dtd = s.up(td, W);
    int *dmatch = s.up<int>(nullptr, M);
    int *dkf = s.up<int>(nullptr, W), *dcnt = s.up<int>(nullptr, 3 * W);
    double *davg = s.up<double>(nullptr, W);
    int *derr = s.up<int>(nullptr, W);
    if (!s.ok) { ctx_set_error(c, "gfbe_ftab_add_frame: staging allocation failed"); return GFBE_DEVICE_ERROR; }
    s.flush();
    int mmax = 1;
    for (int w = 0; w < W; w++) mmax = std::max(mmax, offset[w + 1] - offset[w]);
    (void)hipMemsetAsync(dmatch, 0xFF, sizeof(int) * (size_t)std::max(M, 1), ctx_stream(c));     // -1: not in the list
    hipLaunchKernelGGL(k_ftab_match, dim3((mmax + 63) / 64, (t->d.F + FT_MATCH_TILE - 1) / FT_MATCH_TILE, W), dim3(64), 0, ctx_stream(c), t->d, t->cur, doff, dfid, dmatch);
    hipLaunchKernelGGL(k_ftab_add, dim3(W), dim3(FT_THREADS), 0, ctx_stream(c), t->d, t->cur, dfc, doff, dfid, dobs, dtd, dmatch, dkf, dcnt, davg);
    // the tables' sticky error flags (only this operation raises them) travel back with the results: one copy, one wait
    (void)hipMemcpyAsync(derr, t->d.err, sizeof(int) * W, hipMemcpyDeviceToDevice, ctx_stream(c));
    s.down(keyframe, dkf, W); s.down(counters, dcnt, 3 * (size_t)W); s.down(avg_parallax, davg, W); s.down(err.data(), derr, W);
  }
  for (int w = 0; w < W; w++)
    if (err[w]) { ctx_set_error(c, err[w] & 1 ? "feature table capacity exceeded" : "a feature received more than WINDOW_SIZE + 1 observations"); return GFBE_BAD_INPUT; }
  return ft_finish(c, t);
}

static gfbe_status ft_erase(gfbe_ctx *c, gfbe_ftab *t, int op, const double *a, const double *b, const int32_t *iarg, const int32_t *off, const int32_t *ids) {
  gfbe_status st = ft_ready(c, t);
  if (st != GFBE_OK) return st;
  const int W = t->d.W;
  {
    Staged s(c, t, (size_t)W * 256 + (off ? (size_t)off[W] * 4 : 0) + 8 * 256, /*defer=*/true);
    double *da = a ? s.up(a, 12 * (size_t)W) : nullptr, *db = b ? s.up(b, 12 * (size_t)W) : nullptr;
    int *di = iarg ? s.up(iarg, W) : nullptr, *doff = off ? s.up(off, W + 1) : nullptr, *dids = off ? s.up(ids, off[W]) : nullptr;
    if (!s.ok) { ctx_set_error(c, "feature table operation: staging allocation failed"); return GFBE_DEVICE_ERROR; }
    s.flush();
    hipLaunchKernelGGL(k_ftab_erase, dim3(W), dim3(FT_THREADS), 0, ctx_stream(c), t->d, t->cur, op, da, db, di, doff, dids);
    hipLaunchKernelGGL(k_ftab_erase_copy, dim3((unsigned)(((size_t)t->d.F * NOBS + 255) / 256), W), dim3(256), 0, ctx_stream(c), t->d, t->cur);
  }
  t->cur = 1 - t->cur;
  return ft_finish(c, t);
}
gfbe_status gfbe_ftab_remove_back_shift_depth(gfbe_ctx *c, gfbe_ftab *t, const double *marg_PR, const double *new_PR) {
  if (!marg_PR || !new_PR) return GFBE_BAD_INPUT;
  return ft_erase(c, t, OP_BACK_SHIFT, marg_PR, new_PR, nullptr, nullptr, nullptr);
}
gfbe_status gfbe_ftab_remove_back(gfbe_ctx *c, gfbe_ftab *t) { return ft_erase(c, t, OP_BACK, nullptr, nullptr, nullptr, nullptr, nullptr); }
gfbe_status gfbe_ftab_remove_front(gfbe_ctx *c, gfbe_ftab *t, const int32_t *frame_count) {
  if (!frame_count) return GFBE_BAD_INPUT;
  return ft_erase(c, t, OP_FRONT, nullptr, nullptr, frame_count, nullptr, nullptr);
}
gfbe_status gfbe_ftab_remove_outlier(gfbe_ctx *c, gfbe_ftab *t, const int32_t *offset, const int32_t *ids) {
  if (!offset) return GFBE_BAD_INPUT;
  return ft_erase(c, t, OP_OUTLIER, nullptr, nullptr, nullptr, offset, ids);
}
gfbe_status gfbe_ftab_remove_failures(gfbe_ctx *c, gfbe_ftab *t) { return ft_erase(c, t, OP_FAILURES, nullptr, nullptr, nullptr, nullptr, nullptr); }

static gfbe_status ft_depth(gfbe_ctx *c, gfbe_ftab *t, int mode, const int32_t *offset, double *x_io, int32_t *count) {
  gfbe_status st = ft_ready(c, t);
  if (st != GFBE_OK) return st;
  const int W = t->d.W;
  if (mode != 0 && (!offset || !x_io)) return GFBE_BAD_INPUT;
  {
    Staged s(c, t, (size_t)W * 16 + (offset ? (size_t)offset[W] * 8 : 0) + 8 * 256, /*defer=*/mode != 2 && !count);
    int *doff = offset ? s.up(offset, W + 1) : nullptr;
    double *dx = offset ? s.up(mode == 1 ? x_io : nullptr, offset[W]) : nullptr;
    int *dcnt = s.up<int>(nullptr, W);
    if (!s.ok) { ctx_set_error(c, "feature table operation: staging allocation failed"); return GFBE_DEVICE_ERROR; }
    s.flush();
    hipLaunchKernelGGL(k_ftab_depth, dim3(W), dim3(FT_THREADS), 0, ctx_stream(c), t->d, t->cur, mode, doff, dx, dcnt);
    if (mode == 2) s.down(x_io, dx, offset[W]);
    s.down(count, dcnt, W);
  }
  return ft_finish(c, t);
}
gfbe_status gfbe_ftab_clear_depth(gfbe_ctx *c, gfbe_ftab *t) { return ft_depth(c, t, 0, nullptr, nullptr, nullptr); }
gfbe_status gfbe_ftab_set_depth(gfbe_ctx *c, gfbe_ftab *t, const int32_t *offset, const double *x) { return ft_depth(c, t, 1, offset, const_cast<double *>(x), nullptr); }
gfbe_status gfbe_ftab_get_depth_vector(gfbe_ctx *c, gfbe_ftab *t, const int32_t *offset, double *x, int32_t *count) { return ft_depth(c, t, 2, offset, x, count); }

gfbe_status gfbe_ftab_triangulate(gfbe_ctx *c, gfbe_ftab *t, const double *poses, const double *tic_ric, int32_t with_depth) {
  gfbe_status st = ft_ready(c, t);
  if (st != GFBE_OK) return st;
  if (!poses || !tic_ric) return GFBE_BAD_INPUT;
  const int W = t->d.W;
  {
    Staged s(c, t, (size_t)W * 144 * 8 + 8 * 256, /*defer=*/true);
    double *dp = s.up(poses, 132 * (size_t)W), *de = s.up(tic_ric, 12 * (size_t)W);
    if (!s.ok) { ctx_set_error(c, "feature table operation: staging allocation failed"); return GFBE_DEVICE_ERROR; }
    s.flush();
    hipLaunchKernelGGL(k_ftab_triangulate, dim3((t->d.F + 255) / 256, W), dim3(256), 0, ctx_stream(c), t->d, t->cur, dp, de, with_depth);
  }
  return ft_finish(c, t);
}

gfbe_status gfbe_ftab_check_outliers(gfbe_ctx *c, gfbe_ftab *t, const double *poses, const double *tic_ric, int32_t mode,
                                     const int32_t *offset, int32_t *ids_out, int32_t *count_out) {
  gfbe_status st = ft_ready(c, t);
  if (st != GFBE_OK) return st;
  if (!poses || !tic_ric || !offset || !ids_out || !count_out) return GFBE_BAD_INPUT;
  const int W = t->d.W;
  const size_t nspec = std::min<size_t>(1024, (size_t)W * t->d.F);
  std::vector<int32_t> spec(nspec);
  {
    Staged s(c, t, (size_t)W * 144 * 8 + (size_t)W * 4 + nspec * 4 + 8 * 256);
    double *dp = s.up(poses, 132 * (size_t)W), *de = s.up(tic_ric, 12 * (size_t)W);
    if (!s.ok) { ctx_set_error(c, "feature table operation: staging allocation failed"); return GFBE_DEVICE_ERROR; }
    int *dcnt = s.up<int>(nullptr, W);      // counts come back through the pinned mirror (a pageable 1 KB copy took 21 ms here)
    int *dspec = nspec ? s.up<int>(nullptr, nspec) : nullptr;
    if (!s.ok) { ctx_set_error(c, "feature table operation: staging allocation failed"); return GFBE_DEVICE_ERROR; }
    s.flush();
    hipLaunchKernelGGL(k_ftab_outliers, dim3(W), dim3(FT_THREADS), sizeof(int) * (size_t)t->d.F, ctx_stream(c), t->d, t->cur, dp, de, mode,
                       t->d.ids_scratch, dcnt);
    hipLaunchKernelGGL(k_ftab_pack, dim3(1), dim3(FT_THREADS), 0, ctx_stream(c), W, t->d.F, dcnt, t->d.ids_scratch, t->d.keep);
    s.down(count_out, dcnt, W);
    // the first ids travel with the counts (one wait instead of two when few features are flagged — the usual case)
    if (dspec) { (void)hipMemcpyAsync(dspec, t->d.keep, sizeof(int32_t) * nspec, hipMemcpyDeviceToDevice, ctx_stream(c)); s.down(spec.data(), dspec, nspec); }
  }
  size_t total = 0;
  for (int w = 0; w < W; w++) total += count_out[w];
  if (total && total <= nspec) {
    size_t run = 0;
    for (int w = 0; w < W; w++) {
      std::memcpy(ids_out + offset[w], spec.data() + run, sizeof(int32_t) * std::min<size_t>(offset[w + 1] - offset[w], count_out[w]));
      run += count_out[w];
    }
  } else if (total) {   // the host copy moves exactly the flagged ids, through the pinned staging mirror (a pageable copy runs at a few 100 MB/s)
    Staged s2(c, t, sizeof(int32_t) * total);
    if (!s2.ok) { ctx_set_error(c, "gfbe_ftab_check_outliers: staging allocation failed"); return GFBE_DEVICE_ERROR; }
    FT_CHECK(c, hipMemcpyAsync(t->stage_h, t->d.keep, sizeof(int32_t) * total, hipMemcpyDeviceToHost, ctx_stream(c)));
    FT_CHECK(c, hipStreamSynchronize(ctx_stream(c)));
    const int32_t *tmp = (const int32_t *)t->stage_h;
    size_t run = 0;
    for (int w = 0; w < W; w++) {
      std::memcpy(ids_out + offset[w], tmp + run, sizeof(int32_t) * std::min<size_t>(offset[w + 1] - offset[w], count_out[w]));
      run += count_out[w];
    }
  }
  return ft_finish(c, t);
}

gfbe_status gfbe_ftab_size(gfbe_ctx *c, gfbe_ftab *t, int32_t *n) {
  gfbe_status st = ft_ready(c, t);
  if (st != GFBE_OK) return st;
  if (!n) return GFBE_BAD_INPUT;
  FT_CHECK(c, hipMemcpyAsync(t->stage_h, t->d.count, sizeof(int) * t->d.W, hipMemcpyDeviceToHost, ctx_stream(c)));   // (through the pinned mirror)
  FT_CHECK(c, hipStreamSynchronize(ctx_stream(c)));
  std::memcpy(n, t->stage_h, sizeof(int) * t->d.W);
  return GFBE_OK;
}

gfbe_status gfbe_ftab_download(gfbe_ctx *c, gfbe_ftab *t, int32_t w, int32_t *id, int32_t *start, int32_t *nobs, double *obs8,
                               double *obs_td, double *depth, int32_t *eflag, int32_t *sflag) {
  gfbe_status st = ft_ready(c, t);
  if (st != GFBE_OK) return st;
  if (w < 0 || w >= t->d.W) return GFBE_BAD_INPUT;
  int n = 0;
  FT_CHECK(c, hipMemcpyAsync(&n, t->d.count + w, sizeof(int), hipMemcpyDeviceToHost, ctx_stream(c)));
  FT_CHECK(c, hipStreamSynchronize(ctx_stream(c)));
  const size_t base = (size_t)w * t->d.F;
  const int b = t->cur;
  hipStream_t s = ctx_stream(c);
#define DN(h, dptr, cnt) if (h && n) FT_CHECK(c, hipMemcpyAsync(h, dptr, sizeof(*h) * (cnt), hipMemcpyDeviceToHost, s))
  DN(id, t->d.id[b] + base, (size_t)n); DN(start, t->d.start[b] + base, (size_t)n); DN(nobs, t->d.nobs[b] + base, (size_t)n);
  DN(eflag, t->d.eflag[b] + base, (size_t)n); DN(sflag, t->d.sflag[b] + base, (size_t)n); DN(depth, t->d.depth[b] + base, (size_t)n);
  DN(obs8, t->d.obs[b] + base * NOBS * OW, (size_t)n * NOBS * OW); DN(obs_td, t->d.td[b] + base * NOBS, (size_t)n * NOBS);
#undef DN
  FT_CHECK(c, hipStreamSynchronize(s));
  return GFBE_OK;
}

void gfbe_slide_window_state(gfbe_state *s, int32_t flag) {   // estimator.cpp:3700-3858 (the para_* view of Rs/Ps/Vs/Bas/Bgs)
  const int W = GFBE_WINDOW_SIZE;
  if (flag == GFBE_MARGIN_OLD) {
    std::memmove(s->para_Pose[0], s->para_Pose[1], sizeof(s->para_Pose[0]) * W);
    std::memmove(s->para_SpeedBias[0], s->para_SpeedBias[1], sizeof(s->para_SpeedBias[0]) * W);
    // frame WINDOW_SIZE keeps the newest values (Ps[WINDOW_SIZE] = Ps[WINDOW_SIZE - 1] after the shift)
  } else if (flag == GFBE_MARGIN_SECOND_NEW) {
    std::memcpy(s->para_Pose[W - 1], s->para_Pose[W], sizeof s->para_Pose[0]);
    std::memcpy(s->para_SpeedBias[W - 1], s->para_SpeedBias[W], sizeof s->para_SpeedBias[0]);
  }
}

}  // extern "C"

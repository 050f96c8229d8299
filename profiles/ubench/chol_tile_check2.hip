// Round 5: the pipelined tile step (chol_inv_tile16_p: the 1/sqrt chain of the next pivot started after ONE update of the current one,
// the inverse's columns dealt over the four row groups) against chol_inv_tile16 — bit for bit — and both against a host computation;
// time per call of each by one wave on an idle CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include profiles/ubench/chol_tile_check2.hip -o profiles/ubench/chol_tile_check2
#include "../../ground-fusion2_amd/csrc/gfbe_solve.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
using namespace gfd;

template <int P>
__global__ __launch_bounds__(64) void k_tile(const double *A, double *W, double *z, int zrow, int reps, long long *cycles, int *okout) {
  __shared__ double T[TB * TB], zl[TB];
  const int lane = threadIdx.x;
  bool ok = true;
  long long t0 = 0, t1 = 0;
  for (int rep = 0; rep < reps; rep++) {
    for (int e = lane; e < TB * TB; e += 64) T[tsw(e / TB, e % TB)] = A[e];
    __syncthreads();
    if (rep == reps - 1) t0 = wall_clock64();
    ok = P ? chol_inv_tile16_p((lds_double *)T, lane, zrow, (lds_double *)zl) : chol_inv_tile16((lds_double *)T, lane, zrow, (lds_double *)zl);
    if (rep == reps - 1) t1 = wall_clock64();
    __syncthreads();
  }
  for (int e = lane; e < TB * TB; e += 64) W[e] = T[tsw(e / TB, e % TB)];
  if (lane < TB) z[lane] = zl[lane];
  if (lane == 0) { *cycles = t1 - t0; *okout = ok; }
}

int main() {
  const int n = TB;
  srand(7);
  std::vector<double> M(n * n), A(n * n), L(n * n, 0.0), Winv(n * n, 0.0);
  for (auto &v : M) v = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double s = (i == j) ? 0.5 : 0.0; for (int k = 0; k < n; k++) s += M[i * n + k] * M[j * n + k]; A[i * n + j] = s; }
  for (int j = 0; j < n; j++) {
    double d = A[j * n + j]; for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k];
    L[j * n + j] = sqrt(d);
    for (int i = j + 1; i < n; i++) { double s = A[i * n + j]; for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k]; L[i * n + j] = s / L[j * n + j]; }
  }
  for (int j = 0; j < n; j++) for (int i = j; i < n; i++) { double s = (i == j) ? 1.0 : 0.0; for (int k = j; k < i; k++) s -= L[i * n + k] * Winv[k * n + j]; Winv[i * n + j] = s / L[i * n + i]; }
  double *dA, *dW, *dz; long long *dc; int *dok;
  hipMalloc(&dA, sizeof(double) * n * n); hipMalloc(&dW, sizeof(double) * n * n); hipMalloc(&dz, sizeof(double) * n); hipMalloc(&dc, 8); hipMalloc(&dok, 4);
  hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
  int rc = 0;
  std::vector<double> W0(n * n), z0(n);
  for (int zrow : {11, -1}) {
    for (int P = 0; P < 2; P++) {
      if (P) hipLaunchKernelGGL(k_tile<1>, dim3(1), dim3(64), 0, 0, dA, dW, dz, zrow, 20, dc, dok);
      else hipLaunchKernelGGL(k_tile<0>, dim3(1), dim3(64), 0, 0, dA, dW, dz, zrow, 20, dc, dok);
      std::vector<double> W(n * n), z(n); long long cyc; int ok;
      hipMemcpy(W.data(), dW, sizeof(double) * n * n, hipMemcpyDeviceToHost); hipMemcpy(z.data(), dz, sizeof(double) * n, hipMemcpyDeviceToHost);
      hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost); hipMemcpy(&ok, dok, 4, hipMemcpyDeviceToHost);
      double ew = 0, ez = 0, sc = 0;
      for (int i = 0; i < n * n; i++) { ew = fmax(ew, fabs(W[i] - Winv[i])); sc = fmax(sc, fabs(Winv[i])); }
      if (zrow >= 0) for (int q = 0; q < zrow; q++) ez = fmax(ez, fabs(z[q] - L[zrow * n + q]));
      printf("%s zrow %2d: ok=%d  max|W - L^-1| = %.3e (scale %.3e)  max|z - L[zrow]| = %.3e  time %.2f us\n", P ? "pipelined " : "round 1-4 ", zrow, ok, ew, sc, ez, cyc * 0.01);
      if (!(ew < 1e-10 * sc && ez < 1e-12) || !ok) rc = 1;
      if (!P) { W0 = W; z0 = z; }
      else {
        const bool same = !memcmp(W0.data(), W.data(), sizeof(double) * n * n) && (zrow < 0 || !memcmp(z0.data(), z.data(), sizeof(double) * zrow));
        printf("   pipelined == round 1-4 bit for bit: %s\n", same ? "yes" : "NO");
        if (!same) rc = 1;
      }
    }
  }
  A[5 * n + 5] = -1.0;      // a tile that is not positive definite must be reported
  hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
  for (int P = 0; P < 2; P++) {
    if (P) hipLaunchKernelGGL(k_tile<1>, dim3(1), dim3(64), 0, 0, dA, dW, dz, -1, 1, dc, dok);
    else hipLaunchKernelGGL(k_tile<0>, dim3(1), dim3(64), 0, 0, dA, dW, dz, -1, 1, dc, dok);
    int ok; hipMemcpy(&ok, dok, 4, hipMemcpyDeviceToHost);
    printf("indefinite tile (%s): ok=%d (expected 0)\n", P ? "pipelined" : "round 1-4", ok);
    if (ok) rc = 1;
  }
  return rc;
}

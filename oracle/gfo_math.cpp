// oracle/gfo_math.cpp — TEST INFRASTRUCTURE ONLY (CPU oracle). See gfo_math.h.
#include "gfo_math.h"
#include <vector>
#include <algorithm>

namespace gfo {

// PartialPivLU inverse (what Eigen's MatrixBase::inverse() does for n > 4; used by
// covariance.inverse() at imu_factor.h:73 and wheel_factor.h:85).
bool lu_inverse(const double *A, double *Ainv, int n) {
  std::vector<double> lu(A, A + n * n);
  std::vector<int> perm(n);
  for (int i = 0; i < n; i++) perm[i] = i;
  for (int k = 0; k < n; k++) {
    int piv = k;
    double best = std::fabs(lu[k * n + k]);
    for (int i = k + 1; i < n; i++)
      if (std::fabs(lu[i * n + k]) > best) { best = std::fabs(lu[i * n + k]); piv = i; }
    if (best == 0.0) return false;
    if (piv != k) {
      for (int j = 0; j < n; j++) std::swap(lu[k * n + j], lu[piv * n + j]);
      std::swap(perm[k], perm[piv]);
    }
    for (int i = k + 1; i < n; i++) {
      lu[i * n + k] /= lu[k * n + k];
      const double f = lu[i * n + k];
      for (int j = k + 1; j < n; j++) lu[i * n + j] -= f * lu[k * n + j];
    }
  }
  // Solve LU X = P I column by column.
  for (int c = 0; c < n; c++) {
    std::vector<double> y(n);
    for (int i = 0; i < n; i++) {
      double s = (perm[i] == c) ? 1.0 : 0.0;
      for (int j = 0; j < i; j++) s -= lu[i * n + j] * y[j];
      y[i] = s;
    }
    for (int i = n - 1; i >= 0; i--) {
      double s = y[i];
      for (int j = i + 1; j < n; j++) s -= lu[i * n + j] * Ainv[j * n + c];
      Ainv[i * n + c] = s / lu[i * n + i];
    }
  }
  return true;
}

bool llt_lower(const double *A, double *L, int n) {
  for (int i = 0; i < n * n; i++) L[i] = 0.0;
  for (int j = 0; j < n; j++) {
    double d = A[j * n + j];
    for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    L[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = s / d;
    }
  }
  return true;
}

bool sqrt_info_from_cov(const double *cov, double *sqrt_info, int n) {
  std::vector<double> inv(n * n), L(n * n);
  if (!lu_inverse(cov, inv.data(), n)) return false;
  if (!llt_lower(inv.data(), L.data(), n)) return false;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) sqrt_info[i * n + j] = L[j * n + i];
  return true;
}

// Symmetric eigen-decomposition by Householder tridiagonalisation followed by the implicit-shift
// QL iteration (the classical EISPACK tred2/tql2 scheme, i.e. the same family of algorithm as
// Eigen::SelfAdjointEigenSolver). Eigenvalues ascending; V column j is eigenvector j.
void sym_eig(const double *A_in, int n, double *w, double *V) {
  std::vector<double> e(n, 0.0);
  double *d = w;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) V[i * n + j] = 0.5 * (A_in[i * n + j] + A_in[j * n + i]);
  // --- Householder reduction to tridiagonal form; V accumulates the transformation ---
  for (int j = 0; j < n; j++) d[j] = V[(n - 1) * n + j];
  for (int i = n - 1; i > 0; i--) {
    double scale = 0.0, h = 0.0;
    for (int k = 0; k < i; k++) scale += std::fabs(d[k]);
    if (scale == 0.0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; j++) { d[j] = V[(i - 1) * n + j]; V[i * n + j] = 0.0; V[j * n + i] = 0.0; }
    } else {
      for (int k = 0; k < i; k++) { d[k] /= scale; h += d[k] * d[k]; }
      double f = d[i - 1];
      double g = std::sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h -= f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; j++) e[j] = 0.0;
      for (int j = 0; j < i; j++) {
        f = d[j];
        V[j * n + i] = f;
        g = e[j] + V[j * n + j] * f;
        for (int k = j + 1; k <= i - 1; k++) { g += V[k * n + j] * d[k]; e[k] += V[k * n + j] * f; }
        e[j] = g;
      }
      f = 0.0;
      for (int j = 0; j < i; j++) { e[j] /= h; f += e[j] * d[j]; }
      const double hh = f / (h + h);
      for (int j = 0; j < i; j++) e[j] -= hh * d[j];
      for (int j = 0; j < i; j++) {
        f = d[j]; g = e[j];
        for (int k = j; k <= i - 1; k++) V[k * n + j] -= (f * e[k] + g * d[k]);
        d[j] = V[(i - 1) * n + j];
        V[i * n + j] = 0.0;
      }
    }
    d[i] = h;
  }
  for (int i = 0; i < n - 1; i++) {
    V[(n - 1) * n + i] = V[i * n + i];
    V[i * n + i] = 1.0;
    const double h = d[i + 1];
    if (h != 0.0) {
      for (int k = 0; k <= i; k++) d[k] = V[k * n + i + 1] / h;
      for (int j = 0; j <= i; j++) {
        double g = 0.0;
        for (int k = 0; k <= i; k++) g += V[k * n + i + 1] * V[k * n + j];
        for (int k = 0; k <= i; k++) V[k * n + j] -= g * d[k];
      }
    }
    for (int k = 0; k <= i; k++) V[k * n + i + 1] = 0.0;
  }
  for (int j = 0; j < n; j++) { d[j] = V[(n - 1) * n + j]; V[(n - 1) * n + j] = 0.0; }
  V[(n - 1) * n + n - 1] = 1.0;
  e[0] = 0.0;
  // --- implicit QL on the tridiagonal (d, e) ---
  for (int i = 1; i < n; i++) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = 2.220446049250313e-16;
  for (int l = 0; l < n; l++) {
    tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
    int m = l;
    while (m < n) { if (std::fabs(e[m]) <= eps * tst1) break; m++; }
    if (m > l) {
      int iter = 0;
      do {
        iter++;
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = std::hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; i++) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
        const double el1 = e[l + 1];
        for (int i = m - 1; i >= l; i--) {
          c3 = c2; c2 = c; s2 = s;
          g = c * e[i];
          h = c * p;
          r = std::hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          for (int k = 0; k < n; k++) {
            h = V[k * n + i + 1];
            V[k * n + i + 1] = s * V[k * n + i] + c * h;
            V[k * n + i] = c * V[k * n + i] - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (std::fabs(e[l]) > eps * tst1 && iter < 200);
    }
    d[l] = d[l] + f;
    e[l] = 0.0;
  }
  // --- sort ascending ---
  for (int i = 0; i < n - 1; i++) {
    int k = i; double p = d[i];
    for (int j = i + 1; j < n; j++) if (d[j] < p) { k = j; p = d[j]; }
    if (k != i) {
      d[k] = d[i]; d[i] = p;
      for (int j = 0; j < n; j++) std::swap(V[j * n + i], V[j * n + k]);
    }
  }
}

}  // namespace gfo

// 3x3 SVD (one-sided Jacobi, f64) and determinant, shared by reg.hip (weighted Procrustes) and
// o3d.hip (Umeyama inside ICP / RANSAC).
#pragma once
#include <hip/hip_runtime.h>

// ---- 3x3 SVD, one-sided Jacobi in f64: A = U diag(s) V^T, s sorted descending ----------------
__device__ inline void svd3(const double A[9], double U[9], double s[3], double V[9]) {
  double a[3][3], v[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { a[i][j] = A[i * 3 + j]; v[i][j] = (i == j) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int qq = p + 1; qq < 3; ++qq) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < 3; ++i) {
          alpha += a[i][p] * a[i][p];
          beta += a[i][qq] * a[i][qq];
          gamma += a[i][p] * a[i][qq];
        }
        off = fmax(off, fabs(gamma) / (sqrt(alpha * beta) + 1e-300));
        if (fabs(gamma) < 1e-300) continue;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        for (int i = 0; i < 3; ++i) {
          double x = a[i][p], y = a[i][qq];
          a[i][p] = c * x - sn * y;
          a[i][qq] = sn * x + c * y;
          x = v[i][p]; y = v[i][qq];
          v[i][p] = c * x - sn * y;
          v[i][qq] = sn * x + c * y;
        }
      }
    if (off < 1e-15) break;
  }
  double sv[3];
  int ord[3] = {0, 1, 2};
  for (int j = 0; j < 3; ++j) sv[j] = sqrt(a[0][j] * a[0][j] + a[1][j] * a[1][j] + a[2][j] * a[2][j]);
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (sv[ord[j]] < sv[ord[j + 1]]) { int t = ord[j]; ord[j] = ord[j + 1]; ord[j + 1] = t; }
  double u[3][3];
  const double tol = 1e-13 * fmax(sv[ord[0]], 1e-300);
  for (int jj = 0; jj < 3; ++jj) {
    const int j = ord[jj];
    s[jj] = sv[j];
    for (int i = 0; i < 3; ++i) V[i * 3 + jj] = v[i][j];
    if (sv[j] > tol) {
      for (int i = 0; i < 3; ++i) u[i][jj] = a[i][j] / sv[j];
    } else {
      // rank deficient: complete the orthonormal basis (Gram-Schmidt against canonical axes)
      double best[3] = {0, 0, 0}, bn = -1.0;
      for (int e = 0; e < 3; ++e) {
        double c[3] = {e == 0 ? 1.0 : 0.0, e == 1 ? 1.0 : 0.0, e == 2 ? 1.0 : 0.0};
        for (int pj = 0; pj < jj; ++pj) {
          double d = c[0] * u[0][pj] + c[1] * u[1][pj] + c[2] * u[2][pj];
          for (int i = 0; i < 3; ++i) c[i] -= d * u[i][pj];
        }
        double n = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        if (n > bn) { bn = n; for (int i = 0; i < 3; ++i) best[i] = c[i] / n; }
      }
      for (int i = 0; i < 3; ++i) u[i][jj] = best[i];
    }
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) U[i * 3 + j] = u[i][j];
}

__device__ __forceinline__ double det3(const double M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
         M[2] * (M[3] * M[7] - M[4] * M[6]);
}


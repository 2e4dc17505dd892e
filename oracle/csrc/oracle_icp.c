/*
 * oracle_icp.c — plain-C restatement of the reference ICP (TEST INFRASTRUCTURE, see oracle/__init__.py).
 *
 * Parity: PINNED through tests/test_oracle_c.py, which replays the golden vectors captured from the
 * reference (tests/golden/icp_*.npz) and cross-checks this file against oracle/icp_oracle.py.
 *
 * Why a C copy next to the NumPy one: the float32 "device recipe" needs a true fused multiply-add
 * (fmaf) to reproduce fma(dz,dz, fma(dy,dy, dx*dx)) bit for bit — NumPy can only emulate it through
 * float64 (double rounding) — and a brute-force all-pairs search at 4096² / 100k sizes is too slow
 * in NumPy.  Everything else follows modules/icp.py line by line:
 *   icp.py:116     p = T_prev · [x y z 1]
 *   icp.py:29-41   nearest target (exact, here brute force, lowest index on ties)
 *   icp.py:43-74   Kabsch: centroids, S = XcᵀYc, R = V diag(1,1,det(VUᵀ)) Uᵀ, t = q̄ − R p̄
 *   icp.py:76-98   error = Σ‖p−q‖² [ / (bbox diagonal² · N) ]
 *   icp.py:119     T = T_step · T_prev
 *   icp.py:163-181 stop rules
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RECIPE_F64 0     /* everything double, as the reference */
#define RECIPE_DEVICE 1  /* float32 points + float32 argmin (fmaf), float64 moments/solve */

/* ---- nearest neighbour ---------------------------------------------------------------------- */
/* p32/q32: n×d and m×d float32; exact float32 distances with the kernel's operation order */
void oracle_nn_f32(const float *p, int64_t n, const float *q, int64_t m, int d, int32_t *idx, float *d2out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    float best = INFINITY;
    int64_t bi = 0;
    for (int64_t j = 0; j < m; ++j) {
      const float dx = p[d * i] - q[d * j];
      const float dy = p[d * i + 1] - q[d * j + 1];
      float v = fmaf(dy, dy, dx * dx);
      if (d == 3) {
        const float dz = p[d * i + 2] - q[d * j + 2];
        v = fmaf(dz, dz, v);
      }
      if (v < best || j == 0) { best = v; bi = j; }
    }
    idx[i] = (int32_t)bi;
    if (d2out) d2out[i] = best;
  }
}

void oracle_nn_f64(const double *p, int64_t n, const double *q, int64_t m, int d, int32_t *idx) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    double best = INFINITY;
    int64_t bi = 0;
    for (int64_t j = 0; j < m; ++j) {
      double v = 0.0;
      for (int k = 0; k < d; ++k) { const double t = p[d * i + k] - q[d * j + k]; v += t * t; }
      if (v < best || j == 0) { best = v; bi = j; }
    }
    idx[i] = (int32_t)bi;
  }
}

/* ---- 3×3 helpers ---------------------------------------------------------------------------- */
static void mat3_mul(const double *A, const double *B, double *C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j];
      C[3 * i + j] = s;
    }
}
static double det3(const double *A) {
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}

/* symmetric 3×3 eigen-decomposition by cyclic Jacobi: M = E diag(w) Eᵀ (columns of E) */
static void eig3_sym(const double *M, double *E, double *w) {
  double A[9];
  memcpy(A, M, sizeof(A));
  for (int i = 0; i < 9; ++i) E[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = A[3 * p + q];
        if (apq == 0.0) continue;
        const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {            /* A ← A·J */
          const double akp = A[3 * k + p], akq = A[3 * k + q];
          A[3 * k + p] = c * akp - s * akq;
          A[3 * k + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {            /* A ← Jᵀ·A */
          const double apk = A[3 * p + k], aqk = A[3 * q + k];
          A[3 * p + k] = c * apk - s * aqk;
          A[3 * q + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double ekp = E[3 * k + p], ekq = E[3 * k + q];
          E[3 * k + p] = c * ekp - s * ekq;
          E[3 * k + q] = s * ekp + c * ekq;
        }
      }
  }
  w[0] = A[0]; w[1] = A[4]; w[2] = A[8];
}

/* R = V diag(1,1,det(VUᵀ)) Uᵀ for S = U Σ Vᵀ, via the eigenvectors of SᵀS (=V) and U = S V Σ⁻¹;
 * the third singular pair is completed by cross products (its sign cancels in R). */
static void kabsch_rotation(const double *S, double *R) {
  double StS[9], St[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) St[3 * i + j] = S[3 * j + i];
  mat3_mul(St, S, StS);
  double E[9], w[3];
  eig3_sym(StS, E, w);
  int o[3] = {0, 1, 2};                                      /* sort eigenvalues descending */
  for (int a = 0; a < 2; ++a) for (int b = a + 1; b < 3; ++b) if (w[o[b]] > w[o[a]]) { int t = o[a]; o[a] = o[b]; o[b] = t; }
  for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  if (!(w[o[0]] > 0.0)) return;                              /* S = 0 → identity */
  double v[2][3], u[2][3];
  for (int c = 0; c < 2; ++c) {
    for (int i = 0; i < 3; ++i) v[c][i] = E[3 * i + o[c]];
    double nrm = 0;
    for (int i = 0; i < 3; ++i) { u[c][i] = S[3 * i] * v[c][0] + S[3 * i + 1] * v[c][1] + S[3 * i + 2] * v[c][2]; nrm += u[c][i] * u[c][i]; }
    nrm = sqrt(nrm);
    for (int i = 0; i < 3; ++i) u[c][i] /= nrm;
  }
  /* re-orthogonalise u2 against u1 (guards the squared condition number of SᵀS) */
  double dot = u[0][0] * u[1][0] + u[0][1] * u[1][1] + u[0][2] * u[1][2], nrm = 0;
  for (int i = 0; i < 3; ++i) { u[1][i] -= dot * u[0][i]; nrm += u[1][i] * u[1][i]; }
  nrm = sqrt(nrm);
  for (int i = 0; i < 3; ++i) u[1][i] /= nrm;
  const double u3[3] = {u[0][1] * u[1][2] - u[0][2] * u[1][1], u[0][2] * u[1][0] - u[0][0] * u[1][2], u[0][0] * u[1][1] - u[0][1] * u[1][0]};
  const double v3[3] = {v[0][1] * v[1][2] - v[0][2] * v[1][1], v[0][2] * v[1][0] - v[0][0] * v[1][2], v[0][0] * v[1][1] - v[0][1] * v[1][0]};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[3 * i + j] = v[0][i] * u[0][j] + v[1][i] * u[1][j] + v3[i] * u3[j];
  (void)det3;
}

/* get_transform on paired rows (n×3 doubles each) */
void oracle_kabsch(const double *p, const double *q, int64_t n, double *T) {
  double pb[3] = {0, 0, 0}, qb[3] = {0, 0, 0}, S[9] = {0};
  for (int64_t i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { pb[k] += p[3 * i + k]; qb[k] += q[3 * i + k]; }
  for (int k = 0; k < 3; ++k) { pb[k] /= (double)n; qb[k] /= (double)n; }
  for (int64_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) S[3 * a + b] += (p[3 * i + a] - pb[a]) * (q[3 * i + b] - qb[b]);
  double R[9];
  kabsch_rotation(S, R);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j];
    T[4 * i + 3] = qb[i] - (R[3 * i] * pb[0] + R[3 * i + 1] * pb[1] + R[3 * i + 2] * pb[2]);
  }
  T[12] = T[13] = T[14] = 0.0; T[15] = 1.0;
}

/* ---- run_icp --------------------------------------------------------------------------------- */
/* src, tgt: n×3 / m×3 float64 (planar clouds already padded with z = 0).  d_search = 2 restricts
 * the device-recipe search to x,y (what the planar kernel does; identical result since z ≡ 0).
 * Returns the number of iterations executed, or −1 on allocation failure. */
int oracle_run_icp(const double *src, int64_t n, const double *tgt, int64_t m, const double *init_T,
                   double epsilon, int max_iters, double stopping_thresh, int normalize, int recipe,
                   int d_search, double *out_T, double *out_err, int32_t *out_corr) {
  double *p = malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
  double *q = malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
  double *s64 = malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
  double *t64 = malloc(sizeof(double) * 3 * (size_t)m);
  float *p32 = malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
  float *t32 = malloc(sizeof(float) * 3 * (size_t)m);
  int32_t *corr = malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  if (!p || !q || !s64 || !t64 || !p32 || !t32 || !corr) return -1;
  for (int64_t i = 0; i < 3 * n; ++i) s64[i] = (recipe == RECIPE_DEVICE) ? (double)(float)src[i] : src[i];
  for (int64_t i = 0; i < 3 * m; ++i) t64[i] = (recipe == RECIPE_DEVICE) ? (double)(float)tgt[i] : tgt[i];
  for (int64_t j = 0; j < m; ++j) for (int k = 0; k < d_search; ++k) t32[d_search * j + k] = (float)t64[3 * j + k];

  double T[16];
  memcpy(T, init_T, sizeof(T));
  int iteration = 0, executed = 0, has_last = 0;
  double last = 0.0, err = 0.0;
  for (;;) {
    for (int64_t i = 0; i < n; ++i)
      for (int r = 0; r < 3; ++r)
        p[3 * i + r] = T[4 * r] * s64[3 * i] + T[4 * r + 1] * s64[3 * i + 1] + T[4 * r + 2] * s64[3 * i + 2] + T[4 * r + 3];
    if (recipe == RECIPE_DEVICE) {
      for (int64_t i = 0; i < n; ++i) for (int k = 0; k < d_search; ++k) p32[d_search * i + k] = (float)p[3 * i + k];
      oracle_nn_f32(p32, n, t32, m, d_search, corr, NULL);
    } else {
      oracle_nn_f64(p, n, t64, m, 3, corr);
    }
    for (int64_t i = 0; i < n; ++i) memcpy(q + 3 * i, t64 + 3 * (int64_t)corr[i], 3 * sizeof(double));
    double Ts[16], Tn[16];
    oracle_kabsch(p, q, n, Ts);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double s = 0;
        for (int k = 0; k < 4; ++k) s += Ts[4 * i + k] * T[4 * k + j];
        Tn[4 * i + j] = s;
      }
    err = 0.0;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n; ++i)
      for (int k = 0; k < 3; ++k) {
        const double dlt = p[3 * i + k] - q[3 * i + k];
        err += dlt * dlt;
        lo[k] = fmin(lo[k], fmin(p[3 * i + k], q[3 * i + k]));
        hi[k] = fmax(hi[k], fmax(p[3 * i + k], q[3 * i + k]));
      }
    if (normalize) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += (hi[k] - lo[k]) * (hi[k] - lo[k]);
      const double diag = sqrt(s);
      err = err / (diag * diag * (double)n);
    }
    memcpy(T, Tn, sizeof(T));
    ++executed;
    if (err < epsilon) break;
    if (iteration >= max_iters) break;
    if (has_last && fabs(last - err) < stopping_thresh) break;
    last = err; has_last = 1; ++iteration;
  }
  memcpy(out_T, T, sizeof(T));
  *out_err = err;
  if (out_corr) memcpy(out_corr, corr, sizeof(int32_t) * (size_t)n);
  free(p); free(q); free(s64); free(t64); free(p32); free(t32); free(corr);
  return executed;
}

// lslam_solve.h — float64 rigid-transform solve from raw moments + ICP stop rules.
//
// Compiles for the device (hipcc) and for the host (g++, used by the CPU unit tests of the
// solver through liblslam_hostcheck.so).  One thread runs this per ICP iteration; its cost is
// O(1), the O(N) and O(N·M) work lives in the kernels.
//
// Reference semantics restated here:
//   modules/icp.py:43-74   get_transform (Kabsch): R = V diag(1,1,det(V Uᵀ)) Uᵀ, t = q̄ − R p̄
//   modules/icp.py:76-98   get_error (optional bounding-box normalisation)
//   modules/icp.py:119     T = T_step @ T_prev
//   modules/icp.py:172-181 stop rules and their order
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define LSLAM_HD __host__ __device__ inline
#define LSLAM_UNROLL _Pragma("unroll")
#else
#define LSLAM_HD inline
#define LSLAM_UNROLL
#endif

namespace lslam {

// Raw sums over the matched pairs (p = transformed source, q = matched target), float64.
struct Moments {
  double n;          // number of pairs
  double sp[3];      // Σ p
  double sq[3];      // Σ q
  double spq[9];     // Σ p qᵀ  (row-major, spq[3*i+j] = Σ p_i q_j)
  double sse;        // Σ ‖p − q‖²
  double lo[3];      // min over p ∪ q   (only when normalising)
  double hi[3];      // max over p ∪ q
};

// Orthogonal polar factor for planar data, as the reference's SVD delivers it when S has a zero
// third row and column: a rotation if det S₂ > 0, otherwise a REFLECTION with R[2][2] = −1
// (SURVEY §8 I5).  S₂ = 0 (single point) gives the identity, as LAPACK does.
LSLAM_HD void polar_planar(const double S[9], double R[9]) {
  const double s00 = S[0], s01 = S[1], s10 = S[3], s11 = S[4];
  for (int i = 0; i < 9; ++i) R[i] = 0.0;
  R[8] = 1.0;
  if (s00 * s11 - s01 * s10 > 0.0) {
    const double a = s00 + s11, b = s01 - s10;
    const double h = sqrt(a * a + b * b);      // |S| entries are Σ p·q of metre-scale data: no overflow concern
    const double c = a / h, s = b / h;
    R[0] = c; R[1] = -s; R[3] = s; R[4] = c;
    return;
  }
  const double a = s00 - s11, b = s01 + s10;
  const double h = sqrt(a * a + b * b);
  if (h == 0.0) { R[0] = 1.0; R[4] = 1.0; return; }
  const double c = a / h, s = b / h;
  R[0] = c; R[1] = s; R[3] = s; R[4] = -c; R[8] = -1.0;
}

// General 3-D case.  One-sided (Hestenes) Jacobi SVD in float64: rotate column pairs of
// A = S·V until orthogonal; then A = U Σ.  With u₁,u₂,v₁,v₂ the two dominant singular pairs,
//     R = v₁u₁ᵀ + v₂u₂ᵀ + (v₁×v₂)(u₁×u₂)ᵀ
// equals V diag(1,1,det(VUᵀ)) Uᵀ whatever sign LAPACK would pick for the third pair, so no
// SVD sign convention leaks (R is unique whenever rank S ≥ 2).
LSLAM_HD void polar_3d(const double S[9], double R[9]) {
  double A[9], V[9];
  double amax = 0.0;
  for (int i = 0; i < 9; ++i) { A[i] = S[i]; V[i] = 0.0; amax = fmax(amax, fabs(S[i])); }
  V[0] = V[4] = V[8] = 1.0;
  for (int i = 0; i < 9; ++i) R[i] = 0.0;
  if (!(amax > 0.0)) { R[0] = R[4] = R[8] = 1.0; return; }   // S = 0 → identity (LAPACK: U = V = I)

  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    LSLAM_UNROLL
    for (int p = 0; p < 2; ++p)
      LSLAM_UNROLL
      for (int q = p + 1; q < 3; ++q) {
        double al = 0.0, be = 0.0, ga = 0.0;
        for (int i = 0; i < 3; ++i) {
          al += A[3 * i + p] * A[3 * i + p];
          be += A[3 * i + q] * A[3 * i + q];
          ga += A[3 * i + p] * A[3 * i + q];
        }
        const double lim = 1e-16 * sqrt(al * be);
        if (fabs(ga) <= lim || ga == 0.0) continue;
        off = fmax(off, fabs(ga) / fmax(sqrt(al * be), 1e-300));
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 3; ++i) {
          const double ap = A[3 * i + p], aq = A[3 * i + q];
          A[3 * i + p] = c * ap - s * aq;
          A[3 * i + q] = s * ap + c * aq;
          const double vp = V[3 * i + p], vq = V[3 * i + q];
          V[3 * i + p] = c * vp - s * vq;
          V[3 * i + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  // column norms = singular values; order the columns by descending norm with fixed-index swaps
  // (no runtime-indexed arrays: those would live in scratch memory on the GPU)
  double sg[3];
  for (int j = 0; j < 3; ++j)
    sg[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
#define LSLAM_SWAP_COLS(a, b)                                                         \
  if (sg[b] > sg[a]) {                                                                \
    double t_ = sg[a]; sg[a] = sg[b]; sg[b] = t_;                                     \
    for (int i = 0; i < 3; ++i) {                                                     \
      t_ = A[3 * i + a]; A[3 * i + a] = A[3 * i + b]; A[3 * i + b] = t_;              \
      t_ = V[3 * i + a]; V[3 * i + a] = V[3 * i + b]; V[3 * i + b] = t_;              \
    }                                                                                 \
  }
  LSLAM_SWAP_COLS(0, 1)
  LSLAM_SWAP_COLS(1, 2)
  LSLAM_SWAP_COLS(0, 1)
#undef LSLAM_SWAP_COLS
  constexpr int i0 = 0, i1 = 1;
  double u1[3], u2[3], v1[3], v2[3];
  for (int i = 0; i < 3; ++i) {
    u1[i] = A[3 * i + i0] / sg[i0];
    v1[i] = V[3 * i + i0];
    v2[i] = V[3 * i + i1];
  }
  if (sg[i1] > 1e-14 * sg[i0]) {
    for (int i = 0; i < 3; ++i) u2[i] = A[3 * i + i1] / sg[i1];
  } else {
    // rank 1: the reference's answer depends on LAPACK's null-space basis (parity undefined,
    // SURVEY §8 I5).  Deterministic completion: u₂ := the image of v₂ under the minimal
    // rotation taking v₁ to u₁ (so that R is that minimal rotation).
    double ax[3] = {v1[1] * u1[2] - v1[2] * u1[1], v1[2] * u1[0] - v1[0] * u1[2],
                    v1[0] * u1[1] - v1[1] * u1[0]};
    const double cth = v1[0] * u1[0] + v1[1] * u1[1] + v1[2] * u1[2];
    const double k = 1.0 / (1.0 + cth + 1e-300);
    const double d = ax[0] * v2[0] + ax[1] * v2[1] + ax[2] * v2[2];
    double cr[3] = {ax[1] * v2[2] - ax[2] * v2[1], ax[2] * v2[0] - ax[0] * v2[2],
                    ax[0] * v2[1] - ax[1] * v2[0]};
    for (int i = 0; i < 3; ++i) u2[i] = cth * v2[i] + cr[i] + k * d * ax[i];
  }
  const double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2],
                        u1[0] * u2[1] - u1[1] * u2[0]};
  const double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2],
                        v1[0] * v2[1] - v1[1] * v2[0]};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[3 * i + j] = v1[i] * u1[j] + v2[i] * u2[j] + v3[i] * u3[j];
}

// The pieces of T_step from the moments: R (3x3 row-major), the centroids pb = Σp/n, qb = Σq/n.
// S = Σpqᵀ − Σp Σqᵀ / n is the centred cross-covariance Xcᵀ Yc of icp.py:58-61.
LSLAM_HD void step_parts(const Moments& m, bool planar, double R[9], double pb[3], double qb[3]) {
  double S[9];
  const double inv_n = 1.0 / m.n;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) S[3 * i + j] = m.spq[3 * i + j] - m.sp[i] * m.sq[j] * inv_n;
  if (planar) polar_planar(S, R); else polar_3d(S, R);
  for (int i = 0; i < 3; ++i) { pb[i] = m.sp[i] * inv_n; qb[i] = m.sq[i] * inv_n; }
}

// T_step (4x4 row-major): rotation R, translation qb − R·pb (icp.py:65-72).
LSLAM_HD void step_from_moments(const Moments& m, bool planar, double Tstep[16]) {
  double R[9], pb[3], qb[3];
  step_parts(m, planar, R, pb, qb);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Tstep[4 * i + j] = R[3 * i + j];
    Tstep[4 * i + 3] = qb[i] - (R[3 * i] * pb[0] + R[3 * i + 1] * pb[1] + R[3 * i + 2] * pb[2]);
  }
  Tstep[12] = Tstep[13] = Tstep[14] = 0.0;
  Tstep[15] = 1.0;
}

LSLAM_HD void matmul4(const double A[16], const double B[16], double C[16]) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
      C[4 * i + j] = s;
    }
}

// get_error's value: SSE, or SSE / (diag² · n) with diag = sqrt(Σ extent²) (icp.py:91-97;
// sqrt-then-square kept, as the reference does it).
LSLAM_HD double error_from_moments(const Moments& m, bool normalize) {
  if (!normalize) return m.sse;
  double s = 0.0;
  for (int i = 0; i < 3; ++i) { const double e = m.hi[i] - m.lo[i]; s += e * e; }
  const double diag = sqrt(s);
  return m.sse / (diag * diag * m.n);
}

// Loop state of run_icp (icp.py:163-181).
struct StopState {
  double last_err;
  int32_t iteration;   // the reference's `iteration` counter
  int32_t executed;    // icp_iteration calls made so far
  int32_t has_last;
};

LSLAM_HD void stop_init(StopState& s) { s.last_err = 0.0; s.iteration = 0; s.executed = 0; s.has_last = 0; }

// Call after each iteration with that iteration's error; true = leave the loop.
LSLAM_HD bool stop_after_iteration(StopState& s, double err, double epsilon, double thresh,
                                   int32_t max_iters) {
  s.executed += 1;
  if (err < epsilon) return true;
  if (s.iteration >= max_iters) return true;
  if (s.has_last && fabs(s.last_err - err) < thresh) return true;
  s.last_err = err;
  s.has_last = 1;
  s.iteration += 1;
  return false;
}

}  // namespace lslam

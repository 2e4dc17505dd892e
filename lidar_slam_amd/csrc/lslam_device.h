// lslam_device.h — device-side building blocks shared by the ICP kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lslam_solve.h"

namespace lslam {

constexpr int kWave = 64;            // gfx950 wavefront
constexpr float kFar = 3.0e38f;      // padding coordinate: (a − kFar)² overflows to +inf for any finite a

// Squared distance in float32, direct difference (never |a|²+|b|²−2a·b: the cancellation would
// corrupt the argmin at LiDAR ranges).  The exact operation order is part of the contract: the
// fix-up pass and oracle/csrc re-evaluate it and must reproduce the same bits.
__device__ __forceinline__ float dist2(float ax, float ay, float bx, float by) {
  const float dx = ax - bx;
  const float dy = ay - by;
  return __builtin_fmaf(dy, dy, dx * dx);
}
__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx;
  const float dy = ay - by;
  const float dz = az - bz;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// broadcast lane 0's value through scalar registers (the value must already be wave-uniform)
__device__ __forceinline__ double wave_uniform(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, kWave));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, kWave));
  return v;
}


// Element `lane & 15` (row-major) of T_step·T_prev, with T_step = [R | qb − R·pb] built from the moments — the SAME
// operations in the same order as step_from_moments + matmul4 perform for that element (bit-identical), but spread
// over 16 lanes: the one-lane version is a chain of ≈ 150 dependent float64 operations in every ICP iteration of every
// kernel.  All lanes of the wave hold the same R, pb, qb; `scratch` = 16 doubles of LDS nobody else touches; sh_Tprev =
// the current transform in LDS (read here; the caller stores the result back after this call returns).
__device__ __forceinline__ double step_times_prev_lane(const double (&R)[9], const double (&pb)[3], const double (&qb)[3],
                                                       const double* sh_Tprev, double* scratch, int lane) {
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) scratch[k] = R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { scratch[9 + k] = pb[k]; scratch[12 + k] = qb[k]; }
  }
  const int i = (lane >> 2) & 3, j = lane & 3;
  double ts0 = 0.0, ts1 = 0.0, ts2 = 0.0, ts3 = 1.0;           // row i of T_step (row 3: 0 0 0 1)
  if (i < 3) {
    ts0 = scratch[3 * i]; ts1 = scratch[3 * i + 1]; ts2 = scratch[3 * i + 2];
    ts3 = scratch[12 + i] - (ts0 * scratch[9] + ts1 * scratch[10] + ts2 * scratch[11]);
  }
  double s = 0.0;                                               // matmul4's accumulation order
  s += ts0 * sh_Tprev[j];
  s += ts1 * sh_Tprev[4 + j];
  s += ts2 * sh_Tprev[8 + j];
  s += ts3 * sh_Tprev[12 + j];
  return s;
}

// ---- the expanded-form FILTER of the all-pairs sweeps -----------------------
//   g(a, t) = |t|² − 2 a·t  ( = |a − t|² − |a|² )      2 (planar) / 3 (3-D) v_fma per pair, |t|² kept beside the target
// g is only a filter: its float32 error against the exact direct-difference distance is bounded, so the exact
// winner lies in a chunk whose filter minimum is within τ of the best one.  Shared by the batched kernel
// (lslam_icp_batch.hip) and the whole-chip search (lslam_icp_large.hip).
template <int DIM> struct FilterPoint;
template <> struct FilterPoint<2> { using type = float2; };
template <> struct FilterPoint<3> { using type = float4; };

constexpr int kFilterChunk = 32;
constexpr int kFilterStep = 8;                 // targets per load group

template <int DIM> struct FilterRegs;          // one load group in registers
template <> struct FilterRegs<2> { float4 v[4], w[2]; };      // 8 × (x, y) + 8 × |t|²
template <> struct FilterRegs<3> { float4 v[8]; };            // 8 × (x, y, z, |t|²)

template <int DIM>
__device__ __forceinline__ void filter_load(FilterRegs<DIM>& r, const typename FilterPoint<DIM>::type* tile, const float* wt, int step) {
  if constexpr (DIM == 2) {
    const float4* t4 = reinterpret_cast<const float4*>(tile) + step * 4;      // two points per 16 B
    const float4* w4 = reinterpret_cast<const float4*>(wt) + step * 2;        // four |t|² per 16 B
#pragma unroll
    for (int k = 0; k < 4; ++k) r.v[k] = t4[k];
#pragma unroll
    for (int k = 0; k < 2; ++k) r.w[k] = w4[k];
  } else {
    const float4* t4 = tile + step * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) r.v[k] = t4[k];
  }
}

template <int DIM, int P>
__device__ __forceinline__ void filter_eval(const FilterRegs<DIM>& r, const float (&na)[P][DIM], float (&m)[P]) {
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if constexpr (DIM == 2) {
      const float nx = na[p][0], ny = na[p][1];
      const float g0 = __builtin_fmaf(r.v[0].x, nx, __builtin_fmaf(r.v[0].y, ny, r.w[0].x));
      const float g1 = __builtin_fmaf(r.v[0].z, nx, __builtin_fmaf(r.v[0].w, ny, r.w[0].y));
      const float g2 = __builtin_fmaf(r.v[1].x, nx, __builtin_fmaf(r.v[1].y, ny, r.w[0].z));
      const float g3 = __builtin_fmaf(r.v[1].z, nx, __builtin_fmaf(r.v[1].w, ny, r.w[0].w));
      const float g4 = __builtin_fmaf(r.v[2].x, nx, __builtin_fmaf(r.v[2].y, ny, r.w[1].x));
      const float g5 = __builtin_fmaf(r.v[2].z, nx, __builtin_fmaf(r.v[2].w, ny, r.w[1].y));
      const float g6 = __builtin_fmaf(r.v[3].x, nx, __builtin_fmaf(r.v[3].y, ny, r.w[1].z));
      const float g7 = __builtin_fmaf(r.v[3].z, nx, __builtin_fmaf(r.v[3].w, ny, r.w[1].w));
      float mm = fminf(fminf(g0, g1), m[p]);
      mm = fminf(fminf(g2, g3), mm);
      mm = fminf(fminf(g4, g5), mm);
      m[p] = fminf(fminf(g6, g7), mm);
    } else {
      const float nx = na[p][0], ny = na[p][1], nz = na[p][2];
      float mm = m[p];
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        const float g0 = __builtin_fmaf(r.v[k].x, nx, __builtin_fmaf(r.v[k].y, ny, __builtin_fmaf(r.v[k].z, nz, r.v[k].w)));
        const float g1 = __builtin_fmaf(r.v[k + 1].x, nx, __builtin_fmaf(r.v[k + 1].y, ny, __builtin_fmaf(r.v[k + 1].z, nz, r.v[k + 1].w)));
        mm = fminf(fminf(g0, g1), mm);
      }
      m[p] = mm;
    }
  }
}


// planar, four targets per load group: for builds with 8 source points per lane (half the registers of a load group,
// the same 80 vector instructions per group)
struct FilterRegsPlanar4 { float4 v[2], w; };
__device__ __forceinline__ void filter_load4(FilterRegsPlanar4& r, const float2* tile, const float* wt, int step) {
  const float4* t4 = reinterpret_cast<const float4*>(tile) + step * 2;
  r.v[0] = t4[0]; r.v[1] = t4[1];
  r.w = reinterpret_cast<const float4*>(wt)[step];
}
template <int P>
__device__ __forceinline__ void filter_eval4(const FilterRegsPlanar4& r, const float (&na)[P][2], float (&m)[P]) {
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const float nx = na[p][0], ny = na[p][1];
    const float g0 = __builtin_fmaf(r.v[0].x, nx, __builtin_fmaf(r.v[0].y, ny, r.w.x));
    const float g1 = __builtin_fmaf(r.v[0].z, nx, __builtin_fmaf(r.v[0].w, ny, r.w.y));
    const float g2 = __builtin_fmaf(r.v[1].x, nx, __builtin_fmaf(r.v[1].y, ny, r.w.z));
    const float g3 = __builtin_fmaf(r.v[1].z, nx, __builtin_fmaf(r.v[1].w, ny, r.w.w));
    m[p] = fminf(fminf(g2, g3), fminf(fminf(g0, g1), m[p]));
  }
}

// The three smallest chunk minima, each carrying its chunk id — updated with chunk c's minimum mc (c < 256).
// Tracked is  s = mc + |a|²  (≈ the squared distance to the chunk's best target: small and of either sign by at most the
// filter's error E, where mc itself is ≈ −|a|²) with the chunk id written into its 8 low mantissa bits.  The id then rides
// along through v_med3 / v_min for free: 14 issue cycles per chunk and point instead of 22 for values and ids kept apart
// (2 v_cmp, 2 half-rate v_med3, 5 v_cndmask) — 11.5 % of the headline kernel's cycles went there.  What the 8 bits cost: every
// tracked value is off by < 2⁻¹⁵ of ITS OWN magnitude (|s| ≈ d², not |a|²): filter_decide adds that to its window.  The same
// a2 is added to every chunk minimum of a point, so its own rounding shifts all of them alike.
constexpr int kFilterIdBits = 8;
__device__ __forceinline__ void filter_track(float mc, float a2, int c, float& s1, float& s2, float& s3) {
  // written as instructions: from C++ the compiler splits the and-or in two and, because s comes out of integer
  // operations, guards fminf with two canonicalising v_max (8 instead of 5 instructions per point)
  float s;
  const float sum = mc + a2;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(s) : "v"(sum), "v"(~((1u << kFilterIdBits) - 1u)), "s"(c));      // one scalar operand only (constant bus)
  s3 = __builtin_amdgcn_fmed3f(s2, s3, s);                   // s2 ≤ s3 holds: the median is the new third smallest
  s2 = __builtin_amdgcn_fmed3f(s1, s2, s);
  asm("v_min_f32 %0, %1, %2" : "=v"(s1) : "v"(s1), "v"(s));
}
__device__ __forceinline__ int filter_chunk_of(float s) { return (int)(__float_as_uint(s) & ((1u << kFilterIdBits) - 1u)); }

// ---- float64-exact mode: near ties of the float32 search are decided in float64 ---------------------
// The reference searches in float64 (modules/icp.py:29-41).  When the caller also hands over the float64 clouds, the
// float32 search is only a candidate generator: with a_d = T·p_d, t_d the float64 points and a_f, t_f what the sweep
// sees, |(a_d − t_d) − (a_f − t_f)| ≤ ε = 1.001·u·(|a| + F·Rs + Rt)  (u = 2⁻²⁴, Rs / Rt = largest |p| / |t| of the
// problem, F = Frobenius norm of T's linear block ≥ its spectral norm; |a| ≤ Rt + d̂).  If ĵ is the float32 winner
// with recipe value D̂, the float64 winner j* (and every float64 tie of it) has a recipe value
//     D(j*) ≤ lim(D̂) = (√D̂·(1 + 4u) + 2ε)²·(1 + 8u)
// so only targets with D ≤ lim(D̂) are re-evaluated in float64 — usually none besides ĵ.  With the mode off lim(D) = D
// and K1 = 1, C = 0: every comparison below is then bit for bit the float32-only rule.
struct NearTie {
  float eps0;                // 1.001·u·(2 Rt + F·Rs): the part of ε that does not depend on the distance
  float K1, C;               // lim(D) ≤ K1·D + C for every D: the slack of the pruning tests (no square root there)
  bool on;
};
__device__ __forceinline__ NearTie near_tie_setup(bool on, float Rt, float Rs, float F) {
  NearTie nt;
  nt.on = on;
  nt.eps0 = 0.f; nt.K1 = 1.f; nt.C = 0.f;
  if (on) {
    constexpr float kU = 5.97e-8f;
    nt.eps0 = 1.001f * kU * (2.f * Rt + F * Rs);
    // 4 ε d ≤ κ d² + 4 ε²/κ with κ = 2⁻⁹; ε² ≤ 2 ε0² + 2.01 u² D
    nt.K1 = 1.f + 0.001953125f + 1e-5f;
    nt.C = 8.1f * 513.f * nt.eps0 * nt.eps0 + 1e-36f;
  }
  return nt;
}
__device__ __forceinline__ float near_lim(float D, const NearTie& nt) {
  if (!nt.on) return D;
  const float d = sqrtf(D) * 1.000001f;
  const float x = d + 2.f * (nt.eps0 + 6.0e-8f * d);
  return x * x * 1.000002f + 1e-37f;
}

// float64-exact mode, float64 points kept on chip at 4 bytes per coordinate beside their float32 rounding: the residual
// t64 − f64(t32) is a multiple of 2^(e − 54), e = frexp exponent of t32, and smaller than 2²⁹ of them — an int32 count from which
// f64(t32) + ldexp(count, e − 54) restores the float64 coordinate bit for bit (t32 = 0 or subnormal: to within 2⁻¹⁵⁰).
__device__ __forceinline__ int residual_count(double t64, float t32) {
  const int e = __builtin_amdgcn_frexp_expf(t32);
  return (int)rint(ldexp(t64 - (double)t32, 54 - e));
}
__device__ __forceinline__ double rebuild64(float t32, int count) {
  const int e = __builtin_amdgcn_frexp_expf(t32);
  return (double)t32 + ldexp((double)count, e - 54);
}

// After the sweep: does the second chunk have to be re-evaluated (`two`), and is a third chunk (or an overflow risk)
// in play so that the exact sweep must run instead (`third`)?  Rt ≥ the largest |t| swept.  Every comparison is
// written so that a NaN sends the caller to the exact sweep.  In float64-exact mode the chunks kept must hold every
// target within lim(D̂) of the winner, not only the winner: τ grows by lim(d²) − d² (+ 2u·d² for the roundings of lim's
// argument), see DESIGN.md §3.1d.
template <int DIM>
__device__ __forceinline__ void filter_decide(const float (&a)[DIM], float Rt, float s1, float s2, float s3, const NearTie& nt, bool& two, bool& third) {
  constexpr float kU = 5.97e-8f;                             // > 2⁻²⁴
  constexpr float kE = (DIM == 2 ? 4.5f : 6.5f) * kU;
  // the direct recipe rounds the x term four times in the plane (|θ| ≤ 4.01u), five times in space (5.01u): 2θ + margins
  constexpr float kD = (DIM == 2 ? 9.f : 11.f) * kU;
  constexpr float kId = 3.1e-5f;                             // > 2⁻¹⁵: what the chunk id in the low mantissa bits moves a tracked value by (relative)
  float a2;
  if constexpr (DIM == 2) a2 = __builtin_fmaf(a[1], a[1], a[0] * a[0]);
  else a2 = __builtin_fmaf(a[2], a[2], __builtin_fmaf(a[1], a[1], a[0] * a[0]));
  const float an = sqrtf(a2) * 1.0001f;
  const float E = kE * Rt * (Rt + an);
  // s1 ≈ (smallest filter value) + |a|²: ≥ |a − t|² of the filter's winner once the slack terms are added
  const float d2 = fmaxf(s1 * (1.f + 2.f * kId) + fabsf(s1) * kId + E + 3.f * kU * a2, 0.f);
  float tau = (2.f * E + kD * d2) * 1.001f + 1e-30f;
  if (nt.on) tau += (near_lim(d2, nt) - d2) * 1.0001f + 2.f * kU * d2;
  // both sides of the comparison carry an id: each is off by ≤ kId of its own magnitude, and everything compared lies within
  // |s1| + tau of zero
  const float lim = s1 + tau + 2.f * kId * (fabsf(s1) + tau);
  two = !(s2 > lim);
  third = !(s3 > lim) || !(Rt < 1e15f) || !(an < 1e15f);
}

// Position of grid cell c[0..DIM) (BITS bits per axis) along the DIM-dimensional Hilbert curve, DIM·BITS bits (Skilling's
// transpose form, then the bits of the axes interleaved, axis 0 leading).  Consecutive cells share a face: a run of
// consecutive points never bridges two distant parts of the cloud the way it does where the Z curve jumps.
template <int DIM, int BITS>
__device__ __forceinline__ unsigned hilbert_key(const unsigned (&c)[DIM]) {
  unsigned X[DIM];
#pragma unroll
  for (int i = 0; i < DIM; ++i) X[i] = c[i];
  constexpr unsigned M = 1u << (BITS - 1);
  for (unsigned Q = M; Q > 1; Q >>= 1) {                       // inverse undo
    const unsigned Pm = Q - 1;
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
      if (X[i] & Q) X[0] ^= Pm;
      else { const unsigned t = (X[0] ^ X[i]) & Pm; X[0] ^= t; X[i] ^= t; }
    }
  }
#pragma unroll
  for (int i = 1; i < DIM; ++i) X[i] ^= X[i - 1];              // Gray encode
  unsigned t = 0;
  for (unsigned Q = M; Q > 1; Q >>= 1)
    if (X[DIM - 1] & Q) t ^= Q - 1;
#pragma unroll
  for (int i = 0; i < DIM; ++i) X[i] ^= t;
  unsigned key = 0;
  for (int b = BITS - 1; b >= 0; --b)
#pragma unroll
    for (int i = 0; i < DIM; ++i) key = (key << 1) | ((X[i] >> b) & 1u);
  return key;
}

// Accumulator slots of one (p, q) pair stream, in the order they are reduced.
//   planar: n, Σp(2), Σq(2), Σpqᵀ(4), sse            = 10   (+4 bbox when normalising)
//   3-D   : n, Σp(3), Σq(3), Σpqᵀ(9), sse            = 17   (+6 bbox)
template <int DIM> struct AccLayout {
  static constexpr int kSums = 1 + DIM + DIM + DIM * DIM + 1;
  static constexpr int kBox = 2 * DIM;            // lo[DIM], hi[DIM]
  static constexpr int kTotal = kSums + kBox;
};

template <int DIM>
struct PairAcc {
  double s[AccLayout<DIM>::kSums];
  double lo[DIM], hi[DIM];

  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int i = 0; i < AccLayout<DIM>::kSums; ++i) s[i] = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) { lo[i] = INFINITY; hi[i] = -INFINITY; }
  }
  // p, q in float64
  __device__ __forceinline__ void add(const double (&p)[DIM], const double (&q)[DIM], bool box) {
    s[0] += 1.0;
    double e = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
      s[1 + i] += p[i];
      s[1 + DIM + i] += q[i];
#pragma unroll
      for (int j = 0; j < DIM; ++j) s[1 + 2 * DIM + DIM * i + j] += p[i] * q[j];
      const double d = p[i] - q[i];
      e += d * d;
    }
    s[1 + 2 * DIM + DIM * DIM] += e;
    if (box) {
#pragma unroll
      for (int i = 0; i < DIM; ++i) {
        lo[i] = fmin(lo[i], fmin(p[i], q[i]));
        hi[i] = fmax(hi[i], fmax(p[i], q[i]));
      }
    }
  }
  __device__ __forceinline__ void wave_reduce(bool box) {
#pragma unroll
    for (int i = 0; i < AccLayout<DIM>::kSums; ++i) s[i] = wave_sum(s[i]);
    if (box) {
#pragma unroll
      for (int i = 0; i < DIM; ++i) { lo[i] = wave_min(lo[i]); hi[i] = wave_max(hi[i]); }
    }
  }
  __device__ __forceinline__ void store(double* dst) const {   // kTotal doubles
#pragma unroll
    for (int i = 0; i < AccLayout<DIM>::kSums; ++i) dst[i] = s[i];
#pragma unroll
    for (int i = 0; i < DIM; ++i) { dst[AccLayout<DIM>::kSums + i] = lo[i]; dst[AccLayout<DIM>::kSums + DIM + i] = hi[i]; }
  }
};

// Combine `parts` partial accumulator rows (each AccLayout<DIM>::kTotal doubles, fixed order →
// bitwise reproducible) into Moments.
template <int DIM>
__device__ inline void combine_partials(const double* rows, int parts, int row_stride, bool box, Moments& m) {
  constexpr int kS = AccLayout<DIM>::kSums;
  double s[kS];
  double lo[DIM], hi[DIM];
  for (int i = 0; i < kS; ++i) s[i] = 0.0;
  for (int i = 0; i < DIM; ++i) { lo[i] = INFINITY; hi[i] = -INFINITY; }
  for (int r = 0; r < parts; ++r) {
    const double* row = rows + (size_t)r * row_stride;
    for (int i = 0; i < kS; ++i) s[i] += row[i];
    if (box)
      for (int i = 0; i < DIM; ++i) { lo[i] = fmin(lo[i], row[kS + i]); hi[i] = fmax(hi[i], row[kS + DIM + i]); }
  }
  m.n = s[0];
  for (int i = 0; i < 3; ++i) { m.sp[i] = 0.0; m.sq[i] = 0.0; m.lo[i] = 0.0; m.hi[i] = 0.0; }
  for (int i = 0; i < 9; ++i) m.spq[i] = 0.0;
  for (int i = 0; i < DIM; ++i) {
    m.sp[i] = s[1 + i];
    m.sq[i] = s[1 + DIM + i];
    for (int j = 0; j < DIM; ++j) m.spq[3 * i + j] = s[1 + 2 * DIM + DIM * i + j];
    if (box) { m.lo[i] = lo[i]; m.hi[i] = hi[i]; }
  }
  m.sse = s[1 + 2 * DIM + DIM * DIM];
}

}  // namespace lslam

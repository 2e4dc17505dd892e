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

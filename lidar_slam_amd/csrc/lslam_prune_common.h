// lslam_prune_common.h — what the two batched planar kernels with exact chunk pruning share (lslam_icp_pruned.hip: one
// problem per workgroup; lslam_icp_queue.hip: problems in LDS slots, waves take tasks from a queue).
#pragma once
#include <hip/hip_runtime.h>

#include "lslam_device.h"
#include "lslam_kernels.h"

namespace lslam {

#ifndef LSLAM_PRUNE_CHUNK
#define LSLAM_PRUNE_CHUNK 16
#endif
constexpr int kPruneChunk = LSLAM_PRUNE_CHUNK;   // targets per box (8 was measured: see DESIGN §3.1b)
constexpr float kPruneSafety = 0.99999f;

// the next float above a non-negative finite float (or +inf)
__device__ __forceinline__ float next_up(float v) {
  return v < INFINITY ? __uint_as_float(__float_as_uint(v) + 1u) : v;
}

__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- wave reductions through DPP (no LDS round trip per step; the ds_bpermute butterfly costs six of them in a row) ----
// Steps: xor 1, xor 2 (quad_perm), row_half_mirror (quad ↔ neighbouring quad), row_mirror (half row ↔ half row): every lane of a
// row of 16 then holds the row's value; row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3: lane 63 holds the wave's.
// The pairing is that of the xor butterfly taken in the order 1, 2, 4, 8, 16, 32, so a float64 sum is bit-identical to it
// (tools/dpp_probe.hip).
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140, kDppBcast15 = 0x142, kDppBcast31 = 0x143;
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_f(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROWS, 0xf, false));
}
__device__ __forceinline__ float wave_max_dpp(float v) {      // the maximum over the wave, in every lane (wave-uniform)
  v = fmaxf(v, dpp_f<kDppXor1, 0xf>(v, v));
  v = fmaxf(v, dpp_f<kDppXor2, 0xf>(v, v));
  v = fmaxf(v, dpp_f<kDppHalfMirror, 0xf>(v, v));
  v = fmaxf(v, dpp_f<kDppMirror, 0xf>(v, v));
  v = fmaxf(v, dpp_f<kDppBcast15, 0xa>(v, v));
  v = fmaxf(v, dpp_f<kDppBcast31, 0xc>(v, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_min_dpp(float v) {
  v = fminf(v, dpp_f<kDppXor1, 0xf>(v, v));
  v = fminf(v, dpp_f<kDppXor2, 0xf>(v, v));
  v = fminf(v, dpp_f<kDppHalfMirror, 0xf>(v, v));
  v = fminf(v, dpp_f<kDppMirror, 0xf>(v, v));
  v = fminf(v, dpp_f<kDppBcast15, 0xa>(v, v));
  v = fminf(v, dpp_f<kDppBcast31, 0xc>(v, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_d(double v) {            // lanes outside ROWS receive 0.0
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp_last(double v) {      // the sum over the wave, valid in lane 63 only
  v += dpp_d<kDppXor1, 0xf>(v);
  v += dpp_d<kDppXor2, 0xf>(v);
  v += dpp_d<kDppHalfMirror, 0xf>(v);
  v += dpp_d<kDppMirror, 0xf>(v);
  v += dpp_d<kDppBcast15, 0xa>(v);
  v += dpp_d<kDppBcast31, 0xc>(v);
  return v;
}
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_d_old(double old, double v) {      // lanes outside ROWS receive `old`
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROWS, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROWS, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_min_dpp_last(double v) {
  v = fmin(v, dpp_d_old<kDppXor1, 0xf>(v, v));
  v = fmin(v, dpp_d_old<kDppXor2, 0xf>(v, v));
  v = fmin(v, dpp_d_old<kDppHalfMirror, 0xf>(v, v));
  v = fmin(v, dpp_d_old<kDppMirror, 0xf>(v, v));
  v = fmin(v, dpp_d_old<kDppBcast15, 0xa>(v, v));
  v = fmin(v, dpp_d_old<kDppBcast31, 0xc>(v, v));
  return v;
}
__device__ __forceinline__ double wave_max_dpp_last(double v) {
  v = fmax(v, dpp_d_old<kDppXor1, 0xf>(v, v));
  v = fmax(v, dpp_d_old<kDppXor2, 0xf>(v, v));
  v = fmax(v, dpp_d_old<kDppHalfMirror, 0xf>(v, v));
  v = fmax(v, dpp_d_old<kDppMirror, 0xf>(v, v));
  v = fmax(v, dpp_d_old<kDppBcast15, 0xa>(v, v));
  v = fmax(v, dpp_d_old<kDppBcast31, 0xc>(v, v));
  return v;
}

// One chunk of 16 sorted targets against ONE point of this lane, the chunk chosen by the lane (a gather): every target is
// compared, (bd, sec) = the smallest and second smallest distance over everything evaluated for the point so far, at = the LDS
// address of the nearest target.  Lane l visits target (k XOR l) at step k: chunks start on the same banks, and the XOR
// costs one instruction per address (the arithmetic is done on the 32-bit LDS address: see lslam_icp_batch.hip, exact_chunk).
constexpr int kPruneTileAlign = kPruneChunk * 8;
__device__ __forceinline__ void prune_eval_chunk(unsigned tile0, int chunk, int lane, float x, float y, float& bd, float& sec, unsigned& at) {
  typedef float __attribute__((ext_vector_type(2))) Vec;
  typedef const __attribute__((address_space(3))) Vec* LdsPoint;
  unsigned r = (unsigned)lane & (unsigned)(kPruneChunk - 1);
  asm volatile("" : "+v"(r));                                  // opaque: keeps the 16 offsets from being hoisted out of the ICP loop (and spilled)
  const unsigned first = tile0 + (unsigned)chunk * (kPruneChunk * 8u) + r * 8u;
#pragma unroll 1
  for (int h = 0; h < kPruneChunk; h += 8) {
    unsigned a8[8];
    Vec v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { a8[c] = first ^ ((unsigned)(h + c) * 8u); v[c] = *(LdsPoint)(size_t)a8[c]; }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float d = dist2(x, y, v[c].x, v[c].y);
      const bool lt = d < bd;
      sec = __builtin_amdgcn_fmed3f(bd, sec, d);
      at = lt ? a8[c] : at;
      bd = lt ? d : bd;
    }
  }
}


// The same comparison with all 16 reads of the chunk in flight before the first distance is evaluated (the builds with 128
// VGPRs: four waves per SIMD do not hide an LDS round trip per group of three reads, which is how the compiler schedules the
// loop above).  The position is tracked as the step's XOR offset (an inline constant), not as an address register per read.
__device__ __forceinline__ void prune_eval_chunk_wide(unsigned tile0, int chunk, int lane, float x, float y, float& bd, float& sec, unsigned& at) {
  typedef float __attribute__((ext_vector_type(2))) Vec;
  typedef const __attribute__((address_space(3))) Vec* LdsPoint;
  unsigned r = (unsigned)lane & (unsigned)(kPruneChunk - 1);
  asm volatile("" : "+v"(r));
  const unsigned first = tile0 + (unsigned)chunk * (kPruneChunk * 8u) + r * 8u;
  Vec v[kPruneChunk];
#pragma unroll
  for (int c = 0; c < kPruneChunk; ++c) v[c] = *(LdsPoint)(size_t)(first ^ ((unsigned)c * 8u));
  __builtin_amdgcn_sched_barrier(0);
  unsigned kb = 0xffffffffu;
#pragma unroll
  for (int c = 0; c < kPruneChunk; ++c) {
    const float d = dist2(x, y, v[c].x, v[c].y);
    const bool lt = d < bd;
    sec = __builtin_amdgcn_fmed3f(bd, sec, d);
    kb = lt ? (unsigned)c * 8u : kb;
    bd = lt ? d : bd;
  }
  at = kb != 0xffffffffu ? first ^ kb : at;
}

}  // namespace lslam

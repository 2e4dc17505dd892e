// lslam_icp_large_pruned.hip — whole-chip ICP on one large pair with EXACT box pruning (opt-in:
// LSLAM_ICP_PRUNE), planar or 3-D.
//
// Same operator and same answers as the LDS-tiled all-pairs path of lslam_icp_large.hip (exact
// float32 nearest target, lowest original index on ties), but a 100k × 100k iteration no longer
// costs 10¹⁰ pair evaluations:
//   once per problem   both clouds are put into Morton order of a uniform grid over the target's box
//                      (rocPRIM radix sort of (cell key, index)), and a 3-level box hierarchy is built
//                      over the sorted target: chunks of 8/16 points → 64 chunks → 64 × 64 chunks;
//   per iteration      nn_pruned_kernel: every wave owns 64·P consecutive sorted sources, reduces the
//                      box of its transformed points and descends the hierarchy with wave-uniform
//                      (scalar) loads, evaluating a chunk only if its box is not farther than every
//                      lane's current best: the 64 lanes test 64 sibling boxes at once (one coalesced
//                      load + ballot), so a descent is a few dozen dependent loads, not hundreds.  The
//                      bound starts from the previous iteration's match.  No LDS, no barriers.
//                      moments_sorted_kernel / solve_kernel: as in the all-pairs path.
// Rounding cannot break the bound (float32 subtraction, product and fma are monotone, and an extra
// factor 1 − 1e-5 is applied); the test is non-strict, so a skipped box holds neither the argmin nor a
// tie.  Every evaluation keeps the lexicographic minimum of (distance, original index), so ties need no
// second pass; sibling boxes are visited nearest-first so that the bound tightens as early as possible.
#include <hip/hip_runtime.h>

#include <cstring>
#include <cstdlib>

#include <rocprim/device/device_radix_sort.hpp>

#include "lslam_device.h"
#include "lslam_kernels.h"

namespace lslam {

constexpr int kLpThreads = 256;
constexpr int kLpFan = 64;                // children per box at levels 2 and 3
constexpr float kLpSafety = 0.99999f;
constexpr int kLpAxisBits = 10;

template <int DIM> struct SortedPoint;
template <> struct SortedPoint<2> { using type = float2; };
template <> struct SortedPoint<3> { using type = float4; };

__device__ __forceinline__ float lp_wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float lp_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ unsigned spread10(unsigned v, int dim) {        // bit i of v → bit dim·i
  unsigned r = 0;
#pragma unroll
  for (int b = 0; b < kLpAxisBits; ++b) r |= ((v >> b) & 1u) << (dim * b);
  return r;
}

// ---- bounding box of the target (grid frame of the Morton keys) ----------------------------------------
template <int DIM>
__global__ void __launch_bounds__(kLpThreads)
lp_bbox_kernel(const float* __restrict__ pts, int64_t n, float* __restrict__ partial /*grid x 6*/) {
  __shared__ float sh[kLpThreads / 64][6];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = (int64_t)blockIdx.x * kLpThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLpThreads)
    for (int d = 0; d < DIM; ++d) { const float v = pts[i * DIM + d]; lo[d] = fminf(lo[d], v); hi[d] = fmaxf(hi[d], v); }
  for (int d = 0; d < 3; ++d) { lo[d] = lp_wave_min(lo[d]); hi[d] = lp_wave_max(hi[d]); }
  if ((threadIdx.x & 63) == 0) for (int d = 0; d < 3; ++d) { sh[threadIdx.x >> 6][d] = lo[d]; sh[threadIdx.x >> 6][3 + d] = hi[d]; }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = sh[0][threadIdx.x];
    for (int w = 1; w < kLpThreads / 64; ++w) v = threadIdx.x < 3 ? fminf(v, sh[w][threadIdx.x]) : fmaxf(v, sh[w][threadIdx.x]);
    partial[6 * blockIdx.x + threadIdx.x] = v;
  }
}
__global__ void lp_bbox_finish_kernel(const float* __restrict__ partial, int parts, float* __restrict__ grid /*lo[3], scale[3]*/) {
  if (threadIdx.x >= 3) return;
  float lo = INFINITY, hi = -INFINITY;
  for (int i = 0; i < parts; ++i) { lo = fminf(lo, partial[6 * i + threadIdx.x]); hi = fmaxf(hi, partial[6 * i + 3 + threadIdx.x]); }
  grid[threadIdx.x] = lo;
  grid[3 + threadIdx.x] = (hi > lo) ? (float)(1 << kLpAxisBits) / (hi - lo) : 0.f;
}

// key = Morton code of the grid cell of T·p (T = identity for the target, the initial guess for the source)
template <int DIM>
__global__ void __launch_bounds__(kLpThreads)
lp_keys_kernel(const float* __restrict__ pts, int64_t n, const float* __restrict__ grid, LpT T, unsigned* __restrict__ keys,
               unsigned* __restrict__ vals, int curve) {
  const int64_t i = (int64_t)blockIdx.x * kLpThreads + threadIdx.x;
  if (i >= n) return;
  float p[3] = {pts[i * DIM], pts[i * DIM + 1], DIM == 3 ? pts[i * DIM + 2] : 0.f};
  unsigned key = 0, cell[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    const float v = T.m[d][0] * p[0] + T.m[d][1] * p[1] + T.m[d][2] * p[2] + T.m[d][3];
    int c = (int)((v - grid[d]) * grid[3 + d]);
    c = min(max(c, 0), (1 << kLpAxisBits) - 1);
    cell[d] = (unsigned)c;
    key |= spread10((unsigned)c, DIM) << d;
  }
  if (curve == 1) key = hilbert_key<DIM, kLpAxisBits>(cell);   // the cells along the Hilbert curve instead of the Z curve
  keys[i] = key;
  vals[i] = (unsigned)i;
}

template <int DIM>
__global__ void __launch_bounds__(kLpThreads)
lp_gather_kernel(const float* __restrict__ pts, const unsigned* __restrict__ order, int64_t n,
                 typename SortedPoint<DIM>::type* __restrict__ out, int32_t* __restrict__ out_idx) {
  const int64_t i = (int64_t)blockIdx.x * kLpThreads + threadIdx.x;
  if (i >= n) return;
  const int64_t o = order[i];
  if constexpr (DIM == 2) out[i] = make_float2(pts[2 * o], pts[2 * o + 1]);
  else out[i] = make_float4(pts[3 * o], pts[3 * o + 1], pts[3 * o + 2], 0.f);
  out_idx[i] = (int32_t)o;
}

// level-1 boxes over CH consecutive sorted points; upper levels over kLpFan children
template <int DIM>
__global__ void __launch_bounds__(kLpThreads)
lp_boxes1_kernel(const typename SortedPoint<DIM>::type* __restrict__ pts, int64_t m, int64_t nboxes, float4* __restrict__ lo, float4* __restrict__ hi) {
  constexpr int CH = NnChunk<DIM>::value;
  const int64_t c = (int64_t)blockIdx.x * kLpThreads + threadIdx.x;
  if (c >= nboxes) return;
  float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = 0; k < CH && c * CH + k < m; ++k) {
    const auto q = pts[c * CH + k];
    l[0] = fminf(l[0], q.x); h[0] = fmaxf(h[0], q.x); l[1] = fminf(l[1], q.y); h[1] = fmaxf(h[1], q.y);
    if constexpr (DIM == 3) { l[2] = fminf(l[2], q.z); h[2] = fmaxf(h[2], q.z); }
  }
  if (DIM == 2) { l[2] = 0.f; h[2] = 0.f; }
  lo[c] = make_float4(l[0], l[1], l[2], 0.f);
  hi[c] = make_float4(h[0], h[1], h[2], 0.f);
}
__global__ void __launch_bounds__(kLpThreads)
lp_boxes_up_kernel(const float4* __restrict__ clo, const float4* __restrict__ chi, int64_t nchild, int64_t nboxes,
                   float4* __restrict__ lo, float4* __restrict__ hi) {
  const int64_t b = (int64_t)blockIdx.x * kLpThreads + threadIdx.x;
  if (b >= nboxes) return;
  float4 l = make_float4(INFINITY, INFINITY, INFINITY, 0.f), h = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.f);
  for (int k = 0; k < kLpFan && b * kLpFan + k < nchild; ++k) {
    const float4 a = clo[b * kLpFan + k], z = chi[b * kLpFan + k];
    l.x = fminf(l.x, a.x); l.y = fminf(l.y, a.y); l.z = fminf(l.z, a.z);
    h.x = fmaxf(h.x, z.x); h.y = fmaxf(h.y, z.y); h.z = fmaxf(h.z, z.z);
  }
  lo[b] = l; hi[b] = h;
}

__global__ void lp_fill_kernel(int32_t* p, int64_t n, int32_t v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---- the search --------------------------------------------------------------------------------------
template <int DIM>
__device__ __forceinline__ float lp_dist(const float (&a)[DIM], float qx, float qy, float qz) {
  if constexpr (DIM == 2) return dist2(a[0], a[1], qx, qy);
  else return dist2(a[0], a[1], a[2], qx, qy, qz);
}

// value held by lane k (k wave-uniform): v_readlane, no LDS round trip
__device__ __forceinline__ float lane_value(float v, int k) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_amdgcn_readfirstlane(k)));
}

// F64: float64-exact mode (lslam_device.h: NearTie) — every box test compares against reach(best) = the near-tie window, the second
// smallest distance of the evaluated targets is tracked, and a source whose second is within lim(best) goes on a list that
// lp_refine_kernel decides on the float64 points
template <int DIM, int P, bool F64>
__global__ void __launch_bounds__(kLpThreads)
nn_pruned_kernel(LpNnArgs a) {
  constexpr int CH = NnChunk<DIM>::value;
  using SP = typename SortedPoint<DIM>::type;
  if (a.state->done) return;
  const SP* __restrict__ src = reinterpret_cast<const SP*>(a.src_sorted);
  const SP* __restrict__ tgt = reinterpret_cast<const SP*>(a.tgt_sorted);
  const int lane = threadIdx.x & 63;
  const int64_t wave_first = ((int64_t)blockIdx.x * kLpThreads + (threadIdx.x & ~63)) * P;
  if (wave_first >= a.n) return;                            // waves are independent: no barrier below

  double T[3][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) T[r][c] = wave_uniform(a.state->T[4 * r + c]);

  NearTie nt = near_tie_setup(false, 0.f, 0.f, 0.f);
  if constexpr (F64) {
    double f2 = 0.0;
#pragma unroll
    for (int r = 0; r < DIM; ++r)
#pragma unroll
      for (int c = 0; c < DIM; ++c) f2 += T[r][c] * T[r][c];
    nt = near_tie_setup(true, sqrtf(a.bounds[0]) * 1.0001f, sqrtf(a.bounds[1]) * 1.0001f, sqrtf((float)f2) * 1.0001f);
  }
  auto reach = [&](float b) { return F64 ? __builtin_fmaf(b, nt.K1, nt.C) : b; };
  float s[P][DIM];
  float second[P];                                          // F64: second smallest distance among the targets evaluated (not the winner itself)
  float wlo[3] = {INFINITY, INFINITY, INFINITY}, whi[3] = {-INFINITY, -INFINITY, -INFINITY};
  // per source: the best candidate so far as the lexicographic minimum of (distance, ORIGINAL index) —
  // kept exactly at every evaluation, so ties need no second pass
  float best[P];
  int bidx[P];
  int64_t bpos[P];
  bool unseeded = false;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int64_t i = ((int64_t)blockIdx.x * kLpThreads + threadIdx.x) * P + p;
    best[p] = -INFINITY;                                    // lanes without a point never ask for a box
    second[p] = INFINITY;
    bidx[p] = 0x7fffffff;
    bpos[p] = 0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) s[p][d] = 0.f;
    if (i < a.n) {
      const SP v = src[i];
      const double x = v.x, y = v.y;
      double z = 0.0;
      if constexpr (DIM == 3) z = v.z;
#pragma unroll
      for (int r = 0; r < DIM; ++r) {                       // icp.py:116
        double t = T[r][0] * x + T[r][1] * y + T[r][3];
        if constexpr (DIM == 3) t += T[r][2] * z;
        s[p][r] = (float)t;
        wlo[r] = fminf(wlo[r], s[p][r]); whi[r] = fmaxf(whi[r], s[p][r]);
      }
      best[p] = INFINITY;
      const int64_t pv = a.match[i];
      unseeded |= pv < 0;
      if (pv >= 0) {                                        // last iteration's match: a tight bound from the start
        const SP q = tgt[pv];
        float qz = 0.f;
        if constexpr (DIM == 3) qz = q.z;
        best[p] = lp_dist<DIM>(s[p], q.x, q.y, qz);
        bidx[p] = a.tgt_index[pv];
        bpos[p] = pv;
      }
    }
  }
#pragma unroll
  for (int d = 0; d < DIM; ++d) { wlo[d] = lp_wave_min(wlo[d]); whi[d] = lp_wave_max(whi[d]); }

  // Lower bound of the distance² from any point of this wave to a box (conservative, see header).
  auto box_bound = [&](const float4 lo, const float4 hi) -> float {
    const float dx = fmaxf(fmaxf(lo.x - whi[0], wlo[0] - hi.x), 0.f);
    const float dy = fmaxf(fmaxf(lo.y - whi[1], wlo[1] - hi.y), 0.f);
    float lb = __builtin_fmaf(dy, dy, dx * dx);
    if constexpr (DIM == 3) {
      const float dz = fmaxf(fmaxf(lo.z - whi[2], wlo[2] - hi.z), 0.f);
      lb = __builtin_fmaf(dz, dz, lb);
    }
    return lb * kLpSafety;
  };
  // The 64 lanes test 64 sibling boxes at once (one coalesced load); a box matters iff its bound does
  // not exceed the largest current best of the wave.
  auto wave_best = [&]() -> float {
    float w = reach(best[0]);
#pragma unroll
    for (int p = 1; p < P; ++p) w = fmaxf(w, reach(best[p]));
    return lp_wave_max(w);
  };
  // stage 1 — the 64 lanes test 64 sibling boxes at once against the WAVE's box (one coalesced load)
  auto test64 = [&](const float4* __restrict__ lo, const float4* __restrict__ hi, int64_t first, int64_t total, float wmax,
                    float& my_lb, float4& my_lo, float4& my_hi) -> unsigned long long {
    my_lb = INFINITY;
    my_lo = make_float4(0.f, 0.f, 0.f, 0.f);
    my_hi = my_lo;
    const int64_t b = first + lane;
    const bool real = b < total;
    if (real) { my_lo = lo[b]; my_hi = hi[b]; my_lb = box_bound(my_lo, my_hi); }
    return __ballot(real && my_lb <= wmax);                  // (wmax is +inf while a lane has no candidate yet)
  };
  // stage 2 — one candidate box (held by lane k) against every lane's OWN points and OWN best: the
  // 64·P sources of a wave are consecutive on a Morton curve, which now and then jumps across the
  // cloud; the wave's box is then huge, but no single point is near the boxes in between
  auto any_lane_needs = [&](int k, const float4 my_lo, const float4 my_hi) -> bool {
    const float lx = lane_value(my_lo.x, k), ly = lane_value(my_lo.y, k), hx = lane_value(my_hi.x, k), hy = lane_value(my_hi.y, k);
    float lz = 0.f, hz = 0.f;
    if constexpr (DIM == 3) { lz = lane_value(my_lo.z, k); hz = lane_value(my_hi.z, k); }
    bool need = false;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float dx = fmaxf(fmaxf(lx - s[p][0], s[p][0] - hx), 0.f);
      const float dy = fmaxf(fmaxf(ly - s[p][1], s[p][1] - hy), 0.f);
      float lb = __builtin_fmaf(dy, dy, dx * dx);
      if constexpr (DIM == 3) {
        const float dz = fmaxf(fmaxf(lz - s[p][2], s[p][2] - hz), 0.f);
        lb = __builtin_fmaf(dz, dz, lb);
      }
      need |= reach(best[p]) >= lb * kLpSafety;
    }
    return __any(need) != 0;
  };

  // nearest box first: among the lanes still in `mask`, the one with the smallest bound (ties: lowest lane)
  const bool by_distance = __any(unseeded) != 0;            // first iteration: no bound yet, order matters
  auto pick_nearest = [&](unsigned long long mask, float my_lb) -> int {
    if (!by_distance) return __ffsll((long long)mask) - 1;  // seeded: the bound is tight already, take index order
    unsigned long long key = ((mask >> lane) & 1ull) ? (((unsigned long long)__float_as_uint(my_lb)) << 6) | (unsigned)lane : ~0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(key, o, 64);
      key = other < key ? other : key;
    }
    return (int)(key & 63ull);
  };
  // Accepted chunks are evaluated kBatch at a time: the 64 lanes fetch kBatch·CH target points (and
  // their original indices) with ONE round trip, then every point is broadcast from its lane with
  // v_readlane and compared by all lanes.  One memory latency per 64 candidate points instead of one
  // per point.
  constexpr int kBatch = 64 / CH;
  int64_t pend[kBatch];
  int npend = 0;
  auto flush = [&]() -> bool {                              // true if any lane improved
    if (npend == 0) return false;
    const int slot = lane / CH, kin = lane % CH;
    int64_t c_mine = pend[0];
#pragma unroll
    for (int b = 1; b < kBatch; ++b) c_mine = (slot == b) ? pend[b] : c_mine;
    const int64_t j_mine = c_mine * CH + kin;
    SP q_mine;
    q_mine.x = kFar; q_mine.y = kFar;
    if constexpr (DIM == 3) { q_mine.z = kFar; q_mine.w = 0.f; }
    int o_mine = 0x7fffffff;
    long long jj_mine = 0;
    if (slot < npend && j_mine < a.m) { q_mine = tgt[j_mine]; o_mine = a.tgt_index[j_mine]; jj_mine = j_mine; }
    bool improved = false;
    const int total = npend * CH;
#pragma unroll
    for (int l = 0; l < 64; ++l) {
      if (l >= total) break;                                // wave-uniform
      const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q_mine.x), l));
      const float qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q_mine.y), l));
      float qz = 0.f;
      if constexpr (DIM == 3) qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q_mine.z), l));
      const int o = __builtin_amdgcn_readlane(o_mine, l);
      const int jlo = __builtin_amdgcn_readlane((int)(jj_mine & 0xffffffffll), l);
      const int jhi = __builtin_amdgcn_readlane((int)(jj_mine >> 32), l);
      const int64_t j = ((int64_t)jhi << 32) | (uint32_t)jlo;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const float d = lp_dist<DIM>(s[p], qx, qy, qz);
        const bool better = (d < best[p]) || (d == best[p] && o < bidx[p]);
        if constexpr (F64) {                                // the winner's own re-evaluation (the seed is met again in its chunk) does not count
          const float s2 = __builtin_amdgcn_fmed3f(best[p], second[p], d);
          second[p] = (o == bidx[p]) ? second[p] : s2;
        }
        improved |= better;
        best[p] = better ? d : best[p];
        bidx[p] = better ? o : bidx[p];
        bpos[p] = better ? j : bpos[p];
      }
    }
    npend = 0;
    return __any(improved) != 0;
  };

  float wmax = wave_best();
  for (int64_t f3 = 0; f3 < a.n3; f3 += 64) {
    float lb3; float4 lo3, hi3;
    unsigned long long m3 = test64(a.lo3, a.hi3, f3, a.n3, wmax, lb3, lo3, hi3);
    while (m3) {
      const int k3 = pick_nearest(m3, lb3);
      m3 &= ~(1ull << k3);
      if (lane_value(lb3, k3) > wmax) continue;             // the bound has tightened since the ballot
      if (!any_lane_needs(k3, lo3, hi3)) continue;
      const int64_t b3 = f3 + k3;
      float lb2; float4 lo2, hi2;
      unsigned long long m2 = test64(a.lo2, a.hi2, b3 * kLpFan, min(a.n2, (b3 + 1) * kLpFan), wmax, lb2, lo2, hi2);
      while (m2) {
        const int k2 = pick_nearest(m2, lb2);
        m2 &= ~(1ull << k2);
        if (lane_value(lb2, k2) > wmax) continue;
        if (!any_lane_needs(k2, lo2, hi2)) continue;
        const int64_t b2 = b3 * kLpFan + k2;
        float lb1; float4 lo1, hi1;
        unsigned long long m1 = test64(a.lo1, a.hi1, b2 * kLpFan, min(a.n1, (b2 + 1) * kLpFan), wmax, lb1, lo1, hi1);
        while (m1) {
          const int k1 = pick_nearest(m1, lb1);
          m1 &= ~(1ull << k1);
          if (lane_value(lb1, k1) > wmax) continue;
          if (!any_lane_needs(k1, lo1, hi1)) continue;
          // (static slot index: the compiler keeps pend[] in scalar registers)
#pragma unroll
          for (int b = 0; b < kBatch; ++b) if (b == npend) pend[b] = b2 * kLpFan + k1;
          if (++npend == kBatch && flush()) wmax = wave_best();
        }
      }
    }
  }
  if (flush()) wmax = wave_best();

#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int64_t i = ((int64_t)blockIdx.x * kLpThreads + threadIdx.x) * P + p;
    if (i < a.n) {
      a.match[i] = (int32_t)bpos[p];
      if constexpr (F64) {
        const float lim = near_lim(best[p], nt);
        if (second[p] <= lim) {                             // a near tie: decided on the float64 points by lp_refine_kernel
          const int e = atomicAdd(a.flag_count, 1);
          a.flag_src[e] = (int32_t)i;
          a.flag_lim[e] = lim;
        }
      }
    }
  }
}

// float64-exact mode: one workgroup per flagged source scans the whole sorted target; every target whose float32 distance is
// within the window is evaluated in float64 on the float64 points; the nearest wins, lowest ORIGINAL index on float64 ties
template <int DIM>
__global__ void __launch_bounds__(kLpThreads)
lp_refine_kernel(LpRefineArgs a) {
  using SP = typename SortedPoint<DIM>::type;
  __shared__ unsigned long long s_best, s_key;
  if (a.state->done) return;
  const SP* __restrict__ src = reinterpret_cast<const SP*>(a.src_sorted);
  const SP* __restrict__ tgt = reinterpret_cast<const SP*>(a.tgt_sorted);
  const int count = *a.flag_count;
  double T[3][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) T[r][c] = a.state->T[4 * r + c];
  for (int e = blockIdx.x; e < count; e += gridDim.x) {
    const int i = a.flag_src[e];
    const float lim = a.flag_lim[e];
    const SP v = src[i];
    const int so = a.src_index[i];
    float af[DIM];
    double ad[DIM];
    {
      const double x = v.x, y = v.y;
      double z = 0.0;
      if constexpr (DIM == 3) z = v.z;
      const double xd = a.src64[(int64_t)DIM * so], yd = a.src64[(int64_t)DIM * so + 1];
      double zd = 0.0;
      if constexpr (DIM == 3) zd = a.src64[(int64_t)DIM * so + 2];
#pragma unroll
      for (int r = 0; r < DIM; ++r) {
        double t = T[r][0] * x + T[r][1] * y + T[r][3], td = T[r][0] * xd + T[r][1] * yd + T[r][3];
        if constexpr (DIM == 3) { t += T[r][2] * z; td += T[r][2] * zd; }
        af[r] = (float)t;                                   // what the search saw
        ad[r] = td;                                         // icp.py:116 on the float64 point
      }
    }
    if (threadIdx.x == 0) { s_best = ~0ull; s_key = ~0ull; }
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
      // a thread looks at whole chunks: the chunk's box first (a candidate lies within √lim of the point), its CH targets only then
      for (int64_t c = threadIdx.x; c < a.n1; c += kLpThreads) {
        const float4 lo = a.lo1[c], hi = a.hi1[c];
        const float dx = fmaxf(fmaxf(lo.x - af[0], af[0] - hi.x), 0.f), dy = fmaxf(fmaxf(lo.y - af[1], af[1] - hi.y), 0.f);
        float lb = __builtin_fmaf(dy, dy, dx * dx);
        if constexpr (DIM == 3) { const float dz = fmaxf(fmaxf(lo.z - af[2], af[2] - hi.z), 0.f); lb = __builtin_fmaf(dz, dz, lb); }
        if (lb * kLpSafety > lim) continue;
        for (int k = 0; k < NnChunk<DIM>::value; ++k) {
          const int64_t j = c * NnChunk<DIM>::value + k;
          if (j >= a.m) break;
          const SP q = tgt[j];
          float qz = 0.f;
          if constexpr (DIM == 3) qz = q.z;
          if (lp_dist<DIM>(af, q.x, q.y, qz) <= lim) {
            const int o = a.tgt_index[j];
            double ev = 0.0;
#pragma unroll
            for (int r = 0; r < DIM; ++r) { const double df = ad[r] - a.tgt64[(int64_t)DIM * o + r]; ev += df * df; }
            const unsigned long long bits = (unsigned long long)__double_as_longlong(ev);      // non-negative doubles order like their bits
            if (pass == 0) atomicMin(&s_best, bits);
            else if (bits == s_best) atomicMin(&s_key, ((unsigned long long)(unsigned)o << 32) | (unsigned long long)(unsigned)j);
          }
        }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0 && s_key != ~0ull) a.match[i] = (int32_t)(unsigned)(s_key & 0xffffffffull);
    __syncthreads();
  }
}

template <int DIM>
__global__ void __launch_bounds__(kMomThreads)
moments_sorted_kernel(LpMomArgs a) {
  constexpr int kAcc = AccLayout<DIM>::kTotal;
  using SP = typename SortedPoint<DIM>::type;
  __shared__ double sh[kMomThreads / 64][kAcc];
  if (a.state->done) return;
  const SP* __restrict__ src = reinterpret_cast<const SP*>(a.src_sorted);
  const SP* __restrict__ tgt = reinterpret_cast<const SP*>(a.tgt_sorted);
  const bool box = (a.flags & kFlagNormalize) != 0;
  double T[3][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) T[r][c] = a.state->T[4 * r + c];
  PairAcc<DIM> acc;
  acc.clear();
  for (int64_t i = (int64_t)blockIdx.x * kMomThreads + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * kMomThreads) {
    const int32_t j = a.match[i];
    if (a.out_corr) a.out_corr[a.src_index[i]] = a.tgt_index[j];
    double x, y, z = 0.0, p[DIM], q[DIM];
    if (a.src64) {                                          // float64-exact mode: the float64 points themselves (original order)
      const int64_t so = a.src_index[i], to = a.tgt_index[j];
      x = a.src64[DIM * so]; y = a.src64[DIM * so + 1];
      if constexpr (DIM == 3) z = a.src64[DIM * so + 2];
#pragma unroll
      for (int r = 0; r < DIM; ++r) q[r] = a.tgt64[DIM * to + r];
    } else {
      const SP v = src[i];
      const SP w = tgt[j];
      x = v.x; y = v.y;
      if constexpr (DIM == 3) z = v.z;
      q[0] = w.x; q[1] = w.y;
      if constexpr (DIM == 3) q[2] = w.z;
    }
#pragma unroll
    for (int r = 0; r < DIM; ++r) {
      double t = T[r][0] * x + T[r][1] * y + T[r][3];
      if constexpr (DIM == 3) t += T[r][2] * z;
      p[r] = t;
    }
    acc.add(p, q, box);
  }
  acc.wave_reduce(box);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) acc.store(sh[wave]);
  __syncthreads();
  if (threadIdx.x < kAcc) {
    const int k = threadIdx.x;
    constexpr int kS = AccLayout<DIM>::kSums;
    double v = sh[0][k];
    for (int w = 1; w < kMomThreads / 64; ++w) {
      if (k < kS) v += sh[w][k];
      else if (k < kS + DIM) v = fmin(v, sh[w][k]);
      else v = fmax(v, sh[w][k]);
    }
    a.partials[(size_t)blockIdx.x * kAcc + k] = v;
  }
}

// ---- host side ----------------------------------------------------------------------------------------
size_t lp_sort_temp_bytes(int64_t n) {
  size_t b = 0;
  unsigned* k = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, b, k, k, k, k, (size_t)n, 0, 3 * kLpAxisBits);
  return b;
}

static unsigned lp_blocks(int64_t n) { return (unsigned)((n + kLpThreads - 1) / kLpThreads); }

template <int DIM>
static hipError_t lp_sort_cloud(const float* pts, int64_t n, const float* d_grid, const LpT& T, const LpWs& ws, void* sorted,
                                int32_t* index, hipStream_t st) {
  using SP = typename SortedPoint<DIM>::type;
  int curve = 1;
  if (const char* env = getenv("LSLAM_ICP_PRUNE_CURVE")) curve = atoi(env);
  hipLaunchKernelGGL(lp_keys_kernel<DIM>, dim3(lp_blocks(n)), dim3(kLpThreads), 0, st, pts, n, d_grid, T, ws.keys_in, ws.vals_in, curve);
  size_t tb = ws.temp_bytes;
  hipError_t e = rocprim::radix_sort_pairs(ws.temp, tb, ws.keys_in, ws.keys_out, ws.vals_in, ws.vals_out, (size_t)n, 0, DIM * kLpAxisBits, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(lp_gather_kernel<DIM>, dim3(lp_blocks(n)), dim3(kLpThreads), 0, st, pts, ws.vals_out, n, static_cast<SP*>(sorted), index);
  return hipGetLastError();
}

template <int DIM>
static hipError_t lp_prepare_t(const float* d_src, int64_t n, const float* d_tgt, int64_t m, const double* init_T, const LpWs& ws,
                               LpTree* tree, hipStream_t st) {
  using SP = typename SortedPoint<DIM>::type;
  constexpr int CH = NnChunk<DIM>::value;
  int parts = (int)lp_blocks(m);
  if (parts > 1024) parts = 1024;
  hipLaunchKernelGGL(lp_bbox_kernel<DIM>, dim3(parts), dim3(kLpThreads), 0, st, d_tgt, m, ws.bbox_partial);
  hipLaunchKernelGGL(lp_bbox_finish_kernel, dim3(1), dim3(64), 0, st, ws.bbox_partial, parts, ws.grid);
  LpT I{}, T0{};
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) { I.m[r][c] = (r == c) ? 1.f : 0.f; T0.m[r][c] = (float)init_T[4 * r + c]; }
  hipError_t e = lp_sort_cloud<DIM>(d_tgt, m, ws.grid, I, ws, ws.tgt_sorted, ws.tgt_index, st);
  if (e != hipSuccess) return e;
  e = lp_sort_cloud<DIM>(d_src, n, ws.grid, T0, ws, ws.src_sorted, ws.src_index, st);
  if (e != hipSuccess) return e;
  tree->n1 = (m + CH - 1) / CH;
  tree->n2 = (tree->n1 + kLpFan - 1) / kLpFan;
  tree->n3 = (tree->n2 + kLpFan - 1) / kLpFan;
  hipLaunchKernelGGL(lp_boxes1_kernel<DIM>, dim3(lp_blocks(tree->n1)), dim3(kLpThreads), 0, st, static_cast<const SP*>(ws.tgt_sorted), m, tree->n1, ws.lo1, ws.hi1);
  hipLaunchKernelGGL(lp_boxes_up_kernel, dim3(lp_blocks(tree->n2)), dim3(kLpThreads), 0, st, ws.lo1, ws.hi1, tree->n1, tree->n2, ws.lo2, ws.hi2);
  hipLaunchKernelGGL(lp_boxes_up_kernel, dim3(lp_blocks(tree->n3)), dim3(kLpThreads), 0, st, ws.lo2, ws.hi2, tree->n2, tree->n3, ws.lo3, ws.hi3);
  hipLaunchKernelGGL(lp_fill_kernel, dim3(lp_blocks(n)), dim3(kLpThreads), 0, st, ws.match, n, -1);
  return hipGetLastError();
}

hipError_t lp_prepare(int dim, const float* d_src, int64_t n, const float* d_tgt, int64_t m, const double* init_T,
                      const LpWs& ws, LpTree* tree, hipStream_t st) {
  return dim == 2 ? lp_prepare_t<2>(d_src, n, d_tgt, m, init_T, ws, tree, st) : lp_prepare_t<3>(d_src, n, d_tgt, m, init_T, ws, tree, st);
}

hipError_t lp_launch_nn(int dim, const LpNnArgs& a, hipStream_t st) {
  constexpr int P = 1;                 // as many independent waves as possible: the search is latency-bound
  const unsigned blocks = (unsigned)((a.n + (int64_t)kLpThreads * P - 1) / ((int64_t)kLpThreads * P));
  if (a.flag_count) {                                       // float64-exact mode
    hipError_t e = hipMemsetAsync(a.flag_count, 0, sizeof(int32_t), st);
    if (e != hipSuccess) return e;
    if (dim == 2) hipLaunchKernelGGL((nn_pruned_kernel<2, P, true>), dim3(blocks), dim3(kLpThreads), 0, st, a);
    else hipLaunchKernelGGL((nn_pruned_kernel<3, P, true>), dim3(blocks), dim3(kLpThreads), 0, st, a);
    return hipGetLastError();
  }
  if (dim == 2) hipLaunchKernelGGL((nn_pruned_kernel<2, P, false>), dim3(blocks), dim3(kLpThreads), 0, st, a);
  else hipLaunchKernelGGL((nn_pruned_kernel<3, P, false>), dim3(blocks), dim3(kLpThreads), 0, st, a);
  return hipGetLastError();
}

hipError_t lp_launch_refine(int dim, const LpRefineArgs& a, hipStream_t st) {
  if (dim == 2) hipLaunchKernelGGL(lp_refine_kernel<2>, dim3(512), dim3(kLpThreads), 0, st, a);
  else hipLaunchKernelGGL(lp_refine_kernel<3>, dim3(512), dim3(kLpThreads), 0, st, a);
  return hipGetLastError();
}

hipError_t lp_launch_moments(int dim, const LpMomArgs& a, int rows, hipStream_t st) {
  if (dim == 2) hipLaunchKernelGGL(moments_sorted_kernel<2>, dim3(rows), dim3(kMomThreads), 0, st, a);
  else hipLaunchKernelGGL(moments_sorted_kernel<3>, dim3(rows), dim3(kMomThreads), 0, st, a);
  return hipGetLastError();
}

}  // namespace lslam

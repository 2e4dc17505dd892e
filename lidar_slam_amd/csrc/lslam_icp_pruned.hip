// lslam_icp_pruned.hip — the batched planar ICP with EXACT chunk pruning (the default from 256 points per cloud).
//
// Same operator, same results as icp_batch_kernel (modules/icp.py:123-189): every source point still
// gets its exact float32 nearest target, lowest original index on ties.  What changes is how many
// pair evaluations that takes.  Both clouds of a problem are first brought into an order that keeps
// neighbours together (icp_sort_kernel: the order they arrive in when that is compact already — LiDAR
// beam order — else the cells of a uniform grid over the target's bounding box along the Hilbert curve,
// bitonic sort of (cell, index) keys in LDS, once per problem).  Then
//   * 16 consecutive targets (a chunk) are spatially compact: their bounding box is kept in LDS;
//   * the 64·P consecutive sources of a wave are spatially compact too: each iteration the
//     wave reduces the bounding box of its transformed points;
//   * a chunk whose box is farther from the wave's box than every lane's current best distance cannot
//     contain a nearer (or equally near) point for any lane; the others are tested against every point,
//     and each lane compares its points with the chunks they still need.  The best distance starts from
//     the chunk of the previous iteration's match, so from the second iteration on almost everything is skipped.
// The bound is made conservative against float32 rounding (factor 1 − 1e-5) and the comparison is
// non-strict, so a skipped chunk can never hold the argmin or a tie.  Equal distances of two targets
// (duplicate points) are resolved by a rare exhaustive pass ordered by original index.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "lslam_device.h"
#include "lslam_kernels.h"
#include "lslam_prune_common.h"

namespace lslam {

__device__ __forceinline__ unsigned morton5(unsigned x, unsigned y) {      // interleave two 5-bit coordinates
  unsigned r = 0;
#pragma unroll
  for (int b = 0; b < 5; ++b) r |= ((x >> b) & 1u) << (2 * b) | ((y >> b) & 1u) << (2 * b + 1);
  return r;
}

// position of cell (x, y) of a side × side grid along the Hilbert curve: consecutive cells share an edge, so a chunk of
// consecutive points never spans two far-apart corners of the cloud the way it does where the Z curve jumps
__device__ __forceinline__ unsigned hilbert_index(unsigned x, unsigned y, unsigned side) {      // side: cells per axis, a power of two ≤ 64
  unsigned d = 0;
  for (unsigned s = side >> 1; s > 0; s >>= 1) {
    const unsigned rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
    d += s * s * ((3u * rx) ^ ry);
    if (ry == 0) {                                             // rotate / flip the quadrant
      if (rx == 1) { x = side - 1u - x; y = side - 1u - y; }
      const unsigned t = x; x = y; y = t;
    }
  }
  return d;
}

// ascending bitonic sort of `len` (power of two) uint32 keys in LDS
__device__ inline void bitonic_sort(unsigned* keys, int len) {
  for (int k = 2; k <= len; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = threadIdx.x; i < len; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned a = keys[i], b = keys[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
        }
      }
    }
  __syncthreads();
}

// One workgroup per problem: copies of target and source in an order that keeps neighbours together + their original indices.
//
// That order is the order of the cells of a grid over the target's box along the Hilbert curve (indices inside a cell ascending) —
// unless the cloud ARRIVES in a compact order.  A LiDAR scan
// does: consecutive beams hit neighbouring points of the same wall, so 16 consecutive points already form a small box (the
// boxes of a scan add up to about the length of the walls, whatever the number of beams; Morton-sorted points of a filled
// area add up to √(n·W·H)/2).  keep bit 0 / 1: the target / the source stays as it is when the half perimeters of its
// chunk boxes sum to no more than that figure (+ 2(W+H) for clouds on a line).  On the C3 sequence this takes the batch from
// 1.94 to 1.29 ms: the two bitonic sorts were a quarter of it, and the search itself is faster on beam order.  The search
// is exact for any order; the order only decides how much is pruned.
// How it is sorted (round 3): each cloud is read from memory ONCE into LDS; a counting sort by cell (histogram with LDS atomics,
// a scan, a scatter whose order inside a cell is arbitrary, then every cell's few points put in index order by one thread) replaces
// the bitonic sort of (cell, index) keys — the same final order, so the same bits downstream.  Round 2's kernel moved 3.5× the
// clouds through HBM (three passes + a gather of 8 bytes per cache line) and spent 78 barrier-separated steps per sort.  A cell
// holding more than kSortCellMax points (collapsed or very uneven clouds) sends that cloud to the bitonic sort.
constexpr int kSortCellMax = 48;
template <int kSortThreads>       // 256 lanes for LiDAR-sized clouds (seven workgroups per CU), 512 from 2048 points per cloud (C4: 0.70 -> 0.50 ms)
__global__ void __launch_bounds__(kSortThreads)
icp_sort_kernel(IcpBatchArgs a, PruneScratch sc, int len_pow2, int keep_forced, int curve, int cap, int cur_words, int skip64) {
  // LDS sized by the launch for the batch's largest cloud (cap points, cur_words counters ≥ cells + 1 and ≥ len_pow2 keys):
  // LiDAR-sized batches keep seven workgroups per CU, 4096-point clouds two
  extern __shared__ __align__(16) unsigned char smem[];
  float2* pts = reinterpret_cast<float2*>(smem);                                     // the cloud as it arrives
  unsigned* cur = reinterpret_cast<unsigned*>(smem + sizeof(float2) * (size_t)cap);  // per cell: count → start → end; the bitonic keys alias it
  uint16_t* cell = reinterpret_cast<uint16_t*>(cur + cur_words);                     // cell of every point
  uint16_t* ord = cell + cap;                                                        // ord[position] = original index
  unsigned* keys = cur;
  __shared__ float sbox[4];
  __shared__ float red[kSortThreads / 64][4];
  __shared__ unsigned sred[kSortThreads / 64 + 1];
  const int pair = blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t s0 = a.src_off[pair], t0 = a.tgt_off[pair];
  const int n = (int)(a.src_off[pair + 1] - s0), m = (int)(a.tgt_off[pair + 1] - t0);
  const float2* src = reinterpret_cast<const float2*>(a.src) + s0;
  const float2* tgt = reinterpret_cast<const float2*>(a.tgt) + t0;
  // float64-exact mode: the sorted float64 copies — not for problems the running-sums build of the queue kernel iterates (skip64 = 1: those
  // with a cloud of kQueueCertifyPoints or more, 2: all): it reads the few float64 points it needs through the index maps
  const bool copy64 = !(skip64 == 2 || (skip64 == 1 && (n >= kQueueCertifyPoints || m >= kQueueCertifyPoints)));
  // the sorted copies live in pair-local slots of the scratch (slot = pair · stride): the caller's offsets may
  // start anywhere in a larger packed buffer, the scratch is sized for this batch only
  const int64_t ss = (int64_t)pair * sc.src_stride, ts = (int64_t)pair * sc.tgt_stride;

  // bounding box of the staged cloud, the same four numbers in every thread
  auto cloud_box = [&](int cnt, float& x0, float& y0, float& x1, float& y1) {
    x0 = INFINITY; y0 = INFINITY; x1 = -INFINITY; y1 = -INFINITY;
    for (int j = tid; j < cnt; j += kSortThreads) {
      const float2 q = pts[j];
      x0 = fminf(x0, q.x); y0 = fminf(y0, q.y); x1 = fmaxf(x1, q.x); y1 = fmaxf(y1, q.y);
    }
    x0 = wave_min_f(x0); y0 = wave_min_f(y0); x1 = wave_max_f(x1); y1 = wave_max_f(y1);
    __syncthreads();                                         // red / sbox of an earlier call are consumed
    if ((tid & 63) == 0) { red[tid >> 6][0] = x0; red[tid >> 6][1] = y0; red[tid >> 6][2] = x1; red[tid >> 6][3] = y1; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < kSortThreads / 64; ++w) {
        x0 = fminf(x0, red[w][0]); y0 = fminf(y0, red[w][1]); x1 = fmaxf(x1, red[w][2]); y1 = fmaxf(y1, red[w][3]);
      }
      sbox[0] = x0; sbox[1] = y0; sbox[2] = x1; sbox[3] = y1;
    }
    __syncthreads();
    x0 = sbox[0]; y0 = sbox[1]; x1 = sbox[2]; y1 = sbox[3];
  };
  // does the staged cloud arrive in a compact order?  (fixed reduction order: the same answer in every run)
  auto compact_as_given = [&](int cnt, float w, float h) -> bool {
    float sum = 0.f;
    for (int c = tid; c * kPruneChunk < cnt; c += kSortThreads) {
      float cx0 = INFINITY, cy0 = INFINITY, cx1 = -INFINITY, cy1 = -INFINITY;
      for (int k = 0; k < kPruneChunk && c * kPruneChunk + k < cnt; ++k) {
        const float2 q = pts[c * kPruneChunk + k];
        cx0 = fminf(cx0, q.x); cy0 = fminf(cy0, q.y); cx1 = fmaxf(cx1, q.x); cy1 = fmaxf(cy1, q.y);
      }
      sum += (cx1 - cx0) + (cy1 - cy0);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6][0] = sum;
    __syncthreads();
    if (tid == 0) {
      float tot = 0.f;
      for (int wv = 0; wv < kSortThreads / 64; ++wv) tot += red[wv][0];
      sbox[0] = (tot <= 0.5f * sqrtf((float)cnt * w * h) + 2.f * (w + h)) ? 1.f : 0.f;      // NaN / inf extents: false, the cloud is sorted
    }
    __syncthreads();
    return sbox[0] != 0.f;
  };

  // ---- the target: staged, its box is the grid of both clouds ----
  for (int j = tid; j < m; j += kSortThreads) pts[j] = tgt[j];
  __syncthreads();
  float x0, y0, x1, y1;
  cloud_box(m, x0, y0, x1, y1);
  const bool keep_tgt = keep_forced >= 0 ? (keep_forced & 1) != 0 : compact_as_given(m, x1 - x0, y1 - y0);
  // about one point per cell on the Hilbert curve (≤ 64 cells per axis: 12 + 13 key bits; 1 % better than four per cell on C4);
  // the Z-curve keys of round 2 hold 5 bits per axis
  int G = m >= 2048 ? 32 : (m >= 512 ? 16 : 8);
  if (curve >= 1) G *= 2;
  const int ncell = G * G;
  const float sx = (x1 > x0) ? (float)G / (x1 - x0) : 0.f, sy = (y1 > y0) ? (float)G / (y1 - y0) : 0.f;
  auto cell_of = [&](float x, float y) -> unsigned {
    int gx = (int)((x - x0) * sx), gy = (int)((y - y0) * sy);
    gx = min(max(gx, 0), G - 1); gy = min(max(gy, 0), G - 1);
    if (curve >= 1) return hilbert_index((unsigned)gx, (unsigned)gy, (unsigned)G);
    return morton5((unsigned)gx, (unsigned)gy);
  };
  const double* T = a.init_T + (int64_t)16 * pair;
  const float t00 = (float)T[0], t01 = (float)T[1], t03 = (float)T[3], t10 = (float)T[4], t11 = (float)T[5], t13 = (float)T[7];

  // ord[] for the staged cloud of cnt points: by (cell, index); `placed`: the cell is taken of the point moved by the initial transform
  auto order_by_cell = [&](int cnt, bool placed) {
    for (int c = tid; c <= ncell; c += kSortThreads) cur[c] = 0u;
    __syncthreads();
    for (int j = tid; j < cnt; j += kSortThreads) {
      const float2 p = pts[j];
      const unsigned c = placed ? cell_of(t00 * p.x + t01 * p.y + t03, t10 * p.x + t11 * p.y + t13) : cell_of(p.x, p.y);
      cell[j] = (uint16_t)c;
      atomicAdd(&cur[c], 1u);
    }
    __syncthreads();
    // counts → starts (exclusive scan over the cells; every thread owns `per` consecutive cells), and the fullest cell
    const int per = (ncell + kSortThreads - 1) / kSortThreads;
    unsigned mine = 0, most = 0;
    for (int k = 0; k < per; ++k) { const int c = tid * per + k; if (c < ncell) { mine += cur[c]; most = max(most, cur[c]); } }
    unsigned incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += up; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) most = max(most, (unsigned)__shfl_xor((int)most, o, 64));
    if ((tid & 63) == 63) sred[tid >> 6] = incl;
    if ((tid & 63) == 0) red[tid >> 6][1] = __uint_as_float(most);
    __syncthreads();
    unsigned before = incl - mine;
    for (int w = 0; w < (tid >> 6); ++w) before += sred[w];
    unsigned fullest = 0;
    for (int w = 0; w < kSortThreads / 64; ++w) fullest = max(fullest, __float_as_uint(red[w][1]));
    if (fullest > (unsigned)kSortCellMax) {
      // collapsed / very uneven cloud: the bitonic sort of (cell, index) keys (they alias the counters, which are done with)
      __syncthreads();
      for (int j = tid; j < len_pow2; j += kSortThreads) keys[j] = j < cnt ? ((unsigned)cell[j] << 13) | (unsigned)j : 0xffffffffu;
      bitonic_sort(keys, len_pow2);
      for (int j = tid; j < cnt; j += kSortThreads) ord[j] = (uint16_t)(keys[j] & 8191u);
      __syncthreads();
      return;
    }
    for (int k = 0; k < per; ++k) {                            // in place: cur[c] = first position of cell c
      const int c = tid * per + k;
      if (c < ncell) { const unsigned cnt_c = cur[c]; cur[c] = before; before += cnt_c; }
    }
    __syncthreads();
    for (int j = tid; j < cnt; j += kSortThreads) ord[atomicAdd(&cur[cell[j]], 1u)] = (uint16_t)j;      // cur[c] ends as the END of cell c
    __syncthreads();
    // inside a cell the scatter's order is whoever came first: put every cell's points in index order (insertion sort, ≤ kSortCellMax)
    for (int c = tid; c < ncell; c += kSortThreads) {
      const int b = c ? (int)cur[c - 1] : 0, e = (int)cur[c];
      for (int i = b + 1; i < e; ++i) {
        const uint16_t v = ord[i];
        int k = i - 1;
        while (k >= b && ord[k] > v) { ord[k + 1] = ord[k]; --k; }
        ord[k + 1] = v;
      }
    }
    __syncthreads();
  };

  if (keep_tgt) { for (int j = tid; j < m; j += kSortThreads) ord[j] = (uint16_t)j; __syncthreads(); }
  else order_by_cell(m, false);
  for (int j = tid; j < m; j += kSortThreads) {
    const unsigned o = ord[j];
    sc.tgt_sorted[ts + j] = pts[o];
    sc.tgt_index[ts + j] = (uint16_t)o;
    if (a.tgt64 && copy64) sc.tgt64_sorted[ts + j] = reinterpret_cast<const double2*>(a.tgt64)[t0 + o];
  }
  __syncthreads();

  // ---- the source: as given if compact (a rigid motion does not undo that), else placed by its initial transform into the target's grid ----
  {
    float mx = 0.f;                                            // the largest |p|² of the source, for the queue kernel's step bound and near-tie window
    for (int i = tid; i < n; i += kSortThreads) { const float2 q = src[i]; pts[i] = q; mx = fmaxf(mx, __builtin_fmaf(q.y, q.y, q.x * q.x)); }
    mx = wave_max_f(mx);
    if ((tid & 63) == 0) red[tid >> 6][0] = mx;                // (red is free here: every earlier use ended at a barrier)
  }
  __syncthreads();
  if (sc.smax && tid == 0) {
    float mx = red[0][0];
    for (int w = 1; w < kSortThreads / 64; ++w) mx = fmaxf(mx, red[w][0]);
    sc.smax[pair] = mx;
  }
  bool keep_src = keep_forced >= 0 ? (keep_forced & 2) != 0 : false;
  if (keep_forced < 0) {
    float u0, v0, u1, v1;
    cloud_box(n, u0, v0, u1, v1);
    keep_src = compact_as_given(n, u1 - u0, v1 - v0);
  }
  if (keep_src) { for (int i = tid; i < n; i += kSortThreads) ord[i] = (uint16_t)i; __syncthreads(); }
  else order_by_cell(n, true);
  for (int i = tid; i < n; i += kSortThreads) {
    const unsigned o = ord[i];
    sc.src_sorted[ss + i] = pts[o];
    sc.src_index[ss + i] = (uint16_t)o;
    if (a.src64 && copy64) sc.src64_sorted[ss + i] = reinterpret_cast<const double2*>(a.src64)[s0 + o];
    if (a.src64 && !copy64 && sc.src_rank) sc.src_rank[ss + o] = (uint16_t)i;      // (the running-sums build walks the float64 sources in the caller's order)
  }
  // the problem's class (launch_icp_pruned): which kernel iterates a problem depends on its own size, not on the batch it came in
  if (sc.order && tid == 0) {
    const unsigned big = (n >= kQueueCertifyPoints || m >= kQueueCertifyPoints) ? 1u : 0u;
    const unsigned k = atomicAdd(&sc.queue[48 + big], 1u);
    sc.order[big ? (int)gridDim.x - 1 - (int)k : (int)k] = pair;
  }
}

// F64: the float64-exact mode (lslam_device.h: NearTie; args.src64 / args.tgt64 set) — pruning and tie detection use the
// near-tie window lim(D) instead of D, near ties are decided on the float64 points, the moments are taken from them.
// LIST: every lane evaluates the chunks ITS points need, one (point, chunk) at a time with gathered reads, instead of the
// whole wave evaluating every chunk any of its points needs (see the sweep below).
template <int P, bool F64, bool LIST>
__global__ void __launch_bounds__(kIcpBatchMaxThreads, 8)      // ≤ 64 VGPRs: two 16-wave workgroups per CU (measured 1.3× faster than 119 VGPRs / one workgroup)
icp_pruned_kernel(IcpBatchArgs args, PruneScratch sc, PruneLayout lay) {
  constexpr int CHUNK = kPruneChunk;
  constexpr int kAcc = AccLayout<2>::kTotal;
  extern __shared__ __align__(kPruneTileAlign) unsigned char smem[];
  float2* tile = reinterpret_cast<float2*>(smem);                                   // sorted targets, padded
  float4* aabb = reinterpret_cast<float4*>(smem + lay.aabb_off);                    // per chunk {x0,y0,x1,y1}
  float2* sh_src = reinterpret_cast<float2*>(smem + lay.src_off);                   // sorted sources: read twice per iteration, so they live on chip
  uint16_t* sh_prev = reinterpret_cast<uint16_t*>(smem + lay.prev_off);             // last match (sorted target position) of every source
  double* sh_red = reinterpret_cast<double*>(smem + lay.red_off);                   // [waves][kAcc]
  double* sh_T = sh_red + (size_t)kIcpBatchMaxWaves * kAcc;
  // loop state and last error in LDS (wave 0 only, wave-uniform): in registers they were spilled around the solve
  double* sh_err = sh_T + 16;
  StopState* sh_stop = reinterpret_cast<StopState*>(sh_err + 1);
  int* sh_flag = reinterpret_cast<int*>(sh_stop + 1);

  const int pair = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int64_t s0 = args.src_off[pair], t0 = args.tgt_off[pair];
  const int n = (int)(args.src_off[pair + 1] - s0), m = (int)(args.tgt_off[pair + 1] - t0);
  const int64_t ss = (int64_t)pair * sc.src_stride, ts = (int64_t)pair * sc.tgt_stride;   // pair-local scratch slots
  const float2* __restrict__ ssrc = sc.src_sorted + ss;        // Morton-sorted sources
  // original target index of a sorted position: only needed when two targets are equally near (rare) and for the
  // correspondences written at the end, so it stays in global memory (the LDS it took lets the sources live on chip)
  const uint16_t* __restrict__ tidx = sc.tgt_index + ts;
  const bool box = (args.flags & kFlagNormalize) != 0;
  const int nchunks = (m + CHUNK - 1) / CHUNK;

  const double2* __restrict__ gsrc64 = F64 ? sc.src64_sorted + ss : nullptr;        // the float64 clouds in the SORTED order (icp_sort_kernel)
  const double2* __restrict__ gtgt64 = F64 ? sc.tgt64_sorted + ts : nullptr;
  if (tid < 16) sh_T[tid] = args.init_T[(int64_t)16 * pair + tid];
  if (F64 && tid == 0) { sh_flag[1] = 0; sh_flag[2] = 0; }   // bits of max |t|², max |p|²
  if (tid == 0) sh_flag[3] = 0;                              // the next slab of 64·P sources to be searched (phase A)
  float tmax = 0.f, smax = 0.f;
  for (int j = tid; j < nchunks * CHUNK; j += blockDim.x) {
    const float2 v = (j < m) ? sc.tgt_sorted[ts + j] : make_float2(kFar, kFar);
    tile[j] = v;
    if (F64 && j < m) tmax = fmaxf(tmax, __builtin_fmaf(v.y, v.y, v.x * v.x));
  }
  const bool src_lds = lay.src_off != 0xffffffffu;
  for (int i = tid; i < n; i += blockDim.x) {
    const float2 v = ssrc[i];
    if (src_lds) sh_src[i] = v;
    sh_prev[i] = 0xffff;
    if (F64) smax = fmaxf(smax, __builtin_fmaf(v.y, v.y, v.x * v.x));
  }
  __syncthreads();
  float Rt = 0.f, Rs = 0.f;
  if constexpr (F64) {
    tmax = wave_max_f(tmax); smax = wave_max_f(smax);
    if (lane == 0) { atomicMax(&sh_flag[1], __float_as_int(tmax)); atomicMax(&sh_flag[2], __float_as_int(smax)); }
  }
  for (int c = tid; c < nchunks; c += blockDim.x) {        // boxes over the real points only
    float x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
    for (int k = 0; k < CHUNK && c * CHUNK + k < m; ++k) {
      const float2 q = tile[c * CHUNK + k];
      x0 = fminf(x0, q.x); y0 = fminf(y0, q.y); x1 = fmaxf(x1, q.x); y1 = fmaxf(y1, q.y);
    }
    aabb[c] = make_float4(x0, y0, x1, y1);
  }
  // the box of every slab of 64·P consecutive sources, once per problem, in the sources' own frame: per iteration the wave's
  // box is then the box of this box's four corners under T (interval arithmetic on a wave-uniform float4) instead of four
  // 6-step reductions over the transformed points — 24 dependent cross-lane steps per slab and iteration were a fifth of the
  // kernel's wave time.  It is a little larger than the exact box (by the rotation of T), which only lets a few more chunks
  // through the wave-level test; every point is still tested on its own.
  float4* slab_box = reinterpret_cast<float4*>(smem + lay.slab_off);
  const int nslabs = (n + kWave * P - 1) / (kWave * P);
  for (int sl = wave; sl < nslabs; sl += nwaves) {
    float x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = (sl * kWave + lane) * P + p;
      if (i < n) {
        const float2 v = src_lds ? sh_src[i] : ssrc[i];
        x0 = fminf(x0, v.x); y0 = fminf(y0, v.y); x1 = fmaxf(x1, v.x); y1 = fmaxf(y1, v.y);
      }
    }
    x0 = wave_min_f(x0); y0 = wave_min_f(y0); x1 = wave_max_f(x1); y1 = wave_max_f(y1);
    if (lane == 0) slab_box[sl] = make_float4(x0, y0, x1, y1);
  }
  __syncthreads();
  if constexpr (F64) {
    Rt = sqrtf(__int_as_float(sh_flag[1])) * 1.0001f;
    Rs = sqrtf(__int_as_float(sh_flag[2])) * 1.0001f;
  }
  if (tid == 0) { StopState st0; stop_init(st0); *sh_stop = st0; *sh_err = 0.0; }       // read first after the barrier that precedes the solve
  const int per_pass = blockDim.x * P;
  const int passes = n > 0 ? (n + per_pass - 1) / per_pass : 1;
#ifdef LSLAM_PRUNE_PROFILE
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = __builtin_amdgcn_s_memtime();
#define PRUNE_MARK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); prof[k] += now_ - pt; pt = now_; } while (0)
#else
#define PRUNE_MARK(k) do {} while (0)
#endif

  for (;;) {
    double T[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) T[r][c] = wave_uniform(sh_T[4 * r + c]);
    NearTie nt = near_tie_setup(false, 0.f, 0.f, 0.f);
    if constexpr (F64)
      nt = near_tie_setup(true, Rt, Rs, sqrtf((float)(T[0][0] * T[0][0] + T[0][1] * T[0][1] + T[1][0] * T[1][0] + T[1][1] * T[1][1])) * 1.0001f);
    // the bound a chunk must beat: the lane's best itself, or in float64-exact mode everything within its near-tie window
    auto reach = [&](float b) { return F64 ? __builtin_fmaf(b, nt.K1, nt.C) : b; };

    // ---- phase A: nearest neighbours (float32); the matches go to LDS (they also seed the next iteration)
    // With more slabs of 64·P consecutive sorted sources than waves, the waves take them from a counter: the search time of a
    // slab depends on where it lies (how many chunks its points still need), and with a fixed slab-to-wave map the workgroup
    // waited at the barrier for its slowest wave (C4: 22.8 -> 22.4 ms).  The matches go to LDS by source position, so who
    // searched a slab does not matter — but phase B then needs a barrier, which costs LiDAR-sized problems (one slab per
    // wave, nothing to balance) 10 %: they keep the fixed map.
    const bool take = nslabs > nwaves;
    for (int round = 0;; ++round) {
      int slab = wave + nwaves * round;
      if (take) {                                            // (drawing the next slab ahead of the search was measured: 1 % slower — a wave then sits on a slab)
        if (lane == 0) slab = atomicAdd(&sh_flag[3], 1);
        slab = __builtin_amdgcn_readfirstlane(slab);
      }
      if (slab >= nslabs) break;
      const int base = (slab - wave) * kWave * P;            // base + tid·P + p = slab·64·P + lane·P + p
      float ax[P], ay[P];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = base + tid * P + p;
        ax[p] = 0.f; ay[p] = 0.f;
        if (i < n) {
          const float2 v = src_lds ? sh_src[i] : ssrc[i];
          ax[p] = (float)(T[0][0] * (double)v.x + T[0][1] * (double)v.y + T[0][3]);      // icp.py:116
          ay[p] = (float)(T[1][0] * (double)v.x + T[1][1] * (double)v.y + T[1][3]);
        }
      }
      // the wave's box: the slab's box under T, widened by 16 ulps of the magnitudes involved (the points above are rounded from
      // float64, the corners here are float32 products)
      float wx0, wy0, wx1, wy1;
      {
        const float4 sb = slab_box[slab];
        const float r00 = (float)T[0][0], r01 = (float)T[0][1], r03 = (float)T[0][3], r10 = (float)T[1][0], r11 = (float)T[1][1], r13 = (float)T[1][3];
        const float mx = fmaxf(fabsf(sb.x), fabsf(sb.z)), my = fmaxf(fabsf(sb.y), fabsf(sb.w));
        const float ex = 1e-6f * (fabsf(r00) * mx + fabsf(r01) * my + fabsf(r03)) + 1e-30f;
        const float ey = 1e-6f * (fabsf(r10) * mx + fabsf(r11) * my + fabsf(r13)) + 1e-30f;
        wx0 = r03 + fminf(r00 * sb.x, r00 * sb.z) + fminf(r01 * sb.y, r01 * sb.w) - ex;
        wx1 = r03 + fmaxf(r00 * sb.x, r00 * sb.z) + fmaxf(r01 * sb.y, r01 * sb.w) + ex;
        wy0 = r13 + fminf(r10 * sb.x, r10 * sb.z) + fminf(r11 * sb.y, r11 * sb.w) - ey;
        wy1 = r13 + fmaxf(r10 * sb.x, r10 * sb.z) + fmaxf(r11 * sb.y, r11 * sb.w) + ey;
      }
      PRUNE_MARK(0);

      bool tie = false;
      bool near[P];                                          // float64-exact mode: another target within the near-tie window of the winner
      int pos[P];                                            // sorted position of the nearest target
      float bd[P];                                           // its distance
      if constexpr (LIST) {
        typedef const __attribute__((address_space(3))) float2* LdsTile;
        const unsigned tile0 = (unsigned)(size_t)(LdsTile)(const void*)tile;
        // ---- seed: the chunk of last iteration's match (first iteration: of the target at the same relative rank — both clouds
        // are sorted along the same curve), compared target by target by the lane itself: an exact bound before any box test
        float sec[P];                                        // second smallest distance over everything evaluated for the point
        unsigned at[P];                                      // LDS address of the nearest target so far
        int own[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int i = base + tid * P + p;
          bd[p] = (i < n) ? INFINITY : -INFINITY;            // lanes without a point never ask for a chunk
          sec[p] = INFINITY; at[p] = tile0; own[p] = -1;
          if (i < n) {
            const int pv = sh_prev[i];
            own[p] = (pv != 0xffff ? pv : (int)(((long long)i * m) / n)) / CHUNK;
            prune_eval_chunk(tile0, own[p], lane, ax[p], ay[p], bd[p], sec[p], at[p]);
          }
        }
        float wb = -INFINITY;
#pragma unroll
        for (int p = 0; p < P; ++p) wb = fmaxf(wb, bd[p]);
        wb = wave_max_f(wb);                                 // no lane can ask for a chunk farther than this from the wave's box
        PRUNE_MARK(1);
        // ---- sweep: 64 chunk boxes are tested at once, one per lane, against the wave's box and the wave's largest bound.
        // The survivors (a 64-bit ballot, walked with scalar bit operations) are tested against every point of every lane with
        // its CURRENT bound; what a point still needs goes into a 64-bit list per point, and each lane then works through the
        // lists of its points.  (Evaluating a surviving chunk for all 64·P points of the wave because one of them needed it
        // was 58 % of the kernel's instructions.)
        for (int c0 = 0; c0 < nchunks; c0 += 64) {
          bool cand = false;
          if (c0 + lane < nchunks) {
            const float4 b = aabb[c0 + lane];
            const float dx = fmaxf(fmaxf(b.x - wx1, wx0 - b.z), 0.f);
            const float dy = fmaxf(fmaxf(b.y - wy1, wy0 - b.w), 0.f);
            cand = reach(wb) >= __builtin_fmaf(dy, dy, dx * dx) * kPruneSafety;
          }
          unsigned long long todo = __ballot(cand);
          if (!todo) continue;
          unsigned long long list[P];
#pragma unroll
          for (int p = 0; p < P; ++p) list[p] = 0ull;
          while (todo) {
            const int t = __builtin_ctzll(todo);
            todo &= todo - 1;
            // from the register of the lane that tested it (v_readlane), not from LDS again: a read per survivor, each waited
            // for in turn, was a fifth of the kernel's wave time
            const float4 b = aabb[c0 + t];                     // (v_readlane from the lane that tested the box was measured: 10 % slower)
            const unsigned long long bit = 1ull << t;
#pragma unroll
            for (int p = 0; p < P; ++p) {
              const float ox = fmaxf(fmaxf(b.x - ax[p], ax[p] - b.z), 0.f);
              const float oy = fmaxf(fmaxf(b.y - ay[p], ay[p] - b.w), 0.f);
              // the seed chunk was compared already: a second visit would read as a tie of the point with itself
              const bool need = reach(bd[p]) >= __builtin_fmaf(oy, oy, ox * ox) * kPruneSafety && own[p] != c0 + t;
              list[p] |= need ? bit : 0ull;
            }
          }
          PRUNE_MARK(2);
#pragma unroll
          for (int p = 0; p < P; ++p) {
            unsigned long long mk = list[p];
            while (__any(mk != 0ull)) {
              if (mk != 0ull) {
                const int t = __builtin_ctzll(mk);
                mk &= mk - 1;
                prune_eval_chunk(tile0, c0 + t, lane, ax[p], ay[p], bd[p], sec[p], at[p]);
              }
            }
          }
          PRUNE_MARK(3);
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const bool owns = base + tid * P + p < n;
          pos[p] = (int)((at[p] - tile0) >> 3);
          if constexpr (F64) { near[p] = owns && bd[p] < INFINITY && sec[p] <= near_lim(bd[p], nt); tie |= near[p]; }
          else { near[p] = false; tie |= owns && bd[p] < INFINITY && sec[p] == bd[p]; }
        }
      } else {
        // ---- start from last iteration's match: a tight bound before the first chunk ------------
        float best[P], second[P];                            // smallest and second smallest chunk minimum seen so far
        int bchunk[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int i = base + tid * P + p;
          best[p] = (i < n) ? INFINITY : -INFINITY;          // lanes without a point never ask for a chunk
          second[p] = INFINITY;
          bchunk[p] = 0;
          if (i < n) {
            const int pv = sh_prev[i];
            if (pv != 0xffff) {
              const float2 q = tile[pv];
              // a bound, not yet a candidate: its chunk will beat it strictly (float64-exact mode: by more than the near-tie window,
              // so that the bound itself is never mistaken for a second candidate)
              const float ds = dist2(ax[p], ay[p], q.x, q.y);
              best[p] = F64 ? next_up(reach(ds) * 1.000001f) : next_up(ds);
              bchunk[p] = pv / CHUNK;
            } else {
              // first iteration: both clouds are sorted along the same curve, so the target at the same
              // relative rank is usually close — any evaluated point is a valid upper bound
              const int g = (int)(((long long)i * m) / n);
              const float2 q = tile[g];
              const float ds = dist2(ax[p], ay[p], q.x, q.y);
              best[p] = F64 ? next_up(reach(ds) * 1.000001f) : next_up(ds);
              bchunk[p] = g / CHUNK;
            }
          }
        }

        // ---- sweep: 64 chunk boxes are tested at once, one per lane, against the wave's box and the wave's largest
        // bound; the chunks that survive (a 64-bit ballot, walked with scalar bit operations) get a second look with
        // every lane's OWN points and CURRENT bounds before they are evaluated.  (One box test per chunk with a
        // compare → scalar branch each, two levels deep, spent a quarter of the kernel's instructions.)
        const float4* t4 = reinterpret_cast<const float4*>(tile);
        float wb = -INFINITY;
#pragma unroll
        for (int p = 0; p < P; ++p) wb = fmaxf(wb, best[p]);
        wb = wave_max_f(wb);                                   // no lane can ask for a chunk farther than this from the wave's box
        for (int c0 = 0; c0 < nchunks; c0 += 64) {
          bool cand = false;
          if (c0 + lane < nchunks) {
            const float4 b = aabb[c0 + lane];
            const float dx = fmaxf(fmaxf(b.x - wx1, wx0 - b.z), 0.f);
            const float dy = fmaxf(fmaxf(b.y - wy1, wy0 - b.w), 0.f);
            cand = reach(wb) >= __builtin_fmaf(dy, dy, dx * dx) * kPruneSafety;
          }
          unsigned long long todo = __ballot(cand);
          while (todo) {
            const int c = c0 + __builtin_ctzll(todo);
            todo &= todo - 1;
            const float4 b = aabb[c];
            // the sorted sources of a wave sit on a Morton curve that sometimes jumps across the cloud; the wave's
            // box is then large, each point's reach is not
            bool need_own = false;
#pragma unroll
            for (int p = 0; p < P; ++p) {
              const float ox = fmaxf(fmaxf(b.x - ax[p], ax[p] - b.z), 0.f);
              const float oy = fmaxf(fmaxf(b.y - ay[p], ay[p] - b.w), 0.f);
              need_own |= reach(best[p]) >= __builtin_fmaf(oy, oy, ox * ox) * kPruneSafety;
            }
            if (!__any(need_own)) continue;                    // nothing in this chunk can beat or tie any lane
            float4 v[CHUNK / 2];
#pragma unroll
            for (int k = 0; k < CHUNK / 2; ++k) v[k] = t4[c * (CHUNK / 2) + k];
#pragma unroll
            for (int p = 0; p < P; ++p) {
              float mn = dist2(ax[p], ay[p], v[0].x, v[0].y);
              mn = fminf(mn, dist2(ax[p], ay[p], v[0].z, v[0].w));
#pragma unroll
              for (int k = 1; k < CHUNK / 2; ++k) {
                const float d0 = dist2(ax[p], ay[p], v[k].x, v[k].y);
                const float d1 = dist2(ax[p], ay[p], v[k].z, v[k].w);
                mn = fminf(fminf(d0, d1), mn);
              }
              // equal minima of two chunks are found through the VALUES (second smallest == smallest) after the sweep:
              // no per-chunk compare into a scalar mask
              const bool lt = mn < best[p];
              second[p] = __builtin_amdgcn_fmed3f(best[p], second[p], mn);
              bchunk[p] = lt ? c : bchunk[p];
              best[p] = lt ? mn : best[p];
            }
          }
        }

#pragma unroll
        for (int p = 0; p < P; ++p) {
          if constexpr (F64) near[p] = (second[p] <= near_lim(best[p], nt)) && (best[p] < INFINITY);
          else { near[p] = false; tie |= (second[p] == best[p]) && (best[p] < INFINITY); }
        }
        // ---- fix-up inside the winning chunk: position of the minimum; equal distances are left to the pass below ----
        int rot = lane;
        asm volatile("" : "+v"(rot));                          // opaque: keeps the 16 rotated offsets from being hoisted out of the ICP loop (and spilled)
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int j0 = bchunk[p] * CHUNK;
          float sec2 = INFINITY;                               // second smallest distance of the chunk: equal to the smallest ⇔ two equally near targets
          bd[p] = INFINITY; pos[p] = j0;
#pragma unroll
          for (int kk = 0; kk < CHUNK; ++kk) {
            const int k = (kk + rot) & (CHUNK - 1);
            const float2 q = tile[j0 + k];
            const float d = dist2(ax[p], ay[p], q.x, q.y);
            // all in vector registers: a per-element compare into a scalar mask (v_cmp → s_and → s_or chains) was measured
            // 1.5× slower for the whole kernel
            const bool lt = d < bd[p];
            sec2 = __builtin_amdgcn_fmed3f(bd[p], sec2, d);
            pos[p] = lt ? j0 + k : pos[p];
            bd[p] = lt ? d : bd[p];
          }
          if constexpr (F64) { near[p] = (near[p] || (sec2 <= near_lim(bd[p], nt) && bd[p] < INFINITY)) && (base + tid * P + p < n); tie |= near[p]; }
          else tie |= (sec2 == bd[p]) && (bd[p] < INFINITY);
        }
      }
      // ---- equal minima (duplicate or equidistant points, in one chunk or in two): exhaustive pass, lowest ORIGINAL index wins --
      if constexpr (F64) {
        // ---- near ties: every target whose float32 distance is within lim of the winner's is compared in float64
        // (lowest original index on float64 ties).  Chunks are walked with the flagged lanes' own points and windows.
        if (__any(tie)) {
          if (lay.debug_counter && lane == 0) atomicAdd(lay.debug_counter, 1u);
#pragma unroll
          for (int p = 0; p < P; ++p) {
            if (!__any(near[p])) continue;
            const int i = base + tid * P + p;
            const float lim = near[p] ? near_lim(bd[p], nt) : -INFINITY;
            double adx = 0.0, ady = 0.0;
            if (near[p]) {
              const double2 pd = gsrc64[i];
              const double x = pd.x, y = pd.y;
              adx = T[0][0] * x + T[0][1] * y + T[0][3];       // icp.py:116 on the float64 point
              ady = T[1][0] * x + T[1][1] * y + T[1][3];
            }
            double bv = INFINITY;
            int bo = 0x7fffffff, bp = pos[p];
            for (int c = 0; c < nchunks; ++c) {
              const float4 b = aabb[c];
              const float ox = fmaxf(fmaxf(b.x - ax[p], ax[p] - b.z), 0.f);
              const float oy = fmaxf(fmaxf(b.y - ay[p], ay[p] - b.w), 0.f);
              if (!__any(lim >= __builtin_fmaf(oy, oy, ox * ox) * kPruneSafety)) continue;
              for (int k = 0; k < CHUNK; ++k) {
                const int j = c * CHUNK + k;
                const float2 q = tile[j];
                if (dist2(ax[p], ay[p], q.x, q.y) <= lim) {    // rare: a candidate of this lane (padding is at +inf)
                  const int o = tidx[j];
                  const double2 td = gtgt64[j];
                  const double dx = adx - td.x, dy = ady - td.y;
                  const double e = dx * dx + dy * dy;
                  if (e < bv || (e == bv && o < bo)) { bv = e; bo = o; bp = j; }
                }
              }
            }
            if (near[p]) pos[p] = bp;
          }
        }
      } else if (__any(tie)) {
        if (lay.debug_counter && lane == 0) atomicAdd(lay.debug_counter, 1u);
        int oidx[P];
#pragma unroll
        for (int p = 0; p < P; ++p) { bd[p] = INFINITY; oidx[p] = 0x7fffffff; }
        for (int j = 0; j < m; ++j) {
          const float2 q = tile[j];
          const int o = tidx[j];
#pragma unroll
          for (int p = 0; p < P; ++p) {
            const float d = dist2(ax[p], ay[p], q.x, q.y);
            if (d < bd[p] || (d == bd[p] && o < oidx[p])) { bd[p] = d; pos[p] = j; oidx[p] = o; }
          }
        }
      }

#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = base + tid * P + p;
        if (i < n) sh_prev[i] = (uint16_t)pos[p];
      }
      PRUNE_MARK(4);
    }
    if (take) __syncthreads();                               // phase B reads matches that other waves wrote
    PRUNE_MARK(5);

    // ---- phase B: float64 moments of the matched pairs (kept apart from the sweep: the 28 accumulator
    // registers are not live while the search runs) ------------------------------------------------
    PairAcc<2> acc;
    acc.clear();
    for (int pass = 0; pass < passes; ++pass) {
      const int base = pass * per_pass;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = base + tid * P + p;
        if (i < n) {
          const int ps = sh_prev[i];
          double x, y, pd[2], qd[2];
          if constexpr (F64) {                                // the float64 points themselves (original order)
            const double2 pd = gsrc64[i], td = gtgt64[ps];
            x = pd.x; y = pd.y;
            qd[0] = td.x; qd[1] = td.y;
          } else {
            const float2 v = src_lds ? sh_src[i] : ssrc[i];
            const float2 q = tile[ps];
            x = (double)v.x; y = (double)v.y;
            qd[0] = (double)q.x; qd[1] = (double)q.y;
          }
          pd[0] = T[0][0] * x + T[0][1] * y + T[0][3];
          pd[1] = T[1][0] * x + T[1][1] * y + T[1][3];
          acc.add(pd, qd, box);
        }
      }
    }

    PRUNE_MARK(6);
    acc.wave_reduce(box);
    if (lane == 0) acc.store(sh_red + (size_t)wave * kAcc);
    __syncthreads();
    if (wave == 0) {
      constexpr int kS = AccLayout<2>::kSums;
      double col = 0.0;
      {
        // four rows per trip, their loads issued together, sum / min / max chosen with selects: as a rolled loop with a branch
        // per row every load waited for a full LDS round trip (up to 16 per iteration of every problem)
        const int cl = lane < kAcc ? lane : 0;
        const bool is_sum = lane < kS, is_min = lane < kS + 2;
        col = sh_red[cl];
        for (int w0 = 1; w0 < nwaves; w0 += 4) {
          double x[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) x[u] = sh_red[(size_t)(w0 + u < nwaves ? w0 + u : 0) * kAcc + cl];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (!box) { col = (w0 + u < nwaves) ? col + x[u] : col; continue; }      // wave-uniform: without the bounding box every value is a sum
            const double s_ = col + x[u], lo_ = fmin(col, x[u]), hi_ = fmax(col, x[u]);
            col = (w0 + u < nwaves) ? (is_sum ? s_ : (is_min ? lo_ : hi_)) : col;
          }
        }
      }
      double tot[kAcc];
#pragma unroll
      for (int k = 0; k < kAcc; ++k) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(col), k);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(col), k);
        tot[k] = __hiloint2double(hi, lo);
      }
      Moments mo;
      combine_partials<2>(tot, 1, kAcc, box, mo);
      double R[9], pb[3], qb[3];
      step_parts(mo, true, R, pb, qb);
      const double tn = step_times_prev_lane(R, pb, qb, sh_T, sh_red, lane);      // one element of T_step·T_prev per lane (the wave rows in sh_red are consumed)
      const double err = error_from_moments(mo, box);
      StopState st = *sh_stop;
      const bool stop = stop_after_iteration(st, err, args.epsilon, args.stopping_thresh, args.max_iters);
      if (lane < 16) sh_T[lane] = tn;
      if (lane == 0) {
        sh_flag[0] = stop ? 1 : 0;
        sh_flag[3] = 0;                                      // every wave is past its last slab (the barrier above)
        *sh_stop = st;
        *sh_err = err;
      }
    }
    __syncthreads();
    PRUNE_MARK(7);
    if (sh_flag[0]) break;
  }
#ifdef LSLAM_PRUNE_PROFILE
  // cycles per phase, summed over the waves of the launch: [2 + k] of the debug block (printed by the launcher under LSLAM_DEBUG)
  if (lay.debug_counter && lane == 0)
    for (int k = 0; k < 8; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(lay.debug_counter) + 1 + k, prof[k]);
#endif

  if (tid < 16) args.out_T[(int64_t)16 * pair + tid] = sh_T[tid];
  if (tid == 0) {
    args.out_err[pair] = *sh_err;
    args.out_iters[pair] = sh_stop->executed;
  }
  // correspondences of the LAST iteration (icp.py:117, 187-189), in the caller's numbering
  if (args.out_corr)
    for (int i = tid; i < n; i += blockDim.x) args.out_corr[s0 + sc.src_index[ss + i]] = tidx[sh_prev[i]];
}

// ---- launch ------------------------------------------------------------------------------------------
bool icp_pruned_supported(int dim, int64_t max_src_n, int64_t max_tgt_n) {
  return dim == 2 && max_src_n >= 1 && max_src_n <= kPruneMaxPoints && max_tgt_n <= kPruneMaxPoints;
}

hipError_t launch_icp_pruned(IcpBatchArgs a, const PruneScratch& sc, int S, int64_t max_src_n, int64_t max_tgt_n,
                             hipStream_t stream) {
  // 1. sort
  int len = 64;
  while (len < max_src_n || len < max_tgt_n) len <<= 1;
  int keep = -1;                                             // −1: decided per cloud (icp_sort_kernel); 0–3: bit 0 / 1 keeps the target / source order
  if (const char* env = getenv("LSLAM_ICP_PRUNE_KEEP")) keep = atoi(env);
  // the order of the grid cells: 1 = Hilbert curve (default), 0 = Z curve (round 2).  Same results; on the Z curve the chunks that
  // straddle a jump have boxes across half the cloud and survive the box tests of many waves: C4 22.5 -> 17.0 ms with Hilbert
  int curve = 1;
  if (const char* env = getenv("LSLAM_ICP_PRUNE_CURVE")) curve = atoi(env);
  // LDS of the sort: the staged cloud, a counter per cell (the bitonic keys of the fall-back alias them), cell and position per point
  const int64_t big = max_src_n > max_tgt_n ? max_src_n : max_tgt_n;
  const int cap = (int)(((big + 63) / 64) * 64);
  int g_max = max_tgt_n >= 2048 ? 32 : (max_tgt_n >= 512 ? 16 : 8);
  if (curve >= 1) g_max *= 2;
  const int cur_words = ((g_max * g_max + 64 > len ? g_max * g_max + 64 : len) + 3) & ~3;
  const size_t sort_lds = sizeof(float2) * (size_t)cap + sizeof(unsigned) * (size_t)cur_words + 2 * sizeof(uint16_t) * (size_t)cap;
  // A batch that holds problems on both sides of the certificates' threshold is iterated by two launches, one per class: every bit of
  // a problem's result then depends on the problem alone (tests/test_gpu_icp.py: …batch_independent), and the small ones get the LDS
  // geometry of small problems.  Not when LSLAM_ICP_QUEUE_CERTIFY forces one kind of kernel on everything.
  const bool split = sc.order && sc.queue && big >= kQueueCertifyPoints && !getenv("LSLAM_ICP_QUEUE_CERTIFY");
  PruneScratch scs = sc;
  if (!split) scs.order = nullptr;
  // which problems the queue kernel will iterate on running sums (launch_icp_queue decides the same way): they need no sorted float64 copies
  int skip64 = 0;
  {
    bool queue_on = sc.queue != nullptr;
    if (const char* env = getenv("LSLAM_ICP_PRUNE_QUEUE")) queue_on = queue_on && atoi(env) != 0;
    bool incr_ok = queue_on && a.src64 && !(a.flags & kFlagNormalize);
    if (const char* env = getenv("LSLAM_ICP_QUEUE_INCR")) incr_ok = incr_ok && atoi(env) != 0;
    if (incr_ok) {
      if (const char* env = getenv("LSLAM_ICP_QUEUE_CERTIFY")) skip64 = atoi(env) != 0 ? 2 : 0;
      else skip64 = big >= kQueueCertifyPoints ? 1 : 0;
    }
  }
  if (split) {                                               // one memset for both launches: draw counters [0] [44], debug counters, class counts [48] [49] (not the sticky [50])
    const hipError_t ee = hipMemsetAsync(sc.queue, 0, 50 * sizeof(unsigned), stream);
    if (ee != hipSuccess) return ee;
  }
  {
    auto sort_s = [&](auto kernel, int threads) -> hipError_t {
      hipError_t ee = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds);
      if (ee != hipSuccess) return ee;
      hipLaunchKernelGGL(kernel, dim3(S), dim3(threads), sort_lds, stream, a, scs, len, keep, curve, cap, cur_words, skip64);
      return hipSuccess;
    };
    const hipError_t ee = big >= 2048 ? sort_s(icp_sort_kernel<512>, 512) : sort_s(icp_sort_kernel<256>, 256);
    if (ee != hipSuccess) return ee;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // 2. iterate: the task-queue kernel (lslam_icp_queue.hip: problems in LDS slots, waves draw tasks, no barriers; from 2048 points
  // per cloud it also keeps a source's match from one iteration to the next while a proof allows it).  Round 4, same boxes: C3
  // sequence 0.77 ms against 0.87 (float64-exact 0.91 against 0.96), C4 with 50 iterations 2.15 ms against 16.0 (float64-exact 2.4
  // against 17.9).  LSLAM_ICP_PRUNE_QUEUE=0 keeps one problem per workgroup (the kernel below).
  {
    bool queue = true;
    if (const char* env = getenv("LSLAM_ICP_PRUNE_QUEUE")) queue = atoi(env) != 0;
    if (queue && sc.queue) {
      if (!split) return launch_icp_queue(a, scs, S, max_src_n, max_tgt_n, stream);
      const hipError_t ee = launch_icp_queue(a, scs, S, max_src_n, max_tgt_n, stream, 1, true);
      if (ee != hipSuccess) return ee;
      const int64_t cap_small = kQueueCertifyPoints - 1;
      return launch_icp_queue(a, scs, S, max_src_n < cap_small ? max_src_n : cap_small, max_tgt_n < cap_small ? max_tgt_n : cap_small, stream, 0, true);
    }
  }
  // One source per lane, and about four slabs of 64 sources per wave: a workgroup is then a quarter of the waves a problem
  // "needs", several problems share a CU (eight LiDAR-sized ones instead of three), and their barriers and one-wave solves —
  // half of a wave's time in the round-2 geometry — overlap with each other's search; the waves take the slabs from the
  // counter.  tools/prune_threads.py (10 forced iterations, 4096 problems): 384 points 0.41 ms at 128 lanes against 0.43 at
  // two points per lane; 768: 0.69 at 256 against 0.86; 1536: 1.35-1.43 at 256-384 against 1.70; 2048: 1.97 at 512 against
  // 2.44 at 1024; 3072: 3.00 at 768 against 3.09; 4096: 1024 lanes as before.  C3: 0.85 ms against 1.19.
  int P = 1;
  if (const char* env = getenv("LSLAM_ICP_PRUNE_P")) P = atoi(env);
  if (P != 1 && P != 2 && P != 4) P = 1;
  const int natural = (int)((((max_src_n + P - 1) / P + 63) / 64) * 64);
  int threads = (int)((max_src_n / (4 * P)) / 64 * 64);       // rounded down: 1081 points -> 256 lanes (0.85 ms on C3; 320: 0.93)
  if (threads < 128) threads = 128;
  if (S < 256 || threads > natural) threads = natural;          // fewer problems than CUs: every lane the problem can use
  if (threads < 64) threads = 64;
  if (threads > kIcpBatchMaxThreads) threads = kIcpBatchMaxThreads;      // more sources: more slabs per wave
  if (const char* env = getenv("LSLAM_ICP_PRUNE_THREADS")) { const int t = atoi(env); if (t >= 64 && t <= kIcpBatchMaxThreads && t % 64 == 0) threads = t; }
  const int padded = (int)(((max_tgt_n + kPruneChunk - 1) / kPruneChunk) * kPruneChunk);
  const int nchunks = padded / kPruneChunk;
  PruneLayout lay{};
  auto up = [](size_t b) { return (b + 15) & ~(size_t)15; };
  size_t off = (size_t)padded * sizeof(float2);
  lay.aabb_off = (uint32_t)off; off += (size_t)nchunks * sizeof(float4);
  lay.slab_off = (uint32_t)off; off += (size_t)((max_src_n + 64 * P - 1) / (64 * P)) * sizeof(float4);
  lay.src_off = (uint32_t)off; off += up((size_t)max_src_n * sizeof(float2));
  lay.prev_off = (uint32_t)off; off += up((size_t)max_src_n * sizeof(uint16_t));
  lay.red_off = (uint32_t)off;
  const size_t smem = off + sizeof(double) * ((size_t)kIcpBatchMaxWaves * AccLayout<2>::kTotal + 16 + 1) + sizeof(StopState) + 16;
  auto launch = [&](auto kernel) -> hipError_t {
    if (smem > 48 * 1024) {
      hipError_t ee = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (ee != hipSuccess) return ee;
    }
    static unsigned* dbg = nullptr;
    if (getenv("LSLAM_DEBUG")) {
      if (!dbg) { (void)hipMalloc(&dbg, 80); }
      else {
        unsigned long long h[10];
        (void)hipMemcpy(h, dbg, 80, hipMemcpyDeviceToHost);
        fprintf(stderr, "[lslam] exhaustive tie passes of the previous launch: %u\n", (unsigned)h[0]);
#ifdef LSLAM_PRUNE_PROFILE
        double tot = 0; for (int k = 0; k < 8; ++k) tot += (double)h[1 + k];
        fprintf(stderr, "[lslam] pruned kernel, share of wave cycles: transform+box %.3f  seed %.3f  box tests %.3f  lists %.3f  ties+store %.3f  (pass loop) %.3f  moments %.3f  reduce+solve+barriers %.3f\n",
                h[1] / tot, h[2] / tot, h[3] / tot, h[4] / tot, h[5] / tot, h[6] / tot, h[7] / tot, h[8] / tot);
#endif
      }
      (void)hipMemset(dbg, 0, 80);
      lay.debug_counter = dbg;
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, smem);
      fprintf(stderr, "[lslam] icp_pruned_kernel: %d threads, %zu B LDS, %d workgroups per CU\n", threads, smem, nb);
    }
    hipLaunchKernelGGL(kernel, dim3(S), dim3(threads), smem, stream, a, sc, lay);
    return hipGetLastError();
  };
  if ((a.src64 == nullptr) != (a.tgt64 == nullptr)) return hipErrorInvalidValue;
  // per-lane lists (LIST) against whole-wave chunk evaluation, same box: C4 22.4 vs 24.1 ms, float64-exact 23.9 vs 27.5 ms;
  // the LiDAR-sized C3 batch in beam order 1.21 vs 1.37 ms (float64-exact 1.53 vs 2.12 ms).  Only Morton-sorted clouds of
  // LiDAR size in float32 are 3 % slower with lists (1.98 vs 1.91 ms): not worth a second rule
  bool list = true;
  if (const char* env = getenv("LSLAM_ICP_PRUNE_LIST")) list = atoi(env) != 0;
  if (list) {
    if (a.src64) {
      if (P == 1) return launch(icp_pruned_kernel<1, true, true>);
      if (P == 2) return launch(icp_pruned_kernel<2, true, true>);
      return launch(icp_pruned_kernel<4, true, true>);
    }
    if (P == 1) return launch(icp_pruned_kernel<1, false, true>);
    if (P == 2) return launch(icp_pruned_kernel<2, false, true>);
    return launch(icp_pruned_kernel<4, false, true>);
  }
  if (a.src64) {
    if (P == 1) return launch(icp_pruned_kernel<1, true, false>);
    if (P == 2) return launch(icp_pruned_kernel<2, true, false>);
    return launch(icp_pruned_kernel<4, true, false>);
  }
  if (P == 1) return launch(icp_pruned_kernel<1, false, false>);
  if (P == 2) return launch(icp_pruned_kernel<2, false, false>);
  return launch(icp_pruned_kernel<4, false, false>);
}

}  // namespace lslam

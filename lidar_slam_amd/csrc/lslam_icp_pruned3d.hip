// lslam_icp_pruned3d.hip — the batched ICP with EXACT box pruning for 3-D clouds (the planar kernel of
// lslam_icp_pruned.hip carried to three dimensions; same operator, same results as icp_batch_kernel<3,·>:
// modules/icp.py:123-189, every source gets its exact float32 nearest target, lowest original index on ties).
//
// icp_sort3_kernel puts both clouds of a problem into Morton order of a uniform grid over the target's box (16³ cells,
// bitonic sort of (cell, index) keys in LDS, once per problem).  icp_pruned3_kernel then iterates on chip: per wave the
// box of its transformed sources and its largest current bound; 64 chunk boxes (16 consecutive sorted targets each)
// tested per ballot; the survivors re-tested with every lane's own points and current bounds, then evaluated with the
// exact recipe; smallest and second smallest chunk minimum per source, equal minima detected from the values and
// resolved by a rare exhaustive pass ordered by original index; float64 moments in a second phase; Jacobi-SVD solve
// and the reference's stop rules by wave 0.  One workgroup per problem, up to 4096 targets (64 KB of LDS) and 8191
// sources per problem.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "lslam_device.h"
#include "lslam_kernels.h"

namespace lslam {

namespace {

constexpr int kChunk3 = 16;
constexpr int kSort3Threads = 256;
constexpr float kSafety3 = 0.99999f;

__device__ __forceinline__ unsigned morton4x3(unsigned x, unsigned y, unsigned z) {      // interleave three 4-bit coordinates
  unsigned r = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) r |= ((x >> b) & 1u) << (3 * b) | ((y >> b) & 1u) << (3 * b + 1) | ((z >> b) & 1u) << (3 * b + 2);
  return r;
}
__device__ __forceinline__ float wmin3(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wmax3(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float next_up3(float v) {         // the next float above a non-negative finite float (or +inf)
  return v < INFINITY ? __uint_as_float(__float_as_uint(v) + 1u) : v;
}
__device__ inline void bitonic_sort3(unsigned* keys, int len) {
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

}  // namespace

// One workgroup per problem: Morton-sorted copies of target and source (float4: x, y, z, 0) + their original indices.
__global__ void __launch_bounds__(kSort3Threads)
icp_sort3_kernel(IcpBatchArgs a, PruneScratch3 sc, int len_pow2, int curve) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned* keys = reinterpret_cast<unsigned*>(smem);
  __shared__ float sbox[6];
  __shared__ float red[kSort3Threads / 64][6];
  const int pair = blockIdx.x, tid = threadIdx.x;
  const int64_t s0 = a.src_off[pair], t0 = a.tgt_off[pair];
  const int n = (int)(a.src_off[pair + 1] - s0), m = (int)(a.tgt_off[pair + 1] - t0);
  const float* src = a.src + 3 * s0;
  const float* tgt = a.tgt + 3 * t0;
  const int64_t ss = (int64_t)pair * sc.src_stride, ts = (int64_t)pair * sc.tgt_stride;      // pair-local scratch slots

  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int j = tid; j < m; j += kSort3Threads)
#pragma unroll
    for (int d = 0; d < 3; ++d) { const float v = tgt[3 * j + d]; lo[d] = fminf(lo[d], v); hi[d] = fmaxf(hi[d], v); }
#pragma unroll
  for (int d = 0; d < 3; ++d) { lo[d] = wmin3(lo[d]); hi[d] = wmax3(hi[d]); }
  if ((tid & 63) == 0)
#pragma unroll
    for (int d = 0; d < 3; ++d) { red[tid >> 6][d] = lo[d]; red[tid >> 6][3 + d] = hi[d]; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < kSort3Threads / 64; ++w)
#pragma unroll
      for (int d = 0; d < 3; ++d) { lo[d] = fminf(lo[d], red[w][d]); hi[d] = fmaxf(hi[d], red[w][3 + d]); }
#pragma unroll
    for (int d = 0; d < 3; ++d) { sbox[d] = lo[d]; sbox[3 + d] = hi[d]; }
  }
  __syncthreads();
  const int G = m >= 2048 ? 16 : 8;
  float sc3[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) { lo[d] = sbox[d]; hi[d] = sbox[3 + d]; sc3[d] = (hi[d] > lo[d]) ? (float)G / (hi[d] - lo[d]) : 0.f; }
  auto cell_of = [&](float x, float y, float z) -> unsigned {
    int gx = (int)((x - lo[0]) * sc3[0]), gy = (int)((y - lo[1]) * sc3[1]), gz = (int)((z - lo[2]) * sc3[2]);
    gx = min(max(gx, 0), G - 1); gy = min(max(gy, 0), G - 1); gz = min(max(gz, 0), G - 1);
    if (curve == 1) { const unsigned c[3] = {(unsigned)gx, (unsigned)gy, (unsigned)gz}; return hilbert_key<3, 4>(c); }   // (G = 8: the curve of the coarser grid, cells doubled)
    return morton4x3((unsigned)gx, (unsigned)gy, (unsigned)gz);
  };

  // ---- target ----
  for (int j = tid; j < len_pow2; j += kSort3Threads) {
    unsigned k = 0xffffffffu;
    if (j < m) k = (cell_of(tgt[3 * j], tgt[3 * j + 1], tgt[3 * j + 2]) << 13) | (unsigned)j;
    keys[j] = k;
  }
  bitonic_sort3(keys, len_pow2);
  for (int j = tid; j < m; j += kSort3Threads) {
    const unsigned o = keys[j] & 8191u;
    sc.tgt_sorted[ts + j] = make_float4(tgt[3 * o], tgt[3 * o + 1], tgt[3 * o + 2], 0.f);
    sc.tgt_index[ts + j] = (uint16_t)o;
    if (a.tgt64)
      for (int r = 0; r < 3; ++r) sc.tgt64_sorted[3 * (ts + j) + r] = a.tgt64[3 * (t0 + o) + r];
  }
  __syncthreads();
  // ---- source, placed by its initial transform into the target's grid ----
  const double* T = a.init_T + (int64_t)16 * pair;
  float t[3][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) t[r][c] = (float)T[4 * r + c];
  for (int i = tid; i < len_pow2; i += kSort3Threads) {
    unsigned k = 0xffffffffu;
    if (i < n) {
      const float x = src[3 * i], y = src[3 * i + 1], z = src[3 * i + 2];
      k = (cell_of(t[0][0] * x + t[0][1] * y + t[0][2] * z + t[0][3], t[1][0] * x + t[1][1] * y + t[1][2] * z + t[1][3],
                   t[2][0] * x + t[2][1] * y + t[2][2] * z + t[2][3]) << 13) | (unsigned)i;
    }
    keys[i] = k;
  }
  bitonic_sort3(keys, len_pow2);
  for (int i = tid; i < n; i += kSort3Threads) {
    const unsigned o = keys[i] & 8191u;
    sc.src_sorted[ss + i] = make_float4(src[3 * o], src[3 * o + 1], src[3 * o + 2], 0.f);
    sc.src_index[ss + i] = (uint16_t)o;
    if (a.src64)
      for (int r = 0; r < 3; ++r) sc.src64_sorted[3 * (ss + i) + r] = a.src64[3 * (s0 + o) + r];
  }
}

// F64: the float64-exact mode (lslam_device.h: NearTie), as in icp_pruned_kernel
// One chunk of 16 sorted targets against ONE point of this lane, the chunk chosen by the lane (a gather; see prune_eval_chunk in
// lslam_icp_pruned.hip): (bd, sec) = the smallest and second smallest distance over everything compared for the point so far,
// at = the LDS address of the nearest target; lane l visits target (k XOR l) at step k, four reads in flight.
constexpr int kTile3Align = kChunk3 * 16;
__device__ __forceinline__ void prune3_eval_chunk(unsigned tile0, int chunk, int lane, const float (&a)[3], float& bd, float& sec, unsigned& at) {
  typedef float __attribute__((ext_vector_type(4))) Vec;
  typedef const __attribute__((address_space(3))) Vec* LdsPoint;
  unsigned r = (unsigned)lane & (unsigned)(kChunk3 - 1);
  asm volatile("" : "+v"(r));                                  // opaque: keeps the 16 offsets from being hoisted out of the ICP loop
  const unsigned first = tile0 + (unsigned)chunk * (kChunk3 * 16u) + r * 16u;
#pragma unroll 1
  for (int h = 0; h < kChunk3; h += 4) {
    unsigned a4[4];
    Vec v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { a4[c] = first ^ ((unsigned)(h + c) * 16u); v[c] = *(LdsPoint)(size_t)a4[c]; }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d = dist2(a[0], a[1], a[2], v[c].x, v[c].y, v[c].z);
      const bool lt = d < bd;
      sec = __builtin_amdgcn_fmed3f(bd, sec, d);
      at = lt ? a4[c] : at;
      bd = lt ? d : bd;
    }
  }
}

// LIST: per-lane chunk lists with gathered reads instead of whole-wave chunk evaluation (as in the planar kernel)
template <int P, bool F64, bool LIST>
__global__ void __launch_bounds__(kIcpBatchMaxThreads)
icp_pruned3_kernel(IcpBatchArgs args, PruneScratch3 sc, PruneLayout lay) {
  constexpr int CHUNK = kChunk3;
  constexpr int kAcc = AccLayout<3>::kTotal;
  extern __shared__ __align__(kTile3Align) unsigned char smem[];
  float4* tile = reinterpret_cast<float4*>(smem);                                   // sorted targets, padded
  float4* aabb = reinterpret_cast<float4*>(smem + lay.aabb_off);                    // per chunk: lo (x, y, z), hi (x, y, z)
  float4* sh_src = reinterpret_cast<float4*>(smem + lay.src_off);                   // sorted sources (when they fit)
  uint16_t* sh_prev = reinterpret_cast<uint16_t*>(smem + lay.prev_off);             // last match (sorted target position) of every source
  double* sh_red = reinterpret_cast<double*>(smem + lay.red_off);                   // [waves][kAcc]
  double* sh_T = sh_red + (size_t)kIcpBatchMaxWaves * kAcc;
  double* sh_err = sh_T + 16;
  StopState* sh_stop = reinterpret_cast<StopState*>(sh_err + 1);
  int* sh_flag = reinterpret_cast<int*>(sh_stop + 1);

  const int pair = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int64_t s0 = args.src_off[pair], t0 = args.tgt_off[pair];
  const int n = (int)(args.src_off[pair + 1] - s0), m = (int)(args.tgt_off[pair + 1] - t0);
  const int64_t ss = (int64_t)pair * sc.src_stride, ts = (int64_t)pair * sc.tgt_stride;
  const float4* __restrict__ ssrc = sc.src_sorted + ss;
  const uint16_t* __restrict__ tidx = sc.tgt_index + ts;       // original target index: ties and final correspondences only
  const bool box = (args.flags & kFlagNormalize) != 0;
  const int nchunks = (m + CHUNK - 1) / CHUNK;

  const double* __restrict__ gsrc64 = F64 ? sc.src64_sorted + 3 * ss : nullptr;     // the float64 clouds in the SORTED order (icp_sort3_kernel)
  const double* __restrict__ gtgt64 = F64 ? sc.tgt64_sorted + 3 * ts : nullptr;
  if (tid < 16) sh_T[tid] = args.init_T[(int64_t)16 * pair + tid];
  if (F64 && tid == 0) { sh_flag[1] = 0; sh_flag[2] = 0; }   // bits of max |t|², max |p|²
  float tmax = 0.f, smax = 0.f;
  for (int j = tid; j < nchunks * CHUNK; j += blockDim.x) {
    const float4 v = (j < m) ? sc.tgt_sorted[ts + j] : make_float4(kFar, kFar, kFar, 0.f);
    tile[j] = v;
    if (F64 && j < m) tmax = fmaxf(tmax, __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x)));
  }
  const bool src_lds = lay.src_off != 0xffffffffu;
  for (int i = tid; i < n; i += blockDim.x) {
    const float4 v = ssrc[i];
    if (src_lds) sh_src[i] = v;
    sh_prev[i] = 0xffff;
    if (F64) smax = fmaxf(smax, __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x)));
  }
  __syncthreads();
  if constexpr (F64) {
    tmax = wmax3(tmax); smax = wmax3(smax);
    if (lane == 0) { atomicMax(&sh_flag[1], __float_as_int(tmax)); atomicMax(&sh_flag[2], __float_as_int(smax)); }
  }
  for (int c = tid; c < nchunks; c += blockDim.x) {          // boxes over the real points only
    float4 lo = make_float4(INFINITY, INFINITY, INFINITY, 0.f), hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.f);
    for (int k = 0; k < CHUNK && c * CHUNK + k < m; ++k) {
      const float4 q = tile[c * CHUNK + k];
      lo.x = fminf(lo.x, q.x); lo.y = fminf(lo.y, q.y); lo.z = fminf(lo.z, q.z);
      hi.x = fmaxf(hi.x, q.x); hi.y = fmaxf(hi.y, q.y); hi.z = fmaxf(hi.z, q.z);
    }
    aabb[2 * c] = lo; aabb[2 * c + 1] = hi;
  }
  if (tid == 0) { StopState st0; stop_init(st0); *sh_stop = st0; *sh_err = 0.0; }
  __syncthreads();
  float Rt = 0.f, Rs = 0.f;
  if constexpr (F64) {
    Rt = sqrtf(__int_as_float(sh_flag[1])) * 1.0001f;
    Rs = sqrtf(__int_as_float(sh_flag[2])) * 1.0001f;
  }
  const int per_pass = blockDim.x * P;
  const int passes = n > 0 ? (n + per_pass - 1) / per_pass : 1;

  // lower bound of the distance² from the box [lo, hi] to the box [blo, bhi] (a point when lo == hi)
  auto box_dist = [](const float (&lo)[3], const float (&hi)[3], const float4 blo, const float4 bhi) -> float {
    const float dx = fmaxf(fmaxf(blo.x - hi[0], lo[0] - bhi.x), 0.f);
    const float dy = fmaxf(fmaxf(blo.y - hi[1], lo[1] - bhi.y), 0.f);
    const float dz = fmaxf(fmaxf(blo.z - hi[2], lo[2] - bhi.z), 0.f);
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) * kSafety3;
  };

  for (;;) {
    double T[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) T[r][c] = wave_uniform(sh_T[4 * r + c]);
    NearTie nt = near_tie_setup(false, 0.f, 0.f, 0.f);
    if constexpr (F64) {
      double f2 = 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) f2 += T[r][c] * T[r][c];
      nt = near_tie_setup(true, Rt, Rs, sqrtf((float)f2) * 1.0001f);
    }
    // the bound a chunk must beat: the lane's best itself, or in float64-exact mode everything within its near-tie window
    auto reach = [&](float b) { return F64 ? __builtin_fmaf(b, nt.K1, nt.C) : b; };

    // ---- phase A: nearest neighbours (float32); the matches go to LDS (they also seed the next iteration)
    for (int pass = 0; pass < passes; ++pass) {
      const int base = pass * per_pass;
      if (base + (tid & ~63) * P >= n) continue;             // this wave owns no source in this pass
      float a[P][3];
      float wlo[3] = {INFINITY, INFINITY, INFINITY}, whi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = base + tid * P + p;
        a[p][0] = a[p][1] = a[p][2] = 0.f;
        if (i < n) {
          const float4 v = src_lds ? sh_src[i] : ssrc[i];
#pragma unroll
          for (int r = 0; r < 3; ++r) {                      // icp.py:116
            a[p][r] = (float)(T[r][0] * (double)v.x + T[r][1] * (double)v.y + T[r][2] * (double)v.z + T[r][3]);
            wlo[r] = fminf(wlo[r], a[p][r]); whi[r] = fmaxf(whi[r], a[p][r]);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) { wlo[r] = wmin3(wlo[r]); whi[r] = wmax3(whi[r]); }

      bool tie = false;
      bool near[P];                                          // float64-exact mode: another target within the near-tie window of the winner
      int pos[P];
      float bd[P];
      if constexpr (LIST) {
        typedef const __attribute__((address_space(3))) float4* LdsTile;
        const unsigned tile0 = (unsigned)(size_t)(LdsTile)(const void*)tile;
        // ---- seed: the chunk of last iteration's match (first iteration: of the target at the same rank along the curve), compared
        // target by target by the lane itself
        float sec[P];
        unsigned at[P];
        int own[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int i = base + tid * P + p;
          bd[p] = (i < n) ? INFINITY : -INFINITY;            // lanes without a point never ask for a chunk
          sec[p] = INFINITY; at[p] = tile0; own[p] = -1;
          if (i < n) {
            int pv = sh_prev[i];
            if (pv == 0xffff) pv = (int)(((long long)i * m) / n);
            own[p] = pv / CHUNK;
            prune3_eval_chunk(tile0, own[p], lane, a[p], bd[p], sec[p], at[p]);
          }
        }
        float wb = -INFINITY;
#pragma unroll
        for (int p = 0; p < P; ++p) wb = fmaxf(wb, bd[p]);
        wb = wmax3(wb);
        // ---- sweep: 64 chunk boxes per ballot against the wave's box and largest bound; each survivor is tested against every
        // point with its current bound; what a point still needs goes into its 64-bit list, and every lane works through its lists
        for (int c0 = 0; c0 < nchunks; c0 += 64) {
          bool cand = false;
          if (c0 + lane < nchunks) cand = reach(wb) >= box_dist(wlo, whi, aabb[2 * (c0 + lane)], aabb[2 * (c0 + lane) + 1]);
          unsigned long long todo = __ballot(cand);
          if (!todo) continue;
          unsigned long long list[P];
#pragma unroll
          for (int p = 0; p < P; ++p) list[p] = 0ull;
          while (todo) {
            const int t = __builtin_ctzll(todo);
            todo &= todo - 1;
            const float4 blo = aabb[2 * (c0 + t)], bhi = aabb[2 * (c0 + t) + 1];
            const unsigned long long bit = 1ull << t;
#pragma unroll
            for (int p = 0; p < P; ++p) {
              // the seed chunk was compared already: a second visit would read as a tie of the point with itself
              const bool need = reach(bd[p]) >= box_dist(a[p], a[p], blo, bhi) && own[p] != c0 + t;
              list[p] |= need ? bit : 0ull;
            }
          }
#pragma unroll
          for (int p = 0; p < P; ++p) {
            unsigned long long mk = list[p];
            while (__any(mk != 0ull)) {
              if (mk != 0ull) {
                const int t = __builtin_ctzll(mk);
                mk &= mk - 1;
                prune3_eval_chunk(tile0, c0 + t, lane, a[p], bd[p], sec[p], at[p]);
              }
            }
          }
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const bool owns = base + tid * P + p < n;
          pos[p] = (int)((at[p] - tile0) >> 4);
          if constexpr (F64) { near[p] = owns && bd[p] < INFINITY && sec[p] <= near_lim(bd[p], nt); tie |= near[p]; }
          else { near[p] = false; tie |= owns && bd[p] < INFINITY && sec[p] == bd[p]; }
        }
      } else {
        // ---- start from last iteration's match (first iteration: the target at the same rank along the curve) ----
        float best[P], second[P];                              // smallest and second smallest chunk minimum seen so far
        int bchunk[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int i = base + tid * P + p;
          best[p] = (i < n) ? INFINITY : -INFINITY;            // lanes without a point never ask for a chunk
          second[p] = INFINITY;
          bchunk[p] = 0;
          if (i < n) {
            int pv = sh_prev[i];
            if (pv == 0xffff) pv = (int)(((long long)i * m) / n);
            const float4 q = tile[pv];
            // a bound, not yet a candidate: its chunk will beat it strictly (float64-exact mode: by more than the near-tie window)
            const float ds = dist2(a[p][0], a[p][1], a[p][2], q.x, q.y, q.z);
            best[p] = F64 ? next_up3(reach(ds) * 1.000001f) : next_up3(ds);
            bchunk[p] = pv / CHUNK;
          }
        }

        // ---- sweep: 64 chunk boxes per ballot against the wave's box and largest bound; survivors get a second look with
        // every lane's own points and current bounds, then are evaluated with the exact recipe
        float wb = -INFINITY;
#pragma unroll
        for (int p = 0; p < P; ++p) wb = fmaxf(wb, best[p]);
        wb = wmax3(wb);
        for (int c0 = 0; c0 < nchunks; c0 += 64) {
          bool cand = false;
          if (c0 + lane < nchunks) cand = reach(wb) >= box_dist(wlo, whi, aabb[2 * (c0 + lane)], aabb[2 * (c0 + lane) + 1]);
          unsigned long long todo = __ballot(cand);
          while (todo) {
            const int c = c0 + __builtin_ctzll(todo);
            todo &= todo - 1;
            const float4 blo = aabb[2 * c], bhi = aabb[2 * c + 1];
            bool need_own = false;
#pragma unroll
            for (int p = 0; p < P; ++p) need_own |= reach(best[p]) >= box_dist(a[p], a[p], blo, bhi);
            if (!__any(need_own)) continue;                    // nothing in this chunk can beat or tie any lane
#pragma unroll
            for (int p = 0; p < P; ++p) {
              float mn = INFINITY;
#pragma unroll
              for (int k = 0; k < CHUNK; k += 2) {
                const float4 q0 = tile[c * CHUNK + k], q1 = tile[c * CHUNK + k + 1];
                mn = fminf(fminf(dist2(a[p][0], a[p][1], a[p][2], q0.x, q0.y, q0.z), dist2(a[p][0], a[p][1], a[p][2], q1.x, q1.y, q1.z)), mn);
              }
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
        asm volatile("" : "+v"(rot));                          // opaque: keeps the 16 rotated offsets from being hoisted out of the ICP loop
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int j0 = bchunk[p] * CHUNK;
          float sec2 = INFINITY;
          bd[p] = INFINITY; pos[p] = j0;
#pragma unroll
          for (int kk = 0; kk < CHUNK; ++kk) {
            const int k = (kk + rot) & (CHUNK - 1);
            const float4 q = tile[j0 + k];
            const float d = dist2(a[p][0], a[p][1], a[p][2], q.x, q.y, q.z);
            const bool lt = d < bd[p];
            sec2 = __builtin_amdgcn_fmed3f(bd[p], sec2, d);
            pos[p] = lt ? j0 + k : pos[p];
            bd[p] = lt ? d : bd[p];
          }
          if constexpr (F64) { near[p] = (near[p] || (sec2 <= near_lim(bd[p], nt) && bd[p] < INFINITY)) && (base + tid * P + p < n); tie |= near[p]; }
          else tie |= (sec2 == bd[p]) && (bd[p] < INFINITY);
        }
      }
      // ---- equal minima (duplicate or equidistant points): exhaustive pass, lowest ORIGINAL index wins --
      if constexpr (F64) {
        // ---- near ties: every target whose float32 distance is within lim of the winner's is compared in float64
        // (lowest original index on float64 ties); chunks are walked with the flagged lanes' own points and windows
        if (__any(tie)) {
#pragma unroll
          for (int p = 0; p < P; ++p) {
            if (!__any(near[p])) continue;
            const int i = base + tid * P + p;
            const float lim = near[p] ? near_lim(bd[p], nt) : -INFINITY;
            double ad[3] = {0.0, 0.0, 0.0};
            if (near[p]) {
              const double x = gsrc64[3 * i], y = gsrc64[3 * i + 1], z = gsrc64[3 * i + 2];
#pragma unroll
              for (int r = 0; r < 3; ++r) ad[r] = T[r][0] * x + T[r][1] * y + T[r][2] * z + T[r][3];      // icp.py:116 on the float64 point
            }
            double bv = INFINITY;
            int bo = 0x7fffffff, bp = pos[p];
            for (int c = 0; c < nchunks; ++c) {
              if (!__any(lim >= box_dist(a[p], a[p], aabb[2 * c], aabb[2 * c + 1]))) continue;
              for (int k = 0; k < CHUNK; ++k) {
                const int j = c * CHUNK + k;
                const float4 q = tile[j];
                if (dist2(a[p][0], a[p][1], a[p][2], q.x, q.y, q.z) <= lim) {      // rare: a candidate of this lane (padding is at +inf)
                  const int o = tidx[j];
                  double e = 0.0;
#pragma unroll
                  for (int r = 0; r < 3; ++r) { const double df = ad[r] - gtgt64[3 * j + r]; e += df * df; }
                  if (e < bv || (e == bv && o < bo)) { bv = e; bo = o; bp = j; }
                }
              }
            }
            if (near[p]) pos[p] = bp;
          }
        }
      } else if (__any(tie)) {
        int oidx[P];
#pragma unroll
        for (int p = 0; p < P; ++p) { bd[p] = INFINITY; oidx[p] = 0x7fffffff; }
        for (int j = 0; j < m; ++j) {
          const float4 q = tile[j];
          const int o = tidx[j];
#pragma unroll
          for (int p = 0; p < P; ++p) {
            const float d = dist2(a[p][0], a[p][1], a[p][2], q.x, q.y, q.z);
            if (d < bd[p] || (d == bd[p] && o < oidx[p])) { bd[p] = d; pos[p] = j; oidx[p] = o; }
          }
        }
      }
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = base + tid * P + p;
        if (i < n) sh_prev[i] = (uint16_t)pos[p];
      }
    }

    // ---- phase B: float64 moments of the matched pairs ----------------------------------------------------------
    PairAcc<3> acc;
    acc.clear();
    for (int pass = 0; pass < passes; ++pass) {
      const int base = pass * per_pass;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = base + tid * P + p;
        if (i < n) {
          double x, y, z, pd[3], qd[3];
          if constexpr (F64) {                                // the float64 points themselves (original order)
            const int to = sh_prev[i];
            x = gsrc64[3 * i]; y = gsrc64[3 * i + 1]; z = gsrc64[3 * i + 2];
            qd[0] = gtgt64[3 * to]; qd[1] = gtgt64[3 * to + 1]; qd[2] = gtgt64[3 * to + 2];
          } else {
            const float4 v = src_lds ? sh_src[i] : ssrc[i];
            const float4 q = tile[sh_prev[i]];
            x = (double)v.x; y = (double)v.y; z = (double)v.z;
            qd[0] = (double)q.x; qd[1] = (double)q.y; qd[2] = (double)q.z;
          }
#pragma unroll
          for (int r = 0; r < 3; ++r) pd[r] = T[r][0] * x + T[r][1] * y + T[r][2] * z + T[r][3];
          acc.add(pd, qd, box);
        }
      }
    }
    acc.wave_reduce(box);
    if (lane == 0) acc.store(sh_red + (size_t)wave * kAcc);
    __syncthreads();
    if (wave == 0) {
      constexpr int kS = AccLayout<3>::kSums;
      double col = 0.0;
      {
        // four rows per trip, their loads issued together, sum / min / max chosen with selects: as a rolled loop with a branch
        // per row every load waited for a full LDS round trip (up to 16 per iteration of every problem)
        const int cl = lane < kAcc ? lane : 0;
        const bool is_sum = lane < kS, is_min = lane < kS + 3;
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
      combine_partials<3>(tot, 1, kAcc, box, mo);
      double R[9], pb[3], qb[3];
      step_parts(mo, false, R, pb, qb);
      const double tn = step_times_prev_lane(R, pb, qb, sh_T, sh_red, lane);      // one element of T_step·T_prev per lane (the wave rows in sh_red are consumed)
      const double err = error_from_moments(mo, box);
      StopState st = *sh_stop;
      const bool stop = stop_after_iteration(st, err, args.epsilon, args.stopping_thresh, args.max_iters);
      if (lane < 16) sh_T[lane] = tn;
      if (lane == 0) {
        sh_flag[0] = stop ? 1 : 0;
        *sh_stop = st;
        *sh_err = err;
      }
    }
    __syncthreads();
    if (sh_flag[0]) break;
  }

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
bool icp_pruned3_supported(int64_t max_src_n, int64_t max_tgt_n) {
  return max_src_n >= 1 && max_src_n <= kPrune3MaxSrc && max_tgt_n >= 1 && max_tgt_n <= kPrune3MaxTgt;
}

hipError_t launch_icp_pruned3(IcpBatchArgs a, const PruneScratch3& sc, int S, int64_t max_src_n, int64_t max_tgt_n, hipStream_t stream) {
  int len = 64;
  while (len < max_src_n || len < max_tgt_n) len <<= 1;
  int curve = 1;
  if (const char* env = getenv("LSLAM_ICP_PRUNE_CURVE")) curve = atoi(env);
  hipLaunchKernelGGL(icp_sort3_kernel, dim3(S), dim3(kSort3Threads), sizeof(unsigned) * (size_t)len, stream, a, sc, len, curve);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  int P = 1;
  if (const char* env = getenv("LSLAM_ICP_PRUNE_P")) { const int v = atoi(env); if (v == 1 || v == 2) P = v; }
  // Workgroup size: with at least one problem per CU, a quarter of the lanes a problem "needs" (several passes per wave, several
  // problems per CU whose barriers and solves overlap; as in the planar kernel) — tools/batch3d_probe.py, 10 forced iterations:
  // 250 points 0.77 ms at 128 lanes against 0.96 at 256; 500: 0.76 at 256 against 1.04 at 512; 1000: 0.81 at 256 against 1.22 at
  // 1024; 2000: 1.01 at 512 against 1.19; 4000: 1024 lanes (1.59).  Fewer problems than CUs: every lane the problem can use.
  const int natural = (int)((((max_src_n + P - 1) / P + 63) / 64) * 64);
  int64_t want = max_src_n / (4 * P);
  const int64_t half = max_src_n / (2 * P) < 256 ? max_src_n / (2 * P) : 256;
  if (want < half) want = half;
  int threads = (int)(((want + 63) / 64) * 64);
  if (threads < 128) threads = 128;
  if (S < 256 || threads > natural) threads = natural;
  if (threads < 64) threads = 64;
  if (threads > kIcpBatchMaxThreads) threads = kIcpBatchMaxThreads;      // more sources: several passes
  if (const char* env = getenv("LSLAM_ICP_PRUNE_THREADS")) { const int t = atoi(env); if (t >= 64 && t <= kIcpBatchMaxThreads && t % 64 == 0) threads = t; }
  const int padded = (int)(((max_tgt_n + kChunk3 - 1) / kChunk3) * kChunk3);
  const int nchunks = padded / kChunk3;
  PruneLayout lay{};
  auto up = [](size_t b) { return (b + 15) & ~(size_t)15; };
  size_t off = (size_t)padded * sizeof(float4);
  lay.aabb_off = (uint32_t)off; off += (size_t)nchunks * 2 * sizeof(float4);
  const size_t tail = up((size_t)max_src_n * sizeof(uint16_t)) + sizeof(double) * ((size_t)kIcpBatchMaxWaves * AccLayout<3>::kTotal + 17) + sizeof(StopState) + 16;
  if (off + up((size_t)max_src_n * sizeof(float4)) + tail > 156 * 1024) lay.src_off = 0xffffffffu;     // sources stay in global memory (L2)
  else { lay.src_off = (uint32_t)off; off += up((size_t)max_src_n * sizeof(float4)); }
  lay.prev_off = (uint32_t)off; off += up((size_t)max_src_n * sizeof(uint16_t));
  lay.red_off = (uint32_t)off;
  const size_t smem = off + sizeof(double) * ((size_t)kIcpBatchMaxWaves * AccLayout<3>::kTotal + 17) + sizeof(StopState) + 16;
  auto launch = [&](auto kernel) -> hipError_t {
    if (smem > 48 * 1024) {
      hipError_t ee = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (ee != hipSuccess) return ee;
    }
    hipLaunchKernelGGL(kernel, dim3(S), dim3(threads), smem, stream, a, sc, lay);
    return hipGetLastError();
  };
  if ((a.src64 == nullptr) != (a.tgt64 == nullptr)) return hipErrorInvalidValue;
  bool list = true;                                            // LSLAM_ICP_PRUNE_LIST=0: every surviving chunk evaluated by the whole wave (round 2)
  if (const char* env = getenv("LSLAM_ICP_PRUNE_LIST")) list = atoi(env) != 0;
  if (list) {
    if (a.src64) return P == 2 ? launch(icp_pruned3_kernel<2, true, true>) : launch(icp_pruned3_kernel<1, true, true>);
    return P == 2 ? launch(icp_pruned3_kernel<2, false, true>) : launch(icp_pruned3_kernel<1, false, true>);
  }
  if (a.src64) return P == 2 ? launch(icp_pruned3_kernel<2, true, false>) : launch(icp_pruned3_kernel<1, true, false>);
  return P == 2 ? launch(icp_pruned3_kernel<2, false, false>) : launch(icp_pruned3_kernel<1, false, false>);
}

}  // namespace lslam

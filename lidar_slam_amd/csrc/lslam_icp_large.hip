// lslam_icp_large.hip — ICP on ONE large cloud pair with the whole chip on it (config "100k × 100k").
//
// Per iteration (modules/icp.py:100-121), three launches on one stream, no host round trip:
//   nn_tiled_kernel   grid = (source blocks) × (target splits).  A workgroup owns 256·P source points
//                     in registers (transformed on load: icp.py:116 fused) and sweeps its share of the
//                     target through LDS tiles with wave-uniform ds_read_b128 — the same
//                     distance / v_min3 / (min, chunk) inner loop as the batched kernel.  Per source
//                     the partial result of split s is stored as a 64-bit key
//                     (float bits of d² << 32 | index) in keys[s][i]; the minimum over s is the exact
//                     argmin with the lowest index on ties (no atomics, no reset pass).
//   moments_kernel    reduces the keys over the splits, gathers the matched targets, accumulates the
//                     float64 moments per workgroup, optionally writes correspondences.
//   solve_kernel      one wave: fixed-order sum of the partial rows → Kabsch solve → stop rules →
//                     new T and the `done` flag in device memory.
// Every kernel returns at once when `done` is set, so the host may enqueue iterations ahead and
// poll the flag only now and then.
// Below 8192 source points (one source point per lane) the three launches are one — three dependent
// launches cost ≈ 17 µs however little work they carry: nn_tiled_kernel<DIM,1,true> reduces the splits
// into ONE key row with 64-bit atomicMin, each workgroup takes a ticket when its keys are out, and the
// workgroup drawing the last ticket does the moments, the solve and the stop rules, and resets keys
// and ticket for the next launch.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "lslam_device.h"
#include "lslam_kernels.h"

namespace lslam {

constexpr unsigned long long kKeyInit = ~0ull;

// min over the split rows of the keys (exact argmin, lowest index on ties)
__device__ __forceinline__ unsigned long long reduce_keys(const unsigned long long* keys, int64_t n, int splits, int64_t i) {
  unsigned long long k = keys[i];
  for (int s = 1; s < splits; ++s) {
    const unsigned long long v = keys[(int64_t)s * n + i];
    k = v < k ? v : k;
  }
  return k;
}

// SEC (float64-exact mode): also the second smallest distance of the split, one float per (split, source) in a.sec
template <int DIM, int P, bool FUSE, bool SEC = false>
__global__ void __launch_bounds__(kNnThreads)
nn_tiled_kernel(NnArgs a) {
  constexpr int CHUNK = NnChunk<DIM>::value;
  using TP = typename std::conditional<DIM == 2, float2, float4>::type;
  extern __shared__ __align__(16) unsigned char smem[];
  TP* tile = reinterpret_cast<TP*>(smem);

  if (a.state && a.state->done) return;
  const int tid = threadIdx.x;

  double T[3][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) T[r][c] = a.state ? a.state->T[4 * r + c] : a.T[4 * r + c];

  // ---- my source points ----------------------------------------------------------------------
  const int64_t base = (int64_t)blockIdx.x * (kNnThreads * P);
  float s[P][DIM];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int64_t i = base + (int64_t)p * kNnThreads + tid;
    double x = 0.0, y = 0.0, z = 0.0;
    if (i < a.n) {
      x = a.src[DIM * i]; y = a.src[DIM * i + 1];
      if constexpr (DIM == 3) z = a.src[DIM * i + 2];
    }
#pragma unroll
    for (int r = 0; r < DIM; ++r) {
      double v = T[r][0] * x + T[r][1] * y + T[r][3];
      if constexpr (DIM == 3) v += T[r][2] * z;
      s[p][r] = (float)v;
    }
  }

  // ---- my share of the target ------------------------------------------------------------------
  // splits are cut on CHUNK boundaries so that chunk ids are global
  const int64_t chunks_total = (a.m + CHUNK - 1) / CHUNK;
  const int64_t c_begin = (int64_t)blockIdx.y * a.chunks_per_split;      // whole tiles per split
  const int64_t c_end = min(chunks_total, c_begin + a.chunks_per_split);

  float best[P], second[P];
  int bchunk[P];
#pragma unroll
  for (int p = 0; p < P; ++p) { best[p] = INFINITY; second[p] = INFINITY; bchunk[p] = (int)c_begin; }

  const int tile_chunks = a.tile_points / CHUNK;
  for (int64_t c0 = c_begin; c0 < c_end; c0 += tile_chunks) {
    const int nch = (int)min((int64_t)tile_chunks, c_end - c0);
    const int64_t first = c0 * CHUNK;
    __syncthreads();
    for (int j = tid; j < nch * CHUNK; j += kNnThreads) {
      const int64_t g = first + j;
      TP v;
      if constexpr (DIM == 2) {
        v = make_float2(kFar, kFar);
        if (g < a.m) v = *reinterpret_cast<const float2*>(a.tgt + 2 * g);
      } else {
        v = make_float4(kFar, kFar, kFar, 0.f);
        if (g < a.m) { v.x = a.tgt[3 * g]; v.y = a.tgt[3 * g + 1]; v.z = a.tgt[3 * g + 2]; }
      }
      tile[j] = v;
    }
    __syncthreads();

    if constexpr (DIM == 2) {
      const float4* t4 = reinterpret_cast<const float4*>(tile);
      for (int c = 0; c < nch; ++c) {
        float4 v[CHUNK / 2];
#pragma unroll
        for (int k = 0; k < CHUNK / 2; ++k) v[k] = t4[c * (CHUNK / 2) + k];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          float mn = dist2(s[p][0], s[p][1], v[0].x, v[0].y);
          mn = fminf(mn, dist2(s[p][0], s[p][1], v[0].z, v[0].w));
#pragma unroll
          for (int k = 1; k < CHUNK / 2; ++k) {
            const float d0 = dist2(s[p][0], s[p][1], v[k].x, v[k].y);
            const float d1 = dist2(s[p][0], s[p][1], v[k].z, v[k].w);
            mn = fminf(fminf(d0, d1), mn);
          }
          const bool lt = mn < best[p];
          if constexpr (SEC) second[p] = __builtin_amdgcn_fmed3f(best[p], second[p], mn);
          bchunk[p] = lt ? (int)(c0 + c) : bchunk[p];
          best[p] = lt ? mn : best[p];
        }
      }
    } else {
      for (int c = 0; c < nch; ++c) {
        float4 v[CHUNK];
#pragma unroll
        for (int k = 0; k < CHUNK; ++k) v[k] = tile[c * CHUNK + k];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          float mn = dist2(s[p][0], s[p][1], s[p][2], v[0].x, v[0].y, v[0].z);
#pragma unroll
          for (int k = 1; k + 1 < CHUNK; k += 2) {
            const float d0 = dist2(s[p][0], s[p][1], s[p][2], v[k].x, v[k].y, v[k].z);
            const float d1 = dist2(s[p][0], s[p][1], s[p][2], v[k + 1].x, v[k + 1].y, v[k + 1].z);
            mn = fminf(fminf(d0, d1), mn);
          }
          mn = fminf(mn, dist2(s[p][0], s[p][1], s[p][2], v[CHUNK - 1].x, v[CHUNK - 1].y, v[CHUNK - 1].z));
          const bool lt = mn < best[p];
          if constexpr (SEC) second[p] = __builtin_amdgcn_fmed3f(best[p], second[p], mn);
          bchunk[p] = lt ? (int)(c0 + c) : bchunk[p];
          best[p] = lt ? mn : best[p];
        }
      }
    }
  }
  unsigned long long* __restrict__ keys = a.keys + (int64_t)blockIdx.y * a.n;   // this split's row
  // ---- fix-up from global memory (the winning chunk may have left LDS) + store -------------------
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int64_t i = base + (int64_t)p * kNnThreads + tid;
    if (i >= a.n) continue;
    if (c_begin >= c_end) {                                  // empty split: neutral key
      if constexpr (!FUSE) keys[i] = kKeyInit;
      if constexpr (SEC) a.sec[(int64_t)blockIdx.y * a.n + i] = INFINITY;
      continue;
    }
    const int64_t j0 = (int64_t)bchunk[p] * CHUNK;
    float bd = INFINITY, sec2 = INFINITY;
    int64_t bi = j0;
#pragma unroll
    for (int k = 0; k < CHUNK; ++k) {
      const int64_t j = j0 + k;
      const bool in = j < a.m;
      const float bx = in ? a.tgt[DIM * j] : kFar;
      const float by = in ? a.tgt[DIM * j + 1] : kFar;
      float d;
      if constexpr (DIM == 2) {
        d = dist2(s[p][0], s[p][1], bx, by);
      } else {
        const float bz = in ? a.tgt[DIM * j + 2] : kFar;
        d = dist2(s[p][0], s[p][1], s[p][2], bx, by, bz);
      }
      if constexpr (SEC) sec2 = __builtin_amdgcn_fmed3f(bd, sec2, d);
      if (d < bd || k == 0) { bd = d; bi = j; }
    }
    if constexpr (SEC) a.sec[(int64_t)blockIdx.y * a.n + i] = fminf(second[p], sec2);
    const unsigned long long key = ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned long long)(uint32_t)bi;
    if constexpr (FUSE) atomicMin(&a.keys[i], key);         // one key row shared by the splits (reset by the tail)
    else keys[i] = key;
  }

  // ---- fused tail for small problems ---------------------------------------------------------------
  // Every workgroup takes a ticket once its keys are stored (stores → workgroup barrier → agent-scope
  // release → atomic); the one that draws the last ticket acquires and finishes the iteration alone.
  if constexpr (FUSE) {
  __shared__ int sh_last;
  __shared__ double sh_acc[kNnThreads / 64][AccLayout<DIM>::kTotal];
  __syncthreads();                                          // all key stores of this workgroup are issued and waited for
  if (tid == 0) {
    __threadfence();                                        // release (agent scope)
    const int total = (int)(gridDim.x * gridDim.y);
    sh_last = (atomicAdd(&a.fuse_state->ticket, 1) == total - 1) ? 1 : 0;
    if (sh_last) __threadfence();                           // acquire: the other workgroups' keys
  }
  __syncthreads();
  if (!sh_last) return;

  constexpr int kAcc = AccLayout<DIM>::kTotal;
  const bool box = (a.flags & kFlagNormalize) != 0;
  PairAcc<DIM> acc;
  acc.clear();
  constexpr int U = 4;                                      // keys, then points, of U sources in flight per thread
  for (int64_t i0 = tid; i0 < a.n; i0 += (int64_t)U * kNnThreads) {
    uint32_t j[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * kNnThreads;
      j[u] = (i < a.n) ? (uint32_t)(a.keys[i] & 0xffffffffull) : 0u;
    }
    double pp[U][DIM], qq[U][DIM];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * kNnThreads;
      if (i >= a.n) continue;
      a.keys[i] = kKeyInit;                                 // ready for the next iteration's atomic mins
      if (a.out_corr) a.out_corr[i] = (int32_t)j[u];
      const double x = a.src[DIM * i], y = a.src[DIM * i + 1];
      double z = 0.0;
      if constexpr (DIM == 3) z = a.src[DIM * i + 2];
#pragma unroll
      for (int r = 0; r < DIM; ++r) {
        double v = T[r][0] * x + T[r][1] * y + T[r][3];
        if constexpr (DIM == 3) v += T[r][2] * z;
        pp[u][r] = v;
        qq[u][r] = (double)a.tgt[DIM * (int64_t)j[u] + r];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i0 + (int64_t)u * kNnThreads < a.n) acc.add(pp[u], qq[u], box);
  }
  acc.wave_reduce(box);
  if ((tid & 63) == 0) acc.store(sh_acc[tid >> 6]);
  __syncthreads();
  if (tid != 0) return;
  Moments mo;
  combine_partials<DIM>(&sh_acc[0][0], kNnThreads / 64, kAcc, box, mo);
  LargeState* st = a.fuse_state;
  double Tstep[16], Tprev[16], Tnew[16];
  step_from_moments(mo, DIM == 2, Tstep);
  for (int i = 0; i < 16; ++i) Tprev[i] = st->T[i];
  matmul4(Tstep, Tprev, Tnew);
  for (int i = 0; i < 16; ++i) st->T[i] = Tnew[i];
  const double err = error_from_moments(mo, box);
  st->err = err;
  StopState ss = st->stop;
  const bool stop = stop_after_iteration(ss, err, a.epsilon, a.stopping_thresh, a.max_iters);
  st->stop = ss;
  if (stop) st->done = 1;
  st->ticket = 0;                                           // ready for the next launch
  }
}

// ---- the same search with the expanded-form filter in front of the exact recipe ------------------------------
// Sweep: g = |t|² − 2a·t (2 / 3 v_fma + a minimum per pair instead of the direct difference), |t|² beside the target
// in LDS (planar: a float array after the tile; 3-D: the w of the float4); per source the three smallest minima over
// chunks of 32 targets and the ids of the first two.  Then the exact recipe on the chunks within the rounding bound τ
// (lslam_device.h: filter_decide).  A split is ONE tile here, so those chunks are still in LDS (gathering them back
// from global memory cost as much as the sweep: 12-byte points, one cache line per lane and load); a wave in which a
// third chunk is within τ re-sweeps the tile with the exact recipe (rare: a split holds 1/splits of the target's
// density).  The key written is the same exact (d² bits, index) as nn_tiled_kernel writes.
template <int DIM, int P, bool SEC = false>
__global__ void __launch_bounds__(kNnThreads)
nn_tiled_filter_kernel(NnArgs a) {
  constexpr int CHUNK = NnChunk<DIM>::value, FCH = kFilterChunk;
  using TP = typename FilterPoint<DIM>::type;
  extern __shared__ __align__(16) unsigned char smem[];
  TP* tile = reinterpret_cast<TP*>(smem);
  float* wt = reinterpret_cast<float*>(smem + (size_t)a.tile_points * sizeof(TP));   // planar only
  __shared__ int sh_wmax;

  if (a.state && a.state->done) return;
  const int tid = threadIdx.x, lane = tid & 63;

  double T[3][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) T[r][c] = a.state ? a.state->T[4 * r + c] : a.T[4 * r + c];

  const int64_t base = (int64_t)blockIdx.x * (kNnThreads * P);
  float na[P][DIM];                                           // −2 · transformed source
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int64_t i = base + (int64_t)p * kNnThreads + tid;
    double x = 0.0, y = 0.0, z = 0.0;
    if (i < a.n) {
      x = a.src[DIM * i]; y = a.src[DIM * i + 1];
      if constexpr (DIM == 3) z = a.src[DIM * i + 2];
    }
#pragma unroll
    for (int r = 0; r < DIM; ++r) {                           // icp.py:116
      double v = T[r][0] * x + T[r][1] * y + T[r][3];
      if constexpr (DIM == 3) v += T[r][2] * z;
      na[p][r] = -2.f * (float)v;
    }
  }

  const int64_t first_pt = (int64_t)blockIdx.y * a.chunks_per_split * CHUNK;         // this split = one tile
  const int cnt = (int)max((int64_t)0, min((int64_t)a.tile_points, a.m - first_pt));
  unsigned long long* __restrict__ keys = a.keys + (int64_t)blockIdx.y * a.n;
  const int padded = (cnt + FCH - 1) / FCH * FCH;

  float wmax = 0.f;
  if (tid == 0) sh_wmax = 0;
  for (int j = tid; j < padded; j += kNnThreads) {
    // padding: far coordinates (exact recipe → +inf) and |t|² = +inf (filter → +inf; an fma rounds once: never a NaN)
    float w = INFINITY;
    const int64_t g = first_pt + j;
    if constexpr (DIM == 2) {
      float2 v = make_float2(kFar, kFar);
      if (j < cnt) { v = *reinterpret_cast<const float2*>(a.tgt + 2 * g); w = __builtin_fmaf(v.y, v.y, v.x * v.x); wmax = fmaxf(wmax, w); }
      tile[j] = v;
      wt[j] = w;
    } else {
      float4 v = make_float4(kFar, kFar, kFar, INFINITY);
      if (j < cnt) {
        v.x = a.tgt[3 * g]; v.y = a.tgt[3 * g + 1]; v.z = a.tgt[3 * g + 2];
        v.w = __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x));
        wmax = fmaxf(wmax, v.w);
      }
      tile[j] = v;
    }
  }
  __syncthreads();
  // the largest |t|² of the split bounds the filter's rounding error
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
  if (lane == 0) atomicMax(&sh_wmax, __float_as_int(wmax));   // non-negative floats order like their bits

  float b1[P], b2[P], b3[P], a2[P];                           // b: the three smallest (chunk minimum + |a|²), chunk id in the low mantissa bits
  int i1[P], i2[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    b1[p] = b2[p] = b3[p] = INFINITY;
    // |a|² from a = −na/2 (exact), in the operation order filter_decide uses
    if constexpr (DIM == 2) a2[p] = __builtin_fmaf(-0.5f * na[p][1], -0.5f * na[p][1], (-0.5f * na[p][0]) * (-0.5f * na[p][0]));
    else a2[p] = __builtin_fmaf(-0.5f * na[p][2], -0.5f * na[p][2], __builtin_fmaf(-0.5f * na[p][1], -0.5f * na[p][1], (-0.5f * na[p][0]) * (-0.5f * na[p][0])));
  }
  for (int c = 0; c < padded / FCH; ++c) {
    float m[P];
#pragma unroll
    for (int p = 0; p < P; ++p) m[p] = INFINITY;
#pragma unroll 1
    for (int h = 0; h < FCH / kFilterStep; ++h) {
      FilterRegs<DIM> A;
      filter_load<DIM>(A, tile, wt, c * (FCH / kFilterStep) + h);
      filter_eval<DIM, P>(A, na, m);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) filter_track(m[p], a2[p], c, b1[p], b2[p], b3[p]);
  }
#pragma unroll
  for (int p = 0; p < P; ++p) { i1[p] = filter_chunk_of(b1[p]); i2[p] = filter_chunk_of(b2[p]); }
  __syncthreads();                                            // every wave's maximum is in
  const float Rt = sqrtf(__int_as_float(sh_wmax)) * 1.0001f;

  // float64-exact mode: the chunks kept must hold every target within the near-tie window of the winner (filter_decide widens τ);
  // the window needs bounds on |t| (this split's Rt ≤ the cloud's) and |p| (the cloud's, from a.bounds) and T's norm
  NearTie nt = near_tie_setup(false, 0.f, 0.f, 0.f);
  if constexpr (SEC) {
    double f2 = 0.0;
#pragma unroll
    for (int r = 0; r < DIM; ++r)
#pragma unroll
      for (int c = 0; c < DIM; ++c) f2 += T[r][c] * T[r][c];
    nt = near_tie_setup(true, sqrtf(a.bounds[0]) * 1.0001f, sqrtf(a.bounds[1]) * 1.0001f, sqrtf((float)f2) * 1.0001f);
  }
  float s[P][DIM];
  bool two[P], fall = false;
#pragma unroll
  for (int p = 0; p < P; ++p) {
#pragma unroll
    for (int r = 0; r < DIM; ++r) s[p][r] = -0.5f * na[p][r];      // exact
    bool third;
    filter_decide<DIM>(s[p], Rt, b1[p], b2[p], b3[p], nt, two[p], third);
    fall |= (base + (int64_t)p * kNnThreads + tid < a.n) && third;
  }

  // exact recipe: 64-bit key (distance bits, index inside the tile) — distances are non-negative floats, so one
  // unsigned compare is the (distance, lowest index) rule
  unsigned long long key[P];
  float ebd[P], esec[P];                                      // SEC: smallest and second smallest exact distance evaluated
#pragma unroll
  for (int p = 0; p < P; ++p) { key[p] = kKeyInit; ebd[p] = INFINITY; esec[p] = INFINITY; }
  auto eval = [&](int j, unsigned long long& k, float& bd, float& sc, const float (&pt)[DIM]) {
    const TP v = tile[j];
    float d;
    if constexpr (DIM == 2) d = dist2(pt[0], pt[1], v.x, v.y);
    else d = dist2(pt[0], pt[1], pt[2], v.x, v.y, v.z);
    const unsigned long long cand = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)j;
    k = cand < k ? cand : k;
    if constexpr (SEC) { sc = __builtin_amdgcn_fmed3f(bd, sc, d); bd = fminf(bd, d); }
  };
  if (cnt > 0) {
    if (__any(fall)) {
      for (int j = 0; j < cnt; ++j)                           // wave-uniform address: a broadcast read per target
#pragma unroll
        for (int p = 0; p < P; ++p) eval(j, key[p], ebd[p], esec[p], s[p]);
    } else {
      int rot = lane;
      asm volatile("" : "+v"(rot));                           // opaque: the 32 rotated offsets are not worth 32 live registers
#pragma unroll
      for (int p = 0; p < P; ++p) {
#pragma unroll 8
        for (int kk = 0; kk < FCH; ++kk) eval(i1[p] * FCH + ((kk + rot) & (FCH - 1)), key[p], ebd[p], esec[p], s[p]);   // rotated start: all chunks begin on the same banks
        if (__any(two[p])) {
          unsigned long long k2 = key[p];
          float bd2 = ebd[p], sc2 = esec[p];
#pragma unroll 8
          for (int kk = 0; kk < FCH; ++kk) eval(i2[p] * FCH + ((kk + rot) & (FCH - 1)), k2, bd2, sc2, s[p]);
          key[p] = two[p] ? k2 : key[p];
          if constexpr (SEC) { ebd[p] = two[p] ? bd2 : ebd[p]; esec[p] = two[p] ? sc2 : esec[p]; }
        }
      }
    }
  }
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int64_t i = base + (int64_t)p * kNnThreads + tid;
    if (i >= a.n) continue;
    // padding targets evaluate to +inf and never win against a real one; the index becomes global here
    keys[i] = cnt > 0 ? key[p] + (unsigned long long)first_pt : kKeyInit;
    // every target within the window of the split's winner was evaluated above, so the second smallest of those is what counts
    if constexpr (SEC) a.sec[(int64_t)blockIdx.y * a.n + i] = cnt > 0 ? esec[p] : INFINITY;
  }
}

// keys → (index, d²).  Used by the NN-only entry point.
__global__ void nn_finalize_kernel(const unsigned long long* keys, int64_t n, int splits, int32_t* out_idx, float* out_d2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = reduce_keys(keys, n, splits, i);
  if (out_idx) out_idx[i] = (int32_t)(uint32_t)(k & 0xffffffffull);
  if (out_d2) out_d2[i] = __uint_as_float((uint32_t)(k >> 32));
}

// ---- moments -------------------------------------------------------------------------------------
template <int DIM>
__global__ void __launch_bounds__(kMomThreads)
moments_kernel(MomArgs a) {
  constexpr int kAcc = AccLayout<DIM>::kTotal;
  __shared__ double sh[kMomThreads / 64][kAcc];
  if (a.state->done) return;
  const bool box = (a.flags & kFlagNormalize) != 0;
  double T[3][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) T[r][c] = a.state->T[4 * r + c];

  PairAcc<DIM> acc;
  acc.clear();
  for (int64_t i = (int64_t)blockIdx.x * kMomThreads + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * kMomThreads) {
    const uint32_t j = (uint32_t)(reduce_keys(a.keys, a.n, a.splits, i) & 0xffffffffull);
    if (a.out_corr) a.out_corr[i] = (int32_t)j;
    double x = a.src[DIM * i], y = a.src[DIM * i + 1], z = 0.0;
    if constexpr (DIM == 3) z = a.src[DIM * i + 2];
    double p[DIM], q[DIM];
#pragma unroll
    for (int r = 0; r < DIM; ++r) {
      double v = T[r][0] * x + T[r][1] * y + T[r][3];
      if constexpr (DIM == 3) v += T[r][2] * z;
      p[r] = v;
      q[r] = (double)a.tgt[DIM * (int64_t)j + r];
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

// ---- float64-exact mode of the three-launch path ------------------------------------------------------------
// max |p|², max |t|² of the float32 clouds (once per call): bounds[0] = target, bounds[1] = source, as float bits via atomicMax
__global__ void __launch_bounds__(256)
cloud_bounds_kernel(const float* src, int64_t n, const float* tgt, int64_t m, int dim, int* bounds) {
  float smax = 0.f, tmax = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float w = 0.f;
    for (int r = 0; r < dim; ++r) { const float c = src[dim * i + r]; w = __builtin_fmaf(c, c, w); }
    smax = fmaxf(smax, w);
  }
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < m; j += (int64_t)gridDim.x * 256) {
    float w = 0.f;
    for (int r = 0; r < dim; ++r) { const float c = tgt[dim * j + r]; w = __builtin_fmaf(c, c, w); }
    tmax = fmaxf(tmax, w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { smax = fmaxf(smax, __shfl_xor(smax, o, 64)); tmax = fmaxf(tmax, __shfl_xor(tmax, o, 64)); }
  if ((threadIdx.x & 63) == 0) { atomicMax(&bounds[0], __float_as_int(tmax)); atomicMax(&bounds[1], __float_as_int(smax)); }
}

// moments_kernel in float64-exact mode: the keys and second smallest distances of the splits say which sources have a
// near tie (lslam_device.h: NearTie); those are re-examined by the whole workgroup — every target whose float32 distance is
// within the window is evaluated in float64 — and all pairs are accumulated from the float64 points.
template <int DIM>
__global__ void __launch_bounds__(kMomThreads)
moments64_kernel(MomArgs a) {
  constexpr int kAcc = AccLayout<DIM>::kTotal;
  __shared__ double sh[kMomThreads / 64][kAcc];
  __shared__ double s_ent[kMomThreads][8];                    // near-tie entries: lim, af[3], ad[3], source index (as double: < 2^31)
  __shared__ unsigned long long s_best[kMomThreads];
  __shared__ int s_bj[kMomThreads];
  __shared__ int s_count;
  if (a.state->done) return;
  const bool box = (a.flags & kFlagNormalize) != 0;
  const int tid = threadIdx.x;
  double T[3][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) T[r][c] = a.state->T[4 * r + c];
  double f2 = 0.0;
#pragma unroll
  for (int r = 0; r < DIM; ++r)
#pragma unroll
    for (int c = 0; c < DIM; ++c) f2 += T[r][c] * T[r][c];
  const NearTie nt = near_tie_setup(true, sqrtf(a.bounds[0]) * 1.0001f, sqrtf(a.bounds[1]) * 1.0001f, sqrtf((float)f2) * 1.0001f);

  PairAcc<DIM> acc;
  acc.clear();
  for (int64_t base = (int64_t)blockIdx.x * kMomThreads; base < a.n; base += (int64_t)gridDim.x * kMomThreads) {
    if (tid == 0) s_count = 0;
    __syncthreads();
    const int64_t i = base + tid;
    double pd[DIM];
    int my_entry = -1;                                         // this thread's near-tie entry: its OWNER adds the pair to its sums (which slot the
                                                               // atomic handed out must not decide who sums what: the sums are bitwise reproducible)
    if (i < a.n) {
      unsigned long long kmin = kKeyInit;
      float v1 = INFINITY, v2 = INFINITY;
      for (int sp = 0; sp < a.splits; ++sp) {
        const unsigned long long k = a.keys[(int64_t)sp * a.n + i];
        kmin = k < kmin ? k : kmin;
        const float b = (k == kKeyInit) ? INFINITY : __uint_as_float((unsigned)(k >> 32)), sc = a.sec[(int64_t)sp * a.n + i];
        v2 = __builtin_amdgcn_fmed3f(v1, v2, b); v1 = fminf(v1, b);
        v2 = __builtin_amdgcn_fmed3f(v1, v2, sc); v1 = fminf(v1, sc);
      }
      const uint32_t j = (uint32_t)(kmin & 0xffffffffull);
      const double x = a.src64[DIM * i], y = a.src64[DIM * i + 1];
      double z = 0.0;
      if constexpr (DIM == 3) z = a.src64[DIM * i + 2];
#pragma unroll
      for (int r = 0; r < DIM; ++r) {                          // icp.py:116 on the float64 point
        double v = T[r][0] * x + T[r][1] * y + T[r][3];
        if constexpr (DIM == 3) v += T[r][2] * z;
        pd[r] = v;
      }
      const float lim = near_lim(v1, nt);
      if (v2 <= lim) {                                         // near tie: the workgroup decides it below
        const int e = atomicAdd(&s_count, 1);
        my_entry = e;
        s_ent[e][0] = (double)lim;
#pragma unroll
        for (int r = 0; r < DIM; ++r) {
          double v = T[r][0] * (double)a.src[DIM * i] + T[r][1] * (double)a.src[DIM * i + 1] + T[r][3];     // what the search saw
          if constexpr (DIM == 3) v += T[r][2] * (double)a.src[DIM * i + 2];
          s_ent[e][1 + r] = (double)(float)v;
          s_ent[e][4 + r] = pd[r];
        }
        s_ent[e][7] = (double)i;
        s_best[e] = ~0ull;
        s_bj[e] = 0x7fffffff;
      } else {
        if (a.out_corr) a.out_corr[i] = (int32_t)j;
        double q[DIM];
#pragma unroll
        for (int r = 0; r < DIM; ++r) q[r] = a.tgt64[DIM * (int64_t)j + r];
        acc.add(pd, q, box);
      }
    }
    __syncthreads();
    const int ne = s_count;
    if (ne > 0) {
      for (int pass = 0; pass < 2; ++pass) {
        for (int e = 0; e < ne; ++e) {
          const float lim = (float)s_ent[e][0];
          float ap[DIM];
#pragma unroll
          for (int r = 0; r < DIM; ++r) ap[r] = (float)s_ent[e][1 + r];
          for (int64_t j = tid; j < a.m; j += kMomThreads) {
            float d;
            if constexpr (DIM == 2) d = dist2(ap[0], ap[1], a.tgt[2 * j], a.tgt[2 * j + 1]);
            else d = dist2(ap[0], ap[1], ap[2], a.tgt[3 * j], a.tgt[3 * j + 1], a.tgt[3 * j + 2]);
            if (d <= lim) {                                    // a candidate: float64 decides (lowest index on float64 ties)
              double ev = 0.0;
#pragma unroll
              for (int r = 0; r < DIM; ++r) { const double df = s_ent[e][4 + r] - a.tgt64[DIM * j + r]; ev += df * df; }
              const unsigned long long bits = (unsigned long long)__double_as_longlong(ev);
              if (pass == 0) atomicMin(&s_best[e], bits);
              else if (bits == s_best[e]) atomicMin(&s_bj[e], (int)j);
            }
          }
        }
        __syncthreads();
      }
      if (my_entry >= 0) {                                     // the source's own thread finishes its entry
        const int bj = s_bj[my_entry];
        if (a.out_corr) a.out_corr[i] = bj;
        double q[DIM];
#pragma unroll
        for (int r = 0; r < DIM; ++r) q[r] = a.tgt64[DIM * (int64_t)bj + r];
        acc.add(pd, q, box);
      }
      __syncthreads();
    }
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

template <int DIM>
__global__ void __launch_bounds__(64)
solve_kernel(SolveArgs a) {
  constexpr int kAcc = AccLayout<DIM>::kTotal;
  constexpr int kS = AccLayout<DIM>::kSums;
  LargeState* st = a.state;
  if (st->done) return;
  const bool box = (a.flags & kFlagNormalize) != 0;
  const int lane = threadIdx.x;
  // lane-strided rows, then a shuffle tree: a fixed order for a fixed grid → reproducible
  double v[kAcc];
  for (int k = 0; k < kS; ++k) v[k] = 0.0;
  for (int k = 0; k < DIM; ++k) { v[kS + k] = INFINITY; v[kS + DIM + k] = -INFINITY; }
  for (int r = lane; r < a.rows; r += 64) {
    const double* row = a.partials + (size_t)r * kAcc;
    for (int k = 0; k < kS; ++k) v[k] += row[k];
    for (int k = 0; k < DIM; ++k) { v[kS + k] = fmin(v[kS + k], row[kS + k]); v[kS + DIM + k] = fmax(v[kS + DIM + k], row[kS + DIM + k]); }
  }
  for (int k = 0; k < kS; ++k) v[k] = wave_sum(v[k]);
  for (int k = 0; k < DIM; ++k) { v[kS + k] = wave_min(v[kS + k]); v[kS + DIM + k] = wave_max(v[kS + DIM + k]); }
  if (lane != 0) return;

  Moments mo;
  combine_partials<DIM>(v, 1, kAcc, box, mo);
  double Tstep[16], Tprev[16], Tnew[16];
  step_from_moments(mo, DIM == 2, Tstep);
  for (int i = 0; i < 16; ++i) Tprev[i] = st->T[i];
  matmul4(Tstep, Tprev, Tnew);
  for (int i = 0; i < 16; ++i) st->T[i] = Tnew[i];
  const double err = error_from_moments(mo, box);
  st->err = err;
  StopState ss = st->stop;
  const bool stop = stop_after_iteration(ss, err, a.epsilon, a.stopping_thresh, a.max_iters);
  st->stop = ss;
  if (stop) st->done = 1;
}

// ---- launchers -----------------------------------------------------------------------------------
static bool nn_filter_enabled() {
  const char* e = getenv("LSLAM_ICP_FILTER");
  return !e || atoi(e) != 0;
}

NnGeometry nn_geometry(int dim, int64_t n, int64_t m) {
  NnGeometry g{};
  const int CHUNK = (dim == 2) ? NnChunk<2>::value : NnChunk<3>::value;
  const int pt_bytes = (dim == 2) ? 8 : 16;
  g.P = (n >= 65536) ? 4 : (n >= 8192 ? 2 : 1);
  const int64_t per_block = (int64_t)kNnThreads * g.P;
  const int64_t sb = (n + per_block - 1) / per_block;
  // Aim at ~8 workgroups per CU so that the last partial round of workgroups is short; a split is
  // a whole number of LDS tiles, a tile at least kNnTileMin points.
  int64_t want = (kNnTargetWorkgroups + sb - 1) / sb;
  if (want < 1) want = 1;
  const int tile_max = (dim == 2) ? kNnTilePlanar : kNnTile3D;
  static const int tile_min = [] { const char* e = getenv("LSLAM_NN_TILE_MIN"); const int v = e ? atoi(e) : kNnTileMin; return v < 16 ? 16 : v; }();
  g.grid_x = (unsigned)sb;
  // Several source points per lane (the non-fused launches): the expanded-form filter search, whose split is ONE
  // tile of whole filter chunks — as many splits as tiles (keys: 8 bytes per source and split).  Planar tiles also
  // hold |t|² (12 bytes per point): 2048 points keep four workgroups per CU.  Planar clouds take it from 65536 source
  // points (four per lane) only: with two points per lane the planar sweep has too few vector instructions per LDS read
  // to gain (tools/large_planar_probe.py: 14.5 vs 16.6 ms at 100 000², 3.4 vs 3.3 ms at 40 000²).
  if (g.P > 1 && (dim == 3 || g.P == 4) && nn_filter_enabled()) {
    const int ftile_max = (dim == 2) ? 2048 : tile_max;
    int64_t tile = ((m + want - 1) / want + kFilterChunk - 1) / kFilterChunk * kFilterChunk;
    if (tile < tile_min) tile = (tile_min + kFilterChunk - 1) / kFilterChunk * kFilterChunk;
    if (tile > ftile_max) tile = ftile_max;
    const int64_t splits = (m + tile - 1) / tile;
    // one key row per split: keep the key workspace below 2 GiB (beyond that the multi-tile splits of the direct sweep)
    if (splits <= kNnMaxSplitsFilter && (double)n * (double)splits * 8.0 <= 2147483648.0) {
      g.filter = true;
      g.splits = (int)splits;
      g.tile_points = (int)tile;
      g.chunks_per_split = tile / CHUNK;
      g.smem = (size_t)tile * (dim == 2 ? 12 : 16);
      return g;
    }
  }
  if (want > kNnMaxSplits) want = kNnMaxSplits;
  int64_t tile = ((m + want - 1) / want + CHUNK - 1) / CHUNK * CHUNK;
  if (tile < tile_min) tile = (tile_min + CHUNK - 1) / CHUNK * CHUNK;
  if (tile > tile_max) tile = tile_max;
  const int64_t tiles_total = (m + tile - 1) / tile;
  int64_t tiles_per_split = tiles_total / want;
  if (tiles_per_split < 1) tiles_per_split = 1;
  const int64_t splits = (tiles_total + tiles_per_split - 1) / tiles_per_split;
  g.splits = (int)splits;
  g.tile_points = (int)tile;
  g.chunks_per_split = tiles_per_split * (tile / CHUNK);
  g.smem = (size_t)tile * pt_bytes;
  return g;
}

hipError_t launch_nn(NnArgs a, int dim, const NnGeometry& g, hipStream_t stream) {
  a.tile_points = g.tile_points;
  a.chunks_per_split = g.chunks_per_split;
  if (a.n <= 0) return hipSuccess;
  const dim3 grid(g.grid_x, (unsigned)g.splits), block(kNnThreads);
  const bool fuse = a.fuse_state != nullptr;                // the caller sets it only when g.P == 1
  const bool sec = a.sec != nullptr;                        // float64-exact mode (never with the fused tail)
  if (fuse && sec) return hipErrorInvalidValue;
  // the expanded-form filter sweep is the default of the search with several points per lane (LSLAM_ICP_FILTER=0:
  // direct-difference sweep); nn_geometry decides, the two kernels need different splits
#define LSLAM_NN_LAUNCH(D)                                                                                                        \
  do {                                                                                                                            \
    if (fuse) hipLaunchKernelGGL((nn_tiled_kernel<D, 1, true>), grid, block, g.smem, stream, a);                                  \
    else if (g.filter && g.P == 2) { if (sec) hipLaunchKernelGGL((nn_tiled_filter_kernel<D, 2, true>), grid, block, g.smem, stream, a);  \
                                     else hipLaunchKernelGGL((nn_tiled_filter_kernel<D, 2>), grid, block, g.smem, stream, a); }   \
    else if (g.filter) { if (sec) hipLaunchKernelGGL((nn_tiled_filter_kernel<D, 4, true>), grid, block, g.smem, stream, a);       \
                         else hipLaunchKernelGGL((nn_tiled_filter_kernel<D, 4>), grid, block, g.smem, stream, a); }               \
    else if (g.P == 1) { if (sec) hipLaunchKernelGGL((nn_tiled_kernel<D, 1, false, true>), grid, block, g.smem, stream, a);       \
                         else hipLaunchKernelGGL((nn_tiled_kernel<D, 1, false>), grid, block, g.smem, stream, a); }               \
    else if (g.P == 2) { if (sec) hipLaunchKernelGGL((nn_tiled_kernel<D, 2, false, true>), grid, block, g.smem, stream, a);       \
                         else hipLaunchKernelGGL((nn_tiled_kernel<D, 2, false>), grid, block, g.smem, stream, a); }               \
    else { if (sec) hipLaunchKernelGGL((nn_tiled_kernel<D, 4, false, true>), grid, block, g.smem, stream, a);                     \
           else hipLaunchKernelGGL((nn_tiled_kernel<D, 4, false>), grid, block, g.smem, stream, a); }                             \
  } while (0)
  if (dim == 2) LSLAM_NN_LAUNCH(2); else LSLAM_NN_LAUNCH(3);
#undef LSLAM_NN_LAUNCH
  return hipGetLastError();
}

hipError_t launch_cloud_bounds(const float* src, int64_t n, const float* tgt, int64_t m, int dim, float* bounds, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(bounds, 0, 8, stream);
  if (e != hipSuccess) return e;
  const int64_t mx = n > m ? n : m;
  int blocks = (int)((mx + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(cloud_bounds_kernel, dim3(blocks), dim3(256), 0, stream, src, n, tgt, m, dim, reinterpret_cast<int*>(bounds));
  return hipGetLastError();
}

hipError_t launch_nn_finalize(const unsigned long long* keys, int64_t n, int splits, int32_t* out_idx, float* out_d2, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(nn_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, keys, n, splits, out_idx, out_d2);
  return hipGetLastError();
}

int moments_rows(int64_t n) {
  int64_t b = (n + kMomThreads - 1) / kMomThreads;
  if (b > kMomMaxBlocks) b = kMomMaxBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

hipError_t launch_moments(MomArgs a, int dim, int rows, hipStream_t stream) {
  if (a.src64) {                                            // float64-exact mode
    if (dim == 2) hipLaunchKernelGGL(moments64_kernel<2>, dim3(rows), dim3(kMomThreads), 0, stream, a);
    else hipLaunchKernelGGL(moments64_kernel<3>, dim3(rows), dim3(kMomThreads), 0, stream, a);
    return hipGetLastError();
  }
  if (dim == 2) hipLaunchKernelGGL(moments_kernel<2>, dim3(rows), dim3(kMomThreads), 0, stream, a);
  else hipLaunchKernelGGL(moments_kernel<3>, dim3(rows), dim3(kMomThreads), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_solve(SolveArgs a, int dim, hipStream_t stream) {
  if (dim == 2) hipLaunchKernelGGL(solve_kernel<2>, dim3(1), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL(solve_kernel<3>, dim3(1), dim3(64), 0, stream, a);
  return hipGetLastError();
}

}  // namespace lslam

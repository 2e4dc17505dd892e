// lslam_icp_batch.hip — S independent ICP problems, one workgroup per problem, on-chip to convergence.
//
// Replaces the loop  poses_from_scan_matching → run_icp → icp_iteration  of the reference
// (modules/localization.py:115-124, modules/icp.py:123-189) for a whole batch in ONE launch.
//
// Per workgroup (= one source/target pair):
//   * the target cloud is staged ONCE into LDS (float2 / float4 per point, tail padded with
//     far-away points so the inner loop has no bounds checks); clouds larger than a tile are
//     streamed tile by tile;
//   * every lane owns P source points in registers.  Per iteration: transform them with the
//     current T (float64 → float32), then sweep the LDS tile with wave-uniform (broadcast)
//     ds_read_b128: per CHUNK targets 4·CHUNK VALU ops for the distances (direct difference),
//     CHUNK/2 v_min3 (half rate on gfx950: as dear as one v_min per pair) and three ops to record
//     (min, chunk id) — 4.86 instructions, ≈ 5.4 issue slots per pair.  A fix-up pass re-evaluates only the winning
//     chunk to recover the exact index (lowest index on ties);
//   * float64 moments Σp, Σq, Σpqᵀ, Σ‖p−q‖² (+ joint bounding box) → wave64 shuffle reduction →
//     LDS → lane 0 solves (closed form for planar scans, Jacobi SVD for 3-D), applies the
//     reference's stop rules and publishes T through LDS.
// HBM traffic per problem: the two clouds once, 16+16 doubles of transforms; everything else is
// on chip.  The kernel is float32-VALU-issue bound.
//
// FILTER build (big clouds, target resident in LDS): the all-pairs sweep runs on the EXPANDED form
//   g(a, t) = |t|² − 2 a·t  ( = |a − t|² − |a|² )      2 (planar) / 3 (3-D) v_fma per pair
// with |t|² kept beside the target in LDS.  g is only a filter: its float32 error against the exact
// direct-difference distance is bounded by E = c·2⁻²⁴·R(R + |a|) (R = largest |t| of the problem), so the
// exact winner lies in a chunk whose filter minimum is within τ = 2E + 9·2⁻²⁴·d² of the best one.  The sweep
// keeps the three smallest chunk minima (two with their chunk id).  Every lane then re-evaluates the best chunk
// of each of its points with the exact recipe; the few points with a second chunk within τ (≈ 1 % on C4), or a
// third (≈ 10⁻⁴: then any target may be the winner), are searched one at a time by the whole wave (coop_exact).
// Only when many points of a wave have a third chunk within τ (lattices, clouds far from the origin) does the
// wave fall back to the exact sweep — results are identical by construction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "lslam_device.h"
#include "lslam_kernels.h"

namespace lslam {

constexpr int kTileAlign = kFilterChunk * 16;            // one filter chunk of the widest point (float4): see exact_chunk
template <int DIM> struct TilePoint;
template <> struct TilePoint<2> { using type = float2; static constexpr int kFloats = 2; };
template <> struct TilePoint<3> { using type = float4; static constexpr int kFloats = 4; };

// ---- target staging ---------------------------------------------------------------------------
template <int DIM>
__device__ __forceinline__ void stage_tile(const float* __restrict__ g, int64_t count, int padded,
                                           typename TilePoint<DIM>::type* tile) {
  for (int j = threadIdx.x; j < padded; j += blockDim.x) {
    if constexpr (DIM == 2) {
      float2 v = make_float2(kFar, kFar);
      if (j < count) v = *reinterpret_cast<const float2*>(g + 2 * (int64_t)j);
      tile[j] = v;
    } else {
      float4 v = make_float4(kFar, kFar, kFar, 0.f);
      if (j < count) { v.x = g[3 * (int64_t)j]; v.y = g[3 * (int64_t)j + 1]; v.z = g[3 * (int64_t)j + 2]; }
      tile[j] = v;
    }
  }
}

// ---- the all-pairs sweep over one LDS tile ----------------------------------------------------
// best[p]: smallest distance seen so far; bchunk[p]: global chunk id holding it (strict '<', so
// the earliest chunk wins ties); second[p]: the second smallest chunk minimum (float64-exact mode: is another
// chunk within reach of a near tie?).
template <int DIM, int P, int CHUNK, bool SEC>
__device__ __forceinline__ void sweep_tile(const typename TilePoint<DIM>::type* tile, int nchunks,
                                           int chunk_base, const float (&a)[P][DIM],
                                           float (&best)[P], float (&second)[P], int (&bchunk)[P]) {
  if constexpr (DIM == 2) {
    // a record step (min, chunk id) covers CHUNK targets; they are loaded LOAD at a time so that a
    // 32-target chunk does not need 64 registers of target data
    constexpr int LOAD = CHUNK > 16 ? 16 : CHUNK;
    const float4* t4 = reinterpret_cast<const float4*>(tile);    // two planar points per 16 B
    for (int c = 0; c < nchunks; ++c) {
      float m[P];
#pragma unroll
      for (int h = 0; h < CHUNK / LOAD; ++h) {
        float4 v[LOAD / 2];
#pragma unroll
        for (int k = 0; k < LOAD / 2; ++k) v[k] = t4[c * (CHUNK / 2) + h * (LOAD / 2) + k];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float e0 = dist2(a[p][0], a[p][1], v[0].x, v[0].y);
          const float e1 = dist2(a[p][0], a[p][1], v[0].z, v[0].w);
          float mm = (h == 0) ? fminf(e0, e1) : fminf(fminf(e0, e1), m[p]);
#pragma unroll
          for (int k = 1; k < LOAD / 2; ++k) {
            const float d0 = dist2(a[p][0], a[p][1], v[k].x, v[k].y);
            const float d1 = dist2(a[p][0], a[p][1], v[k].z, v[k].w);
            mm = fminf(fminf(d0, d1), mm);
          }
          m[p] = mm;
        }
      }
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const bool lt = m[p] < best[p];
        if constexpr (SEC) second[p] = __builtin_amdgcn_fmed3f(best[p], second[p], m[p]);      // second smallest chunk minimum (best ≤ second holds)
        bchunk[p] = lt ? (chunk_base + c) : bchunk[p];
        best[p] = fminf(best[p], m[p]);
      }
    }
  } else {
    for (int c = 0; c < nchunks; ++c) {
      float4 v[CHUNK];
#pragma unroll
      for (int k = 0; k < CHUNK; ++k) v[k] = tile[c * CHUNK + k];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        float m = dist2(a[p][0], a[p][1], a[p][2], v[0].x, v[0].y, v[0].z);
#pragma unroll
        for (int k = 1; k + 1 < CHUNK; k += 2) {
          const float d0 = dist2(a[p][0], a[p][1], a[p][2], v[k].x, v[k].y, v[k].z);
          const float d1 = dist2(a[p][0], a[p][1], a[p][2], v[k + 1].x, v[k + 1].y, v[k + 1].z);
          m = fminf(fminf(d0, d1), m);
        }
        m = fminf(m, dist2(a[p][0], a[p][1], a[p][2], v[CHUNK - 1].x, v[CHUNK - 1].y, v[CHUNK - 1].z));
        const bool lt = m < best[p];
        if constexpr (SEC) second[p] = __builtin_amdgcn_fmed3f(best[p], second[p], m);
        bchunk[p] = lt ? (chunk_base + c) : bchunk[p];
        best[p] = fminf(best[p], m);
      }
    }
  }
}


// ---- the all-pairs FILTER sweep (expanded form) -------------------------------------------------
// na[p] = −2·a[p]; per chunk of kFilterChunk targets: the minimum of g, then the update of the three smallest
// chunk minima b1 ≤ b2 ≤ b3 and the chunk ids of the first two.  (Issuing a group's loads one group ahead of
// its use — two register buffers held apart with scheduling barriers — was measured: no gain, 24 more VGPRs.)
template <int DIM, int P>
__device__ __forceinline__ void filter_sweep(const typename TilePoint<DIM>::type* tile, const float* wt, int nchunks,
                                             const float (&na)[P][DIM], const float (&a2)[P], float (&b1)[P], float (&b2)[P], float (&b3)[P]) {
  constexpr int kGroups = kFilterChunk / kFilterStep;          // 4 load groups per chunk
  for (int c = 0; c < nchunks; ++c) {
    float m[P];
#pragma unroll
    for (int p = 0; p < P; ++p) m[p] = INFINITY;
    // NOT unrolled: the compiler would hoist all 24 loads of a chunk to its top and spill
    if constexpr (DIM == 2 && P == 8) {
#pragma unroll 1
      for (int h = 0; h < 2 * kGroups; ++h) {                  // four targets per group: 80 vector instructions per 3 LDS reads
        FilterRegsPlanar4 A;
        filter_load4(A, tile, wt, c * 2 * kGroups + h);
        filter_eval4<P>(A, na, m);
      }
    } else {
#pragma unroll 1
      for (int h = 0; h < kGroups; ++h) {
        FilterRegs<DIM> A;
        filter_load<DIM>(A, tile, wt, c * kGroups + h);
        filter_eval<DIM, P>(A, na, m);
      }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) filter_track(m[p], a2[p], c, b1[p], b2[p], b3[p]);
  }
}

// exact distances of the targets of one filter chunk: minimum of the 64-bit key (distance bits, index) — distances
// are non-negative floats, whose order is the order of their bit patterns, so one unsigned compare is the
// lexicographic (distance, lowest index) rule
// The low word of the key is the target's BYTE OFFSET in the tile (index × sizeof(point)): it orders like the index and is
// what the address arithmetic produces anyway; exact_index() turns a finished key into the index.
template <int DIM>
__device__ __forceinline__ int exact_index(unsigned long long key) { return (int)((unsigned)key / (unsigned)sizeof(typename TilePoint<DIM>::type)); }

template <int DIM, bool SEC>
__device__ __forceinline__ void exact_chunk(const typename TilePoint<DIM>::type* tile, int chunk, int lane, const float (&a)[DIM],
                                            unsigned long long& best, float& bd, float& sec) {
  using TP = typename TilePoint<DIM>::type;
  constexpr unsigned kB = (unsigned)sizeof(TP);
  // every chunk begins on the same LDS banks, so lane l starts at target (l mod 32) of its chunk and visits target
  // (k XOR l) at step k: one v_xor per address (a rotation costs an add, an and and a shift-add), conflict-free like the rotation.
  // Opaque: otherwise the 32 offsets are hoisted out of the ICP loop as 32 live registers (spilled, reloaded at every use)
  unsigned r = (unsigned)lane & (unsigned)(kFilterChunk - 1);
  asm volatile("" : "+v"(r));
  // the arithmetic is done on the 32-bit LDS address itself (an address_space(3) pointer as an integer; through the generic
  // pointer the compiler adds the tile base to every offset again).  The tile starts the dynamic LDS block, which is aligned
  // to a chunk (kTileAlign), so the XOR stays inside the chunk.
  typedef float __attribute__((ext_vector_type(TilePoint<DIM>::kFloats))) Vec;
  typedef const __attribute__((address_space(3))) Vec* LdsPoint;
  const unsigned tile0 = (unsigned)(size_t)(LdsPoint)(const void*)tile;
  const unsigned first = tile0 + (unsigned)chunk * (kFilterChunk * kB) + r * kB;
  unsigned long long k = ~0ull;                              // low word: the LDS address (orders like the index)
  // eight targets per trip, their loads issued together: one LDS round trip per eight targets instead of one per target
#pragma unroll 1
  for (int h = 0; h < kFilterChunk; h += 8) {
    unsigned at[8];
    Vec v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      at[c] = first ^ ((unsigned)(h + c) * kB);
      v[c] = *(LdsPoint)(size_t)at[c];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float d;
      if constexpr (DIM == 2) d = dist2(a[0], a[1], v[c].x, v[c].y);
      else d = dist2(a[0], a[1], a[2], v[c].x, v[c].y, v[c].z);
      const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | at[c];
      k = key < k ? key : k;
      if constexpr (SEC) {                                   // float64-exact mode: smallest and second smallest distance seen (a near tie?)
        sec = __builtin_amdgcn_fmed3f(bd, sec, d);
        bd = fminf(bd, d);
      }
    }
  }
  best = k - tile0;                                          // 32 keys were compared: k is one of them
}

// ONE source point (wave-uniform coordinates) against the targets [j0, j0 + cnt) of the resident tile, the 64 lanes together:
// for the rare points whose filter decision names a second chunk (cnt = kFilterChunk) or cannot be confined to two chunks
// (cnt = the whole tile).  Handling those few points one by one costs the wave a few hundred instructions; evaluating the
// second chunk for all 64 lanes because one of them asked, or re-running the whole sweep with the exact recipe because one
// point of 512 had three close chunks, cost it 2 % and 7 % of the C4 kernel.  Every lane returns the same values.
template <int DIM, bool SEC>
__device__ __forceinline__ void coop_exact(const typename TilePoint<DIM>::type* tile, int j0, int cnt, const float (&a)[DIM], int lane,
                                           unsigned long long& best, float& bd, float& sec) {
  using TP = typename TilePoint<DIM>::type;
  best = ~0ull; bd = INFINITY; sec = INFINITY;
  for (int j = j0 + lane; j < j0 + cnt; j += kWave) {
    const TP v = tile[j];
    float d;
    if constexpr (DIM == 2) d = dist2(a[0], a[1], v.x, v.y);
    else d = dist2(a[0], a[1], a[2], v.x, v.y, v.z);
    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | ((unsigned)j * (unsigned)sizeof(TP));
    best = key < best ? key : best;
    if constexpr (SEC) { sec = __builtin_amdgcn_fmed3f(bd, sec, d); bd = fminf(bd, d); }
  }
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(best, o, kWave);
    best = other < best ? other : best;
    if constexpr (SEC) {                                     // two multisets: the second smallest of the union
      const float b2 = __shfl_xor(bd, o, kWave), s2 = __shfl_xor(sec, o, kWave);
      sec = fminf(fmaxf(bd, b2), fminf(sec, s2));
      bd = fminf(bd, b2);
    }
  }
}

template <int DIM, int P, int CHUNK, bool FILTER, bool F64>
__device__ __forceinline__ void icp_batch_body(const IcpBatchArgs& args) {
  using TP = typename TilePoint<DIM>::type;
  constexpr int kAcc = AccLayout<DIM>::kTotal;
  extern __shared__ __align__(kTileAlign) unsigned char smem[];
  TP* tile = reinterpret_cast<TP*>(smem);
  float* wtile = reinterpret_cast<float*>(smem + args.tile_bytes);            // |t|² per target (planar FILTER build; 3-D keeps it in .w)
  float* sh_src = reinterpret_cast<float*>(smem + args.tile_bytes + args.w_bytes);   // the source cloud, DIM floats per point
  // float64-exact mode: the float64 clouds on chip as residual counts beside their float32 copies (lslam_device.h: rebuild64) — the
  // moments of every iteration then come from LDS instead of re-reading 16 bytes per source and per matched target from memory
  int* res_tgt = reinterpret_cast<int*>(smem + args.tile_bytes + args.w_bytes + args.src_bytes);        // DIM per target of the tile
  int* res_src = res_tgt + (size_t)args.tile_points * DIM;                                                 // DIM per source
  double* sh_red = reinterpret_cast<double*>(smem + args.tile_bytes + args.w_bytes + args.src_bytes + args.res_bytes);   // [waves][kAcc]
  double* sh_T = sh_red + (size_t)kIcpBatchMaxWaves * kAcc;                   // current transform, 16
  // loop state of run_icp (icp.py:163-181) and the last error: wave 0 is the only reader and writer, and the values are
  // wave-uniform — in LDS they cost nothing, in registers they were spilled around the solve (stores in every iteration)
  double* sh_err = sh_T + 16;
  StopState* sh_stop = reinterpret_cast<StopState*>(sh_err + 1);
  int* sh_flag = reinterpret_cast<int*>(sh_stop + 1);                         // [0] stop, [1] bits of max |t|² (FILTER)

  const int pair = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid / kWave;
  const int nwaves = blockDim.x / kWave;

  const int64_t s0 = args.src_off[pair];
  const int n = (int)(args.src_off[pair + 1] - s0);
  const int64_t t0 = args.tgt_off[pair];
  const int m = (int)(args.tgt_off[pair + 1] - t0);
  const float* __restrict__ gsrc = args.src + (int64_t)DIM * s0;
  const float* __restrict__ gtgt = args.tgt + (int64_t)DIM * t0;
  const bool box = (args.flags & kFlagNormalize) != 0;
  // float64-exact mode (lslam_device.h: NearTie): the same clouds in float64 decide near ties and feed the moments
  constexpr bool f64 = F64;                                     // the launcher picks the F64 builds when args.src64 is set
  const double* __restrict__ gsrc64 = f64 ? args.src64 + (int64_t)DIM * s0 : nullptr;
  const double* __restrict__ gtgt64 = f64 ? args.tgt64 + (int64_t)DIM * t0 : nullptr;

  if (tid < 16) sh_T[tid] = args.init_T[(int64_t)16 * pair + tid];
  if (tid == 0) { sh_flag[1] = 0; sh_flag[2] = 0; sh_flag[3] = 0; }       // bits of max |t|², max |p|² (non-negative floats order like their bits)

  const int tile_cap = args.tile_points;                       // multiple of CHUNK (and of kFilterChunk in the FILTER build)
  const int ntiles = (m + tile_cap - 1) / tile_cap;
  const bool resident = FILTER || (ntiles <= 1);               // whole target stays in LDS (the launcher picks the FILTER build only then)
  constexpr int kPad = FILTER ? kFilterChunk : CHUNK;
  const int m_padded = ((m + kPad - 1) / kPad) * kPad;
  if (resident) stage_tile<DIM>(gtgt, m, m_padded, tile);
  // the source cloud is read every iteration (twice): keep it in LDS too when it fits, so that HBM/L2
  // see each cloud exactly once per problem
  const bool src_resident = (size_t)n * DIM * sizeof(float) <= args.src_bytes;
  if (src_resident)
    for (int i = tid; i < n * DIM; i += blockDim.x) sh_src[i] = gsrc[i];
  // source point i as doubles, from LDS or from memory: two explicit paths (one generic pointer for both would make
  // every read a flat load through a 64-bit per-point address)
  auto load_src = [&](int i, double& x, double& y, double& z) {
    if constexpr (DIM == 2) {
      const float2 v = src_resident ? reinterpret_cast<const float2*>(sh_src)[i] : reinterpret_cast<const float2*>(gsrc)[i];
      x = v.x; y = v.y;
    } else {
      if (src_resident) { x = sh_src[3 * i]; y = sh_src[3 * i + 1]; z = sh_src[3 * i + 2]; }
      else { x = gsrc[3 * (int64_t)i]; y = gsrc[3 * (int64_t)i + 1]; z = gsrc[3 * (int64_t)i + 2]; }
    }
  };
  __syncthreads();
  const bool res_resident = F64 && args.res_bytes != 0 && resident && src_resident;
  if constexpr (F64) {
    if (res_resident) {
      for (int j = tid; j < m; j += blockDim.x) {
        const TP v = tile[j];
        res_tgt[DIM * j] = residual_count(gtgt64[(int64_t)DIM * j], v.x);
        res_tgt[DIM * j + 1] = residual_count(gtgt64[(int64_t)DIM * j + 1], v.y);
        if constexpr (DIM == 3) res_tgt[DIM * j + 2] = residual_count(gtgt64[(int64_t)DIM * j + 2], v.z);
      }
      for (int i = tid; i < n * DIM; i += blockDim.x) res_src[i] = residual_count(gsrc64[i], sh_src[i]);
    }
  }
  float Rt = 0.f;                                              // ≥ the largest |t| of the problem
  if constexpr (FILTER) {
    // |t|² beside every target (+inf on the padding: g = +inf there, the padding coordinates never produce a NaN
    // because an fma only rounds once); the largest one bounds the filter's rounding error
    float wmax = 0.f;
    for (int j = tid; j < m_padded; j += blockDim.x) {
      const TP v = tile[j];
      float w;
      if constexpr (DIM == 2) w = __builtin_fmaf(v.y, v.y, v.x * v.x);
      else w = __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x));
      if (j < m) wmax = fmaxf(wmax, w); else w = INFINITY;
      if constexpr (DIM == 2) wtile[j] = w;
      else tile[j].w = w;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o, kWave));
    if (lane == 0) atomicMax(&sh_flag[1], __float_as_int(wmax));      // non-negative floats order like their bits
  }
  float Rs = 0.f;                                              // ≥ the largest |p| of the problem (float64-exact mode)
  if (f64) {
    float smax = 0.f, tmax = 0.f;
    for (int i = tid; i < n; i += blockDim.x) {
      float w = 0.f;
#pragma unroll
      for (int r = 0; r < DIM; ++r) { const float c = gsrc[(int64_t)DIM * i + r]; w = __builtin_fmaf(c, c, w); }
      smax = fmaxf(smax, w);
    }
    if constexpr (!FILTER)
      for (int j = tid; j < m; j += blockDim.x) {
        float w = 0.f;
#pragma unroll
        for (int r = 0; r < DIM; ++r) { const float c = gtgt[(int64_t)DIM * j + r]; w = __builtin_fmaf(c, c, w); }
        tmax = fmaxf(tmax, w);
      }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) { smax = fmaxf(smax, __shfl_xor(smax, o, kWave)); tmax = fmaxf(tmax, __shfl_xor(tmax, o, kWave)); }
    if (lane == 0) { atomicMax(&sh_flag[2], __float_as_int(smax)); if (!FILTER) atomicMax(&sh_flag[1], __float_as_int(tmax)); }
  }
  if (FILTER || f64) {
    __syncthreads();
    Rt = sqrtf(__int_as_float(sh_flag[1])) * 1.0001f;
    Rs = sqrtf(__int_as_float(sh_flag[2])) * 1.0001f;
  }

  if (tid == 0) { StopState st0; stop_init(st0); *sh_stop = st0; *sh_err = 0.0; }       // read first after the barrier that precedes the solve
  const int src_per_pass = blockDim.x * P;
  int f_tries = 0, f_fails = 0;                                // wave-uniform: filter passes attempted / fallen back

  // One pass = blockDim·P source points: transform, sweep all targets, fix up the index, add the
  // pair to the float64 moments.  Only the float32 copies of the points live across the sweep;
  // the float64 transform is re-evaluated afterwards from the (L2-resident) source.
  auto run_pass = [&](int base, const double (&T)[3][4], const NearTie& nt, PairAcc<DIM>& acc) {
    // waves whose first lane is already past the last source point skip the sweep and the fix-up
    // (scans of a batch differ in size; the launch is sized for the largest)
    const bool wave_live = base + (tid & ~(kWave - 1)) * P < n;
    float a[P][DIM];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = base + tid * P + p;                  // lane-consecutive: tail waves own no point at all
      double x = 0.0, y = 0.0, z = 0.0;
      if (i < n) load_src(i, x, y, z);
#pragma unroll
      for (int r = 0; r < DIM; ++r) {                        // icp.py:116
        double v = T[r][0] * x + T[r][1] * y + T[r][3];
        if constexpr (DIM == 3) v += T[r][2] * z;
        a[p][r] = (float)v;
      }
    }

    int bi[P];                                               // matched target index and its coordinates
    float q[P][DIM];
    bool matched = false;

    // ---- float64-exact mode: the targets whose float32 distance is within lim of the winner's are compared in float64.
    // scope (per lane) 0: nothing to do; 1: chunk cA; 2: chunks cA and cB; 3: any target.  csz = targets per chunk.
    // Wave-uniform loops with broadcast (or rotated) reads; the float64 evaluation sits in a rare divergent branch.
    auto refine_point = [&](const float (&ap)[DIM], int i, float lim, int scope, int cA, int cB, int csz, int& bj) {
      if (!__any(scope != 0)) return;
      double ad[DIM];
      if (scope) {
        double x, y, z = 0.0;
        x = gsrc64[(int64_t)DIM * i]; y = gsrc64[(int64_t)DIM * i + 1];
        if constexpr (DIM == 3) z = gsrc64[(int64_t)DIM * i + 2];
#pragma unroll
        for (int r = 0; r < DIM; ++r) {                        // icp.py:116 on the float64 point
          double v = T[r][0] * x + T[r][1] * y + T[r][3];
          if constexpr (DIM == 3) v += T[r][2] * z;
          ad[r] = v;
        }
      }
      double bv = INFINITY;
      int bjn = bj;
      auto look = [&](int j) {
        float bx, by, bz = 0.f;
        if (resident) {
          const TP v = tile[j];
          bx = v.x; by = v.y;
          if constexpr (DIM == 3) bz = v.z;
        } else {
          const bool in = j < m;
          bx = in ? gtgt[(int64_t)DIM * j] : kFar;
          by = in ? gtgt[(int64_t)DIM * j + 1] : kFar;
          if constexpr (DIM == 3) bz = in ? gtgt[(int64_t)DIM * j + 2] : kFar;
        }
        float d;
        if constexpr (DIM == 2) d = dist2(ap[0], ap[1], bx, by);
        else d = dist2(ap[0], ap[1], ap[2], bx, by, bz);
        if (scope && d <= lim) {                               // rare: a candidate of this lane
          double e = 0.0;
#pragma unroll
          for (int r = 0; r < DIM; ++r) { const double df = ad[r] - gtgt64[(int64_t)DIM * j + r]; e += df * df; }
          if (e < bv || (e == bv && j < bjn)) { bv = e; bjn = j; }
        }
      };
      if (__any(scope == 3)) {
        for (int j = 0; j < m; ++j) look(j);
      } else {
        int rot = lane;
        asm volatile("" : "+v"(rot));
        for (int kk = 0; kk < csz; ++kk) look(cA * csz + ((kk + rot) & (csz - 1)));
        if (__any(scope == 2))
          for (int kk = 0; kk < csz; ++kk) look(cB * csz + ((kk + rot) & (csz - 1)));
      }
      if (scope) bj = bjn;
    };

    // ---- FILTER build: expanded-form sweep, then the exact recipe on the chunks that can hold the winner ----
    if constexpr (FILTER) {
      // the filter is dropped for this wave once it fell back in at least half of its attempts (lattices, clouds far
      // from the origin): from then on the exact sweep runs directly
      if (resident && wave_live && 2 * f_fails < f_tries + 1) {
        float b1[P], b2[P], b3[P], na[P][DIM], a2[P];        // b: the three smallest (chunk minimum + |a|²), chunk id in the low mantissa bits
        int i1[P], i2[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          b1[p] = b2[p] = b3[p] = INFINITY;
#pragma unroll
          for (int r = 0; r < DIM; ++r) na[p][r] = -2.f * a[p][r];
          if constexpr (DIM == 2) a2[p] = __builtin_fmaf(a[p][1], a[p][1], a[p][0] * a[p][0]);
          else a2[p] = __builtin_fmaf(a[p][2], a[p][2], __builtin_fmaf(a[p][1], a[p][1], a[p][0] * a[p][0]));
        }
        filter_sweep<DIM, P>(tile, wtile, m_padded / kFilterChunk, na, a2, b1, b2, b3);
#pragma unroll
        for (int p = 0; p < P; ++p) { i1[p] = filter_chunk_of(b1[p]); i2[p] = filter_chunk_of(b2[p]); }
        // a = −na/2 exactly: re-deriving it here keeps `a` out of the registers that are live across the sweep
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
          for (int r = 0; r < DIM; ++r) a[p][r] = -0.5f * na[p][r];
        // two[p]: a second chunk can hold the winner; third[p]: more than two can (then any target may)
        bool two[P], third[P];
        int nthird = 0;                                        // wave-uniform
#pragma unroll
        for (int p = 0; p < P; ++p) {
          filter_decide<DIM>(a[p], Rt, b1[p], b2[p], b3[p], nt, two[p], third[p]);
          const bool owns = base + tid * P + p < n;
          two[p] &= owns; third[p] &= owns;
          nthird += __popcll(__ballot(third[p]));
        }
        f_tries = __builtin_amdgcn_readfirstlane(f_tries + 1);   // wave-uniform: stays in a scalar register (as a vector value it was spilled, one scratch store per iteration and lane)
        if (args.debug && lane == 0) atomicAdd(args.debug, 1u);
        // a few such points are searched one by one by the whole wave (coop_exact: ≈ m/8 instructions each, against ≈ 5·m·P for
        // the exact sweep); on lattices and on clouds far from the origin most points are: exact sweep
        if (nthird > 8 * P) {
          f_fails = __builtin_amdgcn_readfirstlane(f_fails + 1);
          if (args.debug && lane == 0) atomicAdd(args.debug + 1, 1u);
        } else {
#pragma unroll
          for (int p = 0; p < P; ++p) {
            unsigned long long k1 = ~0ull;
            float bd = INFINITY, sec = INFINITY;
            exact_chunk<DIM, F64>(tile, i1[p], lane, a[p], k1, bd, sec);
            const unsigned long long m3 = __ballot(third[p]);
            unsigned long long todo = __ballot(two[p]) | m3;
            while (todo) {
              const int l = __builtin_ctzll(todo);
              todo &= todo - 1;
              float au[DIM];
#pragma unroll
              for (int r = 0; r < DIM; ++r) au[r] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[p][r]), l));
              const bool all = (m3 >> l) & 1ull;
              const int c2 = __builtin_amdgcn_readlane(i2[p], l);
              unsigned long long kc;
              float bc, sc;
              coop_exact<DIM, F64>(tile, all ? 0 : c2 * kFilterChunk, all ? m_padded : kFilterChunk, au, lane, kc, bc, sc);
              if (lane == l) {
                if (all) { k1 = kc; bd = bc; sec = sc; }       // the whole tile: supersedes the first chunk
                else {
                  k1 = kc < k1 ? kc : k1;
                  sec = fminf(fmaxf(bd, bc), fminf(sec, sc));
                  bd = fminf(bd, bc);
                }
              }
            }
            bi[p] = exact_index<DIM>(k1);
            if constexpr (F64) {                               // every target within lim of the winner lies in the chunks searched (τ was widened)
              const int i = base + tid * P + p;
              const float lim = near_lim(bd, nt);
              refine_point(a[p], i, lim, (i < n && sec <= lim) ? (third[p] ? 3 : (two[p] ? 2 : 1)) : 0, i1[p], i2[p], kFilterChunk, bi[p]);
            }
            const TP v = tile[bi[p]];
            q[p][0] = v.x; q[p][1] = v.y;
            if constexpr (DIM == 3) q[p][2] = v.z;
          }
          matched = true;
        }
      }
    }

    if (!matched) {
      // ---- nearest neighbour over all targets, exact recipe ----------------------------------------
      float best[P], second[P];
      int bchunk[P];
#pragma unroll
      for (int p = 0; p < P; ++p) { best[p] = INFINITY; second[p] = INFINITY; bchunk[p] = 0; }
      if (resident) {
        if (wave_live) sweep_tile<DIM, P, CHUNK, F64>(tile, (m + CHUNK - 1) / CHUNK, 0, a, best, second, bchunk);
      } else {
        for (int t = 0; t < ntiles; ++t) {
          const int first = t * tile_cap;
          const int cnt = min(tile_cap, m - first);
          __syncthreads();                                     // previous tile fully consumed
          stage_tile<DIM>(gtgt + (int64_t)DIM * first, cnt, ((cnt + CHUNK - 1) / CHUNK) * CHUNK, tile);
          __syncthreads();
          if (wave_live) sweep_tile<DIM, P, CHUNK, F64>(tile, (cnt + CHUNK - 1) / CHUNK, first / CHUNK, a, best, second, bchunk);
        }
      }

      // ---- fix-up: exact index inside the winning chunk ----------------------------------------------
      if (!wave_live) return;
      int rot = lane;
      asm volatile("" : "+v"(rot));                          // opaque: keeps the CHUNK rotated offsets from being hoisted out of the ICP loop (and spilled)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int j0 = bchunk[p] * CHUNK;
        float bd = INFINITY, sec2 = INFINITY;                  // smallest and second smallest distance inside the chunk
        bi[p] = j0;
#pragma unroll
        for (int r = 0; r < DIM; ++r) q[p][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < CHUNK; ++kk) {
          // every lane starts at a different point of its chunk: all chunks begin on the same LDS
          // banks, an unrotated gather would be a 32-way bank conflict
          const int k = (kk + rot) & (CHUNK - 1);
          const int j = j0 + k;
          float bx, by, bz = 0.f;
          if (resident) {
            const TP v = tile[j];
            bx = v.x; by = v.y;
            if constexpr (DIM == 3) bz = v.z;
          } else {
            const bool in = j < m;
            bx = in ? gtgt[(int64_t)DIM * j] : kFar;
            by = in ? gtgt[(int64_t)DIM * j + 1] : kFar;
            if constexpr (DIM == 3) bz = in ? gtgt[(int64_t)DIM * j + 2] : kFar;
          }
          float d;
          if constexpr (DIM == 2) d = dist2(a[p][0], a[p][1], bx, by);
          else d = dist2(a[p][0], a[p][1], a[p][2], bx, by, bz);
          if constexpr (F64) sec2 = __builtin_amdgcn_fmed3f(bd, sec2, d);
          if (kk == 0 || d < bd || (d == bd && j < bi[p])) {      // lowest index on ties, whatever the visiting order
            bd = d; bi[p] = j; q[p][0] = bx; q[p][1] = by;
            if constexpr (DIM == 3) q[p][2] = bz;
          }
        }
        if constexpr (F64) {
          const int i = base + tid * P + p;
          const float lim = near_lim(bd, nt);
          const int scope = (i < n) ? (second[p] <= lim ? 3 : (sec2 <= lim ? 1 : 0)) : 0;
          refine_point(a[p], i, lim, scope, bchunk[p], 0, CHUNK, bi[p]);
        }
      }
    }

    // ---- the matched pairs go to the float64 moments -------------------------------------------------
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = base + tid * P + p;
      if (i < n) {
        double x, y, z = 0.0;
        double pd[DIM], qd[DIM];
        if (f64 && res_resident) {                             // the float64 points themselves, restored from their on-chip residuals
          x = rebuild64(sh_src[DIM * i], res_src[DIM * i]); y = rebuild64(sh_src[DIM * i + 1], res_src[DIM * i + 1]);
          if constexpr (DIM == 3) z = rebuild64(sh_src[DIM * i + 2], res_src[DIM * i + 2]);
          const TP v = tile[bi[p]];
          qd[0] = rebuild64(v.x, res_tgt[DIM * bi[p]]); qd[1] = rebuild64(v.y, res_tgt[DIM * bi[p] + 1]);
          if constexpr (DIM == 3) qd[2] = rebuild64(v.z, res_tgt[DIM * bi[p] + 2]);
        } else if (f64) {                                      // the float64 points themselves: icp.py:116-118 as the reference computes them
          x = gsrc64[(int64_t)DIM * i]; y = gsrc64[(int64_t)DIM * i + 1];
          if constexpr (DIM == 3) z = gsrc64[(int64_t)DIM * i + 2];
#pragma unroll
          for (int r = 0; r < DIM; ++r) qd[r] = gtgt64[(int64_t)DIM * bi[p] + r];
        } else {
          load_src(i, x, y, z);
#pragma unroll
          for (int r = 0; r < DIM; ++r) qd[r] = (double)q[p][r];
        }
#pragma unroll
        for (int r = 0; r < DIM; ++r) {
          double v = T[r][0] * x + T[r][1] * y + T[r][3];
          if constexpr (DIM == 3) v += T[r][2] * z;
          pd[r] = v;
        }
        acc.add(pd, qd, box);
        if (args.out_corr) args.out_corr[s0 + i] = bi[p];
      }
    }
  };

  for (;;) {
    // the transform is wave-uniform: move it to scalar registers
    double T[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) T[r][c] = wave_uniform(sh_T[4 * r + c]);

    // float64-exact mode: the near-tie window of this iteration (F = Frobenius norm of T's linear block)
    float Fn = 0.f;
    if (f64) {
      double f2 = 0.0;
#pragma unroll
      for (int r = 0; r < DIM; ++r)
#pragma unroll
        for (int c = 0; c < DIM; ++c) f2 += T[r][c] * T[r][c];
      Fn = sqrtf((float)f2) * 1.0001f;
    }
    const NearTie nt = near_tie_setup(f64, Rt, Rs, Fn);

    PairAcc<DIM> acc;
    acc.clear();
    const int passes = n > 0 ? (n + src_per_pass - 1) / src_per_pass : 1;
    if (passes == 1) {
      run_pass(0, T, nt, acc);           // common case: the zeroed accumulators never live across the sweep
    } else {
      for (int pass = 0; pass < passes; ++pass) run_pass(pass * src_per_pass, T, nt, acc);
    }

    // ---- block reduction: wave rows → LDS → wave 0 combines and solves -----------------------
    acc.wave_reduce(box);
    if (lane == 0) acc.store(sh_red + (size_t)wave * kAcc);
    __syncthreads();
    if (wave == 0) {
      // lane k sums column k over the waves in wave order (fixed order → bitwise reproducible) ...
      constexpr int kS = AccLayout<DIM>::kSums;
      double col = 0.0;
      {
        // four rows per trip, their loads issued together, sum / min / max chosen with selects: as a rolled loop with a branch
        // per row every load waited for a full LDS round trip (up to 16 per iteration of every problem)
        const int cl = lane < kAcc ? lane : 0;
        const bool is_sum = lane < kS, is_min = lane < kS + DIM;
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
      // ... the totals reach every lane through scalar registers, and the (uniform) solve runs once
      double tot[kAcc];
#pragma unroll
      for (int k = 0; k < kAcc; ++k) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(col), k);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(col), k);
        tot[k] = __hiloint2double(hi, lo);
      }
      Moments mo;
      combine_partials<DIM>(tot, 1, kAcc, box, mo);
      double R[9], pb[3], qb[3];
      step_parts(mo, DIM == 2, R, pb, qb);
      const double tn = step_times_prev_lane(R, pb, qb, sh_T, sh_red, lane);      // icp.py:119, one element per lane (the wave rows in sh_red are consumed)
      const double err = error_from_moments(mo, box);       // icp.py:120 (pre-update residual)
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
}

// Two builds of the same body:
//   icp_batch_kernel        up to 1024 lanes per problem, 128 VGPRs: big clouds (C4), chunk = IcpChunk;
//   icp_batch_small_kernel  capped at 64 VGPRs (8 waves per SIMD) with a smaller
//                           chunk: LiDAR-sized problems (≤ ~1000 points) leave most of an iteration's
//                           wall time in barriers and the one-lane solve, so several workgroups must
//                           be resident per CU to keep the VALUs fed.
template <int DIM, int P, bool F64>
__global__ void __launch_bounds__(kIcpBatchMaxThreads)
icp_batch_kernel(IcpBatchArgs args) {
  icp_batch_body<DIM, P, IcpChunk<DIM>::value, false, F64>(args);
}

// the big build with the expanded-form filter sweep in front of the exact recipe (target resident in LDS)
template <int DIM, int P, bool F64>
__global__ void __launch_bounds__(kIcpBatchMaxThreads)
icp_batch_filter_kernel(IcpBatchArgs args) {
  icp_batch_body<DIM, P, IcpChunk<DIM>::value, true, F64>(args);
}

template <int DIM, int P, bool F64>
__global__ void __launch_bounds__(kIcpSmallMaxThreads, 8)
icp_batch_small_kernel(IcpBatchArgs args) {
  icp_batch_body<DIM, P, IcpChunkSmall<DIM>::value, false, F64>(args);
}

// ---- launch ------------------------------------------------------------------------------------
enum class Build { kBig, kSmall, kFilter };

template <int DIM, int P, bool F64>
static hipError_t launch_one_mode(const IcpBatchArgs& a, int S, int threads, size_t smem, Build build, hipStream_t stream) {
  const void* fn = build == Build::kSmall ? reinterpret_cast<const void*>(icp_batch_small_kernel<DIM, P, F64>)
                 : build == Build::kFilter ? reinterpret_cast<const void*>(icp_batch_filter_kernel<DIM, P, F64>)
                                           : reinterpret_cast<const void*>(icp_batch_kernel<DIM, P, F64>);
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
  }
  if (build == Build::kSmall) hipLaunchKernelGGL((icp_batch_small_kernel<DIM, P, F64>), dim3(S), dim3(threads), smem, stream, a);
  else if (build == Build::kFilter) hipLaunchKernelGGL((icp_batch_filter_kernel<DIM, P, F64>), dim3(S), dim3(threads), smem, stream, a);
  else hipLaunchKernelGGL((icp_batch_kernel<DIM, P, F64>), dim3(S), dim3(threads), smem, stream, a);
  return hipGetLastError();
}

// the float64-exact builds exist for 1, 2 and 4 points per lane (the launcher never picks 8 in that mode)
template <int DIM, int P>
static hipError_t launch_one(const IcpBatchArgs& a, int S, int threads, size_t smem, Build build, hipStream_t stream) {
  // (P = 8 in this mode was built and measured in round 4: 824 bytes of scratch per lane, and its two 8-wave workgroups per CU
  // have no LDS left for the float64 residuals — 319 ms on C4 against 190 ms for the P = 4 build)
  if constexpr (P <= 4)
    if (a.src64) return launch_one_mode<DIM, P, true>(a, S, threads, smem, build, stream);
  return launch_one_mode<DIM, P, false>(a, S, threads, smem, build, stream);
}

hipError_t launch_icp_batch(IcpBatchArgs a, int dim, int S, int64_t max_src_n, int64_t max_tgt_n,
                            const IcpBatchTuning& tune, hipStream_t stream) {
  // points per lane
  int P = tune.points_per_lane;
  if (P == 0) P = (max_src_n <= 512) ? 1 : ((max_src_n <= 2048) ? 2 : 4);
  if (P != 1 && P != 2 && P != 4 && !(P == 8 && dim == 2 && !a.src64)) return hipErrorInvalidValue;
  if ((a.src64 == nullptr) != (a.tgt64 == nullptr)) return hipErrorInvalidValue;
  int64_t lanes = (max_src_n + P - 1) / P;
  int threads = (int)(((lanes + kWave - 1) / kWave) * kWave);
  if (threads < kWave) threads = kWave;
  if (threads > kIcpBatchMaxThreads) threads = kIcpBatchMaxThreads;
  if (tune.threads > 0) threads = tune.threads;
  // the 64-VGPR build (three workgroups per CU) pays for planar LiDAR-sized problems only: in 3-D it spills 250–500
  // registers and the 128-VGPR filter build is 15–45 % faster at every size from 64 to 2500 points (tools/batch3d_probe.py)
  bool small = dim == 2 && threads <= kIcpSmallAutoThreads && max_tgt_n <= kIcpSmallMaxTargets;
  if (tune.variant == 1) small = false;
  if (tune.variant == 2 && threads <= kIcpSmallMaxThreads) small = true;
  const int CHUNK = small ? ((dim == 2) ? IcpChunkSmall<2>::value : IcpChunkSmall<3>::value)
                          : ((dim == 2) ? IcpChunk<2>::value : IcpChunk<3>::value);

  // LDS tile
  const int pt_bytes = (dim == 2) ? 8 : 16;
  int cap = (dim == 2) ? kIcpTileMaxPlanar : kIcpTileMax3D;
  if (tune.tile_points > 0) cap = tune.tile_points;
  cap = ((cap + 31) / 32) * 32;                              // a multiple of every chunk size
  // the expanded-form filter needs the whole target in LDS (one tile); it is the default of the big build
  const bool filter = !small && max_tgt_n <= cap && tune.filter != 2;
  // Planar filter build, big clouds in big batches: EIGHT source points per lane in half as many lanes (80 vector
  // instructions per 3 LDS reads instead of per 6) and, up to 4096 source points, two 8-wave workgroups per CU
  // instead of one of 16.  It pays only while every CU always holds two problems (tools/p8_crossover.sh,
  // profiles/r02/p8_crossover.txt: 194 vs 210 ms on C4, 49 vs 53 ms at 8192 points, but 12.5 vs 11.5 ms for 512
  // problems and 15.5 vs 10.9 ms at 2600 points).
  if (filter && dim == 2 && !a.src64 && tune.points_per_lane == 0 && tune.threads == 0 && max_src_n >= 3584 && S >= 2048) {
    P = 8;
    lanes = (max_src_n + P - 1) / P;
    threads = (int)(((lanes + kWave - 1) / kWave) * kWave);
    if (threads > kIcpBatchMaxThreads) threads = kIcpBatchMaxThreads;
  }
  const int pad = filter ? kFilterChunk : CHUNK;
  int64_t need = ((max_tgt_n + pad - 1) / pad) * pad;
  if (need < pad) need = pad;
  const int tile_points = (int)(need < cap ? need : cap);
  a.tile_points = tile_points;
  a.tile_bytes = (uint32_t)(((size_t)tile_points * pt_bytes + 15) & ~(size_t)15);
  a.w_bytes = (filter && dim == 2) ? (uint32_t)(((size_t)tile_points * sizeof(float) + 15) & ~(size_t)15) : 0u;
  const int kAcc = (dim == 2) ? AccLayout<2>::kTotal : AccLayout<3>::kTotal;
  // keep the source cloud in LDS as well: the small build while four workgroups still fit a CU; the big build
  // (one workgroup per CU for its 128 VGPRs × up to 16 waves) while the whole workgroup stays below 128 KiB
  // (… or, for the 8-wave workgroups of the P = 8 build, while two of them fit a CU)
  const size_t budget = small ? 36 * 1024 : (threads <= 512 ? 78 * 1024 : 128 * 1024);
  const size_t src_need = (((size_t)max_src_n * dim * sizeof(float)) + 15) & ~(size_t)15;
  a.src_bytes = (a.tile_bytes + a.w_bytes + src_need <= budget && tune.src_in_lds != 2) || tune.src_in_lds == 1 ? (uint32_t)src_need : 0u;
  // float64-exact mode: the residual counts of both clouds on chip as well while everything fits one CU's LDS (the big builds run one
  // workgroup per CU anyway; C4: 32 + 16 + 32 + 64 KB): HBM then sees each cloud once per problem in this mode too
  a.res_bytes = 0;
  {
    const size_t fixed = sizeof(double) * ((size_t)kIcpBatchMaxWaves * kAcc + 16 + 1) + sizeof(StopState) + 16;
    const size_t res_need = (((size_t)tile_points + (size_t)max_src_n) * dim * sizeof(int) + 15) & ~(size_t)15;
    if (a.src64 && !small && a.src_bytes != 0 && max_tgt_n <= tile_points &&
        a.tile_bytes + a.w_bytes + a.src_bytes + res_need + fixed <= 160 * 1024 && !getenv("LSLAM_ICP_NO_RESIDUALS"))
      a.res_bytes = (uint32_t)res_need;
  }
  const size_t smem = a.tile_bytes + a.w_bytes + a.src_bytes + a.res_bytes + sizeof(double) * ((size_t)kIcpBatchMaxWaves * kAcc + 16 + 1) + sizeof(StopState) + 16;
  const Build build = small ? Build::kSmall : (filter ? Build::kFilter : Build::kBig);
  static unsigned* dbg = nullptr;
  if (getenv("LSLAM_DEBUG")) {
    if (!dbg) (void)hipMalloc(&dbg, 8);
    else { unsigned h[2] = {0, 0}; (void)hipMemcpy(h, dbg, 8, hipMemcpyDeviceToHost); fprintf(stderr, "[lslam] previous batch launch: %u filter passes, %u fell back to the exact sweep\n", h[0], h[1]); }
    (void)hipMemset(dbg, 0, 8);
    a.debug = dbg;
    fprintf(stderr, "[lslam] icp_batch build %d (0 big, 1 small, 2 filter), P %d, %d threads, %zu B LDS\n", (int)build, P, threads, smem);
  }

  if (dim == 2) {
    if (P == 1) return launch_one<2, 1>(a, S, threads, smem, build, stream);
    if (P == 2) return launch_one<2, 2>(a, S, threads, smem, build, stream);
    if (P == 8) return launch_one<2, 8>(a, S, threads, smem, build, stream);      // experiment knob (LSLAM_ICP_POINTS_PER_LANE=8)
    return launch_one<2, 4>(a, S, threads, smem, build, stream);
  }
  if (P == 1) return launch_one<3, 1>(a, S, threads, smem, build, stream);
  if (P == 2) return launch_one<3, 2>(a, S, threads, smem, build, stream);
  return launch_one<3, 4>(a, S, threads, smem, build, stream);
}

}  // namespace lslam

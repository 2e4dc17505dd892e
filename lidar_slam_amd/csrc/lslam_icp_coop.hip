// lslam_icp_coop.hip — ONE ICP problem of LiDAR-scan size (config "single scan pair, 1081 × 1081, 50 iterations"),
// iterated to convergence inside ONE launch by a small group of co-resident workgroups.
//
// Replaces the loop of modules/icp.py:167-181 (run_icp → icp_iteration) for clouds of up to 8191 source points whose
// target fits one LDS tile.  The whole-chip path of lslam_icp_large.hip spends one dependent launch per iteration on
// such a pair (≈ 11 µs each: launch, two global round trips for the fix-up, an atomic key row, an agent-scope fence, a
// ticket, one workgroup gathering every match).  Here nothing leaves the chip between iterations except one row of
// float64 moments per workgroup:
//   * G = ⌈n / SPW⌉ workgroups of 1024 lanes stay resident for the whole run (G ≤ 128 ≪ 256 CUs; one per CU).  Workgroup g
//     owns SPW source points and keeps the WHOLE target in its LDS (staged once);
//   * a wave holds SPW sources × SUB target slices (SPW·SUB = 64): every lane sweeps its slice with the exact
//     direct-difference recipe (lslam_device.h: dist2), fixes up the index inside the winning chunk, and leaves a 64-bit
//     key (distance bits ‖ index) in LDS; wave 0 takes the minimum over the 16·SUB slices (= the exact argmin, lowest
//     index on ties), accumulates the float64 moments of its SPW pairs and reduces them over the wave;
//   * the ONLY exchange between workgroups: every workgroup publishes its row of moments as 8-byte {epoch, 32 bits}
//     granules (write-through agent-scope stores: the data is the flag, no fence — guide, Guideline 16 R2), sweeps the
//     rows of all G workgroups until every tag carries this iteration's epoch, sums them in workgroup order (so every
//     workgroup obtains bit-identical totals), solves and applies the stop rules redundantly.  All workgroups therefore
//     take the same decision in the same iteration and leave together; rows are double-buffered by epoch parity (a
//     workgroup can run at most one epoch ahead of the slowest reader of its row).
// float64-exact mode (F64, lslam_device.h: NearTie): a source whose second smallest float32 distance lies within
// lim(D̂) of the smallest is re-examined by all waves — every target of their slices within lim is evaluated in float64
// on the float64 points — and the moments are taken from the float64 points.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "lslam_device.h"
#include "lslam_kernels.h"

namespace lslam {

using gu64 = __attribute__((address_space(1))) unsigned long long;
using gu32 = __attribute__((address_space(1))) unsigned;

constexpr int kCoopThreads = 1024;
constexpr int kCoopWaves = kCoopThreads / 64;

template <int DIM> struct CoopChunk;
template <> struct CoopChunk<2> { static constexpr int value = 16; };
template <> struct CoopChunk<3> { static constexpr int value = 8; };

// wall clock in 100 MHz ticks (s_memrealtime): bounds every spin so that the grid always drains
__device__ __forceinline__ unsigned long long coop_now() { return wall_clock64(); }
constexpr unsigned long long kCoopTimeoutTicksDefault = 5000000ull;      // 50 ms of the 100 MHz wall clock per hand-off (a healthy one takes microseconds); the host then takes the launch-per-iteration path

// values of a workgroup's row: Σp, Σq, Σpqᵀ, Σ‖p−q‖² (the pair count is n for everybody) [+ joint box when normalising]
template <int DIM> struct CoopRow {
  static constexpr int kSums = DIM + DIM + DIM * DIM + 1;
  static constexpr int kVals = kSums + 2 * DIM;
  static constexpr int kVP = kVals <= 16 ? 16 : 32;          // lanes per segment of the transposed reduction
};

template <int DIM, int SUB, bool F64>
__global__ void __launch_bounds__(kCoopThreads)
icp_coop_kernel(CoopArgs a) {
  constexpr int SPW = 64 / SUB;                   // source points per workgroup
  constexpr int NS = kCoopWaves * SUB;            // target slices
  constexpr int CH = CoopChunk<DIM>::value;
  constexpr int kSums = CoopRow<DIM>::kSums, kVals = CoopRow<DIM>::kVals, VP = CoopRow<DIM>::kVP, NSEG = 64 / VP;
  using TP = typename std::conditional<DIM == 2, float2, float4>::type;

  extern __shared__ __align__(16) unsigned char smem[];
  TP* tile = reinterpret_cast<TP*>(smem);
  double* tile64 = reinterpret_cast<double*>(smem + a.tile_bytes);                          // F64: the float64 target, DIM doubles per point
  float4* abox = reinterpret_cast<float4*>(smem + a.tile_bytes + a.tile64_bytes);             // per chunk: {x0, y0, x1, y1} / {lo, hi}
  unsigned long long* skey = reinterpret_cast<unsigned long long*>(smem + a.tile_bytes + a.tile64_bytes + a.box_bytes);   // [NS][SPW]
  float* ssec = reinterpret_cast<float*>(skey + NS * SPW);                                  // [NS][SPW]
  double* sred = reinterpret_cast<double*>(ssec + NS * SPW);                                // [kVals][65] transposed moments of wave 0
  double* sent = sred + 32 * 65;                                                            // near-tie entries [64][8]: lim, af[3], ad[3], source lane
  unsigned long long* sbest = reinterpret_cast<unsigned long long*>(sent + 64 * 8);         // [64] float64 distance bits of the best candidate
  int* sbj = reinterpret_cast<int*>(sbest + 64);                                            // [64] its index
  int* smatch = sbj + 64;                                                                   // [SPW] last iteration's match of every source (seeds the next search)
  unsigned* rows32 = reinterpret_cast<unsigned*>(smatch + 64);                              // [G][kVals][2] every workgroup's row
  double* sh_T = reinterpret_cast<double*>(rows32 + (size_t)gridDim.x * kVals * 2);
  double* sh_err = sh_T + 16;
  StopState* sh_stop = reinterpret_cast<StopState*>(sh_err + 1);
  int* sh_flag = reinterpret_cast<int*>(sh_stop + 1);     // [0] stop, [1] near-tie entries, [2] bits of max |t|², [3] bits of max |p|², [4] failed

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = blockIdx.x, G = gridDim.x;
  const int sl = lane % SPW, sub = lane / SPW;
  const int slice = wave * SUB + sub;
  const int n = a.n, m = a.m;
  const int i = wg * SPW + sl;                   // my source point
  const bool valid = i < n;
  const bool box = (a.flags & kFlagNormalize) != 0;
  const int nvals = box ? kVals : kSums;         // values exchanged per workgroup and iteration

  // ---- stage the target once, my source point, the start transform ---------------------------------
  const int nchunks = (m + CH - 1) / CH;
  float tmax = 0.f;
  for (int j = tid; j < nchunks * CH; j += kCoopThreads) {
    TP v;
    if constexpr (DIM == 2) {
      v = make_float2(kFar, kFar);
      if (j < m) v = *reinterpret_cast<const float2*>(a.tgt + 2 * (int64_t)j);
      if (F64 && j < m) tmax = fmaxf(tmax, __builtin_fmaf(v.y, v.y, v.x * v.x));
    } else {
      v = make_float4(kFar, kFar, kFar, 0.f);
      if (j < m) { v.x = a.tgt[3 * (int64_t)j]; v.y = a.tgt[3 * (int64_t)j + 1]; v.z = a.tgt[3 * (int64_t)j + 2]; }
      if (F64 && j < m) tmax = fmaxf(tmax, __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x)));
    }
    tile[j] = v;
  }
  if constexpr (F64)
    for (int j = tid; j < m * DIM; j += kCoopThreads) tile64[j] = a.tgt64[j];
  float pf[DIM];
  double pd[DIM];
#pragma unroll
  for (int r = 0; r < DIM; ++r) {
    pf[r] = valid ? a.src[(int64_t)DIM * i + r] : 0.f;
    pd[r] = F64 ? (valid ? a.src64[(int64_t)DIM * i + r] : 0.0) : (double)pf[r];
  }
  if (tid < 16) sh_T[tid] = a.state->T[tid];
  if (tid < SPW) smatch[tid] = (int)(((long long)min(wg * SPW + tid, n - 1) * m) / n);      // first iteration: the target at the same rank (scans are ordered alike)
  if (tid == 0) {
    StopState st0; stop_init(st0); *sh_stop = st0; *sh_err = 0.0;
    sh_flag[0] = 0; sh_flag[1] = 0; sh_flag[2] = 0; sh_flag[3] = 0; sh_flag[4] = 0;
  }
  __syncthreads();
  // bounding box of every chunk of CH consecutive targets (real points only): consecutive points of a scan are neighbours in
  // space, so most chunks can be skipped from their box alone (below); clouds in arbitrary order lose nothing but the test
  for (int c = tid; c < nchunks; c += kCoopThreads) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < CH && c * CH + k < m; ++k) {
      const TP q = tile[c * CH + k];
      lo[0] = fminf(lo[0], q.x); hi[0] = fmaxf(hi[0], q.x);
      lo[1] = fminf(lo[1], q.y); hi[1] = fmaxf(hi[1], q.y);
      if constexpr (DIM == 3) { lo[2] = fminf(lo[2], q.z); hi[2] = fmaxf(hi[2], q.z); }
    }
    if constexpr (DIM == 2) abox[c] = make_float4(lo[0], lo[1], hi[0], hi[1]);
    else { abox[2 * c] = make_float4(lo[0], lo[1], lo[2], 0.f); abox[2 * c + 1] = make_float4(hi[0], hi[1], hi[2], 0.f); }
  }
  float Rt = 0.f, Rs = 0.f;
  if constexpr (F64) {
    float smax = 0.f;
    for (int k = tid; k < n; k += kCoopThreads) {       // over ALL sources (a workgroup's own would do): every workgroup uses the same window
      float w = 0.f;
#pragma unroll
      for (int r = 0; r < DIM; ++r) { const float c = a.src[(int64_t)DIM * k + r]; w = __builtin_fmaf(c, c, w); }
      smax = fmaxf(smax, w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { smax = fmaxf(smax, __shfl_xor(smax, o, 64)); tmax = fmaxf(tmax, __shfl_xor(tmax, o, 64)); }
    if (lane == 0) { atomicMax(&sh_flag[2], __float_as_int(tmax)); atomicMax(&sh_flag[3], __float_as_int(smax)); }
    __syncthreads();
    Rt = sqrtf(__int_as_float(sh_flag[2])) * 1.0001f;
    Rs = sqrtf(__int_as_float(sh_flag[3])) * 1.0001f;
  }

  __syncthreads();                                 // boxes (and, in float64-exact mode, the bounds) are in
  // my slice of the target: chunks slice, slice + NS, … — neighbouring chunks (what the sources of a workgroup typically
  // need) go to different waves, i.e. different SIMDs

  // the rows: [parity][workgroup][value] 16-byte granules {lo32, epoch, hi32, epoch} — each 8-byte half carries its tag
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.gran, 0, (int)a.gran_bytes, 0x00020000);
  int match = 0;                                  // wave 0: the last iteration's correspondence of source sl
  unsigned epoch = 0;
  bool failed = false;

#ifdef LSLAM_COOP_PROFILE
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = coop_now();
#define COOP_MARK(k) do { const unsigned long long now_ = coop_now(); prof[k] += now_ - pt; pt = now_; } while (0)
#else
#define COOP_MARK(k) do {} while (0)
#endif
  for (;;) {
    ++epoch;
    double T[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) T[r][c] = wave_uniform(sh_T[4 * r + c]);
    NearTie nt = near_tie_setup(false, 0.f, 0.f, 0.f);
    if constexpr (F64) {
      double f2 = 0.0;
#pragma unroll
      for (int r = 0; r < DIM; ++r)
#pragma unroll
        for (int c = 0; c < DIM; ++c) f2 += T[r][c] * T[r][c];
      nt = near_tie_setup(true, Rt, Rs, sqrtf((float)f2) * 1.0001f);
    }

    // ---- search: my source against my slice of the target (exact float32 recipe) ---------------------
    float af[DIM];
    [[maybe_unused]] double ptf[DIM];                                 // T·p on the float32 point, float64 (what the float32-only mode feeds the moments)
#pragma unroll
    for (int r = 0; r < DIM; ++r) {                  // icp.py:116
      double v = T[r][0] * (double)pf[0] + T[r][1] * (double)pf[1] + T[r][3];
      if constexpr (DIM == 3) v += T[r][2] * (double)pf[2];
      ptf[r] = v;
      af[r] = (float)v;
    }
    auto d2 = [&](const float (&p)[DIM], const TP& v) -> float {
      if constexpr (DIM == 2) return dist2(p[0], p[1], v.x, v.y);
      else return dist2(p[0], p[1], p[2], v.x, v.y, v.z);
    };
    // a bound from last iteration's match (not yet a candidate: its own chunk beats it strictly; in float64-exact mode by
    // more than the near-tie window, so that the bound is never mistaken for a second candidate) …
    auto reach = [&](float b) { return F64 ? __builtin_fmaf(b, nt.K1, nt.C) : b; };
    const float ds = d2(af, tile[smatch[sl]]);
    const float bound = F64 ? __uint_as_float(__float_as_uint(reach(ds) * 1.000001f) + 1u) : __uint_as_float(__float_as_uint(ds) + 1u);
    // … lets a chunk be skipped when its box is farther away than everything this lane could still accept
    float best = INFINITY, second = INFINITY;
    int bchunk = slice;
    for (int c = slice; c < nchunks; c += NS) {
      float bd2;
      if constexpr (DIM == 2) {
        const float4 b = abox[c];
        const float ox = fmaxf(fmaxf(b.x - af[0], af[0] - b.z), 0.f), oy = fmaxf(fmaxf(b.y - af[1], af[1] - b.w), 0.f);
        bd2 = __builtin_fmaf(oy, oy, ox * ox);
      } else {
        const float4 lo = abox[2 * c], hi = abox[2 * c + 1];
        const float ox = fmaxf(fmaxf(lo.x - af[0], af[0] - hi.x), 0.f), oy = fmaxf(fmaxf(lo.y - af[1], af[1] - hi.y), 0.f),
                    oz = fmaxf(fmaxf(lo.z - af[2], af[2] - hi.z), 0.f);
        bd2 = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
      }
      if (!__any(reach(fminf(best, bound)) >= bd2 * 0.99999f)) continue;      // no lane of this wave can find a nearer (or equally near, or near-tie) target here
      float mn = INFINITY;
#pragma unroll
      for (int k = 0; k < CH; ++k) mn = fminf(mn, d2(af, tile[c * CH + k]));
      const bool lt = mn < best;
      second = __builtin_amdgcn_fmed3f(best, second, mn);
      bchunk = lt ? c : bchunk;
      best = lt ? mn : best;
    }
    unsigned long long key = 0x7f800000ffffffffull;       // an empty slice: distance +inf (all ones would read back as a NaN)
    float sec = second;
    if (best < INFINITY) {                                  // this slice evaluated at least one chunk for this lane's wave
      float bd = INFINITY, sec2 = INFINITY;
      int rot = lane;
      asm volatile("" : "+v"(rot));
#pragma unroll
      for (int kk = 0; kk < CH; ++kk) {
        const int j = bchunk * CH + ((kk + rot) & (CH - 1));
        const float d = d2(af, tile[j]);
        const unsigned long long kj = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)j;
        key = kj < key ? kj : key;
        sec2 = __builtin_amdgcn_fmed3f(bd, sec2, d);
        bd = fminf(bd, d);
      }
      sec = fminf(second, sec2);
    }
    skey[slice * SPW + sl] = key;
    ssec[slice * SPW + sl] = sec;
    COOP_MARK(0);
    __syncthreads();
    COOP_MARK(1);

    // ---- wave 0: the minimum over the slices (and the two smallest distances seen) --------------------
    if (wave == 0) {
      unsigned long long kmin = ~0ull;
      float v1 = INFINITY, v2 = INFINITY;
      {
        // all 16 loads first, then the fold: issued one by one behind the fold's dependency each load costs a full LDS round trip
        unsigned long long ks[kCoopWaves];
        float scs[kCoopWaves];
#pragma unroll
        for (int u = 0; u < kCoopWaves; ++u) { ks[u] = skey[(sub + u * SUB) * SPW + sl]; scs[u] = ssec[(sub + u * SUB) * SPW + sl]; }
#pragma unroll
        for (int u = 0; u < kCoopWaves; ++u) {
          const unsigned long long k = ks[u];
          kmin = k < kmin ? k : kmin;
          const float b = __uint_as_float((unsigned)(k >> 32)), sc = scs[u];
          v2 = __builtin_amdgcn_fmed3f(v1, v2, b); v1 = fminf(v1, b);
          v2 = __builtin_amdgcn_fmed3f(v1, v2, sc); v1 = fminf(v1, sc);
        }
      }
#pragma unroll
      for (int o = SPW; o < 64; o <<= 1) {             // lanes sl, sl + SPW, … hold the same source
        const unsigned long long ok = __shfl_xor(kmin, o, 64);
        const float o1 = __shfl_xor(v1, o, 64), o2 = __shfl_xor(v2, o, 64);
        kmin = ok < kmin ? ok : kmin;
        const float hi = fmaxf(v1, o1);
        v1 = fminf(v1, o1);
        v2 = fminf(hi, fminf(v2, o2));
      }
      match = (int)(unsigned)kmin;
      if constexpr (F64) {
        // near ties: one entry per flagged source {lim, af, ad, lane}; the whole workgroup examines them below
        const float l = near_lim(v1, nt);
        const bool flagged = valid && sub == 0 && v2 <= l;
        const unsigned long long mask = __ballot(flagged);
        if (flagged) {
          const int e = __popcll(mask & ((1ull << lane) - 1ull));
          double* en = sent + e * 8;
          en[0] = (double)l;
#pragma unroll
          for (int r = 0; r < DIM; ++r) {
            double v = T[r][0] * pd[0] + T[r][1] * pd[1] + T[r][3];      // icp.py:116 on the float64 point
            if constexpr (DIM == 3) v += T[r][2] * pd[2];
            en[1 + r] = (double)af[r];
            en[4 + r] = v;
          }
          en[7] = (double)lane;
          sbest[e] = ~0ull;
          sbj[e] = 0x7fffffff;
        }
        if (lane == 0) sh_flag[1] = __popcll(mask);
      }
    }
    if constexpr (F64) {
      __syncthreads();
      const int ne = sh_flag[1];
      if (ne > 0) {                                    // some source of this workgroup has a near tie: every thread looks at its share of the target
        for (int pass = 0; pass < 2; ++pass) {
          for (int e = 0; e < ne; ++e) {
            const double* en = sent + e * 8;
            const float l = (float)en[0];
            float ap[DIM];
#pragma unroll
            for (int r = 0; r < DIM; ++r) ap[r] = (float)en[1 + r];
            for (int j = tid; j < m; j += kCoopThreads) {
              if (d2(ap, tile[j]) <= l) {              // a candidate: its float64 distance decides (lowest index on float64 ties)
                double ev = 0.0;
#pragma unroll
                for (int r = 0; r < DIM; ++r) { const double df = en[4 + r] - tile64[DIM * j + r]; ev += df * df; }
                const unsigned long long bits = (unsigned long long)__double_as_longlong(ev);     // non-negative doubles order like their bits
                if (pass == 0) atomicMin(&sbest[e], bits);
                else if (bits == sbest[e]) atomicMin(&sbj[e], j);
              }
            }
          }
          __syncthreads();
        }
        if (wave == 0) {
          for (int e = 0; e < ne; ++e) {
            const int owner = (int)sent[e * 8 + 7];
            const int bj = sbj[e];
            if (lane % SPW == owner) match = bj;       // every sub-lane of that source
          }
        }
      }
    }

    if (wave == 0 && sub == 0) smatch[sl] = match;          // read by every wave after the barrier that ends the iteration
    COOP_MARK(2);
    // ---- wave 0: float64 moments of my pairs → transposed reduction → this workgroup's row → all rows → solve ------
    if (wave == 0) {
      double vals[kVals];
      {
        double p[DIM], q[DIM];
#pragma unroll
        for (int r = 0; r < DIM; ++r) {
          if constexpr (F64) {
            double v = T[r][0] * pd[0] + T[r][1] * pd[1] + T[r][3];
            if constexpr (DIM == 3) v += T[r][2] * pd[2];
            p[r] = v;
          } else {
            p[r] = ptf[r];
          }
        }
        if constexpr (F64) {
#pragma unroll
          for (int r = 0; r < DIM; ++r) q[r] = tile64[DIM * match + r];
        } else {
          const TP v = tile[match];
          q[0] = (double)v.x; q[1] = (double)v.y;
          if constexpr (DIM == 3) q[2] = (double)v.z;
        }
        const bool mine = valid && sub == 0;
        double e = 0.0;
#pragma unroll
        for (int r = 0; r < DIM; ++r) {
          vals[r] = mine ? p[r] : 0.0;
          vals[DIM + r] = mine ? q[r] : 0.0;
#pragma unroll
          for (int c = 0; c < DIM; ++c) vals[2 * DIM + DIM * r + c] = mine ? p[r] * q[c] : 0.0;
          const double d = p[r] - q[r];
          e += d * d;
          vals[kSums + r] = mine ? fmin(p[r], q[r]) : INFINITY;
          vals[kSums + DIM + r] = mine ? fmax(p[r], q[r]) : -INFINITY;
        }
        vals[kSums - 1] = mine ? e : 0.0;
      }
      // transposed through LDS: lane (value v, segment g) folds that value over the 64 / NSEG source lanes of its segment
      // in lane order, then the segments are folded — a fixed order, 4× fewer operations than a shuffle tree per value
#pragma unroll
      for (int v = 0; v < kVals; ++v)
        if (v < nvals) sred[v * 65 + lane] = vals[v];
      const int rv = lane % VP, rg = lane / VP;
      double tot = 0.0;
      const bool is_sum = rv < kSums, is_min = rv < kSums + DIM;
      // sum / min / max chosen with selects, not branches: the loads of a rolled, branchy loop are issued one LDS round trip at a time
      // (without the bounding box every value is a sum: a wave-uniform case, float64 min / max are slow instructions)
      auto fold = [&](double acc_, double x_) {
        if (!box) return acc_ + x_;
        const double s_ = acc_ + x_, lo_ = fmin(acc_, x_), hi_ = fmax(acc_, x_);
        return is_sum ? s_ : (is_min ? lo_ : hi_);
      };
      {
        const double* row = sred + (rv < nvals ? rv : 0) * 65 + rg * (64 / NSEG);
        double xs[64 / NSEG];
#pragma unroll
        for (int k = 0; k < 64 / NSEG; ++k) xs[k] = row[k];
        tot = xs[0];
#pragma unroll
        for (int k = 1; k < 64 / NSEG; ++k) tot = fold(tot, xs[k]);
      }
#pragma unroll
      for (int o = VP; o < 64; o <<= 1) tot = fold(tot, __shfl_xor(tot, o, 64));
      COOP_MARK(3);
      // publish: lane v < nvals owns value v
      typedef unsigned v4u __attribute__((ext_vector_type(4)));
      const int row_bytes = kVals * 16;
      const int par_base = (int)(epoch & 1) * G * row_bytes;
      if (lane < nvals) {
        v4u g;
        g.x = (unsigned)__double2loint(tot); g.y = epoch; g.z = (unsigned)__double2hiint(tot); g.w = epoch;
        __builtin_amdgcn_raw_buffer_store_b128(g, rsrc, par_base + wg * row_bytes + lane * 16, 0, 16);     // aux 16 = sc1: write-through, agent scope
      }
      COOP_MARK(4);
      // gather: sweep every workgroup's granules until both halves of each carry this epoch
      const int total = G * nvals;
      const unsigned long long t0 = coop_now();
      for (;;) {
        bool ok = true;
        for (int t = lane; t < total; t += 64) {
          const int w = t / nvals, v = t - w * nvals;
          const v4u g = __builtin_amdgcn_raw_buffer_load_b128(rsrc, par_base + w * row_bytes + v * 16, 0, 16);
          const bool here = g.y == epoch && g.w == epoch;
          if (here) { rows32[(w * kVals + v) * 2] = g.x; rows32[(w * kVals + v) * 2 + 1] = g.z; }
          ok &= here;
        }
        if (__all(ok)) break;
        if (coop_now() - t0 > a.timeout_ticks) { failed = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      failed = __any(failed);
      COOP_MARK(5);
      // totals in workgroup order (every workgroup: the same bits)
      // totals: lane (value v, quarter qr) folds workgroups qr, qr + 4, … in that order, then the quarters are folded — a fixed
      // order, the same in every workgroup (bit-identical totals), a quarter of the dependent chain
      double col = 0.0;
      {
        const int cv = lane & 15, qr = lane >> 4;
        if (cv < nvals || VP == 32) {
          for (int half = 0; half < (VP == 32 ? 2 : 1); ++half) {          // 3-D rows have up to 22 values: lanes take v and v + 16
            const int v = cv + 16 * half;
            if (v >= nvals) continue;
            // a quarter without a workgroup holds the neutral element; four rows per trip, their loads issued together
            const bool vsum = v < kSums, vmin = v < kSums + DIM;
            const double neutral = vsum ? 0.0 : (vmin ? INFINITY : -INFINITY);
            double acc = neutral;
            for (int w0 = qr; w0 < G; w0 += 16) {
              unsigned lo[4], hi[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int w = w0 + 4 * u, wc = w < G ? w : qr;                 // clamped: a valid row, ignored below
                lo[u] = rows32[(wc * kVals + v) * 2]; hi[u] = rows32[(wc * kVals + v) * 2 + 1];
              }
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const double x = (w0 + 4 * u < G) ? __hiloint2double((int)hi[u], (int)lo[u]) : neutral;
                if (!box) { acc += x; continue; }                               // wave-uniform: sums only
                const double s_ = acc + x, lo_ = fmin(acc, x), hi_ = fmax(acc, x);
                acc = vsum ? s_ : (vmin ? lo_ : hi_);
              }
            }
            sred[(v * 4 + qr)] = acc;
          }
        }
        if (lane < nvals) {
          const double q0 = sred[lane * 4], q1 = sred[lane * 4 + 1], q2 = sred[lane * 4 + 2], q3 = sred[lane * 4 + 3];
          if (!box) col = (q0 + q1) + (q2 + q3);
          else col = (lane < kSums) ? ((q0 + q1) + (q2 + q3)) : ((lane < kSums + DIM) ? fmin(fmin(q0, q1), fmin(q2, q3)) : fmax(fmax(q0, q1), fmax(q2, q3)));
        }
      }
      Moments mo;
      mo.n = (double)n;
#pragma unroll
      for (int k = 0; k < 3; ++k) { mo.sp[k] = 0.0; mo.sq[k] = 0.0; mo.lo[k] = 0.0; mo.hi[k] = 0.0; }
#pragma unroll
      for (int k = 0; k < 9; ++k) mo.spq[k] = 0.0;
      auto bcast = [&](int k) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(col), k);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(col), k);
        return __hiloint2double(hi, lo);
      };
#pragma unroll
      for (int r = 0; r < DIM; ++r) {
        mo.sp[r] = bcast(r);
        mo.sq[r] = bcast(DIM + r);
#pragma unroll
        for (int c = 0; c < DIM; ++c) mo.spq[3 * r + c] = bcast(2 * DIM + DIM * r + c);
        if (box) { mo.lo[r] = bcast(kSums + r); mo.hi[r] = bcast(kSums + DIM + r); }
      }
      mo.sse = bcast(kSums - 1);
      double R[9], pb[3], qb[3];
      step_parts(mo, DIM == 2, R, pb, qb);
      const double tn = step_times_prev_lane(R, pb, qb, sh_T, sred, lane);      // icp.py:119, one element per lane (sred is consumed)
      const double err = error_from_moments(mo, box);           // icp.py:120 (pre-update residual)
      StopState st = *sh_stop;
      const bool stop = stop_after_iteration(st, err, a.epsilon, a.stopping_thresh, a.max_iters);
      if (lane < 16 && !failed) sh_T[lane] = tn;
      if (lane == 0) {
        if (!failed) {
          *sh_stop = st;
          *sh_err = err;
        }
        sh_flag[0] = (stop || failed) ? 1 : 0;
        sh_flag[4] = failed ? 1 : 0;
      }
      COOP_MARK(6);
    }
    __syncthreads();
    COOP_MARK(7);
    if (sh_flag[0]) break;
  }
#ifdef LSLAM_COOP_PROFILE
  if (wg == 0 && tid == 0) for (int k = 0; k < 8; ++k) reinterpret_cast<unsigned long long*>(a.failed + 8)[k] = prof[k];   // 100 MHz ticks per phase, printed by the host wrapper
#endif

  // ---- results: every workgroup holds the same T; workgroup 0 writes it ------------------------------------------
  if (wave == 0 && valid && sub == 0 && a.out_corr) a.out_corr[i] = match;
  if (sh_flag[4] && tid == 0) atomicExch(a.failed, 1u);
  if (wg == 0) {
    if (tid < 16) a.state->T[tid] = sh_T[tid];
    if (tid == 0) { a.state->err = *sh_err; a.state->stop = *sh_stop; a.state->done = 1; }
  }
}

// ---- launch ------------------------------------------------------------------------------------------
// float64-exact mode keeps the float64 target in LDS as well (8·DIM bytes per point beside the float32 tile)
static int64_t coop_max_targets(int dim, bool f64) {
  if (!f64) return dim == 2 ? kIcpTileMaxPlanar : kIcpTileMax3D;
  return dim == 2 ? 4096 : 2048;
}

int icp_coop_sub(int64_t n);

// CUs the co-resident launch may count on: every workgroup needs a CU of its own (≥ 40 KB of LDS and 16 waves each), so the
// grid must not exceed the device's CU count — a partitioned device (CPX: 32 CUs) or a smaller part takes the launch-per-
// iteration path instead.  LSLAM_ICP_COOP_MAX_CUS caps the figure (tests; a service that shares the device).
static int coop_cus() {
  static int cus = -1;
  if (cus < 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    cus = n;
  }
  int c = cus;
  if (const char* e = getenv("LSLAM_ICP_COOP_MAX_CUS")) { const int v = atoi(e); if (v >= 0 && v < c) c = v; }
  return c;
}

bool icp_coop_supported(int dim, int64_t n, int64_t m, bool f64) {
  if (const char* e = getenv("LSLAM_ICP_COOP")) if (atoi(e) == 0) return false;
  if (!(n >= 1 && n < 8192 && m >= 1 && m <= coop_max_targets(dim, f64))) return false;
  const int spw = 64 / icp_coop_sub(n);
  const int64_t G = (n + spw - 1) / spw;
  // half of the CUs at most: the workgroups must all become resident while other streams of the process may hold CUs too
  return G * 2 <= (int64_t)coop_cus();
}

int icp_coop_sub(int64_t n) {
  int sub = 1;                                                   // measured on the 1081² pair (tools/c2_probe.py): 8.7 µs per iteration against 9.8 (2) and 13.7 (4)
  if (const char* e = getenv("LSLAM_ICP_COOP_SUB")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) sub = v; }
  while (sub > 1 && (n + (64 / sub) - 1) / (64 / sub) > 128) sub >>= 1;      // at most 128 workgroups: one per CU with room to spare
  return sub;
}

size_t icp_coop_gran_bytes(int dim, int64_t n) {
  const int sub = icp_coop_sub(n), spw = 64 / sub;
  const size_t G = (size_t)((n + spw - 1) / spw);
  const size_t kVals = dim == 2 ? CoopRow<2>::kVals : CoopRow<3>::kVals;
  return 2 * G * kVals * 16;
}

hipError_t launch_icp_coop(CoopArgs a, int dim, hipStream_t stream) {
  const int sub = icp_coop_sub(a.n), spw = 64 / sub;
  const int G = (a.n + spw - 1) / spw;
  const int CH = dim == 2 ? CoopChunk<2>::value : CoopChunk<3>::value;
  const int kVals = dim == 2 ? CoopRow<2>::kVals : CoopRow<3>::kVals;
  const bool f64 = a.src64 != nullptr;
  const int padded = (a.m + CH - 1) / CH * CH;
  a.tile_bytes = (uint32_t)(((size_t)padded * (dim == 2 ? 8 : 16) + 15) & ~(size_t)15);
  a.tile64_bytes = f64 ? (uint32_t)(((size_t)a.m * dim * 8 + 15) & ~(size_t)15) : 0u;
  a.gran_bytes = (uint32_t)icp_coop_gran_bytes(dim, a.n);
  a.box_bytes = (uint32_t)((size_t)(padded / CH) * (dim == 2 ? 16 : 32));
  a.timeout_ticks = kCoopTimeoutTicksDefault;
  if (const char* e = getenv("LSLAM_ICP_COOP_TIMEOUT_US")) { const long long us = atoll(e); if (us >= 0) a.timeout_ticks = (unsigned long long)us * 100ull; }
  const size_t slots = (size_t)kCoopWaves * 64;                  // NS·SPW
  const size_t smem = a.tile_bytes + a.tile64_bytes + a.box_bytes + 64 * 4 + slots * (8 + 4) + (32 * 65 + 64 * 8) * 8 + 64 * (8 + 4) + (size_t)G * kVals * 2 * 4 +
                      17 * 8 + sizeof(StopState) + 32;
  // every polled word starts at zero (epochs count from 1 within the launch)
  hipError_t e = hipMemsetAsync(a.gran, 0, a.gran_bytes, stream);
  if (e != hipSuccess) return e;
  auto go = [&](auto kernel) -> hipError_t {
    if (smem > 48 * 1024) {
      hipError_t ee = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (ee != hipSuccess) return ee;
    }
    hipLaunchKernelGGL(kernel, dim3(G), dim3(kCoopThreads), smem, stream, a);
    return hipGetLastError();
  };
  if (dim == 2) {
    if (sub == 4) return f64 ? go(icp_coop_kernel<2, 4, true>) : go(icp_coop_kernel<2, 4, false>);
    if (sub == 2) return f64 ? go(icp_coop_kernel<2, 2, true>) : go(icp_coop_kernel<2, 2, false>);
    return f64 ? go(icp_coop_kernel<2, 1, true>) : go(icp_coop_kernel<2, 1, false>);
  }
  if (sub == 4) return f64 ? go(icp_coop_kernel<3, 4, true>) : go(icp_coop_kernel<3, 4, false>);
  if (sub == 2) return f64 ? go(icp_coop_kernel<3, 2, true>) : go(icp_coop_kernel<3, 2, false>);
  return f64 ? go(icp_coop_kernel<3, 1, true>) : go(icp_coop_kernel<3, 1, false>);
}

}  // namespace lslam

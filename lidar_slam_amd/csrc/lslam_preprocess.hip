// lslam_preprocess.hip — the LiDAR pre-processing in front of the ICP path, on the device
// (reference: modules/localization.py:201-307, chained at main.py:84-87).
//
//   polar_mask_kernel    get_lidar_data (localization.py:252-307): range gate + polar→Cartesian
//                        (+ the lidar→body offset).  cos/sin of the 1081 beam angles come from the
//                        host (np.linspace / np.cos, identical to the reference's), so x = r·cos + off
//                        is bit-exact.
//   dbscan_mask_kernel   DBSCAN_outliers_removal (localization.py:201-221) needs only "label != −1":
//                        a point survives iff it is a core point (≥ min_samples points within eps,
//                        itself included) or lies within eps of a core point.  Two all-pairs float64
//                        passes out of LDS, one workgroup per scan; the neighbourhood test is
//                        d² = dx·dx + dy·dy ≤ eps² with sklearn's operation order (no FMA).
//   radial_*_kernel      statistical_outliers_removal (localization.py:223-250): mean and std of the
//                        point ranges over ALL scans (two-pass, float64), then range < mean + k·std.
//   offsets / scatter    ordered stream compaction per scan → CSR output (scan order and point
//                        order are preserved, as boolean indexing does in the reference).
// Everything stays in device memory between the stages; the host sees only the final CSR.
#include <hip/hip_runtime.h>

#include "lslam_device.h"
#include "lslam_kernels.h"

namespace lslam {

constexpr int kPreThreads = 256;

__device__ __forceinline__ int block_sum_int(int v, int* sh /*>= blockDim/64 ints*/) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  int t = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  return t;
}

// ---- get_lidar_data ----------------------------------------------------------------------------
__global__ void __launch_bounds__(kPreThreads)
polar_mask_kernel(const double* __restrict__ ranges, int32_t B, const double* __restrict__ cos_a,
                  const double* __restrict__ sin_a, double off_x, double off_y, double rmin, double rmax,
                  double* __restrict__ xy_full, uint8_t* __restrict__ keep, int32_t* __restrict__ counts) {
  __shared__ int sh[kPreThreads / 64];
  const int k = blockIdx.x;
  int mine = 0;
  for (int b = threadIdx.x; b < B; b += kPreThreads) {
    const int64_t i = (int64_t)k * B + b;
    const double r = ranges[i];
    const bool ok = (r >= rmin) && (r <= rmax);
    xy_full[2 * i] = r * cos_a[b] + off_x;
    xy_full[2 * i + 1] = r * sin_a[b] + off_y;
    keep[i] = ok;
    mine += ok;
  }
  const int total = block_sum_int(mine, sh);
  if (threadIdx.x == 0) counts[k] = total;
}

// ---- DBSCAN noise mask ---------------------------------------------------------------------------
// core: uint8 per point (global scratch, so that scans larger than one LDS tile work too)
// Box pruning (round 3): the points of a tile are grouped 16 at a time in the order they arrive — beam order: 16 consecutive
// beams are neighbours on a wall — with a float64 box per group in LDS.  A wave (64 consecutive points, compact for the same
// reason) skips every group whose box is farther than eps from the wave's box, with a margin of 1e-12 relative against the
// rounding of the distance it compares (a skipped group cannot hold a point the full test would have counted).  On scans in
// beam order about an eighth of the groups survive; on points in arbitrary order every group does and the result is the same.
constexpr int kDbGroup = 16;
template <int PASS>
__global__ void __launch_bounds__(kPreThreads)
dbscan_mask_kernel(const double* __restrict__ xy, const int64_t* __restrict__ off, double eps2, int32_t min_samples,
                   uint8_t* __restrict__ core, uint8_t* __restrict__ keep, int32_t* __restrict__ counts, int tile_cap) {
  extern __shared__ __align__(16) unsigned char smem[];
  double2* tile = reinterpret_cast<double2*>(smem);
  double* gbox = reinterpret_cast<double*>(tile + tile_cap);              // per group {x0, y0, x1, y1}
  uint8_t* tcore = reinterpret_cast<uint8_t*>(gbox + 4 * (tile_cap / kDbGroup));
  __shared__ int sh[kPreThreads / 64];
  const int k = blockIdx.x;
  const int64_t p0 = off[k];
  const int n = (int)(off[k + 1] - p0);
  int kept = 0;
  for (int base = 0; base < n || base == 0; base += kPreThreads) {
    const int i = base + threadIdx.x;
    const bool live = i < n;
    double xi = 0.0, yi = 0.0;
    if (live) { xi = xy[2 * (p0 + i)]; yi = xy[2 * (p0 + i) + 1]; }
    // the wave's box (lanes without a point contribute nothing)
    double wx0 = live ? xi : INFINITY, wy0 = live ? yi : INFINITY, wx1 = live ? xi : -INFINITY, wy1 = live ? yi : -INFINITY;
    wx0 = wave_min(wx0); wy0 = wave_min(wy0); wx1 = wave_max(wx1); wy1 = wave_max(wy1);
    int cnt = 0;          // PASS 0: neighbours within eps;  PASS 1: core neighbours within eps
    for (int t0 = 0; t0 < n; t0 += tile_cap) {
      const int tn = min(tile_cap, n - t0);
      __syncthreads();
      for (int j = threadIdx.x; j < tn; j += kPreThreads) {
        tile[j] = make_double2(xy[2 * (p0 + t0 + j)], xy[2 * (p0 + t0 + j) + 1]);
        if (PASS == 1) tcore[j] = core[p0 + t0 + j];
      }
      __syncthreads();
      const int ngroups = (tn + kDbGroup - 1) / kDbGroup;
      for (int g = threadIdx.x; g < ngroups; g += kPreThreads) {
        double x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
        for (int j = g * kDbGroup; j < min(tn, (g + 1) * kDbGroup); ++j) {
          const double2 q = tile[j];
          x0 = fmin(x0, q.x); y0 = fmin(y0, q.y); x1 = fmax(x1, q.x); y1 = fmax(y1, q.y);
        }
        gbox[4 * g] = x0; gbox[4 * g + 1] = y0; gbox[4 * g + 2] = x1; gbox[4 * g + 3] = y1;
      }
      __syncthreads();
      for (int g = 0; g < ngroups; ++g) {
        // wave-uniform (the same LDS words and the same wave box in every lane)
        const double gx = fmax(fmax(gbox[4 * g] - wx1, wx0 - gbox[4 * g + 2]), 0.0);
        const double gy = fmax(fmax(gbox[4 * g + 1] - wy1, wy0 - gbox[4 * g + 3]), 0.0);
        if ((gx * gx + gy * gy) * (1.0 - 1e-12) > eps2) continue;
        if (live) {
          const int j1 = min(tn, (g + 1) * kDbGroup);
          for (int j = g * kDbGroup; j < j1; ++j) {
            const double2 q = tile[j];                       // wave-uniform address: LDS broadcast
            const double dx = xi - q.x, dy = yi - q.y;
            const double d2 = dx * dx + dy * dy;             // sklearn rdist order, no FMA (-ffp-contract=off)
            if (PASS == 0) cnt += (d2 <= eps2);
            else cnt += (d2 <= eps2) & (tcore[j] != 0);
          }
        }
      }
    }
    if (live) {
      if (PASS == 0) core[p0 + i] = cnt >= min_samples;
      else {
        const bool kp = core[p0 + i] != 0 || cnt > 0;
        keep[p0 + i] = kp;
        kept += kp;
      }
    }
    if (base + kPreThreads >= n) break;
  }
  if (PASS == 1) {
    const int total = block_sum_int(kept, sh);
    if (threadIdx.x == 0) counts[k] = total;
  }
}

// ---- radial statistics -----------------------------------------------------------------------------
// mode 0: Σ range;  mode 1: Σ (range − mean)²   → one partial per block
__global__ void __launch_bounds__(kPreThreads)
radial_partial_kernel(const double* __restrict__ xy, int64_t total, int mode, const double* __restrict__ stats,
                      double* __restrict__ partial) {
  __shared__ double sh[kPreThreads / 64];
  const double mean = mode ? stats[0] : 0.0;
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kPreThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kPreThreads) {
    const double x = xy[2 * i], y = xy[2 * i + 1];
    const double d = sqrt(x * x + y * y);
    s += mode ? (d - mean) * (d - mean) : d;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kPreThreads / 64; ++w) t += sh[w];
    partial[blockIdx.x] = t;
  }
}

// stats = {mean, std, threshold}; mode 0 fills mean, mode 1 fills std and threshold
__global__ void radial_finish_kernel(const double* __restrict__ partial, int nparts, int64_t total, int mode,
                                     double k_std, double* __restrict__ stats) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double t = 0.0;
  for (int i = 0; i < nparts; ++i) t += partial[i];
  if (mode == 0) stats[0] = t / (double)total;
  else {
    stats[1] = sqrt(t / (double)total);
    stats[2] = stats[0] + k_std * stats[1];
  }
}

__global__ void __launch_bounds__(kPreThreads)
radial_mask_kernel(const double* __restrict__ xy, const int64_t* __restrict__ off, const double* __restrict__ stats,
                   uint8_t* __restrict__ keep, int32_t* __restrict__ counts) {
  __shared__ int sh[kPreThreads / 64];
  const int k = blockIdx.x;
  const int64_t p0 = off[k];
  const int n = (int)(off[k + 1] - p0);
  const double thr = stats[2];
  int mine = 0;
  for (int i = threadIdx.x; i < n; i += kPreThreads) {
    const double x = xy[2 * (p0 + i)], y = xy[2 * (p0 + i) + 1];
    const bool ok = sqrt(x * x + y * y) < thr;                      // localization.py:244
    keep[p0 + i] = ok;
    mine += ok;
  }
  const int total = block_sum_int(mine, sh);
  if (threadIdx.x == 0) counts[k] = total;
}

// ---- compaction --------------------------------------------------------------------------------------
// exclusive prefix sum of K counts into K+1 offsets (single workgroup, chunks of 1024 with a carry)
__global__ void __launch_bounds__(1024)
offsets_scan_kernel(const int32_t* __restrict__ counts, int32_t K, int64_t* __restrict__ out_off) {
  __shared__ long long sh[1024];
  __shared__ long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < K; base += 1024) {
    const int i = base + threadIdx.x;
    const long long v = (i < K) ? counts[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                 // Hillis–Steele inclusive scan
      const long long add = (threadIdx.x >= (unsigned)o) ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < K) out_off[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) out_off[K] = carry;
}

// ordered scatter of the kept points of scan k to out_xy[out_off[k] ...]; in_stride_off: offsets of the
// input (nullptr = dense rows of B points)
__global__ void __launch_bounds__(kPreThreads)
compact_scatter_kernel(const double* __restrict__ xy, const int64_t* __restrict__ in_off, int32_t B,
                       const uint8_t* __restrict__ keep, const int64_t* __restrict__ out_off,
                       double* __restrict__ out_xy) {
  __shared__ int wave_cnt[kPreThreads / 64];
  __shared__ int running;
  const int k = blockIdx.x;
  const int64_t p0 = in_off ? in_off[k] : (int64_t)k * B;
  const int n = in_off ? (int)(in_off[k + 1] - p0) : B;
  const int64_t q0 = out_off[k];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  for (int base = 0; base < n; base += kPreThreads) {
    const int i = base + threadIdx.x;
    const bool kp = (i < n) && keep[p0 + i] != 0;
    const unsigned long long m = __ballot(kp);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int prior = running;
    for (int w = 0; w < wave; ++w) prior += wave_cnt[w];
    if (kp) {
      const int64_t dst = q0 + prior + before;
      out_xy[2 * dst] = xy[2 * (p0 + i)];
      out_xy[2 * dst + 1] = xy[2 * (p0 + i) + 1];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < kPreThreads / 64; ++w) t += wave_cnt[w];
      running += t;
    }
    __syncthreads();
  }
}

__global__ void f64_to_f32_kernel(const double* __restrict__ in, float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}
hipError_t launch_f64_to_f32(const double* in, float* out, int64_t n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(f64_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
  return hipGetLastError();
}

// ---- launchers ---------------------------------------------------------------------------------------
hipError_t launch_polar_mask(const double* ranges, int32_t K, int32_t B, const double* cos_a, const double* sin_a,
                             double off_x, double off_y, double rmin, double rmax, double* xy_full, uint8_t* keep,
                             int32_t* counts, hipStream_t st) {
  if (K <= 0) return hipSuccess;
  hipLaunchKernelGGL(polar_mask_kernel, dim3(K), dim3(kPreThreads), 0, st, ranges, B, cos_a, sin_a, off_x, off_y, rmin, rmax, xy_full, keep, counts);
  return hipGetLastError();
}

hipError_t launch_dbscan_mask(const double* xy, const int64_t* off, int32_t K, int64_t max_n, double eps, int32_t min_samples,
                              uint8_t* core, uint8_t* keep, int32_t* counts, hipStream_t st) {
  if (K <= 0) return hipSuccess;
  int tile = (int)(max_n < 64 ? 64 : (max_n > 4096 ? 4096 : max_n));
  tile = (tile + 15) & ~15;
  const size_t smem = (size_t)tile * (sizeof(double2) + 1) + (size_t)(tile / 16) * 4 * sizeof(double) + 16;      // points, core flags, one box per 16 points
  const double eps2 = eps * eps;                                   // sklearn: reduced radius = r²
  hipLaunchKernelGGL(dbscan_mask_kernel<0>, dim3(K), dim3(kPreThreads), smem, st, xy, off, eps2, min_samples, core, keep, counts, tile);
  hipLaunchKernelGGL(dbscan_mask_kernel<1>, dim3(K), dim3(kPreThreads), smem, st, xy, off, eps2, min_samples, core, keep, counts, tile);
  return hipGetLastError();
}

hipError_t launch_radial_mask(const double* xy, const int64_t* off, int32_t K, int64_t total, double k_std, double* partial /*kRadialParts*/,
                              double* stats /*3*/, uint8_t* keep, int32_t* counts, hipStream_t st) {
  if (K <= 0) return hipSuccess;
  int parts = (int)((total + kPreThreads - 1) / kPreThreads);
  if (parts > kRadialParts) parts = kRadialParts;
  if (parts < 1) parts = 1;
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(radial_partial_kernel, dim3(parts), dim3(kPreThreads), 0, st, xy, total, mode, stats, partial);
    hipLaunchKernelGGL(radial_finish_kernel, dim3(1), dim3(64), 0, st, partial, parts, total, mode, k_std, stats);
  }
  hipLaunchKernelGGL(radial_mask_kernel, dim3(K), dim3(kPreThreads), 0, st, xy, off, stats, keep, counts);
  return hipGetLastError();
}

hipError_t launch_compact(const double* xy, const int64_t* in_off, int32_t B, const uint8_t* keep, const int32_t* counts,
                          int32_t K, int64_t* out_off, double* out_xy, hipStream_t st) {
  hipLaunchKernelGGL(offsets_scan_kernel, dim3(1), dim3(1024), 0, st, counts, K, out_off);
  if (K > 0) hipLaunchKernelGGL(compact_scatter_kernel, dim3(K), dim3(kPreThreads), 0, st, xy, in_off, B, keep, out_off, out_xy);
  return hipGetLastError();
}

}  // namespace lslam

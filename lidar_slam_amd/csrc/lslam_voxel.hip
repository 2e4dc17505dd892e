// lslam_voxel.hip — voxel_downsample (modules/icp.py:4-27) on the device.
//
// The reference floor-quantises every point to an integer voxel index, takes np.unique(axis=0) of the
// indices (→ voxels in lexicographic order) and averages the points of each voxel in a Python loop.
// Here: float64 min reduction → one 63-bit key per point (21 bits per axis, x most significant, so key
// order = lexicographic index order) → stable radix sort of (key, point id) with rocPRIM → segment
// heads → one thread per voxel averages its points in their original order.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "lslam_device.h"
#include "lslam_kernels.h"

namespace lslam {

constexpr int kVoxThreads = 256;
constexpr int kVoxAxisBits = 21;

__global__ void __launch_bounds__(kVoxThreads)
voxel_min_kernel(const double* __restrict__ pts, int64_t n, int dim, double* __restrict__ partial /*grid x 3*/) {
  __shared__ double sh[kVoxThreads / 64][3];
  double m[3] = {INFINITY, INFINITY, INFINITY};
  for (int64_t i = (int64_t)blockIdx.x * kVoxThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kVoxThreads)
    for (int d = 0; d < dim; ++d) m[d] = fmin(m[d], pts[i * dim + d]);
  for (int d = 0; d < 3; ++d) m[d] = wave_min(m[d]);
  if ((threadIdx.x & 63) == 0) for (int d = 0; d < 3; ++d) sh[threadIdx.x >> 6][d] = m[d];
  __syncthreads();
  if (threadIdx.x < 3) {
    double v = sh[0][threadIdx.x];
    for (int w = 1; w < kVoxThreads / 64; ++w) v = fmin(v, sh[w][threadIdx.x]);
    partial[3 * blockIdx.x + threadIdx.x] = v;
  }
}

__global__ void voxel_min_finish_kernel(const double* __restrict__ partial, int parts, double* __restrict__ lo) {
  if (threadIdx.x >= 3) return;
  double v = INFINITY;
  for (int i = 0; i < parts; ++i) v = fmin(v, partial[3 * i + threadIdx.x]);
  lo[threadIdx.x] = v;
}

// key = (ix << 42) | (iy << 21) | iz with i = floor((p − min) / voxel)  (icp.py:19); *bad is raised
// if an index does not fit 21 bits
__global__ void __launch_bounds__(kVoxThreads)
voxel_key_kernel(const double* __restrict__ pts, int64_t n, int dim, const double* __restrict__ lo, double voxel,
                 unsigned long long* __restrict__ keys, uint32_t* __restrict__ ids, int* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * kVoxThreads + threadIdx.x;
  if (i >= n) return;
  unsigned long long key = 0;
  for (int d = 0; d < dim; ++d) {
    const double q = floor((pts[i * dim + d] - lo[d]) / voxel);
    if (!(q >= 0.0 && q < (double)(1u << kVoxAxisBits))) { *bad = 1; }
    key = (key << kVoxAxisBits) | (unsigned long long)(q >= 0.0 ? (unsigned long long)q & ((1ull << kVoxAxisBits) - 1) : 0ull);
  }
  keys[i] = key;
  ids[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(kVoxThreads)
voxel_heads_kernel(const unsigned long long* __restrict__ keys, int64_t n, uint32_t* __restrict__ head) {
  const int64_t i = (int64_t)blockIdx.x * kVoxThreads + threadIdx.x;
  if (i >= n) return;
  head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// seg[i] = inclusive scan of head − 1; a head element records where its segment starts
__global__ void __launch_bounds__(kVoxThreads)
voxel_starts_kernel(const uint32_t* __restrict__ head, const uint32_t* __restrict__ seg_incl, int64_t n,
                    int64_t* __restrict__ starts) {
  const int64_t i = (int64_t)blockIdx.x * kVoxThreads + threadIdx.x;
  if (i >= n) return;
  if (head[i]) starts[seg_incl[i] - 1] = i;
  if (i == n - 1) starts[seg_incl[i]] = n;
}

__global__ void __launch_bounds__(kVoxThreads)
voxel_mean_kernel(const double* __restrict__ pts, int dim, const uint32_t* __restrict__ ids,
                  const int64_t* __restrict__ starts, int64_t nvox, double* __restrict__ out) {
  const int64_t v = (int64_t)blockIdx.x * kVoxThreads + threadIdx.x;
  if (v >= nvox) return;
  const int64_t a = starts[v], b = starts[v + 1];
  double s[3] = {0.0, 0.0, 0.0};
  for (int64_t i = a; i < b; ++i) {               // stable sort: original point order inside a voxel
    const int64_t p = ids[i];
    for (int d = 0; d < dim; ++d) s[d] += pts[p * dim + d];
  }
  const double cnt = (double)(b - a);
  for (int d = 0; d < dim; ++d) out[v * dim + d] = s[d] / cnt;
}

size_t voxel_temp_bytes(int64_t n) {
  size_t a = 0, b = 0;
  unsigned long long* k = nullptr; uint32_t* v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, a, k, k, v, v, (size_t)n, 0, 3 * kVoxAxisBits);
  (void)rocprim::inclusive_scan(nullptr, b, v, v, (size_t)n, rocprim::plus<uint32_t>());
  return a > b ? a : b;
}

// all buffers on the device; returns the number of voxels through *h_nvox (synchronises the stream once)
hipError_t run_voxel_downsample(const double* pts, int64_t n, int dim, double voxel, VoxelWs ws, double* out,
                                int64_t* h_nvox, int* h_bad, hipStream_t st) {
  int parts = (int)((n + kVoxThreads - 1) / kVoxThreads);
  if (parts > 1024) parts = 1024;
  const unsigned blocks = (unsigned)((n + kVoxThreads - 1) / kVoxThreads);
  hipLaunchKernelGGL(voxel_min_kernel, dim3(parts), dim3(kVoxThreads), 0, st, pts, n, dim, ws.partial);
  hipLaunchKernelGGL(voxel_min_finish_kernel, dim3(1), dim3(64), 0, st, ws.partial, parts, ws.lo);
  hipError_t e = hipMemsetAsync(ws.bad, 0, sizeof(int), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(voxel_key_kernel, dim3(blocks), dim3(kVoxThreads), 0, st, pts, n, dim, ws.lo, voxel, ws.keys_in, ws.ids_in, ws.bad);
  size_t tb = ws.temp_bytes;
  e = rocprim::radix_sort_pairs(ws.temp, tb, ws.keys_in, ws.keys_out, ws.ids_in, ws.ids_out, (size_t)n, 0, 3 * kVoxAxisBits, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(voxel_heads_kernel, dim3(blocks), dim3(kVoxThreads), 0, st, ws.keys_out, n, ws.ids_in /*reused: head flags*/);
  tb = ws.temp_bytes;
  uint32_t* seg = reinterpret_cast<uint32_t*>(ws.keys_in);          // reused: inclusive scan of the flags
  e = rocprim::inclusive_scan(ws.temp, tb, ws.ids_in, seg, (size_t)n, rocprim::plus<uint32_t>(), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(voxel_starts_kernel, dim3(blocks), dim3(kVoxThreads), 0, st, ws.ids_in, seg, n, ws.starts);
  uint32_t last = 0;
  e = hipMemcpyAsync(&last, seg + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return e;
  e = hipMemcpyAsync(h_bad, ws.bad, sizeof(int), hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  *h_nvox = last;
  if (*h_bad) return hipSuccess;
  hipLaunchKernelGGL(voxel_mean_kernel, dim3((unsigned)((last + kVoxThreads - 1) / kVoxThreads)), dim3(kVoxThreads), 0, st,
                     pts, dim, ws.ids_out, ws.starts, (int64_t)last, out);
  return hipGetLastError();
}

}  // namespace lslam

// lslam_ogm.hip — per-scan log-odds ray casting (modules/ogm.py:149-231) on a grid resident in HBM.
//
// Two launches per scan, in stream order (scans must be applied in order: the ±clip after every
// scan is not associative):
//   ogm_raycast_kernel  one wave64 per ray.  The end cell comes from the float64 world→grid
//                       formula (ogm.py:117-118; float64 is what makes the indices bit-exact);
//                       every Bresenham cell is computed independently from the closed form
//                       minor(k) = (k·dy + dx − 1 − ⌊dx/2⌋) / dx of the reference's error sequence
//                       (ogm.py:211), cells outside the map are dropped AFTER tracing (ogm.py:175-181),
//                       the last valid cell gets +1 and every other valid cell −1 (ogm.py:184-185)
//                       as int32 atomics into a scratch grid; the touched bounding box is recorded.
//   ogm_apply_kernel    l ← f32(clip(f64(l) + cnt·log 4, ±20)) on the touched box, scratch re-zeroed.
// The integer scratch grid is the bit-exact contract; it also makes the float result independent
// of the order in which rays land (atomics commute on integers).
#include <hip/hip_runtime.h>

#include "lslam_device.h"
#include "lslam_kernels.h"

namespace lslam {

constexpr int kRayThreads = 256;          // 4 rays per workgroup
__device__ __forceinline__ int box_init(int i) { return (i & 2) ? -0x7fffffff : 0x7fffffff; }   // {min,min,max,max}

__device__ __forceinline__ int world2grid(double v, double vmin, double res) {
  return (int)ceil((v - vmin) / res) - 1;                 // ogm.py:117-118
}

__global__ void __launch_bounds__(kRayThreads)
ogm_raycast_kernel(OgmGrid g, OgmScan sc) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * (kRayThreads / 64) + (threadIdx.x >> 6);
  if (blockIdx.x == 0 && threadIdx.x < 4) g.bbox[4 * (sc.parity ^ 1) + threadIdx.x] = box_init(threadIdx.x);
  if (ray >= sc.n) return;

  // ray origin: (x, y) + (0.13323, 0), NOT rotated (ogm.py:165-166)
  const int sx = world2grid(sc.x + 0.13323, g.min_x, g.res);
  const int sy = world2grid(sc.y + 0.0, g.min_y, g.res);
  // beam end: rotate about the robot origin, translate (ogm.py:160-162)
  const double zx = sc.scan_xy[2 * (int64_t)ray], zy = sc.scan_xy[2 * (int64_t)ray + 1];
  const double wx = zx * sc.c + zy * (-sc.s) + sc.x;
  const double wy = zx * sc.s + zy * sc.c + sc.y;
  const int ex = world2grid(wx, g.min_x, g.res);
  const int ey = world2grid(wy, g.min_y, g.res);

  int dx = abs(ex - sx), dy = abs(ey - sy);
  const bool steep = dy > dx;
  if (steep) { const int t = dx; dx = dy; dy = t; }
  const int sgx = (sx <= ex) ? 1 : -1, sgy = (sy <= ey) ? 1 : -1;
  const int ncell = dx + 1;
  const int bias = dx - 1 - dx / 2;
  const bool small = dx < 46340;                 // k·dy + bias < 2³¹: 32-bit division is enough

  auto cell = [&](int k, int& cx, int& cy) {
    int minor = 0;
    if (dy != 0)
      minor = small ? (int)(((unsigned)k * (unsigned)dy + (unsigned)bias) / (unsigned)dx)
                    : (int)(((long long)k * dy + bias) / dx);
    if (steep) { cx = sx + sgx * minor; cy = sy + sgy * k; }
    else       { cx = sx + sgx * k;     cy = sy + sgy * minor; }
    return cx >= 0 && cx < g.width && cy >= 0 && cy < g.height;
  };

  // pass 1: last valid k and the touched box
  int klast = -1, bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = -0x7fffffff, by1 = -0x7fffffff;
  for (int k = lane; k < ncell; k += 64) {
    int cx, cy;
    if (cell(k, cx, cy)) {
      klast = k;                                // k grows with the loop: the lane's last valid
      bx0 = min(bx0, cx); by0 = min(by0, cy); bx1 = max(bx1, cx); by1 = max(by1, cy);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    klast = max(klast, __shfl_xor(klast, o, 64));
    bx0 = min(bx0, __shfl_xor(bx0, o, 64)); by0 = min(by0, __shfl_xor(by0, o, 64));
    bx1 = max(bx1, __shfl_xor(bx1, o, 64)); by1 = max(by1, __shfl_xor(by1, o, 64));
  }
  if (klast < 0) return;                        // no valid cell: the ray is skipped (ogm.py:182-183)
  if (lane == 0) {
    int* b = g.bbox + 4 * sc.parity;
    atomicMin(b + 0, bx0); atomicMin(b + 1, by0); atomicMax(b + 2, bx1); atomicMax(b + 3, by1);
  }
  // pass 2: −1 on every valid cell but the last valid one, +1 there
  for (int k = lane; k < ncell; k += 64) {
    int cx, cy;
    if (cell(k, cx, cy)) atomicAdd(g.counts + (int64_t)cx * g.height + cy, (k == klast) ? 1 : -1);
  }
}

__global__ void __launch_bounds__(256)
ogm_apply_kernel(OgmGrid g, int parity) {
  const int* b = g.bbox + 4 * parity;
  const int x0 = b[0], y0 = b[1], x1 = b[2], y1 = b[3];
  if (x1 < x0) return;
  // blockIdx.x tiles y (contiguous in memory), blockIdx.y tiles x
  const int cy = y0 + blockIdx.x * 64 + (threadIdx.x & 63);
  const int cxb = x0 + blockIdx.y * 16 + (threadIdx.x >> 6) * 4;
  if (cy > y1) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int cx = cxb + r;
    if (cx > x1) break;
    const int64_t idx = (int64_t)cx * g.height + cy;
    const int c = g.counts[idx];
    if (c != 0) {
      double l = (double)g.logodds[idx] + (double)c * g.step;
      l = fmin(fmax(l, -g.clip), g.clip);
      g.logodds[idx] = (float)l;
      g.counts[idx] = 0;
    }
  }
}

// re-zero the scratch grid on the touched box without applying it (scan_counts entry point)
__global__ void __launch_bounds__(256)
ogm_clear_kernel(OgmGrid g, int parity) {
  const int* b = g.bbox + 4 * parity;
  const int x0 = b[0], y0 = b[1], x1 = b[2], y1 = b[3];
  if (x1 < x0) return;
  const int cy = y0 + blockIdx.x * 64 + (threadIdx.x & 63);
  const int cxb = x0 + blockIdx.y * 16 + (threadIdx.x >> 6) * 4;
  if (cy > y1) return;
  for (int r = 0; r < 4; ++r) {
    const int cx = cxb + r;
    if (cx > x1) break;
    g.counts[(int64_t)cx * g.height + cy] = 0;
  }
}

__global__ void ogm_binarize_kernel(OgmGrid g, uint8_t* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)g.width * g.height) return;
  const float pmf = 1.0f / (1.0f + expf(g.logodds[i]));        // ogm.py:61
  out[i] = pmf > 0.5f ? 1 : 0;                                 // ogm.py:62-64
}

__global__ void ogm_init_kernel(OgmGrid g) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 8) g.bbox[i] = box_init((int)i & 3);
  if (i >= (int64_t)g.width * g.height) return;
  g.logodds[i] = 0.f;
  g.counts[i] = 0;
}

hipError_t launch_ogm_raycast(const OgmGrid& g, const OgmScan& sc, hipStream_t stream) {
  const int blocks = (sc.n + (kRayThreads / 64) - 1) / (kRayThreads / 64);
  hipLaunchKernelGGL(ogm_raycast_kernel, dim3(blocks > 0 ? blocks : 1), dim3(kRayThreads), 0, stream, g, sc);
  return hipGetLastError();
}
static dim3 box_grid(const OgmGrid& g) { return dim3((g.height + 63) / 64, (g.width + 15) / 16); }
hipError_t launch_ogm_apply(const OgmGrid& g, int parity, hipStream_t stream) {
  hipLaunchKernelGGL(ogm_apply_kernel, box_grid(g), dim3(256), 0, stream, g, parity);
  return hipGetLastError();
}
hipError_t launch_ogm_clear_counts(const OgmGrid& g, int parity, hipStream_t stream) {
  hipLaunchKernelGGL(ogm_clear_kernel, box_grid(g), dim3(256), 0, stream, g, parity);
  return hipGetLastError();
}
hipError_t launch_ogm_binarize(const OgmGrid& g, uint8_t* out, hipStream_t stream) {
  const int64_t n = (int64_t)g.width * g.height;
  hipLaunchKernelGGL(ogm_binarize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, out);
  return hipGetLastError();
}
hipError_t launch_ogm_init(const OgmGrid& g, hipStream_t stream) {
  const int64_t n = (int64_t)g.width * g.height;
  hipLaunchKernelGGL(ogm_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g);
  return hipGetLastError();
}

}  // namespace lslam

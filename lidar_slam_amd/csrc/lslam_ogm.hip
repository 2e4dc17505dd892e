// lslam_ogm.hip — log-odds ray casting (modules/ogm.py:149-231) on a grid resident in HBM.
//
// The reference traces one scan at a time and clips the whole grid after every scan
// (ogm.py:172-188), so scans must take effect in order (the clip is not associative).
// What IS independent is the integer "hits − misses" grid of each scan.  Scans are therefore
// processed in groups of up to kOgmPlanes:
//   ogm_raycast_kernel  grid = (ray blocks, scans of the group).  Scan b accumulates its int32
//                       hit−miss counts into its own plane b.  One wave64 per ray:
//                         * end cell from the float64 world→grid formula (ogm.py:117-118; float64 is
//                           what makes the indices bit-exact);
//                         * the in-map part of the ray is an interval [k_first, k_last] of the major
//                           axis, found in O(1) (the reference filters cells after tracing,
//                           ogm.py:175-181; the hit goes to the LAST valid cell, ogm.py:185);
//                         * every cell from the closed form minor(k) = ⌊(k·dy + dx − 1 − ⌊dx/2⌋)/dx⌋
//                           of the reference's error sequence (ogm.py:211), as a multiply-high;
//                         * −1 / +1 as int32 atomics — into an LDS window around the ray origin
//                           (all rays of a scan leave one cell, so nearby cells see hundreds of
//                           updates), flushed once per workgroup; straight to HBM beyond it.
//   ogm_apply_kernel    one thread per cell walks the planes IN SCAN ORDER:
//                       l ← f32(clip(f64(l) + cnt·log 4, ±20)), plane cell re-zeroed.
// Integer counts make the result independent of the order in which rays land.
#include <hip/hip_runtime.h>

#include "lslam_device.h"
#include "lslam_kernels.h"

namespace lslam {

constexpr int kRayThreads = 1024;         // 16 waves per workgroup, one ray per wave
constexpr int kRaysPerBlock = 16;
constexpr int kWin = 96;                  // LDS window (cells per side) centred on the ray origin
__device__ __forceinline__ int box_init(int i) { return (i & 2) ? -0x7fffffff : 0x7fffffff; }   // {min,min,max,max}

__device__ __forceinline__ int world2grid(double v, double vmin, double res) {
  return (int)ceil((v - vmin) / res) - 1;                 // ogm.py:117-118
}

__global__ void __launch_bounds__(kRayThreads)
ogm_raycast_kernel(OgmGrid g, const OgmScan* __restrict__ scans, int parity) {
  __shared__ int win[kWin * kWin];
  __shared__ int sbox[4];
  const int b = blockIdx.y;
  const OgmScan sc = scans[b];
  // the other parity's boxes are free now (their apply launch is complete): reset them
  if (blockIdx.x == 0 && b == 0 && threadIdx.x < 4 * kOgmPlanes) g.bbox[4 * kOgmPlanes * (parity ^ 1) + threadIdx.x] = box_init(threadIdx.x & 3);
  if ((int)blockIdx.x * kRaysPerBlock >= sc.n) return;
  int* __restrict__ plane = g.counts + (int64_t)b * g.width * g.height;

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < kWin * kWin; i += kRayThreads) win[i] = 0;
  if (threadIdx.x < 4) sbox[threadIdx.x] = box_init(threadIdx.x);
  __syncthreads();

  // ray origin: (x, y) + (0.13323, 0), NOT rotated (ogm.py:165-166)
  const int sx = world2grid(sc.x + 0.13323, g.min_x, g.res);
  const int sy = world2grid(sc.y + 0.0, g.min_y, g.res);
  const int wx0 = sx - kWin / 2, wy0 = sy - kWin / 2;

  const int ray = blockIdx.x * kRaysPerBlock + wave;
  if (ray < sc.n) {
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
    const int bias = dx - 1 - dx / 2;
    // major/minor axis views of the start cell, the step signs and the map extent
    const int sM = steep ? sy : sx, sm = steep ? sx : sy;
    const int gM = steep ? sgy : sgx, gm = steep ? sgx : sgy;
    const int WM = steep ? g.height : g.width, Wm = steep ? g.width : g.height;

    // valid k range in O(1).  Cell k = (sM + gM·k, sm + gm·u(k)), u(k) = ⌊(k·dy + bias)/dx⌋ is
    // non-decreasing, so "inside the map" is an interval of k.
    long long k_first = 0, k_last = dx;
    k_first = max(k_first, (long long)(gM > 0 ? -sM : sM - (WM - 1)));
    k_last = min(k_last, (long long)(gM > 0 ? WM - 1 - sM : sM));
    const long long ulo = gm > 0 ? -(long long)sm : (long long)sm - (Wm - 1);
    const long long uhi = gm > 0 ? (long long)Wm - 1 - sm : (long long)sm;
    if (dy == 0) {
      if (ulo > 0 || uhi < 0) k_last = -1;
    } else {
      if (ulo > 0) k_first = max(k_first, (ulo * dx - bias + dy - 1) / dy);          // u(k) ≥ ulo
      if (uhi < 0) k_last = -1;
      else k_last = min(k_last, ((uhi + 1) * dx - bias - 1) / dy);                  // u(k) ≤ uhi
    }
    if (k_first <= k_last) {                    // else: no valid cell, the ray is skipped (ogm.py:182-183)
      const bool small = dx < 46340;            // k·dy + bias < 2³¹: 32-bit multiply-high division
      const unsigned magic = dx > 1 ? (unsigned)(0x100000000ull / (unsigned)dx) : 0xffffffffu;
      const double inv_dx = 1.0 / (double)max(dx, 1);
      auto minor_of = [&](int k) -> int {
        if (dy == 0) return 0;
        if (small) {
          const unsigned num = (unsigned)k * (unsigned)dy + (unsigned)bias;
          unsigned q = __umulhi(num, magic);
          unsigned r = num - q * (unsigned)dx;
          if (r >= (unsigned)dx) { q += 1; r -= (unsigned)dx; }
          if (r >= (unsigned)dx) { q += 1; }
          return (int)q;
        }
        const long long num = (long long)k * dy + bias;
        long long q = (long long)((double)num * inv_dx);
        const long long r = num - q * dx;
        q += (r >= dx) ? 1 : 0;
        q -= (r < 0) ? 1 : 0;
        return (int)q;
      };
      auto cell_of = [&](int k, int& cx, int& cy) {
        const int u = minor_of(k);
        const int cM = sM + gM * k, cm = sm + gm * u;
        cx = steep ? cm : cM;
        cy = steep ? cM : cm;
      };

      if (lane == 0) {                          // touched box = box of the two end cells (monotone line)
        int ax, ay, bx, by;
        cell_of((int)k_first, ax, ay);
        cell_of((int)k_last, bx, by);
        atomicMin(&sbox[0], min(ax, bx)); atomicMin(&sbox[1], min(ay, by));
        atomicMax(&sbox[2], max(ax, bx)); atomicMax(&sbox[3], max(ay, by));
      }
      // −1 on every valid cell but the last valid one, +1 there (ogm.py:184-185)
      const int kl = (int)k_last;
      for (int k = (int)k_first + lane; k <= kl; k += 64) {
        int cx, cy;
        cell_of(k, cx, cy);
        if ((unsigned)cx >= (unsigned)g.width || (unsigned)cy >= (unsigned)g.height) continue;   // cannot happen; guards the store
        const int v = (k == kl) ? 1 : -1;
        const int ux = cx - wx0, uy = cy - wy0;
        if ((unsigned)ux < (unsigned)kWin && (unsigned)uy < (unsigned)kWin) atomicAdd(&win[ux * kWin + uy], v);
        else atomicAdd(plane + (int64_t)cx * g.height + cy, v);
      }
    }
  }
  __syncthreads();
  // flush the window and the box: one global atomic per touched window cell per workgroup
  for (int i = threadIdx.x; i < kWin * kWin; i += kRayThreads) {
    const int v = win[i];
    if (v != 0) {
      const int cx = wx0 + i / kWin, cy = wy0 + i % kWin;     // inside the map: only valid cells were added
      atomicAdd(plane + (int64_t)cx * g.height + cy, v);
    }
  }
  if (threadIdx.x == 0 && sbox[2] >= sbox[0]) {
    int* bb = g.bbox + 4 * (kOgmPlanes * parity + b);
    atomicMin(bb + 0, sbox[0]); atomicMin(bb + 1, sbox[1]); atomicMax(bb + 2, sbox[2]); atomicMax(bb + 3, sbox[3]);
  }
}

// One thread per cell; planes visited in scan order.  mode 0: apply + clear, mode 1: clear only.
__global__ void __launch_bounds__(256)
ogm_apply_kernel(OgmGrid g, int nplanes, int parity, int mode) {
  const int* boxes = g.bbox + 4 * kOgmPlanes * parity;
  // union of the touched boxes (wave-uniform: scalar loads)
  int x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = -0x7fffffff, y1 = -0x7fffffff;
  for (int b = 0; b < nplanes; ++b) {
    x0 = min(x0, boxes[4 * b]); y0 = min(y0, boxes[4 * b + 1]);
    x1 = max(x1, boxes[4 * b + 2]); y1 = max(y1, boxes[4 * b + 3]);
  }
  if (x1 < x0) return;
  // one cell per thread: blockIdx.x tiles y (contiguous in memory) by 64, blockIdx.y tiles x by 4
  const int cy = y0 + blockIdx.x * 64 + (threadIdx.x & 63);
  const int cx = x0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (cy > y1 || cx > x1) return;
  const int64_t stride = (int64_t)g.width * g.height;
  const int64_t idx = (int64_t)cx * g.height + cy;
  // all plane reads first (independent loads in flight), then the order-dependent update
  int c[kOgmPlanes];
#pragma unroll
  for (int b = 0; b < kOgmPlanes; ++b) {
    c[b] = 0;
    if (b < nplanes) {
      const int* bb = boxes + 4 * b;
      if (cx >= bb[0] && cx <= bb[2] && cy >= bb[1] && cy <= bb[3]) c[b] = g.counts[b * stride + idx];
    }
  }
  float l = g.logodds[idx];
  bool dirty = false;
#pragma unroll
  for (int b = 0; b < kOgmPlanes; ++b) {
    if (c[b] != 0) {
      g.counts[b * stride + idx] = 0;
      double v = (double)l + (double)c[b] * g.step;
      v = fmin(fmax(v, -g.clip), g.clip);
      l = (float)v;
      dirty = true;
    }
  }
  if (dirty && mode == 0) g.logodds[idx] = l;
}

__global__ void ogm_binarize_kernel(OgmGrid g, uint8_t* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)g.width * g.height) return;
  const float pmf = 1.0f / (1.0f + expf(g.logodds[i]));        // ogm.py:61
  out[i] = pmf > 0.5f ? 1 : 0;                                 // ogm.py:62-64
}

__global__ void ogm_init_kernel(OgmGrid g) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 8 * kOgmPlanes) g.bbox[i] = box_init((int)i & 3);
  const int64_t cells = (int64_t)g.width * g.height;
  if (i >= cells) return;
  g.logodds[i] = 0.f;
  for (int b = 0; b < kOgmPlanes; ++b) g.counts[b * cells + i] = 0;
}

hipError_t launch_ogm_raycast(const OgmGrid& g, const OgmScan* d_scans, int nscans, int max_rays, int parity, hipStream_t stream) {
  int blocks = (max_rays + kRaysPerBlock - 1) / kRaysPerBlock;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(ogm_raycast_kernel, dim3(blocks, nscans), dim3(kRayThreads), 0, stream, g, d_scans, parity);
  return hipGetLastError();
}
hipError_t launch_ogm_apply(const OgmGrid& g, int nscans, int parity, int clear_only, hipStream_t stream) {
  hipLaunchKernelGGL(ogm_apply_kernel, dim3((g.height + 63) / 64, (g.width + 3) / 4), dim3(256), 0, stream, g, nscans, parity, clear_only);
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

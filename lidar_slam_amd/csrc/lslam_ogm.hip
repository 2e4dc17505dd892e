// lslam_ogm.hip — log-odds ray casting (modules/ogm.py:149-231) on a grid resident in HBM.
//
// The reference traces one scan at a time and clips the whole grid after every scan
// (ogm.py:172-188), so scans must take effect in order (the clip is not associative).
// What IS independent is the integer "hits − misses" grid of each scan.  Three kernels, none of which
// issues a scattered access to HBM:
//   ogm_rays_kernel   once per call, one thread per ray: the end cell from the float64 world→grid
//                     formula (ogm.py:117-118; float64 is what makes the indices bit-exact), the part
//                     [k_first, k_last] of the ray inside the map in O(1) (the reference filters cells
//                     after tracing, ogm.py:175-181; the hit goes to the LAST valid cell, ogm.py:185)
//                     → a 16-byte record per ray, the box of cells each scan touches, and the box of
//                     every 64 consecutive rays.
//   ogm_tile_kernel   per group of up to kOgmPlanes scans, one workgroup per (scan, 64×64-cell map tile
//                     meeting the scan's box).  A wave takes 64 rays at a time (skipped as a whole when
//                     their common box misses the tile), one ray per lane: exact interval of the ray on
//                     the tile's major axis, reject on the minor axis, then the lane walks its ≤ 64 cells
//                     with the reference's own error accumulator (ogm.py:211; started anywhere along the ray
//                     from the closed form minor(k) = ⌊(k·dy + dx − 1 − ⌊dx/2⌋)/dx⌋, a multiply-high) adding
//                     −1 / +1 with LDS atomics (odd tile pitch: no systematic bank conflicts).  The tile ∩
//                     box is then written to the scan's count plane with coalesced row stores, zeros
//                     included.  Planes are int16 (|count| ≤ rays of the scan), int32 above 32767 rays.
//   ogm_apply_kernel  one thread per cell walks the planes IN SCAN ORDER inside each plane's box:
//                     l ← clip(l + cnt·log 4, ±20) in exact arithmetic: a cell is (base, n) with value
//                     base + n·log 4; a scan adds its count to n, and only a clip (or a host write) moves
//                     the base.  Integer adds and two compares per plane; the one float64 rounding happens
//                     when the grid is read (ogm_values_kernel).
// A plane is only ever read inside the box its tile kernel has just written, so planes are never cleared.
// Integer counts make the result independent of the order in which rays land.
#include <hip/hip_runtime.h>

#include "lslam_device.h"
#include "lslam_kernels.h"

namespace lslam {

constexpr int kTile = 64;                 // map tile (cells per side) accumulated in LDS
constexpr int kTilePitch = kTile + 1;     // odd: the 64 cells a wave adds to along an x-major ray sit in 64 banks
constexpr int kTileThreads = 256;
static_assert(kOgmPlanes <= 128, "the apply kernel stages the boxes of a group in two registers per lane");
constexpr int kBoxEmpty = kOgmBoxEmpty;     // boxes are {min_x, min_y, −max_x, −max_y}, all reduced with atomicMin, memset(0x7f) = empty

__device__ __forceinline__ int world2grid(double v, double vmin, double res) {
  return (int)ceil((v - vmin) / res) - 1;                 // ogm.py:117-118
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}

// The cells of the ray (sx,sy) → (ex,ey) that lie inside the map, in the reference's Bresenham (ogm.py:190-231).
// Cell k = (sM + gM·k, sm + gm·u(k)) in (major, minor) axes with u(k) = ⌊(k·dy + bias)/dx⌋ non-decreasing, so
// "inside the map" is an interval [k_first, k_last] of k, found in O(1) (the reference filters after tracing,
// ogm.py:175-181; the hit goes to the LAST valid cell, ogm.py:185).
struct RayGeom {
  int dx, dy, bias;          // after the steep swap: dx ≥ dy
  int sM, sm, gM, gm;        // start cell and step signs in (major, minor) axes
  int k_first, k_last;       // valid interval (empty: k_first > k_last)
  unsigned magic;            // ⌊(2³² − 1)/dx⌋ for the multiply-high division
  bool steep, small;

  // line geometry only (integer; cheap)
  __device__ __forceinline__ void line(int sx, int sy, int ex, int ey) {
    dx = abs(ex - sx); dy = abs(ey - sy);
    steep = dy > dx;
    if (steep) { const int t = dx; dx = dy; dy = t; }
    const int sgx = (sx <= ex) ? 1 : -1, sgy = (sy <= ey) ? 1 : -1;
    bias = dx - 1 - dx / 2;
    sM = steep ? sy : sx; sm = steep ? sx : sy;
    gM = steep ? sgy : sgx; gm = steep ? sgx : sgy;
    small = dx < 46340;                       // k·dy + bias < 2³¹: 32-bit multiply-high division
    // ⌊(2³² − 1)/dx⌋: one less than ⌊2³²/dx⌋ when dx is a power of two; the quotient estimate below is then
    // still at most one short (num < 2³¹), which the fix-up absorbs — and this is a 32-bit division
    magic = dx > 1 ? 0xffffffffu / (unsigned)dx : 0xffffffffu;
  }
  // + the interval of k inside the map; false if it is empty
  __device__ __forceinline__ bool clip_to_map(const OgmGrid& g) {
    const int WM = steep ? g.height : g.width, Wm = steep ? g.width : g.height;
    long long kf = 0, kl = dx;
    kf = max(kf, (long long)(gM > 0 ? -sM : sM - (WM - 1)));
    kl = min(kl, (long long)(gM > 0 ? WM - 1 - sM : sM));
    const long long ulo = gm > 0 ? -(long long)sm : (long long)sm - (Wm - 1);
    const long long uhi = gm > 0 ? (long long)Wm - 1 - sm : (long long)sm;
    if (dy == 0) {
      if (ulo > 0 || uhi < 0) kl = -1;
    } else {
      if (ulo > 0) kf = max(kf, (ulo * dx - bias + dy - 1) / dy);          // u(k) ≥ ulo
      if (uhi < 0) kl = -1;
      else kl = min(kl, ((uhi + 1) * dx - bias - 1) / dy);                  // u(k) ≤ uhi
    }
    if (kf > kl) { k_first = 0; k_last = -1; return false; }     // no valid cell: the ray is skipped (ogm.py:182-183)
    k_first = (int)kf; k_last = (int)kl;
    return true;
  }
  // minor(k) = ⌊(k·dy + dx − 1 − ⌊dx/2⌋)/dx⌋: the closed form of the reference's error sequence (ogm.py:211)
  __device__ __forceinline__ int minor_of(int k) const {
    if (dy == 0) return 0;
    if (small) {
      const unsigned num = (unsigned)k * (unsigned)dy + (unsigned)bias;
      unsigned q = __umulhi(num, magic);
      unsigned r = num - q * (unsigned)dx;
      if (r >= (unsigned)dx) { q += 1; r -= (unsigned)dx; }
      if (r >= (unsigned)dx) { q += 1; }
      return (int)q;
    }
    const long long num = (long long)k * dy + bias;       // rays whose end lies ≥ 46340 cells away: rare, slow path
    long long q = (long long)((double)num * (1.0 / (double)dx));
    const long long r = num - q * dx;
    q += (r >= dx) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return (int)q;
  }
  __device__ __forceinline__ void cell_of(int k, int& cx, int& cy) const {
    const int u = minor_of(k);
    const int cM = sM + gM * k, cm = sm + gm * u;
    cx = steep ? cm : cM;
    cy = steep ? cM : cm;
  }
};


// ---- per ray: end cell, valid interval; per scan: touched box ------------------------------------------
__global__ void __launch_bounds__(256)
ogm_rays_kernel(OgmGrid g, const OgmScan* __restrict__ scans, int4* __restrict__ rays, int* __restrict__ boxes, int4* __restrict__ batch_boxes) {
  const int b = blockIdx.y;
  const OgmScan sc = scans[b];
  const int i = blockIdx.x * 256 + threadIdx.x;
  // ray origin: (x, y) + (0.13323, 0), NOT rotated (ogm.py:165-166)
  const int sx = world2grid(sc.x + 0.13323, g.min_x, g.res);
  const int sy = world2grid(sc.y + 0.0, g.min_y, g.res);
  int bx0 = kBoxEmpty, by0 = kBoxEmpty, nx1 = kBoxEmpty, ny1 = kBoxEmpty;
  if (i < sc.n) {
    // beam end: rotate about the robot origin, translate (ogm.py:160-162)
    const double zx = sc.scan_xy[2 * (int64_t)i], zy = sc.scan_xy[2 * (int64_t)i + 1];
    const double wx = zx * sc.c + zy * (-sc.s) + sc.x;
    const double wy = zx * sc.s + zy * sc.c + sc.y;
    const int ex = world2grid(wx, g.min_x, g.res);
    const int ey = world2grid(wy, g.min_y, g.res);
    RayGeom r;
    r.line(sx, sy, ex, ey);
    if (r.clip_to_map(g)) {                                 // touched box = box of the two end cells (monotone line)
      int ax, ay, cx, cy;
      r.cell_of(r.k_first, ax, ay);
      r.cell_of(r.k_last, cx, cy);
      bx0 = min(ax, cx); by0 = min(ay, cy); nx1 = -max(ax, cx); ny1 = -max(ay, cy);
    }
    rays[sc.ray_base + i] = make_int4(ex, ey, r.k_first, r.k_last);
  }
  bx0 = wave_min_i(bx0); by0 = wave_min_i(by0); nx1 = wave_min_i(nx1); ny1 = wave_min_i(ny1);
  if ((threadIdx.x & 63) == 0 && (int)(blockIdx.x * 256 + threadIdx.x) < sc.n) {
    batch_boxes[sc.batch_base + (blockIdx.x * 256 + threadIdx.x) / 64] = make_int4(bx0, by0, nx1, ny1);   // of these 64 rays
    if (bx0 != kBoxEmpty) {
      int* bb = boxes + 4 * (int64_t)b;
      atomicMin(bb + 0, bx0); atomicMin(bb + 1, by0); atomicMin(bb + 2, nx1); atomicMin(bb + 3, ny1);
    }
  }
}

// ---- per (scan, map tile): hit − miss counts of the tile in LDS, then one coalesced store ------------------
template <typename CT>
__global__ void __launch_bounds__(kTileThreads)
ogm_tile_kernel(OgmGrid g, const OgmScan* __restrict__ scans, const int4* __restrict__ rays, const int* __restrict__ boxes,
                const int4* __restrict__ batch_boxes) {
  __shared__ int tile[kTile * kTilePitch];
  const int b = blockIdx.z;
  const int* bb = boxes + 4 * (int64_t)b;
  const int box_x0 = bb[0], box_y0 = bb[1], box_x1 = -bb[2], box_y1 = -bb[3];
  if (box_x0 == kBoxEmpty) return;
  // tiles are those of the map grid, counted from the first one the scan's box meets (x is the slow axis of the arrays)
  const int tx0 = (box_x0 / kTile + blockIdx.y) * kTile, ty0 = (box_y0 / kTile + blockIdx.x) * kTile;
  const int tx1 = tx0 + kTile - 1, ty1 = ty0 + kTile - 1;
  if (tx0 > box_x1 || tx1 < box_x0 || ty0 > box_y1 || ty1 < box_y0) return;
  const OgmScan sc = scans[b];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < kTile * kTilePitch; i += kTileThreads) tile[i] = 0;
  __syncthreads();
  const int sx = world2grid(sc.x + 0.13323, g.min_x, g.res);
  const int sy = world2grid(sc.y + 0.0, g.min_y, g.res);

  for (int base = wave * 64; base < sc.n; base += kTileThreads) {
    // 64 rays at a time, one per lane — unless the box of the 64 misses the tile (rays come in angular order)
    const int4 q = batch_boxes[sc.batch_base + base / 64];
    if (q.x == kBoxEmpty || tx0 > -q.z || tx1 < q.x || ty0 > -q.w || ty1 < q.y) continue;
    const int i = base + lane;
    if (i >= sc.n) continue;
    const int4 rec = rays[sc.ray_base + i];
    if (rec.z > rec.w) continue;                            // no valid cell: the ray is skipped (ogm.py:182-183)
    RayGeom r;
    r.line(sx, sy, rec.x, rec.y);
    // the ray's cells inside this tile: exact interval on the major axis (unit steps), cut to the valid part
    const int tM0 = r.steep ? ty0 : tx0, tM1 = r.steep ? ty1 : tx1;
    const int tm0 = r.steep ? tx0 : ty0;
    const int ka = max(rec.z, r.gM > 0 ? tM0 - r.sM : r.sM - tM1);
    const int kb = min(rec.w, r.gM > 0 ? tM1 - r.sM : r.sM - tM0);
    if (ka > kb) continue;
    // minor axis: monotone along the ray, so its two ends bound it
    const int ua = r.minor_of(ka), ub = r.minor_of(kb);
    const int ma = r.sm + r.gm * ua - tm0, mb = r.sm + r.gm * ub - tm0;
    if (max(ma, mb) < 0 || min(ma, mb) >= kTile) continue;
    // walk the cells with the reference's own error accumulator, started at k = ka:
    // rem = (k·dy + bias) mod dx, the minor coordinate steps when rem wraps
    unsigned rem = (unsigned)((long long)ka * r.dy + r.bias - (long long)ua * r.dx);
    const unsigned udx = (unsigned)r.dx, udy = (unsigned)r.dy;
    int um = ma;                                            // tile-local minor coordinate
    const int uM = r.sM + r.gM * ka - tM0;                  // tile-local major coordinate, stays inside [0, kTile)
    int addr = r.steep ? um * kTilePitch + uM : uM * kTilePitch + um;
    const int stepM = r.gM * (r.steep ? 1 : kTilePitch), stepm = r.gm * (r.steep ? kTilePitch : 1);
    const int klast = rec.w;
    for (int k = ka; k <= kb; ++k) {
      // −1 on every valid cell but the last valid one, +1 there (ogm.py:184-185)
      if ((unsigned)um < (unsigned)kTile) atomicAdd(&tile[addr], (k == klast) ? 1 : -1);
      rem += udy;
      if (rem >= udx) { rem -= udx; um += r.gm; addr += stepm; }
      addr += stepM;
    }
  }
  __syncthreads();
  // tile ∩ box → plane, rows of up to 64 consecutive cells (256 B), zeros included: the apply kernel reads
  // exactly the box, and nothing else of this plane is ever read
  CT* __restrict__ plane = reinterpret_cast<CT*>(g.counts) + (int64_t)b * g.width * g.height;
  const int fx0 = max(tx0, box_x0), fx1 = min(tx1, box_x1);
  const int cy = ty0 + lane;
  if (cy >= box_y0 && cy <= box_y1)
    for (int cx = fx0 + wave; cx <= fx1; cx += kTileThreads / 64)
      plane[(int64_t)cx * g.height + cy] = (CT)tile[(cx - tx0) * kTilePitch + lane];
}

__device__ __forceinline__ int clamp_int(double x) { return (int)fmin(fmax(x, -2.0e9), 2.0e9); }

// One cell of the grid while a group of scans is applied to it.
struct CellState {
  float base;
  int n, hi, lo;             // clipped high after a scan ⇔ n > hi, low ⇔ n < lo
  bool dirty, moved;
  __device__ __forceinline__ CellState(float base_, int n_, double step, double clip)
      : base(base_), n(n_), hi(clamp_int(floor((clip - (double)base_) / step))), lo(clamp_int(ceil((-clip - (double)base_) / step))),
        dirty(false), moved(false) {}
  __device__ __forceinline__ void add(int c, float fclip, int hi_top, int lo_top, int hi_bottom, int lo_bottom) {
    if (c == 0) return;
    n += c;
    dirty = true;
    if (n > hi) { base = fclip; n = 0; hi = hi_top; lo = lo_top; moved = true; }
    else if (n < lo) { base = -fclip; n = 0; hi = hi_bottom; lo = lo_bottom; moved = true; }
  }
};

// Two cells (64 apart in y) per thread; planes visited in scan order, each only inside its scan's box.
// A wave covers 128 consecutive y at one x, so "does this plane's box meet my cells" is decided per wave
// on scalars (boxes are staged in two registers per lane and broadcast with readlane); lanes outside the
// box in y drop the value they loaded.  All loads of a chunk of planes are issued before the first use.
// (x0, y0, x1, y1) is the union of the group's boxes, computed by the host, which sizes the grid to it.
template <typename CT>
__global__ void __launch_bounds__(256)
ogm_apply_kernel(OgmGrid g, const int* __restrict__ boxes, int nplanes, int x0, int y0, int x1, int y1) {
  const int lane = threadIdx.x & 63;
  const CT* __restrict__ counts = reinterpret_cast<const CT*>(g.counts);
  const int4* boxes4 = reinterpret_cast<const int4*>(boxes);
  const int4 none = make_int4(kBoxEmpty, kBoxEmpty, kBoxEmpty, kBoxEmpty);
  const int4 boxA = lane < nplanes ? boxes4[lane] : none;
  const int4 boxB = 64 + lane < nplanes ? boxes4[64 + lane] : none;
  const int wy0 = y0 + blockIdx.x * 128;
  const int cx = __builtin_amdgcn_readfirstlane(x0 + blockIdx.y * 4 + (threadIdx.x >> 6));
  if (wy0 > y1 || cx > x1) return;                          // whole wave
  const int wy1 = min(wy0 + 127, y1);
  const bool live0 = wy0 + lane <= y1, live1 = wy0 + 64 + lane <= y1;
  const int cy0 = min(wy0 + lane, y1), cy1 = min(wy0 + 64 + lane, y1);   // idle lanes shadow the last live one (in-bounds loads, no store)
  const int64_t stride = (int64_t)g.width * g.height;
  const int64_t idx0 = (int64_t)cx * g.height + cy0, idx1 = (int64_t)cx * g.height + cy1;
  constexpr int kChunk = 16;
  // value = base + n·step.  "Clipped after this scan" is a comparison of n with two integers that depend on
  // the base only: base + n·step > clip ⇔ n > ⌊(clip − base)/step⌋, and symmetrically below.
  const double step = g.step, clip = g.clip;
  CellState a(g.logodds[idx0], g.steps[idx0], step, clip), b(g.logodds[idx1], g.steps[idx1], step, clip);
  const int hi_at_top = 0, lo_at_top = clamp_int(ceil(-2.0 * clip / step));            // thresholds once the base is +clip
  const int hi_at_bottom = clamp_int(floor(2.0 * clip / step)), lo_at_bottom = 0;     // … or −clip
  const float fclip = (float)clip;
  for (int b0 = 0; b0 < nplanes; b0 += kChunk) {
    const bool second = b0 >= 64;
    const int4 bx = second ? boxB : boxA;
    // pass 1: nothing but the loads (no use of a loaded value, so nothing waits) …
    int r0[kChunk], r1[kChunk];                             // 32-bit: two int16 packed into a register would be a use of the load
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
      const int src = (b0 & 63) + j;
      const int px0 = __builtin_amdgcn_readlane(bx.x, src), py0 = __builtin_amdgcn_readlane(bx.y, src);
      const int px1 = -__builtin_amdgcn_readlane(bx.z, src), py1 = -__builtin_amdgcn_readlane(bx.w, src);
      r0[j] = 0; r1[j] = 0;
      if (cx >= px0 && cx <= px1 && wy0 <= py1 && wy1 >= py0) {      // scalar; an empty box fails cx >= px0
        const CT* pl = counts + (int64_t)(b0 + j) * stride;
        r0[j] = pl[idx0];
        r1[j] = pl[idx1];
      }
    }
    // … pass 2: in scan order, drop what lies outside the plane's box in y, add
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
      const int src = (b0 & 63) + j;
      const int py0 = __builtin_amdgcn_readlane(bx.y, src), py1 = -__builtin_amdgcn_readlane(bx.w, src);
      const int c0 = (cy0 >= py0 && cy0 <= py1) ? (int)r0[j] : 0;
      const int c1 = (cy1 >= py0 && cy1 <= py1) ? (int)r1[j] : 0;
      a.add(c0, fclip, hi_at_top, lo_at_top, hi_at_bottom, lo_at_bottom);
      b.add(c1, fclip, hi_at_top, lo_at_top, hi_at_bottom, lo_at_bottom);
    }
  }
  if (a.dirty && live0) { g.steps[idx0] = a.n; if (a.moved) g.logodds[idx0] = a.base; }
  if (b.dirty && live1) { g.steps[idx1] = b.n; if (b.moved) g.logodds[idx1] = b.base; }
}

__device__ __forceinline__ float cell_value(const OgmGrid& g, int64_t i) {
  return (float)((double)g.logodds[i] + (double)g.steps[i] * g.step);
}

// out = f32(base + n·step); with out == g.logodds the grid is folded in place (n ← 0), which a change of the
// step size needs
__global__ void ogm_values_kernel(OgmGrid g, float* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)g.width * g.height) return;
  const float v = cell_value(g, i);
  out[i] = v;
  if (out == g.logodds) g.steps[i] = 0;
}

__global__ void ogm_binarize_kernel(OgmGrid g, uint8_t* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)g.width * g.height) return;
  const float pmf = 1.0f / (1.0f + expf(cell_value(g, i)));    // ogm.py:61
  out[i] = pmf > 0.5f ? 1 : 0;                                 // ogm.py:62-64
}

__global__ void ogm_init_kernel(OgmGrid g) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (int64_t)g.width * g.height) { g.logodds[i] = 0.f; g.steps[i] = 0; }
}

hipError_t launch_ogm_rays(const OgmGrid& g, const OgmScan* d_scans, int nscans, int max_rays, int4* d_rays, int* d_boxes, int4* d_batch_boxes,
                           hipStream_t stream) {
  if (nscans <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(d_boxes, 0x7f, sizeof(int) * 4 * (size_t)nscans, stream);
  if (e != hipSuccess) return e;
  int blocks = (max_rays + 255) / 256;
  if (blocks < 1) blocks = 1;
  for (int k0 = 0; k0 < nscans; k0 += 32768) {              // grid.y limit
    const int cnt = nscans - k0 < 32768 ? nscans - k0 : 32768;
    hipLaunchKernelGGL(ogm_rays_kernel, dim3(blocks, cnt), dim3(256), 0, stream, g, d_scans + k0, d_rays, d_boxes + 4 * (size_t)k0, d_batch_boxes);
  }
  return hipGetLastError();
}
// scans [0, nscans) of the group go to planes 0 … nscans−1 (nscans ≤ kOgmPlanes).  The count of a cell is at most
// the number of rays of the scan in magnitude: int16 planes unless a scan has more than 32767 rays.
hipError_t launch_ogm_tiles(const OgmGrid& g, const OgmScan* d_scans, int nscans, const int4* d_rays, const int* d_boxes,
                            const int4* d_batch_boxes, bool wide, int tiles_x, int tiles_y, hipStream_t stream) {
  if (nscans <= 0 || tiles_x <= 0 || tiles_y <= 0) return hipSuccess;
  const dim3 grid(tiles_y, tiles_x, nscans);
  if (wide) hipLaunchKernelGGL(ogm_tile_kernel<int32_t>, grid, dim3(kTileThreads), 0, stream, g, d_scans, d_rays, d_boxes, d_batch_boxes);
  else hipLaunchKernelGGL(ogm_tile_kernel<int16_t>, grid, dim3(kTileThreads), 0, stream, g, d_scans, d_rays, d_boxes, d_batch_boxes);
  return hipGetLastError();
}
hipError_t launch_ogm_apply(const OgmGrid& g, const int* d_boxes, int nscans, bool wide, int x0, int y0, int x1, int y1, hipStream_t stream) {
  if (nscans <= 0 || x1 < x0 || y1 < y0) return hipSuccess;
  const dim3 grid((y1 - y0) / 128 + 1, (x1 - x0) / 4 + 1);
  if (wide) hipLaunchKernelGGL(ogm_apply_kernel<int32_t>, grid, dim3(256), 0, stream, g, d_boxes, nscans, x0, y0, x1, y1);
  else hipLaunchKernelGGL(ogm_apply_kernel<int16_t>, grid, dim3(256), 0, stream, g, d_boxes, nscans, x0, y0, x1, y1);
  return hipGetLastError();
}
// how many map tiles a box spans per axis (the tile kernel counts tiles from the first one the box meets)
void ogm_tile_span(const int* box, int* tiles_x, int* tiles_y) {
  if (box[0] == kBoxEmpty) { *tiles_x = *tiles_y = 0; return; }
  *tiles_x = (-box[2]) / kTile - box[0] / kTile + 1;
  *tiles_y = (-box[3]) / kTile - box[1] / kTile + 1;
}
hipError_t launch_ogm_binarize(const OgmGrid& g, uint8_t* out, hipStream_t stream) {
  const int64_t n = (int64_t)g.width * g.height;
  hipLaunchKernelGGL(ogm_binarize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, out);
  return hipGetLastError();
}
hipError_t launch_ogm_values(const OgmGrid& g, float* out, hipStream_t stream) {
  const int64_t n = (int64_t)g.width * g.height;
  hipLaunchKernelGGL(ogm_values_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, out);
  return hipGetLastError();
}
hipError_t launch_ogm_init(const OgmGrid& g, hipStream_t stream) {
  const int64_t n = (int64_t)g.width * g.height;
  hipLaunchKernelGGL(ogm_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g);
  return hipGetLastError();
}

}  // namespace lslam

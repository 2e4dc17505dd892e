// lslam_ops.hip — the small float64 operators of modules/icp.py on paired rows:
//   get_transform (icp.py:43-74)  → Kabsch from moments
//   get_error     (icp.py:76-98)  → Σ‖p−q‖², optionally bounding-box normalised
// One 1024-thread workgroup strides over the rows (these calls see at most a few 1e5 rows and are
// O(N)); wave64 shuffle reduction, LDS across waves, lane 0 solves.
#include <hip/hip_runtime.h>

#include "lslam_device.h"
#include "lslam_kernels.h"

namespace lslam {

constexpr int kPairOpThreads = 1024;

__global__ void __launch_bounds__(kPairOpThreads)
pair_op_kernel(PairOpArgs a) {
  constexpr int kAcc = AccLayout<3>::kTotal;
  __shared__ double sh[kPairOpThreads / 64][kAcc];
  PairAcc<3> acc;
  acc.clear();
  const bool box = a.normalize != 0;
  for (int64_t i = threadIdx.x; i < a.n; i += kPairOpThreads) {
    double p[3] = {0.0, 0.0, 0.0}, q[3] = {0.0, 0.0, 0.0};
    for (int c = 0; c < a.cols && c < 3; ++c) {
      p[c] = a.p[i * a.p_stride + c];
      q[c] = a.q[i * a.q_stride + c];
    }
    acc.add(p, q, box);
  }
  acc.wave_reduce(box);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) acc.store(sh[wave]);
  __syncthreads();
  if (threadIdx.x != 0) return;
  Moments mo;
  combine_partials<3>(&sh[0][0], kPairOpThreads / 64, kAcc, box, mo);
  if (a.out_T) {
    double T[16];
    step_from_moments(mo, /*planar=*/false, T);
    for (int i = 0; i < 16; ++i) a.out_T[i] = T[i];
  }
  if (a.out_err) {
    // get_error sums over ALL columns handed in; columns beyond 3 do not occur in the reference
    a.out_err[0] = error_from_moments(mo, box);
  }
}

// ---- device probe: sustained float32 VALU issue rate --------------------------------------------
// 64 independent-enough v_fma_f32 per loop trip on every SIMD of the chip.  The ICP search kernels
// are bound by VALU issue slots at whatever clock the chip sustains under an all-VALU load (≈1.9 GHz,
// it varies by a few percent from device to device); bench.py prices them against this measurement
// as well as against the 2.4 GHz datasheet peak.
__global__ void __launch_bounds__(256) valu_probe_kernel(float* out, int iters, float seed) {
  // 64 independent v_fma_f32 per trip, written in assembly: in C++ the compiler either packs them (v_pk_fma_f32 issues
  // at half the instruction rate) or chains them through one register (measured 4 cycles per instruction instead of 2)
  // ONE asm statement holds initialisation, loop and reduction: the physical registers it names (v8, v20–v27) carry state
  // from instruction to instruction, which separate asm statements cannot promise (clobber lists do not keep a register
  // live between statements); the loop counter is a scalar operand the statement owns ("+s").
  float r = seed + threadIdx.x, s;
  int n = iters;
#define LSLAM_FMA8 "v_fma_f32 v20, v8, 1.0, 0.5\n v_fma_f32 v21, v8, 1.0, 0.5\n v_fma_f32 v22, v8, 1.0, 0.5\n v_fma_f32 v23, v8, 1.0, 0.5\n" \
                   "v_fma_f32 v24, v8, 1.0, 0.5\n v_fma_f32 v25, v8, 1.0, 0.5\n v_fma_f32 v26, v8, 1.0, 0.5\n v_fma_f32 v27, v8, 1.0, 0.5\n"
  asm volatile("v_mov_b32 v8, %2\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"
               "v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n"
               "s_cmp_lt_i32 %1, 1\n s_cbranch_scc1 2f\n"
               "1:\n"
               LSLAM_FMA8 LSLAM_FMA8 LSLAM_FMA8 LSLAM_FMA8 LSLAM_FMA8 LSLAM_FMA8 LSLAM_FMA8 LSLAM_FMA8
               "s_sub_i32 %1, %1, 1\n s_cmp_gt_i32 %1, 0\n s_cbranch_scc1 1b\n"
               "2:\n"
               "v_add_f32 %0, v20, v21\n v_add_f32 %0, %0, v22\n v_add_f32 %0, %0, v23\n v_add_f32 %0, %0, v24\n"
               "v_add_f32 %0, %0, v25\n v_add_f32 %0, %0, v26\n v_add_f32 %0, %0, v27\n"
               : "=&v"(s), "+s"(n)
               : "v"(r)
               : "v8", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "scc");
#undef LSLAM_FMA8
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

hipError_t run_valu_probe(float* d_scratch /*2048*256 floats*/, hipStream_t stream, double* lane_ops_per_s) {
  const int blocks = 2048, threads = 256, iters = 12000;
  hipEvent_t e0, e1;
  hipError_t e;
  if ((e = hipEventCreate(&e0)) != hipSuccess) return e;
  if ((e = hipEventCreate(&e1)) != hipSuccess) return e;
  hipLaunchKernelGGL(valu_probe_kernel, dim3(blocks), dim3(threads), 0, stream, d_scratch, 2000, 1.0f);   // ramp the clocks
  (void)hipEventRecord(e0, stream);
  hipLaunchKernelGGL(valu_probe_kernel, dim3(blocks), dim3(threads), 0, stream, d_scratch, iters, 1.0f);
  (void)hipEventRecord(e1, stream);
  e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (e != hipSuccess) return e;
  *lane_ops_per_s = (double)blocks * threads * (double)iters * 64.0 / (ms * 1e-3);
  return hipSuccess;
}

hipError_t launch_pair_op(const PairOpArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(pair_op_kernel, dim3(1), dim3(kPairOpThreads), 0, stream, a);
  return hipGetLastError();
}

}  // namespace lslam

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

hipError_t launch_pair_op(const PairOpArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(pair_op_kernel, dim3(1), dim3(kPairOpThreads), 0, stream, a);
  return hipGetLastError();
}

}  // namespace lslam

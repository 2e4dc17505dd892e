// Checks the DPP wave reductions of lslam_prune_common.h against the ds_bpermute butterfly on the GPU (gfx950):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I lidar_slam_amd/csrc -I include tools/dpp_probe.hip -o tools/dpp_probe && tools/dpp_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include "lslam_prune_common.h"
using namespace lslam;
__global__ void probe(const float* f, const double* d, float* fo, double* dout) {
  const int lane = threadIdx.x;
  const float v = f[lane];
  fo[lane] = wave_max_dpp(v);
  fo[64 + lane] = wave_min_dpp(v);
  const double s = wave_sum_dpp_last(d[lane]);
  dout[lane] = s;
  // reference tree: pairwise in the order xor 1, 2, 4, 8, 16, 32
  double r = d[lane];
  for (int o = 1; o < 64; o <<= 1) r += __shfl_xor(r, o, 64);
  dout[64 + lane] = r;
}
int main() {
  float hf[64]; double hd[64];
  srand(3);
  for (int i = 0; i < 64; ++i) { hf[i] = (float)rand() / RAND_MAX * 100.f - 50.f; hd[i] = ((double)rand() / RAND_MAX - 0.5) * 1e6 + 1e-7 * rand(); }
  float *df, *dfo; double *dd, *ddo;
  hipMalloc(&df, 256); hipMalloc(&dfo, 512); hipMalloc(&dd, 512); hipMalloc(&ddo, 1024);
  hipMemcpy(df, hf, 256, hipMemcpyHostToDevice); hipMemcpy(dd, hd, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, df, dd, dfo, ddo);
  float ofo[128]; double odo[128];
  hipMemcpy(ofo, dfo, 512, hipMemcpyDeviceToHost); hipMemcpy(odo, ddo, 1024, hipMemcpyDeviceToHost);
  float mx = -INFINITY, mn = INFINITY;
  for (int i = 0; i < 64; ++i) { mx = fmaxf(mx, hf[i]); mn = fminf(mn, hf[i]); }
  int bad = 0;
  for (int i = 0; i < 64; ++i) { if (ofo[i] != mx) ++bad; if (ofo[64 + i] != mn) ++bad; }
  printf("max/min over the wave: %d lanes wrong (expect 0)\n", bad);
  printf("sum in lane 63: %.17g   butterfly 1..32: %.17g   %s\n", odo[63], odo[64 + 63], odo[63] == odo[64 + 63] ? "bit-identical" : "DIFFERENT");
  return bad != 0 || odo[63] != odo[127];
}

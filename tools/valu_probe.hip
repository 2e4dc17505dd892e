// valu_probe.hip — measures the float32 VALU issue rates the ICP kernels are priced against.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float float2_t __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) probe(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 1.0001f, c = 0.5f;
  float2_t p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const float2_t pb = {b, b}, pc = {c, c};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {        // 8 independent v_fma_f32
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
        a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
      }
    } else if (MODE == 1) { // 4 independent v_pk_fma_f32 (8 fma)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        p0 = __builtin_elementwise_fma(p0, pb, pc); p1 = __builtin_elementwise_fma(p1, pb, pc);
        p2 = __builtin_elementwise_fma(p2, pb, pc); p3 = __builtin_elementwise_fma(p3, pb, pc);
      }
    } else if (MODE == 2) { // sub, mul mix: v_sub + v_mul + v_min3 like the ICP loop
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float d0 = a0 - b, d1 = a1 - c; float m0 = d0 * d0; m0 = __builtin_fmaf(d1, d1, m0);
        float d2 = a2 - b, d3 = a3 - c; float m1 = d2 * d2; m1 = __builtin_fmaf(d3, d3, m1);
        a4 = fminf(fminf(m0, m1), a4);
        a0 += 1.0f; a2 += 1.0f;
      }
    }
  }
  if (MODE == 1) { a0 = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
static int run(const char* name, double ops_per_iter_per_lane, double flops_per_iter_per_lane) {
  const int blocks = 256 * 8, threads = 256, iters = 20000;
  float* out;
  CHECK(hipMalloc(&out, sizeof(float) * blocks * threads));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 100, 1.0f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double lanes = (double)blocks * threads;
  printf("%-28s %8.3f ms  %7.2f T lane-ops/s  %7.2f TFLOP/s\n", name, ms,
         lanes * iters * ops_per_iter_per_lane / (ms * 1e-3) / 1e12, lanes * iters * flops_per_iter_per_lane / (ms * 1e-3) / 1e12);
  CHECK(hipFree(out));
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  printf("%s  CUs=%d  clock=%d MHz  LDS/block=%zu\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000, p.sharedMemPerBlock);
  if (run<0>("v_fma_f32 x64/iter", 64, 128)) return 1;
  if (run<1>("v_pk_fma_f32 x32/iter", 32, 128)) return 1;
  if (run<2>("icp mix (11 ops x8)/iter", 88, 8 * 2 * 5)) return 1;
  return 0;
}

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
    if (MODE >= 3) {
      // integer / compare+select / float64 / conversion rates (what the occupancy-grid kernels are made of)
      int i0 = __float_as_int(a0), i1 = __float_as_int(a1), i2 = __float_as_int(a2), i3 = __float_as_int(a3);
      int i4 = __float_as_int(a4), i5 = __float_as_int(a5), i6 = __float_as_int(a6), i7 = __float_as_int(a7);
      double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (MODE == 3) {        // v_add_u32
          i0 += i1; i1 += i2; i2 += i3; i3 += i4; i4 += i5; i5 += i6; i6 += i7; i7 += i0;
        } else if (MODE == 4) { // v_cmp + v_cndmask pairs
          i0 = i0 > i1 ? i2 : i0; i1 = i1 > i2 ? i3 : i1; i2 = i2 > i3 ? i4 : i2; i3 = i3 > i4 ? i5 : i3;
          i4 = i4 > i5 ? i6 : i4; i5 = i5 > i6 ? i7 : i5; i6 = i6 > i7 ? i0 : i6; i7 = i7 > i0 ? i1 : i7;
        } else if (MODE == 5) { // v_mul_lo_u32
          i0 *= i1; i1 *= i2; i2 *= i3; i3 *= i4; i4 *= i5; i5 *= i6; i6 *= i7; i7 *= i0;
        } else if (MODE == 6) { // v_fma_f64
          d0 = __builtin_fma(d0, 1.0000001, 0.5); d1 = __builtin_fma(d1, 1.0000001, 0.5); d2 = __builtin_fma(d2, 1.0000001, 0.5); d3 = __builtin_fma(d3, 1.0000001, 0.5);
          d0 = __builtin_fma(d0, 1.0000001, 0.5); d1 = __builtin_fma(d1, 1.0000001, 0.5); d2 = __builtin_fma(d2, 1.0000001, 0.5); d3 = __builtin_fma(d3, 1.0000001, 0.5);
        } else if (MODE == 7) { // v_cvt_f64_f32 + v_cvt_f32_f64 round trips
          a0 = (float)((double)a0); a1 = (float)((double)a1); a2 = (float)((double)a2); a3 = (float)((double)a3);
          asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (MODE == 8) { // v_min_f64 / v_max_f64
          d0 = fmin(fmax(d0, d1), d2); d1 = fmin(fmax(d1, d2), d3); d2 = fmin(fmax(d2, d3), d0); d3 = fmin(fmax(d3, d0), d1);
        } else if (MODE == 9) { // v_add_f32 (is plain add at the FMA rate?)
          a0 += b; a1 += b; a2 += b; a3 += b; a4 += b; a5 += b; a6 += b; a7 += b;
        }
      }
      a0 = __int_as_float(i0 ^ __float_as_int(a0)); a1 = __int_as_float(i1 ^ __float_as_int(a1)); a2 = __int_as_float(i2 ^ __float_as_int(a2));
      a3 = __int_as_float(i3 ^ __float_as_int(a3)); a4 = __int_as_float(i4); a5 = __int_as_float(i5); a6 = __int_as_float(i6); a7 = __int_as_float(i7);
      a0 += (float)(d0 + d1 + d2 + d3);
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
  if (run<9>("v_add_f32 x64/iter", 64, 64)) return 1;
  if (run<3>("v_add_u32 x64/iter", 64, 0)) return 1;
  if (run<4>("v_cmp+v_cndmask x64 pairs/iter", 128, 0)) return 1;
  if (run<5>("v_mul_lo_u32 x64/iter", 64, 0)) return 1;
  // (modes 6-8, float64 FMA / conversions / min-max, are kept for experiments but not reported: the compiler folds
  //  parts of those loops and the figures they print are not instruction rates)
  return 0;
}

// valu_probe2.hip — does the VGPR bank of the source operands change the issue rate of 3-source VALU ops on gfx950?
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_probe2.hip -o tools/valu_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(x) x x x x x x x x
// MODE 0: v_fma dst, a, b, c  sources in three different banks (regs mod 4 = 0,1,2), 8 independent dsts
// MODE 1: sources all in the same bank (mod 4 = 0)
// MODE 2: v_fmac (2 src + dst) different banks
// MODE 3: v_min3 three banks
// MODE 4: v_min3 same bank
// MODE 5: v_fma with two sources in the same bank
// MODE 6: v_sub (2 src, different banks)
// MODE 7: v_sub (2 src, same bank)
template <int MODE>
__global__ void __launch_bounds__(256) probe(float* out, int iters) {
  float r = threadIdx.x;
  asm volatile(
      "v_mov_b32 v8, %0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 0.5\n v_mov_b32 v12, %0\n v_mov_b32 v16, 0.5\n v_mov_b32 v13, 2.0\n"
      "v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n"
      :: "v"(r) : "v8", "v9", "v10", "v12", "v13", "v16", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) asm volatile(REP8("v_fma_f32 v20, v8, v9, v10\n v_fma_f32 v21, v8, v9, v10\n v_fma_f32 v22, v8, v9, v10\n v_fma_f32 v23, v8, v9, v10\n v_fma_f32 v24, v8, v9, v10\n v_fma_f32 v25, v8, v9, v10\n v_fma_f32 v26, v8, v9, v10\n v_fma_f32 v27, v8, v9, v10\n") ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    if (MODE == 1) asm volatile(REP8("v_fma_f32 v20, v8, v12, v16\n v_fma_f32 v21, v8, v12, v16\n v_fma_f32 v22, v8, v12, v16\n v_fma_f32 v23, v8, v12, v16\n v_fma_f32 v24, v8, v12, v16\n v_fma_f32 v25, v8, v12, v16\n v_fma_f32 v26, v8, v12, v16\n v_fma_f32 v27, v8, v12, v16\n") ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    if (MODE == 2) asm volatile(REP8("v_fmac_f32 v20, v8, v9\n v_fmac_f32 v21, v8, v9\n v_fmac_f32 v22, v8, v9\n v_fmac_f32 v23, v8, v9\n v_fmac_f32 v24, v8, v9\n v_fmac_f32 v25, v8, v9\n v_fmac_f32 v26, v8, v9\n v_fmac_f32 v27, v8, v9\n") ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    if (MODE == 3) asm volatile(REP8("v_min3_f32 v20, v8, v9, v10\n v_min3_f32 v21, v8, v9, v10\n v_min3_f32 v22, v8, v9, v10\n v_min3_f32 v23, v8, v9, v10\n v_min3_f32 v24, v8, v9, v10\n v_min3_f32 v25, v8, v9, v10\n v_min3_f32 v26, v8, v9, v10\n v_min3_f32 v27, v8, v9, v10\n") ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    if (MODE == 4) asm volatile(REP8("v_min3_f32 v20, v8, v12, v16\n v_min3_f32 v21, v8, v12, v16\n v_min3_f32 v22, v8, v12, v16\n v_min3_f32 v23, v8, v12, v16\n v_min3_f32 v24, v8, v12, v16\n v_min3_f32 v25, v8, v12, v16\n v_min3_f32 v26, v8, v12, v16\n v_min3_f32 v27, v8, v12, v16\n") ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    if (MODE == 5) asm volatile(REP8("v_fma_f32 v20, v8, v12, v9\n v_fma_f32 v21, v8, v12, v9\n v_fma_f32 v22, v8, v12, v9\n v_fma_f32 v23, v8, v12, v9\n v_fma_f32 v24, v8, v12, v9\n v_fma_f32 v25, v8, v12, v9\n v_fma_f32 v26, v8, v12, v9\n v_fma_f32 v27, v8, v12, v9\n") ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    if (MODE == 6) asm volatile(REP8("v_sub_f32 v20, v8, v9\n v_sub_f32 v21, v8, v9\n v_sub_f32 v22, v8, v9\n v_sub_f32 v23, v8, v9\n v_sub_f32 v24, v8, v9\n v_sub_f32 v25, v8, v9\n v_sub_f32 v26, v8, v9\n v_sub_f32 v27, v8, v9\n") ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    if (MODE == 7) asm volatile(REP8("v_sub_f32 v20, v8, v12\n v_sub_f32 v21, v8, v12\n v_sub_f32 v22, v8, v12\n v_sub_f32 v23, v8, v12\n v_sub_f32 v24, v8, v12\n v_sub_f32 v25, v8, v12\n v_sub_f32 v26, v8, v12\n v_sub_f32 v27, v8, v12\n") ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    // MODE 8: fma with one SGPR-like constant source (as the round-1 probe): v_fma v, v, 1.0, 0.5
    if (MODE == 8) asm volatile(REP8("v_fma_f32 v20, v8, 1.0, 0.5\n v_fma_f32 v21, v8, 1.0, 0.5\n v_fma_f32 v22, v8, 1.0, 0.5\n v_fma_f32 v23, v8, 1.0, 0.5\n v_fma_f32 v24, v8, 1.0, 0.5\n v_fma_f32 v25, v8, 1.0, 0.5\n v_fma_f32 v26, v8, 1.0, 0.5\n v_fma_f32 v27, v8, 1.0, 0.5\n") ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    // MODE 9: the filter's real pattern: fma d, x, n, w ; fmac d, y, n2   with operands spread over banks
    if (MODE == 9) asm volatile(REP8("v_fma_f32 v20, v8, v9, v10\n v_fma_f32 v21, v12, v9, v13\n v_fmac_f32 v20, v16, v10\n v_fmac_f32 v21, v8, v10\n v_fma_f32 v22, v8, v9, v10\n v_fma_f32 v23, v12, v9, v13\n v_min3_f32 v24, v20, v21, v24\n v_fmac_f32 v22, v16, v10\n") ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
  }
  float s;
  asm volatile("v_add_f32 %0, v20, v21\n v_add_f32 %0, %0, v22\n v_add_f32 %0, %0, v23\n v_add_f32 %0, %0, v24\n v_add_f32 %0, %0, v25\n v_add_f32 %0, %0, v26\n v_add_f32 %0, %0, v27\n" : "=v"(s));
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static int run(const char* name, int waves_per_simd) {
  const int threads = 256, blocks = 256 * waves_per_simd, iters = 20000;     // 256-thread block = 1 wave per SIMD of a CU
  float* out;
  CHECK(hipMalloc(&out, sizeof(float) * blocks * threads));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 2000);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double instr = (double)blocks * (threads / 64) * iters * 64.0;    // wave-instructions
  printf("%-44s waves/SIMD %d  %8.3f ms  %6.2f T lane-ops/s  %5.2f cycles/instr/SIMD @2.4GHz\n", name, waves_per_simd, ms, instr * 64 / (ms * 1e-3) / 1e12,
         (ms * 1e-3) * 2.4e9 / (instr / 1024.0));
  CHECK(hipFree(out));
  return 0;
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    if (run<0>("v_fma 3 src, three banks", w)) return 1;
    if (run<1>("v_fma 3 src, one bank", w)) return 1;
    if (run<5>("v_fma 3 src, two in one bank", w)) return 1;
    if (run<8>("v_fma 1 vgpr src + constants", w)) return 1;
    if (run<2>("v_fmac 2 src", w)) return 1;
    if (run<3>("v_min3 three banks", w)) return 1;
    if (run<4>("v_min3 one bank", w)) return 1;
    if (run<6>("v_sub two banks", w)) return 1;
    if (run<7>("v_sub one bank", w)) return 1;
    if (run<9>("filter mix", w)) return 1;
  }
  return 0;
}

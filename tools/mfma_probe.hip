// Throughput probe: can the filter's two FMAs per pair ride on v_mfma_f32_32x32x2_f32 while the VALU only keeps the minimum?
//   D(32 targets x 32 sources) = A(32 x 2: target x, y) * B(2 x 32: -2a) + C(|t|^2 per row);  VALU: m[s] = min(m[s], D[r]) for the 16 D registers
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe tools/mfma_probe.hip ; run: tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NB, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) probe(const float2* tgt, const float* w, int m, int iters, float* out) {
  extern __shared__ __align__(16) unsigned char smem[];
  float2* tile = reinterpret_cast<float2*>(smem);
  float* wt = reinterpret_cast<float*>(smem + sizeof(float2) * m);
  for (int j = threadIdx.x; j < m; j += blockDim.x) { tile[j] = tgt[j]; wt[j] = w[j]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
  float b[NB], mn[NB];
#pragma unroll
  for (int s = 0; s < NB; ++s) { b[s] = -2.f * (0.001f * (float)(blockIdx.x * 131 + threadIdx.x * 7 + s * 3 + half)); mn[s] = 3.0e38f; }
  for (int it = 0; it < iters; ++it) {
    for (int blk = 0; blk < m / 32; ++blk) {
      const float* tf = reinterpret_cast<const float*>(tile + blk * 32);
      const float a = tf[2 * col + half];                        // A[i = col][k = half]
      v16f c;
      const float4* w4 = reinterpret_cast<const float4*>(wt + blk * 32);
#pragma unroll
      for (int g = 0; g < 4; ++g) {                               // rows (g*8 + half*4 .. +3)
        const float4 q = w4[2 * g + half];
        c[4 * g] = q.x; c[4 * g + 1] = q.y; c[4 * g + 2] = q.z; c[4 * g + 3] = q.w;
      }
#pragma unroll
      for (int s = 0; s < NB; ++s) {
        const v16f d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[s], c, 0, 0, 0);
        float x = mn[s];
#pragma unroll
        for (int r = 0; r < 16; r += 2) x = fminf(fminf(d[r], d[r + 1]), x);
        mn[s] = x;
      }
    }
#pragma unroll
    for (int s = 0; s < NB; ++s) b[s] += 1e-6f;
  }
  float acc = 0.f;
#pragma unroll
  for (int s = 0; s < NB; ++s) acc += mn[s];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int NB, int WAVES>
void run(int blocks, int m, int iters) {
  std::vector<float2> ht(m); std::vector<float> hw(m);
  for (int j = 0; j < m; ++j) { ht[j] = make_float2(0.01f * (j % 97), 0.02f * (j % 89)); hw[j] = ht[j].x * ht[j].x + ht[j].y * ht[j].y; }
  float2* dt; float* dw; float* out;
  hipMalloc(&dt, sizeof(float2) * m); hipMalloc(&dw, 4 * m); hipMalloc(&out, 4 * blocks * WAVES * 64);
  hipMemcpy(dt, ht.data(), sizeof(float2) * m, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), 4 * m, hipMemcpyHostToDevice);
  const size_t smem = 12 * (size_t)m;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<NB, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NB, WAVES>), dim3(blocks), dim3(WAVES * 64), smem, 0, dt, dw, m, 2, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<NB, WAVES>), dim3(blocks), dim3(WAVES * 64), smem, 0, dt, dw, m, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double pairs = (double)blocks * WAVES * NB * 32 * (double)m * iters;
  printf("NB %d waves/WG %d blocks %d: %.3f ms, %.3e pair evals/s (the VALU kernel: 1.77e13), %s\n", NB, WAVES, blocks, ms, pairs / (ms * 1e-3), hipGetErrorString(hipGetLastError()));
  hipFree(dt); hipFree(dw); hipFree(out);
}

int main() {
  run<4, 16>(512, 4096, 50);
  run<4, 8>(512, 4096, 50);
  run<8, 8>(512, 4096, 50);
  run<8, 4>(512, 4096, 50);
  run<16, 4>(512, 4096, 50);
  run<2, 16>(512, 4096, 50);
  return 0;
}

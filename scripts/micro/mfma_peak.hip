// Micro-benchmark (GPU box): sustained rate of v_mfma_f32_16x16x4_f32 with every SIMD issuing back-to-back, random-ish operands.
// build: hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_peak.hip -o /tmp/mfma_peak ; run: /tmp/mfma_peak [waves_per_simd]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float seed) {
  f32x4 acc[10];
  for (int i = 0; i < 10; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + (threadIdx.x & 15) * 1e-2f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    a += 1e-6f;
  }
  float s = 0.f;
  for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1;
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const int blocks = cus * wps;            // 256 threads = 4 waves = 1 per SIMD per block
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int iters : {2000, 20000, 100000}) {
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 10 * 2048.0;
    printf("CUs %d, %d wave(s)/SIMD, %d iters: %.3f ms  %.1f TF/s  (=> %.2f GHz if 64 flop/clk/SIMD)\n", cus, wps, iters, ms, flops / ms / 1e9,
           flops / ms / 1e9 / (cus * 4 * 64.0) * 1e3 / 1e3);
  }
  return 0;
}

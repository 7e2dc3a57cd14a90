// Micro-benchmark (GPU box): what a dependent tiny launch costs on one stream -- the floor under the BatchNorm bookkeeping kernels.
// build: hipcc -O3 --offload-arch=gfx950 scripts/micro/launch_floor.hip -o /tmp/launch_floor ; run: /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k_empty(float* o) { if (o == nullptr) o[0] = 1.f; }
// one workgroup per channel, 256 threads, sums n_part rows of [2][C] (like bn_finalize) and writes 3 values
__global__ void k_rows(const float* __restrict__ part, int n_part, int C, float* __restrict__ out) {
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < n_part; i += blockDim.x) { s1 += part[((size_t)i * 2) * C + c]; s2 += part[((size_t)i * 2 + 1) * C + c]; }
  __shared__ double sh[2][256];
  sh[0][threadIdx.x] = s1; sh[1][threadIdx.x] = s2;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) { sh[0][threadIdx.x] += sh[0][threadIdx.x + st]; sh[1][threadIdx.x] += sh[1][threadIdx.x + st]; } __syncthreads(); }
  if (threadIdx.x == 0) { const double m = sh[0][0] / n_part; out[c] = (float)m; out[C + c] = (float)(1.0 / sqrt(sh[1][0] / n_part - m * m + 1e-5)); }
}
// a "big" kernel in front of each tiny one, so that the tiny one is a DEPENDENT launch behind real work that wrote its input
__global__ void k_big(float* __restrict__ part, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) part[i] = 1.f + 1e-3f * (i & 255); }

int main() {
  const int C = 64, n_part = 512, reps = 500;
  float *part, *out; hipMalloc(&part, (size_t)n_part * 2 * C * 4); hipMalloc(&out, 2 * C * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto fn) {
    for (int i = 0; i < 20; ++i) fn();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) fn();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-56s %.2f us per iteration\n", name, 1e3 * ms / reps);
  };
  time("big only (512 x 256 threads writing 256 KB)", [&] { hipLaunchKernelGGL(k_big, dim3(512), dim3(256), 0, 0, part, n_part * 2 * C); });
  time("big + empty kernel", [&] { hipLaunchKernelGGL(k_big, dim3(512), dim3(256), 0, 0, part, n_part * 2 * C); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, out); });
  time("big + empty kernel (64 workgroups x 256)", [&] { hipLaunchKernelGGL(k_big, dim3(512), dim3(256), 0, 0, part, n_part * 2 * C); hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, 0, out); });
  time("big + row reduction (bn_finalize-like, 64 ch x 512 rows)", [&] { hipLaunchKernelGGL(k_big, dim3(512), dim3(256), 0, 0, part, n_part * 2 * C); hipLaunchKernelGGL(k_rows, dim3(C), dim3(256), 0, 0, part, n_part, C, out); });
  time("empty kernel alone, back to back", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, out); });
  time("row reduction alone, back to back", [&] { hipLaunchKernelGGL(k_rows, dim3(C), dim3(256), 0, 0, part, n_part, C, out); });
  return 0;
}

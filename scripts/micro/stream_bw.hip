// Micro-benchmark (GPU box): what a streaming kernel can pull from HBM3E on this part -- the practical roof of the HBM-bound kernels
// (combine, classifier, narrow convs).  Variants: copy (1 read + 1 write), combine-like (2 reads + 1 write), read-only sum; plain vs
// nontemporal accesses; grid = k x CUs persistent or one workgroup per 4 KB.
// build: hipcc -O3 --offload-arch=gfx950 scripts/micro/stream_bw.hip -o /tmp/stream_bw ; run: /tmp/stream_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE, bool NT_LD, bool NT_ST>
__global__ __launch_bounds__(256) void stream(const v4f* __restrict__ a, const v4f* __restrict__ b, v4f* __restrict__ o, size_t n4) {
  v4f acc = (v4f){0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 4 * stride) {
    v4f x[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t j = i + u * stride;
      if (j < n4) {
        x[u] = NT_LD ? __builtin_nontemporal_load(a + j) : a[j];
        if (MODE == 1) y[u] = NT_LD ? __builtin_nontemporal_load(b + j) : b[j];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t j = i + u * stride;
      if (j < n4) {
        v4f v = x[u];
        if (MODE == 1) { v.x = fmaxf(v.x, 0.f) + y[u].x; v.y = fmaxf(v.y, 0.f) + y[u].y; v.z = fmaxf(v.z, 0.f) + y[u].z; v.w = fmaxf(v.w, 0.f) + y[u].w; }
        if (MODE == 2) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        else if (NT_ST) __builtin_nontemporal_store(v, o + j);
        else o[j] = v;
      }
    }
  }
  if (MODE == 2 && acc.x + acc.y + acc.z + acc.w == 123.456f) o[0] = acc;
}

template <int MODE, bool NT_LD, bool NT_ST>
static void run(const char* name, const v4f* a, const v4f* b, v4f* o, size_t n4, int grid) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stream<MODE, NT_LD, NT_ST>), dim3(grid), dim3(256), 0, 0, a, b, o, n4);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((stream<MODE, NT_LD, NT_ST>), dim3(grid), dim3(256), 0, 0, a, b, o, n4);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double bytes = (double)n4 * 16 * (MODE == 0 ? 2 : (MODE == 1 ? 3 : 1));
  printf("%-34s grid %6d : %.4f ms  %.0f GB/s\n", name, grid, ms, bytes / ms / 1e6);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const size_t n4 = (size_t)32 * 480 * 640 * 8 / 4;      // one 8-channel full-resolution activation: 314.6 MB
  v4f *a, *b, *o;
  hipMalloc(&a, n4 * 16); hipMalloc(&b, n4 * 16); hipMalloc(&o, n4 * 16);
  hipMemset(a, 0, n4 * 16); hipMemset(b, 0, n4 * 16);
  for (int k : {2, 4, 8, 16}) {
    const int g = cus * k;
    run<0, false, false>("copy", a, b, o, n4, g);
    run<0, false, true>("copy, nontemporal store", a, b, o, n4, g);
    run<0, true, true>("copy, nontemporal load+store", a, b, o, n4, g);
    run<1, false, false>("relu(a)+b", a, b, o, n4, g);
    run<1, false, true>("relu(a)+b, nontemporal store", a, b, o, n4, g);
    run<1, true, true>("relu(a)+b, nontemporal load+store", a, b, o, n4, g);
    run<2, false, false>("read only", a, b, o, n4, g);
    run<2, true, false>("read only, nontemporal", a, b, o, n4, g);
  }
  const int gfull = (int)((n4 + 1023) / 1024);
  run<0, false, false>("copy, one pass per workgroup", a, b, o, n4, gfull);
  run<1, false, true>("relu(a)+b nt store, one pass", a, b, o, n4, gfull);
  return 0;
}

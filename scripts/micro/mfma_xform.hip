// Micro-benchmark (GPU box): can the load transform ride in the shadow of the MFMAs?  8 waves per CU (2 per SIMD), each wave owns a
// 16 x 16 accumulator tile per tap (9 MFMAs per 4-pixel k-step, the 16 -> 16 channel filter gradient), operands read from LDS one
// k-step ahead.  X=0: operands used as read;  X=1: A = (r > 0 ? c0*g + c1 + c2*r : 0) from two LDS reads, B_t = x*sc + sh per tap
// (the BatchNorm/ReLU-backward and BatchNorm-apply load transforms) applied between the read and the MFMA.
// build: hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_xform.hip -o scripts/micro/mfma_xform
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PIX = 512, GW = 34, GPIX = 18 * GW;

template <int X>
__global__ __launch_bounds__(512) void k(float* out, int tiles) {
  __shared__ float pg[PIX * 16], pr[PIX * 16], gg[GPIX * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, l4 = lane >> 4;
  for (int e = tid; e < PIX * 16; e += 512) { pg[e] = 1e-3f * (e & 511) - 0.2f; pr[e] = 1e-3f * ((e * 7) & 511) - 0.1f; }
  for (int e = tid; e < GPIX * 16; e += 512) gg[e] = 1e-3f * ((e * 3) & 1023) - 0.5f;
  __syncthreads();
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int loff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) loff[t] = ((t / 3) * GW + (t % 3) + l4) * 16 + l15;
  const float c0 = 1.f + l15 * 0.01f, c1 = 0.01f * l15, c2 = -0.02f * l15, sc = 0.9f + 0.01f * l15, sh = 0.1f - 0.01f * l15;
  struct Ops { float g, r, b[9]; };
  auto load = [&](Ops& o, int j) {
    const int p0 = 4 * j, ty = p0 >> 5, tx = p0 & 31;
    o.g = pg[(p0 + l4) * 16 + l15];
    if (X) o.r = pr[(p0 + l4) * 16 + l15];
    const float* gj = gg + (ty * GW + tx) * 16;
#pragma unroll
    for (int t = 0; t < 9; ++t) o.b[t] = gj[loff[t]];
  };
  auto mma = [&](const Ops& o) {
    float a = o.g;
    if (X) a = o.r > 0.f ? fmaf(c0, o.g, fmaf(c2, o.r, c1)) : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float b = X ? fmaf(o.b[t], sc, sh) : o.b[t];
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
  };
  const int ksteps = PIX / 4, jl = ksteps - 1;
  for (int t = 0; t < tiles; ++t) {
    Ops o0, o1;
    int j = wave;
    load(o0, j);
    for (; j < ksteps; j += 16) {
      load(o1, j + 8 < jl ? j + 8 : jl);
      mma(o0);
      load(o0, j + 16 < jl ? j + 16 : jl);
      mma(o1);
    }
    __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 512 + tid] = s;
}

template <int X>
static void run(float* out, int blocks, int tiles) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<X>, dim3(blocks), dim3(512), 0, 0, out, tiles);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k<X>, dim3(blocks), dim3(512), 0, 0, out, tiles);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  const double flops = (double)blocks * tiles * (PIX / 4) * 9 * 2048.0;
  printf("transform at read %d, tiles %3d: %.4f ms/launch  %.1f TF/s\n", X, tiles, ms, flops / ms / 1e9);
}
int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  float* out; (void)hipMalloc(&out, (size_t)p.multiProcessorCount * 512 * 4);
  for (int tiles : {16, 200}) { run<0>(out, p.multiProcessorCount, tiles); run<1>(out, p.multiProcessorCount, tiles); }
  return 0;
}

// Micro-benchmark (GPU box): what does feeding v_mfma_f32_16x16x4_f32 from LDS cost?  One MFMA wave per SIMD (256 threads) keeps 36
// independent accumulators (the 64 x 64 filter-gradient tile: 2 A values x 18 B values per 4-pixel k-step) and gets its operands
//   V0: from registers (the ceiling of the instruction mix: 36 MFMAs per iteration, nothing else)
//   V1: from LDS, the reads of k-step j+1 issued before the MFMAs of k-step j (two register sets)
//   V2: from LDS, the reads in front of their own MFMAs
//   V3: V1 with a second, idle wave per SIMD that only takes part in one barrier per 20 k-steps (the producer role with nothing to do)
//   V4: V1 with a second wave per SIMD that streams global memory into the other LDS buffer (the producer role)
//   V5: V0 with 12 vector-ALU instructions per iteration mixed in (address arithmetic stand-in)
// build: hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_feed.hip -o scripts/micro/mfma_feed ; run: scripts/micro/mfma_feed
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SP = 80;                 // floats per staged pixel (64 channels + 16 pad), as in the filter-gradient kernel
constexpr int PIX = 80 + 168;          // pointwise + gathered pixels of one tile
constexpr int BUF = PIX * SP;

template <int V>
__global__ __launch_bounds__(V >= 3 && V <= 4 ? 512 : 256) void feed(float* out, const float* src, int tiles, int ksteps, int rnd) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const bool producer = threadIdx.x >= 256;
  const int l15 = lane & 15, l4 = lane >> 4;
  for (int e = threadIdx.x; e < 2 * BUF; e += blockDim.x) {
    unsigned hsh = (unsigned)e * 2654435761u + blockIdx.x * 40503u;
    hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
    // rnd: full-entropy mantissas and signs, magnitudes 0.5..4 (what a real activation / gradient tile looks like to the multipliers)
    smem[e] = rnd ? __uint_as_float((hsh & 0x80ffffffu) | ((126u + ((hsh >> 24) & 3u)) << 23)) : 1e-3f * (e & 1023);
  }
  __syncthreads();
  if (V >= 3 && producer) {
    for (int t = 0; t < tiles; ++t) {
      if (V == 4) {
        float* dst = smem + ((t + 1) & 1) * BUF;
        const float4* s4 = reinterpret_cast<const float4*>(src) + ((size_t)blockIdx.x * tiles + t) * (BUF / 4);
        for (int e = tid; e < BUF / 4; e += 256 * 4) {
          float4 x[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) x[u] = e + u * 256 < BUF / 4 ? s4[e + u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (e + u * 256 < BUF / 4) {
              float4 v = x[u];
              v.x = fmaxf(v.x * 1.01f + 0.5f, 0.f); v.y = fmaxf(v.y * 1.01f + 0.5f, 0.f); v.z = fmaxf(v.z * 1.01f + 0.5f, 0.f); v.w = fmaxf(v.w * 1.01f + 0.5f, 0.f);
              reinterpret_cast<float4*>(dst)[e + u * 256] = v;
            }
        }
      }
      __syncthreads();
    }
    return;
  }
  f32x4 acc[18][2];
#pragma unroll
  for (int t = 0; t < 18; ++t) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
  int loff[18];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    loff[2 * t] = ((t / 3) * 42 + (t % 3)) * SP + l4 * SP + (wave & 1) * 32 + l15;
    loff[2 * t + 1] = loff[2 * t] + 16;
  }
  const int a_lane = l4 * SP + (wave >> 1) * 32 + l15;
  float av0[2], bv0[18], av1[2], bv1[18];
  auto load_ops = [&](const float* pl, const float* gl, int j, float (&av)[2], float (&bv)[18]) {
    const int p0 = 4 * j;
    const int ty = p0 / 40, tx = p0 - ty * 40;
    av[0] = pl[p0 * SP + a_lane]; av[1] = pl[p0 * SP + a_lane + 16];
    const float* gj = gl + (ty * 42 + tx) * SP;
#pragma unroll
    for (int t = 0; t < 18; ++t) bv[t] = gj[loff[t]];
  };
  auto mfma_ops = [&](const float (&av)[2], const float (&bv)[18]) {
#pragma unroll
    for (int t = 0; t < 18; ++t) {
      acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[t], acc[t][0], 0, 0, 0);
      acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[t], acc[t][1], 0, 0, 0);
    }
  };
  if (V == 0 || V == 5) {
    float a0 = rnd ? smem[tid] : 1.f + tid * 1e-3f, a1 = rnd ? smem[tid + 256] : 0.5f + tid * 1e-3f;
    float bb[18];
#pragma unroll
    for (int t = 0; t < 18; ++t) bb[t] = rnd ? smem[512 + t * 256 + tid] : 0.25f * t + l15 * 1e-2f;
    int x = tid;
    for (int t = 0; t < tiles; ++t)
      for (int j = 0; j < ksteps; ++j) {
        float av[2] = {a0, a1};
        mfma_ops(av, bb);
        if (V == 5) {
#pragma unroll
          for (int u = 0; u < 12; ++u) x = x * 3 + u;
          asm volatile("" : "+v"(x));
        }
        a0 += 1e-6f;
      }
    if (x == 0x12345) out[0] = 1.f;
  } else {
    for (int t = 0; t < tiles; ++t) {
      const float* pl = smem + (t & 1) * BUF;
      const float* gl = pl + 80 * SP;
      if (V == 2) {
        for (int j = 0; j < ksteps; ++j) { load_ops(pl, gl, j, av0, bv0); mfma_ops(av0, bv0); }
      } else {
        const int jl = ksteps - 1;
        load_ops(pl, gl, 0, av0, bv0);
        for (int j = 0; j < ksteps; j += 2) {
          load_ops(pl, gl, j + 1 < jl ? j + 1 : jl, av1, bv1);
          mfma_ops(av0, bv0);
          load_ops(pl, gl, j + 2 < jl ? j + 2 : jl, av0, bv0);
          mfma_ops(av1, bv1);
        }
      }
      if (V >= 3) __syncthreads();
    }
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 18; ++t) s += acc[t][0][0] + acc[t][0][1] + acc[t][0][2] + acc[t][0][3] + acc[t][1][0] + acc[t][1][1] + acc[t][1][2] + acc[t][1][3];
  out[blockIdx.x * 256 + tid] = s;
}

template <int V>
static void run(const char* name, float* out, const float* src, int blocks, int tiles, int ksteps, int rnd) {
  const int threads = (V >= 3 && V <= 4) ? 512 : 256;
  const size_t lds = 2 * BUF * sizeof(float);
  hipFuncSetAttribute(reinterpret_cast<const void*>(feed<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(feed<V>, dim3(blocks), dim3(threads), lds, 0, out, src, tiles, ksteps, rnd);
  hipDeviceSynchronize();
  const int reps = 20;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(feed<V>, dim3(blocks), dim3(threads), lds, 0, out, src, tiles, ksteps, rnd);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flops = (double)blocks * 4 * tiles * ksteps * 36 * 2048.0;
  printf("%-58s %s tiles %3d: %.4f ms/launch  %.1f TF/s\n", name, rnd ? "random data " : "regular data", tiles, ms, flops / ms / 1e9);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount;
  float *out, *src;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  const size_t src_floats = (size_t)blocks * 160 * BUF;
  hipMalloc(&src, src_floats * 4);
  hipMemset(src, 0x3c, src_floats * 4);
  for (int rnd : {0, 1})
  for (int tiles : {8, 160}) {          // 8 tiles x 20 k-steps: the length of the real kernel (~0.1 ms); 160: sustained
    run<0>("V0 registers only", out, src, blocks, tiles, 20, rnd);
    run<5>("V5 registers + 12 VALU per k-step", out, src, blocks, tiles, 20, rnd);
    run<1>("V1 LDS operands, prefetched one k-step ahead", out, src, blocks, tiles, 20, rnd);
    run<2>("V2 LDS operands, read in front of their MFMAs", out, src, blocks, tiles, 20, rnd);
    run<3>("V3 = V1 + idle second wave per SIMD, barrier per tile", out, src, blocks, tiles, 20, rnd);
    run<4>("V4 = V1 + staging second wave per SIMD", out, src, blocks, tiles, 20, rnd);
  }
  return 0;
}

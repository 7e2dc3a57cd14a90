// 3x3 stride-1 convolution of the wide (>= 64-channel) layers with fp32 products formed on the bf16 matrix pipe.
//
// The roles conv_wino.hip serves (forward of Conv2d(k3, pad 1) and the data gradient of the same layer, with the load transforms and
// epilogues of conv_mfma.hip), as a DIRECT convolution: every fp32 operand is split exactly into three bf16 values (x = h + m + l, see
// wgrad_bf3.hip for the arithmetic and its error against fp64) and each multiply-add is six v_mfma_f32_16x16x32_bf16 products.  Nine taps
// x 6/16 of an fp32 MFMA = 3.4 fp32-MFMA equivalents per output element and input channel against 4.0 for Winograd F(2x2,3x3) on the
// fp32 instruction -- without its input / output transforms, which kept the matrix pipe of conv_wino 40 % busy.
//
// GEMM mapping:  D[co][pixel] += W[co][k] * X[k][pixel],  k = (tap, ci), 32 input channels of one tap per MFMA.
//   A = filter, split when it is packed (RCV_OP_PACK layout 3): [plane][tap][ci / 32][co][32 ci] bf16 -- lane (co, g) reads its eight
//       consecutive ci with ONE 16-byte global load; fragments go global (L2 resident, < 1 MB) -> register, three k-steps ahead, no LDS;
//   B = input tile, [pixel][plane h|m|l][32 ci] bf16 in LDS (192 B per pixel): lane (pixel, g) reads 16 bytes; tap shifts are address
//       offsets.  The tile (TH x TW pixels + halo, one 32-channel chunk) is staged by four producer waves (global -> load transform ->
//       split -> three 8-byte writes per channel quad) into one buffer while four consumer waves contract the other.
//   D: lane ends with four consecutive output channels of one pixel -- the epilogue of conv_mfma.hip (bias, ReLU, residual, BatchNorm
//      partial sums) is used as it is.
// Workgroup tile: 64 output channels x (2 x WN) pixel blocks of 16 (WN = 10: 320 pixel slots, WN = 5: 160), pixels taken row-major
// from a TH x TW rectangle (30 x 10 on the 30 x 40 planes of the 128-channel layers: 256 workgroups = one per CU).
// (Measured and not kept, round 3: the same tile on v_mfma_f32_32x32x16_bf16 -- one 32-channel x five 32-pixel blocks per wave, pixel
// pitch 208 B so that the 16-byte reads of a 32-pixel block are conflict free, the shared epilogue with a 32-lane pixel map.  The bare
// MFMA loop of the micro-benchmark is 20 % faster in that form; the kernel was SLOWER: 128 -> 128 0.065 -> 0.069 ms forward,
// 0.070 -> 0.080 data gradient, 64 -> 64 0.074 -> 0.083 / 0.085 -> 0.101, results identical to 1e-6.)
#include <type_traits>
#include "conv_common.h"
#include "conv_epilogue.h"

typedef __bf16 c3_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 c3_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int C3_PITCH = 192;             // bytes per staged pixel (stride-1 form): three planes of 32 bf16
constexpr int C3_NU = 13;                 // staging passes (256 / (CH / 4) pixels each): tiles with halo up to 416 (CH = 32) / 832 (CH = 16) pixels
constexpr int C3_COT = 64;

__device__ __forceinline__ uint32_t c3_pack(float a, float b) {
  const c3_bf16x2 v = {(__bf16)a, (__bf16)b};            // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(uint32_t, v);
}
struct C3Tri { uint32_t h, m, l; };
__device__ __forceinline__ C3Tri c3_split2(float x0, float x1) {
  C3Tri t;
  t.h = c3_pack(x0, x1);
  const float r0 = x0 - __uint_as_float(t.h << 16), r1 = x1 - __uint_as_float(t.h & 0xffff0000u);       // exact
  t.m = c3_pack(r0, r1);
  const float s0 = r0 - __uint_as_float(t.m << 16), s1 = r1 - __uint_as_float(t.m & 0xffff0000u);       // exact
  t.l = c3_pack(s0, s1);
  return t;
}

template <bool TWO>
struct C3Regs {
  float4 x[C3_NU], ax[TWO ? C3_NU : 1];
  bool ok[C3_NU];
};

// all loads of one CH-channel chunk of the tile (CH / 4 threads per pixel, 256 / (CH / 4) pixels per pass)
template <bool TWO, int CH = 32>
__device__ __forceinline__ void c3_load(C3Regs<TWO>& r, const ConvArgs& a, const TileInfo& ti, int c0, int tid, int npix) {
  constexpr int Q = CH / 4, PP = 256 / Q;
  const int q = tid % Q, lp = tid / Q;
#pragma unroll
  for (int u = 0; u < C3_NU; ++u) {
    const int pix = u * PP + lp;
    const int iy = fd_div(pix, a.fdIW), ix = pix - iy * a.IW;
    const int gy = ti.oy0 + iy, gx = ti.ox0 + ix;
    r.ok[u] = pix < npix && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    const uint32_t o = r.ok[u] ? (uint32_t)(((ti.n * a.H + gy) * a.W + gx) * a.Cin + c0 + 4 * q) : 0u;
    r.x[u] = ld4(a.in + o);
    if (TWO) r.ax[u] = ld4(a.in_aux + o);
  }
}
template <int MODE, bool TWO, int CH = 32>
__device__ __forceinline__ void c3_store(const C3Regs<TWO>& r, const ConvArgs& a, char* img, int c0, int tid, int npix) {
  constexpr int Q = CH / 4, PP = 256 / Q;
  const int q = tid % Q, lp = tid / Q;
  float4 k[5];
  if (MODE != RCV_LOAD_PLAIN) {
#pragma unroll
    for (int j = 0; j < 5; ++j) k[j] = ld4(a.in_c + (size_t)j * a.Cin + c0 + 4 * q);
  }
#pragma unroll
  for (int u = 0; u < C3_NU; ++u) {
    const int pix = u * PP + lp;
    float4 v = xform4<MODE>(r.x[u], r.ax[TWO ? u : 0], k);
    if (!r.ok[u]) v = make_float4(0.f, 0.f, 0.f, 0.f);          // zero padding AFTER the transform
    if (pix < npix) {
      const C3Tri lo = c3_split2(v.x, v.y), hi = c3_split2(v.z, v.w);
      char* d = img + pix * (6 * CH) + 8 * q;
      *reinterpret_cast<uint2*>(d) = make_uint2(lo.h, hi.h);
      *reinterpret_cast<uint2*>(d + 2 * CH) = make_uint2(lo.m, hi.m);
      *reinterpret_cast<uint2*>(d + 4 * CH) = make_uint2(lo.l, hi.l);
    }
  }
}

template <int WN, bool TWO>
__global__ __launch_bounds__(512) void conv_bf3_kernel(const ConvArgs a) {
  constexpr int WM = 2;
  extern __shared__ __attribute__((aligned(16))) char smem_c3[];
  const int npix = a.IH * a.IW;
  const int xbytes = a.xl_floats * 4;                       // one input buffer
  float* red = reinterpret_cast<float*>(smem_c3 + 2 * xbytes);
  const bool producer = threadIdx.x >= 256;
  const int tid = producer ? (int)threadIdx.x - 256 : (int)threadIdx.x;
  const TileInfo ti = decode_tile<KIND_GATHER>(a, xcd_remap(blockIdx.x, a.total_tiles), C3_COT);
  const int nchunks = a.nchunks;
  const bool do_stage = !(a.flags & RCV_F_DBG_NOSTAGE), do_mfma = !(a.flags & RCV_F_DBG_NOMFMA);      // (ablation timings: scripts/bench_op.py --flags)

  if (producer) {
    auto stage = [&](int c, char* buf) {
      C3Regs<TWO> r;
      c3_load<TWO>(r, a, ti, 32 * c, tid, npix);
      if (TWO) {
        if (a.in_mode == RCV_LOAD_GRAD_ENC) c3_store<RCV_LOAD_GRAD_ENC, TWO>(r, a, buf, 32 * c, tid, npix);
        else c3_store<RCV_LOAD_GRAD_DEC, TWO>(r, a, buf, 32 * c, tid, npix);
      } else {
        switch (a.in_mode) {
          case RCV_LOAD_PLAIN: c3_store<RCV_LOAD_PLAIN, TWO>(r, a, buf, 32 * c, tid, npix); break;
          case RCV_LOAD_AFFINE: c3_store<RCV_LOAD_AFFINE, TWO>(r, a, buf, 32 * c, tid, npix); break;
          default: c3_store<RCV_LOAD_AFFINE_RELU, TWO>(r, a, buf, 32 * c, tid, npix); break;
        }
      }
    };
    // barrier for barrier the consumer path: 1 + one per chunk (+ the epilogue's when it reduces statistics)
    if (do_stage) stage(0, smem_c3);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks && do_stage) stage(c + 1, smem_c3 + ((c + 1) & 1) * xbytes);
      __syncthreads();
    }
    if (a.stats != RCV_STATS_NONE) __syncthreads();
    return;
  }

  // ---------------- consumer waves ----------------
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  f32x4 acc[WM][WN];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int pixoff[WN];
#pragma unroll
  for (int n = 0; n < WN; ++n) {
    const int p = (wave_n * WN + n) * 16 + l15;
    int ty = fd_div(p, a.fdWt), tx = p - ty * a.Wt;
    if (ty >= a.R) { ty = 0; tx = 0; }
    pixoff[n] = (ty * a.IW + tx) * C3_PITCH + l4 * 16;
  }
  // filter fragments: lane (co, g) of block m reads 16 bytes at ((kstep * CoutP + co) * 32 + 8 g) bf16 of each plane
  const char* wb[WM];
#pragma unroll
  for (int m = 0; m < WM; ++m) {
    int co = ti.co0 + (wave_m * WM + m) * 16 + l15;
    if (co >= a.CoutP) co = a.CoutP - 1;
    wb[m] = reinterpret_cast<const char*>(a.w) + (size_t)co * 64 + l4 * 16;
  }
  const size_t wstep = (size_t)a.CoutP * 64;                      // bytes per k-step (tap, chunk) of one plane
  const size_t wplane = 9 * (size_t)nchunks * wstep;
  c3_bf16x8 A[3][WM][3];                                          // [ring slot][m][plane]
  auto load_a = [&](int tap, int c, c3_bf16x8 (&dst)[WM][3]) {
    if (c >= nchunks) { c = nchunks - 1; }                        // (past the end: a repeated fragment, never used)
    const size_t o = (size_t)(tap * nchunks + c) * wstep;
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dst[m][pl] = *reinterpret_cast<const c3_bf16x8*>(wb[m] + pl * wplane + o);
  };
  load_a(0, 0, A[0]);
  load_a(1, 0, A[1]);
  load_a(2, 0, A[2]);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const char* xb = smem_c3 + (c & 1) * xbytes;
    if (do_mfma)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int tapoff = ((tap / 3) * a.IW + (tap % 3)) * C3_PITCH;
      const c3_bf16x8 (&Ac)[WM][3] = A[tap % 3];
      c3_bf16x8 B[2][3];
      auto load_b = [&](int n, c3_bf16x8 (&dst)[3]) {
        const char* pb = xb + pixoff[n] + tapoff;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) dst[pl] = *reinterpret_cast<const c3_bf16x8*>(pb + pl * 64);
      };
      load_b(0, B[0]);
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        if (n + 1 < WN) load_b(n + 1, B[(n + 1) & 1]);
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};     // smallest products first
#pragma unroll
        for (int e = 0; e < 6; ++e)
#pragma unroll
          for (int m = 0; m < WM; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ac[m][TA[e]], B[n & 1][TB[e]], acc[m][n], 0, 0, 0);
        // the three reads of the next pixel block go behind the first MFMAs of this one (left alone the compiler sinks them to the end
        // of the block, and the next block starts with their latency)
        if (n + 1 < WN) {
#pragma unroll
          for (int e = 0; e < 3; ++e) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // the ring slot of this k-step is free: request the fragments of three k-steps ahead
      {
        int t3 = tap + 3, c3 = c;
        if (t3 >= 9) { t3 -= 9; c3 += 1; }
        if (!(a.flags & RCV_F_DBG_NOSKIP)) load_a(t3, c3, A[tap % 3]);      // (ablation: the fragments of the first three k-steps for all)
      }
    }
    __syncthreads();
  }
  // (EB = 5: the residual / BatchNorm-backward operands of five pixel blocks are requested in one batch -- loaded where they are used they are
  // serialized HBM round trips in the tail of the data-gradient launches)
  if (!(a.flags & RCV_F_DBG_NOEPI)) conv_epilogue<WM, WN, 2, 2, KIND_GATHER, 5>(a, ti, acc, red, tid);
  else if (a.stats != RCV_STATS_NONE) __syncthreads();
}

// Stride-2 form (the 32 -> 64 and 64 -> 128 downsampling convs, and the data gradients of the transposed convs that mirror them).  An
// output pixel needs FOUR input pixels, so the input tile is staged in 16-channel chunks (96 B per pixel: a 10 x 16 output tile with
// its 21 x 33 input pixels is 66 KB per buffer) and a 32-deep k-step spans TWO taps: k = tap * 16 + ci inside a chunk, lane group g
// of the MFMA holds k = 32 ks + 8 g .. + 7 = eight channels of tap (32 ks + 8 g) / 16 -- a per-lane LDS offset per k-step, as in
// convn_bf3.hip.  Nine taps are 4.5 k-steps: five per chunk, the filter (RCV_OP_PACK layout 5: [plane][chunk][5][co][32]) zero in the
// last half.  Filter fragments: a ring of five register sets = one per k-step of a chunk, refilled for the next chunk as soon as a
// k-step is done (five k-steps ahead).
template <int WN, bool TWO>
__global__ __launch_bounds__(512) void conv2_bf3_kernel(const ConvArgs a) {
  constexpr int WM = 2, CH = 16, PITCH = 6 * CH, NKS = 5;
  extern __shared__ __attribute__((aligned(16))) char smem_c3[];
  const int npix = a.IH * a.IW;
  const int xbytes = a.xl_floats * 4;
  float* red = reinterpret_cast<float*>(smem_c3 + 2 * xbytes);
  const bool producer = threadIdx.x >= 256;
  const int tid = producer ? (int)threadIdx.x - 256 : (int)threadIdx.x;
  const TileInfo ti = decode_tile<KIND_GATHER>(a, xcd_remap(blockIdx.x, a.total_tiles), C3_COT);
  const int nchunks = a.nchunks;

  if (producer) {
    auto stage = [&](int c, char* buf) {
      C3Regs<TWO> r;
      c3_load<TWO, CH>(r, a, ti, CH * c, tid, npix);
      if (TWO) {
        if (a.in_mode == RCV_LOAD_GRAD_ENC) c3_store<RCV_LOAD_GRAD_ENC, TWO, CH>(r, a, buf, CH * c, tid, npix);
        else c3_store<RCV_LOAD_GRAD_DEC, TWO, CH>(r, a, buf, CH * c, tid, npix);
      } else {
        switch (a.in_mode) {
          case RCV_LOAD_PLAIN: c3_store<RCV_LOAD_PLAIN, TWO, CH>(r, a, buf, CH * c, tid, npix); break;
          case RCV_LOAD_AFFINE: c3_store<RCV_LOAD_AFFINE, TWO, CH>(r, a, buf, CH * c, tid, npix); break;
          default: c3_store<RCV_LOAD_AFFINE_RELU, TWO, CH>(r, a, buf, CH * c, tid, npix); break;
        }
      }
    };
    stage(0, smem_c3);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks) stage(c + 1, smem_c3 + ((c + 1) & 1) * xbytes);
      __syncthreads();
    }
    if (a.stats != RCV_STATS_NONE) __syncthreads();
    return;
  }

  const int lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  f32x4 acc[WM][WN];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int pixoff[WN];
#pragma unroll
  for (int n = 0; n < WN; ++n) {
    const int p = (wave_n * WN + n) * 16 + l15;
    int ty = fd_div(p, a.fdWt), tx = p - ty * a.Wt;
    if (ty >= a.R) { ty = 0; tx = 0; }
    pixoff[n] = ((2 * ty) * a.IW + 2 * tx) * PITCH;
  }
  int koff[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int k0 = 32 * ks + 8 * l4;
    int tap = k0 >> 4, ci0 = k0 & 15;
    if (tap >= 9) { tap = 0; ci0 = 0; }                      // beyond the ninth tap the filter is zero: read any finite data
    koff[ks] = ((tap / 3) * a.IW + (tap % 3)) * PITCH + 2 * ci0;
  }
  const char* wb[WM];
#pragma unroll
  for (int m = 0; m < WM; ++m) {
    int co = ti.co0 + (wave_m * WM + m) * 16 + l15;
    if (co >= a.CoutP) co = a.CoutP - 1;
    wb[m] = reinterpret_cast<const char*>(a.w) + (size_t)co * 64 + l4 * 16;
  }
  const size_t wstep = (size_t)a.CoutP * 64;
  const size_t wplane = (size_t)NKS * nchunks * wstep;
  c3_bf16x8 A[NKS][WM][3];
  auto load_a = [&](int c, int ks, c3_bf16x8 (&dst)[WM][3]) {
    if (c >= nchunks) c = nchunks - 1;
    const size_t o = (size_t)(c * NKS + ks) * wstep;
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dst[m][pl] = *reinterpret_cast<const c3_bf16x8*>(wb[m] + pl * wplane + o);
  };
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) load_a(0, ks, A[ks]);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const char* xb = smem_c3 + (c & 1) * xbytes;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      c3_bf16x8 B[2][3];
      auto load_b = [&](int n, c3_bf16x8 (&dst)[3]) {
        const char* pb = xb + pixoff[n] + koff[ks];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) dst[pl] = *reinterpret_cast<const c3_bf16x8*>(pb + pl * 2 * CH);
      };
      load_b(0, B[0]);
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        if (n + 1 < WN) load_b(n + 1, B[(n + 1) & 1]);
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int e = 0; e < 6; ++e)
#pragma unroll
          for (int m = 0; m < WM; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ks][m][TA[e]], B[n & 1][TB[e]], acc[m][n], 0, 0, 0);
        if (n + 1 < WN) {
#pragma unroll
          for (int e = 0; e < 3; ++e) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      load_a(c + 1, ks, A[ks]);
    }
    __syncthreads();
  }
  conv_epilogue<WM, WN, 2, 2, KIND_GATHER>(a, ti, acc, red, tid);
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
struct C3Geom { int WN, TH, TW, IH, IW, tiles_x, tiles_y, n_co_tiles, total; size_t lds; };

static bool c3_geometry(const rcv_handle* h, const rcv_op* op, C3Geom* g) {
  const int N = op->i[RCV_I_N], Ho = op->i[RCV_I_HO], Wo = op->i[RCV_I_WO], Cout = op->i[RCV_I_COUT], S = op->i[RCV_I_STRIDE];
  const int n_co = ceil_div(Cout, C3_COT);
  const int pitch = S == 1 ? C3_PITCH : C3_PITCH / 2, per_pass = S == 1 ? 32 : 64;      // stride 2: 16-channel chunks
  double best = 1e30;
  bool found = false;
  for (int WN : {10, 5}) {
    if (S == 2 && WN != 5) continue;                             // (its five-set filter ring leaves registers for five pixel blocks)
    if (const char* ev = RCV_ENV("RCV_BF3_WN")) { if (atoi(ev) != WN && S == 1) continue; }      // experiments build: force the tile size
    const int slots = 2 * WN * 16;
    for (int TW = 4; TW <= Wo && TW <= 64; ++TW) {
      int TH = slots / TW;
      if (TH > Ho) TH = Ho;
      if (TH < 1) continue;
      TH = ceil_div(Ho, ceil_div(Ho, TH));                       // equal row groups
      const int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
      if (IH * IW > C3_NU * per_pass || IW >= 65536) continue;
      const size_t lds = 2 * (size_t)round_up(IH * IW * pitch, 16) + 2 * 2 * C3_COT * sizeof(float) * 2;
      if (lds > (size_t)h->max_lds) continue;
      const long tiles = (long)N * ceil_div(Ho, TH) * ceil_div(Wo, TW) * n_co;
      const long rounds = ceil_div((int)tiles, h->num_cus);
      // time ~ rounds x (pixel slots of the tile + a fixed part: first chunk's staging, epilogue), halo as the tie break
      const double cost = (double)rounds * (slots + 64) * (1.0 + 0.02 * (double)(IH * IW) / (TH * TW * S * S));
      if (cost < best) {
        best = cost; found = true;
        g->WN = WN; g->TH = TH; g->TW = TW; g->IH = IH; g->IW = IW; g->tiles_x = ceil_div(Wo, TW); g->tiles_y = ceil_div(Ho, TH);
        g->n_co_tiles = n_co; g->total = (int)tiles; g->lds = lds;
      }
    }
  }
  return found;
}

bool conv_bf3_wanted(const rcv_handle* h, const rcv_op* op) {
  if (RCV_ENV("RCV_NO_BF3") || (op->flags & RCV_F_MFMA_FP32)) return false;
  const int N = op->i[RCV_I_N], H = op->i[RCV_I_H], W = op->i[RCV_I_W], Cin = op->i[RCV_I_CIN], Cout = op->i[RCV_I_COUT];
  const int S = op->i[RCV_I_STRIDE];
  if (op->kind != RCV_OP_CONV || (S != 1 && S != 2) || op->i[RCV_I_DIL] != 1) return false;
  if (op->i[RCV_I_INMODE] == RCV_LOAD_NCHW || Cout % 4 || Cout < 64) return false;
  if (S == 1 ? (Cin % 32 || Cin < 64) : (Cin % 16 || Cin < 32 || RCV_ENV("RCV_NO_BF3S2") != nullptr)) return false;      // stride 2: layout 5, 16-channel chunks
  if (S == 2 && !RCV_ENV("RCV_BF3S2_ALL")) {
    // Stride 2 only in the FORWARD pass.  Op by op conv2_bf3 beats conv_dma on all four stride-2 launches of the step (32 -> 64: 91 -> 79 us
    // forward, 115 -> 110 as the data gradient of the mirrored transposed conv; 64 -> 128: 77 -> 58, 89 -> 72), but with the backward
    // launches on it the two-stream step measured SLOWER, 5.49 -> 5.55 ms, three interleaved pairs on one box: its 134 KB of LDS per
    // workgroup keep the side stream's filter-gradient workgroups off the CUs that conv_dma's tiles share with them.  Forward only:
    // 5.485 -> 5.468 ms.
    const int m = op->i[RCV_I_INMODE];
    if (m == RCV_LOAD_GRAD_ENC || m == RCV_LOAD_GRAD_DEC) return false;
  }
  if ((long long)N * H * W * Cin >= (1ll << 31)) return false;
  if (RCV_ENV("RCV_BF3_FWD")) {        // experiments build: forward launches only (what the two-stream step makes of the backward ones)
    const int m = op->i[RCV_I_INMODE];
    if (m == RCV_LOAD_GRAD_ENC || m == RCV_LOAD_GRAD_DEC) return false;
  }
  C3Geom g;
  if (!c3_geometry(h, op, &g)) return false;
  // the grid must cover most of the chip (one 512-thread workgroup per CU): small planes stay on the other kernels
  return (long)g.total * 4 >= (long)h->num_cus * 3;
}

bool conv_bf3_supported(const rcv_handle* h, const rcv_op* op, int kind) {
  // the record carries a filter packed in a split layout (rcv_op_filter_layout): 3 = stride 1, 5 = stride 2 (16-channel chunks)
  return kind == KIND_GATHER && ((op->i[RCV_I_AUX0] == 3 && op->i[RCV_I_STRIDE] == 1) || (op->i[RCV_I_AUX0] == 5 && op->i[RCV_I_STRIDE] == 2));
}

int conv_bf3_plan(const rcv_handle* h, const rcv_op* op, ConvPlan* pl) {
  const int Cin = op->i[RCV_I_CIN], Cout = op->i[RCV_I_COUT];
  const int S = op->i[RCV_I_STRIDE];
  RCV_CHECK_ARG(op->kind == RCV_OP_CONV && (S == 1 || S == 2) && op->i[RCV_I_DIL] == 1 && Cin % (S == 1 ? 32 : 16) == 0 && Cout % 4 == 0 &&
                    op->i[RCV_I_INMODE] != RCV_LOAD_NCHW && !(op->flags & RCV_F_MFMA_FP32),
                "split-bf16 conv: needs dilation 1, Cin %% 32 == 0 (stride 1) or %% 16 (stride 2), an NHWC input (got s%d d%d Cin %d Cout %d)", S,
                op->i[RCV_I_DIL], Cin, Cout);
  C3Geom g;
  RCV_CHECK_ARG(c3_geometry(h, op, &g), "split-bf16 conv: no tile fits %dx%d", op->i[RCV_I_HO], op->i[RCV_I_WO]);
  pl->kind = KIND_GATHER; pl->narrow = 0; pl->dma = 0; pl->first = 0; pl->wino = 0; pl->small = 0; pl->bf3 = 1;
  pl->CK = S == 1 ? 32 : 16; pl->CoutV = Cout; pl->CoutP = round_up(Cout, 16);
  pl->R = g.TH; pl->Wt = g.TW; pl->IH = g.IH; pl->IW = g.IW; pl->tiles_x = g.tiles_x; pl->tiles_y = g.tiles_y;
  pl->n_co_tiles = g.n_co_tiles; pl->n_phases = 1; pl->total_tiles = g.total; pl->grid = g.total;
  pl->WN = g.WN; pl->WM = 2;
  pl->xl_floats = round_up(g.IH * g.IW * (S == 1 ? C3_PITCH : C3_PITCH / 2), 16) / 4; pl->wl_floats = 0;
  pl->lds = g.lds;
  return RCV_OK;
}

template <int WN, bool TWO>
static int c3_launch_inst(const ConvPlan& pl, const ConvArgs& a, hipStream_t s) {
  auto kern = conv_bf3_kernel<WN, TWO>;
  static size_t configured[RCV_MAX_DEVICES];
  RCV_ENSURE_LDS(kern, pl.lds, pl.dev, configured);
  hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(512), pl.lds, s, a);
  RCV_HIP(hipGetLastError());
  return RCV_OK;
}

template <bool TWO>
static int c3_launch_s2(const ConvPlan& pl, const ConvArgs& a, hipStream_t s) {
  auto kern = conv2_bf3_kernel<5, TWO>;
  static size_t configured[RCV_MAX_DEVICES];
  RCV_ENSURE_LDS(kern, pl.lds, pl.dev, configured);
  hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(512), pl.lds, s, a);
  RCV_HIP(hipGetLastError());
  return RCV_OK;
}

int conv_bf3_launch(const ConvPlan& pl, const ConvArgs& a, hipStream_t s) {
  const bool two = a.in_mode == RCV_LOAD_GRAD_ENC || a.in_mode == RCV_LOAD_GRAD_DEC;
  if (a.stride == 2) return two ? c3_launch_s2<true>(pl, a, s) : c3_launch_s2<false>(pl, a, s);
  if (pl.WN == 10) return two ? c3_launch_inst<10, true>(pl, a, s) : c3_launch_inst<10, false>(pl, a, s);
  return two ? c3_launch_inst<5, true>(pl, a, s) : c3_launch_inst<5, false>(pl, a, s);
}

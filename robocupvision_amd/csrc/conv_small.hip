// 3x3 convolution on planes so small that the tiled kernels cannot fill the chip (inference only).
//
// LabelProp's three dilated 32/64-channel convs (model.py:546-548: conv1..conv3 at 1/8 resolution) see 2 x 15 x 20 = 600 output pixels
// for one frame pair (BASELINE config 5, validLabelProp.py:132-135).  The tiled kernels give such a layer 8 workgroups, each walking its
// whole K = 9 taps x Cin loop alone: 20..26 us per layer, half of the 0.14 ms latency of the call, on 3 % of the compute units.  Here
//   * one workgroup = ONE 16-pixel x 16-channel MFMA block of the output (600 pixels x 64 channels: 152 workgroups),
//   * its eight waves split K (k-step = one tap x 4 input channels) round robin and meet through LDS in a fixed order,
//   * operands go global -> register -> MFMA (the whole input is 77..154 KB, the filter 74..147 KB: L2 resident; there is nothing to
//     reuse inside a 16 x 16 block that an LDS stage would save), six k-steps of loads in flight per wave.
// Load transform (producer's BatchNorm / BatchNorm + ReLU), zero padding after it, bias and ReLU as in conv_mfma.hip; no statistics,
// no residual: training layers never take this path (conv_small_supported).
#include "conv_common.h"

constexpr int CS_NW = 8;                        // waves per workgroup (K slices)

template <int MODE>
__global__ __launch_bounds__(CS_NW * 64) void conv_small_kernel(const ConvArgs a, int n_co_blk, int total_px) {
  constexpr int NW = CS_NW;
  __shared__ float cs[2][128];                 // load constants (scale, shift) of every input channel
  __shared__ f32x4 red[NW - 1][64];            // accumulators of waves 1..NW-1
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  if (MODE != RCV_LOAD_PLAIN) {
    for (int e = tid; e < 2 * a.Cin; e += NW * 64) cs[e / a.Cin][e % a.Cin] = a.in_c[e];
    __syncthreads();
  }
  const int co_blk = blockIdx.x % n_co_blk, pix_blk = blockIdx.x / n_co_blk;
  const int co0 = co_blk * 16;
  const int p = pix_blk * 16 + l15;
  const bool valid = p < total_px;
  const int pp = valid ? p : 0;
  const int n = pp / (a.Ho * a.Wo), rem = pp - n * (a.Ho * a.Wo);
  const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
  const int by = oy * a.stride - a.dil, bx = ox * a.stride - a.dil;
  const int Q = a.CinP >> 2, nks = 9 * Q;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int U = 6;
  for (int ks0 = wave; ks0 < nks; ks0 += NW * U) {
    float av[U], bv[U], sc[U], sh[U];
    bool inb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ks = ks0 + NW * u;
      const bool live = ks < nks;                                  // (wave uniform)
      const int kc = live ? ks : 0;
      const int tap = kc / Q, c4 = kc - tap * Q;
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int ci = 4 * c4 + l4;
      const int iy = by + ky * a.dil, ix = bx + kx * a.dil;
      inb[u] = live && valid && ci < a.Cin && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      av[u] = live ? a.w[(size_t)(tap * a.CinP + ci) * a.CoutP + co0 + l15] : 0.f;
      bv[u] = inb[u] ? a.in[((size_t)(n * a.H + iy) * a.W + ix) * a.Cin + ci] : 0.f;
      if (MODE != RCV_LOAD_PLAIN) { sc[u] = cs[0][ci < a.Cin ? ci : 0]; sh[u] = cs[1][ci < a.Cin ? ci : 0]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float b = bv[u];
      if (MODE == RCV_LOAD_AFFINE) b = fmaf(b, sc[u], sh[u]);
      if (MODE == RCV_LOAD_AFFINE_RELU) b = fmaxf(fmaf(b, sc[u], sh[u]), 0.f);
      if (MODE != RCV_LOAD_PLAIN && !inb[u]) b = 0.f;              // zero padding AFTER the transform
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], b, acc, 0, 0, 0);
    }
  }
  if (wave > 0) red[wave - 1][lane] = acc;
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int k = 0; k < NW - 1; ++k) { const f32x4 r = red[k][lane]; acc[0] += r[0]; acc[1] += r[1]; acc[2] += r[2]; acc[3] += r[3]; }      // fixed order
    float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
    const int co = co0 + 4 * l4;                                   // D[row = 4 l4 + r][col = l15]: four consecutive channels of pixel l15
    if (valid && co < a.Cout) {
      if (a.flags & RCV_F_BIAS) { const float4 b = ld4(a.bias + co); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
      if (a.flags & RCV_F_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *reinterpret_cast<float4*>(a.out + (size_t)p * a.Cout + co) = v;
    }
  }
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
bool conv_small_supported(const rcv_handle* h, const rcv_op* op, int kind) {
  if (RCV_ENV("RCV_NO_CONV_SMALL")) return false;
  const int mode = op->i[RCV_I_INMODE], Cin = op->i[RCV_I_CIN], Cout = op->i[RCV_I_COUT];
  const long px = (long)op->i[RCV_I_N] * op->i[RCV_I_HO] * op->i[RCV_I_WO];
  // up to 16 pixel blocks per compute unit's worth of channel blocks: beyond that the tiled kernels have enough workgroups of their own
  return kind == KIND_GATHER && op->i[RCV_I_STATS] == RCV_STATS_NONE && !(op->flags & RCV_F_RESID) && op->i[RCV_I_AUX0] == 0 &&
         (mode == RCV_LOAD_PLAIN || mode == RCV_LOAD_AFFINE || mode == RCV_LOAD_AFFINE_RELU) && Cin % 4 == 0 && Cin >= 16 && Cin <= 128 &&
         Cout % 4 == 0 && Cout >= 16 && px * ceil_div(Cout, 16) <= (long)16 * 4 * h->num_cus && px <= 4096;
}

int conv_small_plan(const rcv_handle* h, const rcv_op* op, ConvPlan* pl) {
  const int Cout = op->i[RCV_I_COUT];
  const int px = op->i[RCV_I_N] * op->i[RCV_I_HO] * op->i[RCV_I_WO];
  pl->kind = KIND_GATHER; pl->narrow = 0; pl->dma = 0; pl->first = 0; pl->wino = 0; pl->small = 1;
  pl->CK = 4; pl->CoutV = Cout; pl->CoutP = round_up(Cout, 16);
  pl->R = 1; pl->Wt = 16; pl->tiles_x = ceil_div(px, 16); pl->tiles_y = 1;
  pl->n_co_tiles = pl->CoutP / 16; pl->n_phases = 1;
  pl->total_tiles = pl->tiles_x * pl->n_co_tiles;
  pl->grid = pl->total_tiles;
  pl->lds = 0;
  return RCV_OK;
}

int conv_small_launch(const ConvPlan& pl, const ConvArgs& a, hipStream_t s) {
  const int px = a.N * a.Ho * a.Wo;
  const dim3 block(CS_NW * 64);
  if (a.in_mode == RCV_LOAD_PLAIN) hipLaunchKernelGGL(conv_small_kernel<RCV_LOAD_PLAIN>, dim3(pl.grid), block, 0, s, a, pl.n_co_tiles, px);
  else if (a.in_mode == RCV_LOAD_AFFINE) hipLaunchKernelGGL(conv_small_kernel<RCV_LOAD_AFFINE>, dim3(pl.grid), block, 0, s, a, pl.n_co_tiles, px);
  else hipLaunchKernelGGL(conv_small_kernel<RCV_LOAD_AFFINE_RELU>, dim3(pl.grid), block, 0, s, a, pl.n_co_tiles, px);
  RCV_HIP(hipGetLastError());
  return RCV_OK;
}

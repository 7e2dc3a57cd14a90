// Narrow-layer (8 / 16 / 32 input channels) 3x3 convolutions with fp32 products formed on the bf16 matrix pipe.
//
// On the fp32 matrix instruction even the 8..32-channel layers of the step are matrix-pipe bound: 16 -> 16 at 32 x 240 x 320 moves 315 MB
// (52 us at 6 TB/s) but needs 72 us of v_mfma_f32_16x16x4_f32 at peak -- the fp32 kernel (convs_mfma.hip) measured 117 us with its matrix
// pipe 71 % busy.  With every operand split exactly into three bf16 values and six v_mfma_f32_16x16x32_bf16 products per multiply-add
// (wgrad_bf3.hip has the arithmetic and its error against fp64) the same contraction is 2.0-2.5 x cheaper and these layers become what
// their shapes say they are: HBM streams.
//
// Roles (those of convs_mfma.hip):  KIND_GATHER   conv / data gradient, stride 1 | 2, dilation 1;
//                                   KIND_TMERGED  stride-2 transposed conv as a 2 x 2-tap gather with 4 * Cout virtual output channels.
// GEMM:  D[cov][pixel] += W[cov][k] * X[k][pixel],  k = tap * CIN + ci flattened, 32 k per MFMA: with 8 input channels one MFMA covers
//   four taps, with 32 one tap.  Lane (i, g) holds k = 32 ks + 8 g .. + 7 = eight consecutive channels of ONE tap:
//   A = filter, split when packed (RCV_OP_PACK layouts 3 / 4: [plane][k-step][cov][32 k] bf16, zero beyond the last tap), copied once per
//       persistent workgroup into LDS; lane (cov, g) reads 16 bytes;
//   B = input tile in LDS as [pixel][plane h|m|l][CIN] bf16; lane (pixel, g) reads the 16 bytes at
//       pixel record + tap shift(ks, g) + channel offset(ks, g): one per-lane offset per k-step, computed once.
// Workgroup: persistent over pixel tiles of 64 * WN slots (all output channels); four producer waves stage tile i + 1 (global -> load
// transform -> split -> LDS) while four consumer waves contract tile i and run the epilogue of conv_mfma.hip (bias / ReLU / residual,
// 16-byte NHWC stores, BatchNorm sums kept in registers across tiles: one partial row per workgroup).
#include <type_traits>
#include "conv_common.h"
#include "conv_epilogue.h"

typedef __bf16 n3_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 n3_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int N3_NU = 11;                 // staging passes: tiles with halo up to 11 * (256 / (CIN / 4)) pixels

__device__ __forceinline__ uint32_t n3_pack(float a, float b) {
  const n3_bf16x2 v = {(__bf16)a, (__bf16)b};            // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(uint32_t, v);
}
struct N3Tri { uint32_t h, m, l; };
__device__ __forceinline__ N3Tri n3_split2(float x0, float x1) {
  N3Tri t;
  t.h = n3_pack(x0, x1);
  const float r0 = x0 - __uint_as_float(t.h << 16), r1 = x1 - __uint_as_float(t.h & 0xffff0000u);       // exact
  t.m = n3_pack(r0, r1);
  const float s0 = r0 - __uint_as_float(t.m << 16), s1 = r1 - __uint_as_float(t.m & 0xffff0000u);       // exact
  t.l = n3_pack(s0, s1);
  return t;
}

template <bool TWO>
struct N3Regs {
  float4 x[N3_NU], ax[TWO ? N3_NU : 1];
  bool ok[N3_NU];
};

template <int CIN, bool TWO>
__device__ __forceinline__ void n3_load(N3Regs<TWO>& r, const ConvArgs& a, const TileInfo& ti, int tid, int npix) {
  constexpr int Q = CIN / 4, PP = 256 / Q;
  const int q = tid % Q, lp = tid / Q;
#pragma unroll
  for (int u = 0; u < N3_NU; ++u) {
    const int pix = u * PP + lp;
    const int iy = fd_div(pix, a.fdIW), ix = pix - iy * a.IW;
    const int gy = ti.oy0 + iy, gx = ti.ox0 + ix;
    r.ok[u] = pix < npix && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    const uint32_t o = r.ok[u] ? (uint32_t)(((ti.n * a.H + gy) * a.W + gx) * CIN + 4 * q) : 0u;
    r.x[u] = ld4(a.in + o);
    if (TWO) r.ax[u] = ld4(a.in_aux + o);
  }
}
template <int MODE, int CIN, bool TWO>
__device__ __forceinline__ void n3_store(const N3Regs<TWO>& r, const ConvArgs& a, char* img, int tid, int npix) {
  constexpr int Q = CIN / 4, PP = 256 / Q, PITCH = 6 * CIN;
  const int q = tid % Q, lp = tid / Q;
  float4 k[5];
  if (MODE != RCV_LOAD_PLAIN) {
#pragma unroll
    for (int j = 0; j < 5; ++j) k[j] = ld4(a.in_c + j * CIN + 4 * q);
  }
#pragma unroll
  for (int u = 0; u < N3_NU; ++u) {
    const int pix = u * PP + lp;
    float4 v = xform4<MODE>(r.x[u], r.ax[TWO ? u : 0], k);
    if (!r.ok[u]) v = make_float4(0.f, 0.f, 0.f, 0.f);          // zero padding AFTER the transform
    if (pix < npix) {
      const N3Tri lo = n3_split2(v.x, v.y), hi = n3_split2(v.z, v.w);
      char* d = img + pix * PITCH + 8 * q;
      *reinterpret_cast<uint2*>(d) = make_uint2(lo.h, hi.h);
      *reinterpret_cast<uint2*>(d + 2 * CIN) = make_uint2(lo.m, hi.m);
      *reinterpret_cast<uint2*>(d + 4 * CIN) = make_uint2(lo.l, hi.l);
    }
  }
}

// Epilogue of a tile that lies entirely inside the plane, with everything that does not depend on the tile computed ONCE per (persistent)
// workgroup: the lane's output offsets relative to the tile's first output element, its bias and BatchNorm-backward constants.  The
// shared epilogue (conv_epilogue.h) redoes a division, two multiplies and a 64-bit offset per pixel block per tile -- on the HBM-bound
// layers the consumer waves' epilogue was the longest phase of a tile.  Ragged tiles at the plane's edges take the shared one.
template <int WM, int WN>
struct N3Epi {
  int off[WM][WN];            // element offset from the tile's first output element; < 0: the lane has no output element there
  int co[WM];                 // the lane's first real output channel of block m; < 0: beyond the layer's channels
};
template <int WM, int WN, int KIND>
__device__ __forceinline__ void n3_epi_setup(N3Epi<WM, WN>& E, const ConvArgs& a, int wave, int l15, int l4) {
#pragma unroll
  for (int m = 0; m < WM; ++m) {
    const int cov = m * 16 + 4 * l4;
    int co = cov, chan = cov;
    if (KIND == KIND_TMERGED) { const int ph = cov / a.Cout; co = cov - ph * a.Cout; chan = ((ph >> 1) * a.Wo + (ph & 1)) * a.Cout + co; }
    const bool co_ok = cov < a.CoutV;
    E.co[m] = co_ok ? co : -1;
#pragma unroll
    for (int b = 0; b < WN; ++b) {
      const int p = (wave * WN + b) * 16 + l15;
      const int ty = fd_div(p, a.fdWt), tx = p - ty * a.Wt;
      const int pix = KIND == KIND_TMERGED ? (2 * ty * a.Wo + 2 * tx) * a.Cout : (ty * a.Wo + tx) * a.Cout;
      E.off[m][b] = (ty < a.R && co_ok) ? pix + chan : -1;
    }
  }
}
template <int WM, int WN>
__device__ __forceinline__ void n3_epi_tile(const N3Epi<WM, WN>& E, const ConvArgs& a, size_t base, f32x4 (&acc)[WM][WN], float (&s1)[WM][4], float (&s2)[WM][4]) {
  float* __restrict__ out = a.out + base;
  const float* __restrict__ resid = a.resid + base;
  const float* __restrict__ eaux = a.epi_aux + base;
  const bool bwd_stats = a.stats == RCV_STATS_BWD_ENC || a.stats == RCV_STATS_BWD_DEC;
#pragma unroll
  for (int m = 0; m < WM; ++m) {
    float4 rr[WN], ee[WN];
    if (a.flags & RCV_F_RESID) {
#pragma unroll
      for (int b = 0; b < WN; ++b) rr[b] = ld4(resid + (E.off[m][b] < 0 ? 0 : E.off[m][b]));
    }
    if (bwd_stats) {
#pragma unroll
      for (int b = 0; b < WN; ++b) ee[b] = ld4(eaux + (E.off[m][b] < 0 ? 0 : E.off[m][b]));
    }
    // (per tile from L1: kept in registers across tiles they cost 16 registers per channel block and the 64-virtual-channel variants spilled)
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), e0 = bias, e1 = bias, mu = bias;
    if (E.co[m] >= 0) {
      if (a.flags & RCV_F_BIAS) bias = ld4(a.bias + E.co[m]);
      if (a.stats == RCV_STATS_BWD_DEC) { e0 = ld4(a.epi_c + E.co[m]); e1 = ld4(a.epi_c + a.Cout + E.co[m]); }
      if (bwd_stats) mu = ld4(a.epi_c + 2 * a.Cout + E.co[m]);
    }
#pragma unroll
    for (int b = 0; b < WN; ++b) {
      if (E.off[m][b] < 0) continue;
      float4 v = make_float4(acc[m][b][0] + bias.x, acc[m][b][1] + bias.y, acc[m][b][2] + bias.z, acc[m][b][3] + bias.w);
      if (a.flags & RCV_F_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (a.flags & RCV_F_RESID) { v.x += rr[b].x; v.y += rr[b].y; v.z += rr[b].z; v.w += rr[b].w; }
      *reinterpret_cast<float4*>(out + E.off[m][b]) = v;
      if (a.stats == RCV_STATS_FWD) {
        s1[m][0] += v.x; s1[m][1] += v.y; s1[m][2] += v.z; s1[m][3] += v.w;
        s2[m][0] = fmaf(v.x, v.x, s2[m][0]); s2[m][1] = fmaf(v.y, v.y, s2[m][1]);
        s2[m][2] = fmaf(v.z, v.z, s2[m][2]); s2[m][3] = fmaf(v.w, v.w, s2[m][3]);
      } else if (a.stats == RCV_STATS_BWD_ENC) {
        const float4 e = ee[b];
        s1[m][0] += v.x; s1[m][1] += v.y; s1[m][2] += v.z; s1[m][3] += v.w;
        s2[m][0] = fmaf(v.x, e.x - mu.x, s2[m][0]); s2[m][1] = fmaf(v.y, e.y - mu.y, s2[m][1]);
        s2[m][2] = fmaf(v.z, e.z - mu.z, s2[m][2]); s2[m][3] = fmaf(v.w, e.w - mu.w, s2[m][3]);
      } else if (a.stats == RCV_STATS_BWD_DEC) {
        const float4 e = ee[b];
        const float gx = fmaf(e.x, e0.x, e1.x) > 0.f ? v.x : 0.f;
        const float gy = fmaf(e.y, e0.y, e1.y) > 0.f ? v.y : 0.f;
        const float gz = fmaf(e.z, e0.z, e1.z) > 0.f ? v.z : 0.f;
        const float gw = fmaf(e.w, e0.w, e1.w) > 0.f ? v.w : 0.f;
        s1[m][0] += gx; s1[m][1] += gy; s1[m][2] += gz; s1[m][3] += gw;
        s2[m][0] = fmaf(gx, e.x - mu.x, s2[m][0]); s2[m][1] = fmaf(gy, e.y - mu.y, s2[m][1]);
        s2[m][2] = fmaf(gz, e.z - mu.z, s2[m][2]); s2[m][3] = fmaf(gw, e.w - mu.w, s2[m][3]);
      }
    }
  }
}

template <int CIN, int WM, int WN, int KIND, bool TWO>
__global__ __launch_bounds__(512) void convn_bf3_kernel(const ConvArgs a) {
  constexpr int TAPS = KIND == KIND_GATHER ? 9 : 4;
  constexpr int NKS = (TAPS * CIN + 31) / 32;
  constexpr int COT = 16 * WM;
  constexpr int PITCH = 6 * CIN;
  constexpr int WPLANE = NKS * COT * 64;                    // bytes of one filter plane
  constexpr int WBYTES = 3 * WPLANE;
  extern __shared__ __attribute__((aligned(16))) char smem_n3[];
  char* wl = smem_n3;
  const int xbytes = a.xl_floats * 4;
  char* xb0 = smem_n3 + WBYTES;
  float* red = reinterpret_cast<float*>(xb0 + 2 * xbytes);
  const int npix = a.IH * a.IW;
  const bool producer = threadIdx.x >= 256;
  const int tid = producer ? (int)threadIdx.x - 256 : (int)threadIdx.x;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);           // contiguous tile ranges per XCD at every iteration: halo neighbours share an L2

  // filter: global [plane][k-step][cov][32] -> LDS, verbatim
  for (int e = threadIdx.x; e < WBYTES / 16; e += 512)
    reinterpret_cast<float4*>(wl)[e] = reinterpret_cast<const float4*>(a.w)[e];

  if (producer) {
    auto stage = [&](int tile, char* buf) {
      const TileInfo ti = decode_tile<KIND>(a, tile, COT);
      N3Regs<TWO> r;
      n3_load<CIN, TWO>(r, a, ti, tid, npix);
      if (TWO) {
        if (a.in_mode == RCV_LOAD_GRAD_ENC) n3_store<RCV_LOAD_GRAD_ENC, CIN, TWO>(r, a, buf, tid, npix);
        else n3_store<RCV_LOAD_GRAD_DEC, CIN, TWO>(r, a, buf, tid, npix);
      } else {
        switch (a.in_mode) {
          case RCV_LOAD_PLAIN: n3_store<RCV_LOAD_PLAIN, CIN, TWO>(r, a, buf, tid, npix); break;
          case RCV_LOAD_AFFINE: n3_store<RCV_LOAD_AFFINE, CIN, TWO>(r, a, buf, tid, npix); break;
          default: n3_store<RCV_LOAD_AFFINE_RELU, CIN, TWO>(r, a, buf, tid, npix); break;
        }
      }
    };
    // barrier for barrier the consumer path: 1 + one per tile (+ the statistics reduction's).
    // (Round 3, measured and not kept: a second register set, so that the loads of tile i + 2 are issued before the data of tile i + 1 is
    // consumed -- 8 -> 16 stride 2 forward 0.127 -> 0.125 ms, 16 -> 8 transposed 0.164 -> 0.154: the compiler waits vmcnt(0) at the loop's
    // join points whatever is in flight.  Ablations (scripts/experiments/exp_r3_n3abl.sh) put the HBM-bound forms' time elsewhere: the
    // transposed form spends 90 of its 167 us in the epilogue (conv_epilogue.h stores 16-byte pieces 64 bytes apart, two lanes per
    // output pixel; convs_mfma.hip writes whole output rows per wave), the 8-channel stride-2 form 97 of 128 us in staging.)
    const bool do_stage = !(a.flags & RCV_F_DBG_NOSTAGE);           // (ablation timings: scripts/bench_op.py --flags)
    if (wg < a.total_tiles && do_stage) stage(wg, xb0);
    __syncthreads();
    int it = 0;
    for (int tile = wg; tile < a.total_tiles; tile += gridDim.x, ++it) {
      const int next = tile + gridDim.x;
      if (next < a.total_tiles && do_stage) stage(next, xb0 + ((it + 1) & 1) * xbytes);
      __syncthreads();
    }
    if (a.stats != RCV_STATS_NONE) __syncthreads();
    return;
  }

  // ---------------- consumer waves (4, side by side along the pixels) ----------------
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int s = KIND == KIND_GATHER ? a.stride : 1;
  f32x4 acc[WM][WN];
  float s1[WM][4], s2[WM][4];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[m][r] = 0.f; s2[m][r] = 0.f; }
  int pixoff[WN];
#pragma unroll
  for (int n = 0; n < WN; ++n) {
    const int p = (wave * WN + n) * 16 + l15;
    int ty = fd_div(p, a.fdWt), tx = p - ty * a.Wt;
    if (ty >= a.R) { ty = 0; tx = 0; }
    pixoff[n] = ((ty * s) * a.IW + tx * s) * PITCH;
  }
  // per k-step: tap shift and channel offset of this lane's eight k (k = 32 ks + 8 g: one tap, eight consecutive channels)
  int koff[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int k0 = 32 * ks + 8 * l4;
    int tap = k0 / CIN, ci0 = k0 % CIN;
    if (tap >= TAPS) { tap = 0; ci0 = 0; }                  // beyond the last tap the filter is zero: read any finite data
    const int shift = KIND == KIND_GATHER ? (tap / 3) * a.IW + (tap % 3) : (tap >> 1) * a.IW + (tap & 1);
    koff[ks] = shift * PITCH + 2 * ci0;
  }
  const char* abase = wl + l15 * 64 + l4 * 16;
  constexpr bool FAST_EPI = WM <= 2;       // (the 64-virtual-channel variants have no registers left for the offset table: 45 spills)
  N3Epi<FAST_EPI ? WM : 1, FAST_EPI ? WN : 1> E;
  if constexpr (FAST_EPI) n3_epi_setup<WM, WN, KIND>(E, a, wave, l15, l4);
  const int lim_h = KIND == KIND_GATHER ? a.Ho : a.H, lim_w = KIND == KIND_GATHER ? a.Wo : a.W;

  __syncthreads();
  int it = 0;
  for (int tile = wg; tile < a.total_tiles; tile += gridDim.x, ++it) {
    const char* xb = xb0 + (it & 1) * xbytes;
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    n3_bf16x8 A[2][WM][3], B[2][3];
    auto load_a = [&](int ks, n3_bf16x8 (&dst)[WM][3]) {
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) dst[m][pl] = *reinterpret_cast<const n3_bf16x8*>(abase + pl * WPLANE + (ks * COT + m * 16) * 64);
    };
    auto load_b = [&](int ks, int n, n3_bf16x8 (&dst)[3]) {
      const char* pb = xb + pixoff[n] + koff[ks];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dst[pl] = *reinterpret_cast<const n3_bf16x8*>(pb + pl * 2 * CIN);
    };
    load_a(0, A[0]);
    load_b(0, 0, B[0]);
    if (!(a.flags & RCV_F_DBG_NOMFMA))
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      if (ks + 1 < NKS) load_a(ks + 1, A[(ks + 1) & 1]);
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        const int i = ks * WN + n;                           // step counter: B double buffer
        if (n + 1 < WN) load_b(ks, n + 1, B[(i + 1) & 1]);
        else if (ks + 1 < NKS) load_b(ks + 1, 0, B[(i + 1) & 1]);
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};     // smallest products first
#pragma unroll
        for (int e = 0; e < 6; ++e)
#pragma unroll
          for (int m = 0; m < WM; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ks & 1][m][TA[e]], B[i & 1][TB[e]], acc[m][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const TileInfo ti = decode_tile<KIND>(a, tile, COT);
    // (EB = WN: the residual / BatchNorm-backward operands of all pixel blocks of a channel block are requested in ONE batch; loaded where
    // they are used they are 2 * WN serialized HBM round trips per tile, and nothing hides them here)
    if (!(a.flags & RCV_F_DBG_NOEPI)) {
      bool fast = false;
      if constexpr (FAST_EPI) {
        if (ti.y0 + a.R <= lim_h && ti.x0 + a.Wt <= lim_w) {       // (uniform) the tile lies inside the plane: offsets precomputed
          const int ym = KIND == KIND_TMERGED ? 2 : 1;
          n3_epi_tile<WM, WN>(E, a, ((size_t)(ti.n * a.Ho + ym * ti.y0) * a.Wo + ym * ti.x0) * a.Cout, acc, s1, s2);
          fast = true;
        }
      }
      if (!fast) conv_epilogue_tile<WM, WN, 1, 4, KIND, WN>(a, ti, acc, s1, s2, tid);
    }
    __syncthreads();
  }
  if (a.stats != RCV_STATS_NONE) conv_epilogue_stats<WM, 1, 4, KIND>(a, (size_t)blockIdx.x, 0, s1, s2, red, tid);
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
struct N3Geom { int WN, R, Wt, IH, IW, tiles_x, tiles_y, total, grid, xl_bytes; size_t lds; };

static inline int n3_nks(int kind, int Cin) { return ((kind == KIND_GATHER ? 9 : 4) * Cin + 31) / 32; }

static bool n3_shape_ok(const rcv_op* op, int kind) {
  const int Cin = op->i[RCV_I_CIN], Cout = op->i[RCV_I_COUT];
  if (!(Cin == 8 || Cin == 16 || Cin == 32) || Cout % 4) return false;
  if (op->i[RCV_I_INMODE] == RCV_LOAD_NCHW || op->i[RCV_I_DIL] != 1) return false;
  const int CoutP = round_up(kind == KIND_TMERGED ? 4 * Cout : Cout, 16);
  if (kind == KIND_GATHER) return (CoutP == 16 || CoutP == 32) && (op->i[RCV_I_STRIDE] == 1 || op->i[RCV_I_STRIDE] == 2);
  if (kind == KIND_TMERGED) return (Cin == 16 && CoutP == 32) || (Cin == 32 && CoutP == 64);
  return false;
}

static bool n3_geometry(const rcv_handle* h, const rcv_op* op, int kind, N3Geom* g) {
  const int N = op->i[RCV_I_N], Cin = op->i[RCV_I_CIN], Cout = op->i[RCV_I_COUT], s = op->i[RCV_I_STRIDE];
  const int TH = kind == KIND_GATHER ? op->i[RCV_I_HO] : op->i[RCV_I_H], TW = kind == KIND_GATHER ? op->i[RCV_I_WO] : op->i[RCV_I_W];
  const int CoutP = round_up(kind == KIND_TMERGED ? 4 * Cout : Cout, 16);
  const size_t wbytes = 3 * (size_t)n3_nks(kind, Cin) * CoutP * 64;
  const int per_pass = 256 / (Cin / 4);
  double best = 1e30;
  bool found = false;
  for (int WN : {5, 3}) {
    const int slots = 64 * WN;
    for (int nx = 1; nx <= TW; ++nx) {
      const int wt = ceil_div(TW, nx);
      if (wt > slots) continue;
      if (nx > 1 && wt < 8) break;
      int r = slots / wt < TH ? slots / wt : TH;
      for (; r >= 1; --r) {
        const int rb = ceil_div(TH, ceil_div(TH, r));
        int ih, iw;
        tile_halo(kind, rb, wt, s, 1, &ih, &iw);
        if (ih * iw > N3_NU * per_pass) continue;
        const int xl = round_up(ih * iw * 6 * Cin, 16);
        const size_t lds = wbytes + 2 * (size_t)xl + (size_t)4 * 2 * CoutP * sizeof(float);
        if (lds > (size_t)h->max_lds) continue;
        const long tiles = (long)N * ceil_div(TW, wt) * ceil_div(TH, rb);
        // per tile: the HBM stream of its pixels (with halo) + the MFMA slots + a fixed part
        const double halo = (double)ih * iw / ((double)rb * wt * (kind == KIND_GATHER ? s * s : 1));
        const double cost = (double)tiles * ((double)rb * wt * (1.0 + 0.5 * halo) + 0.35 * slots + 40.0);
        if (cost < best) {
          best = cost; found = true;
          g->WN = WN; g->R = rb; g->Wt = wt; g->IH = ih; g->IW = iw; g->tiles_x = ceil_div(TW, wt); g->tiles_y = ceil_div(TH, rb);
          g->total = (int)tiles; g->xl_bytes = xl; g->lds = lds;
        }
        break;
      }
    }
  }
  if (found) g->grid = g->total < h->num_cus ? g->total : h->num_cus;
  return found;
}

// Would this record run here if its filter were packed in the split layout?  (rcv_op_filter_layout: 3 for a conv, 4 for a merged transposed conv)
bool convn_bf3_wanted(const rcv_handle* h, const rcv_op* op) {
  if (RCV_ENV("RCV_NO_BF3") || RCV_ENV("RCV_NO_BF3N") || (op->flags & RCV_F_MFMA_FP32)) return false;
  const int kind = op->kind == RCV_OP_CONV ? KIND_GATHER : (op->kind == RCV_OP_TCONV ? KIND_TMERGED : -1);
  if (kind < 0 || !n3_shape_ok(op, kind)) return false;
  if ((long long)op->i[RCV_I_N] * op->i[RCV_I_H] * op->i[RCV_I_W] * op->i[RCV_I_CIN] >= (1ll << 31)) return false;
  N3Geom g;
  if (!n3_geometry(h, op, kind, &g)) return false;
  if ((long)g.total * 2 < (long)h->num_cus) return false;   // small planes stay on the other kernels
  if (RCV_ENV("RCV_BF3N_ALL")) return true;
  if (RCV_ENV("RCV_BF3N_FWD")) {       // experiments build: forward launches only
    const int m0 = op->i[RCV_I_INMODE];
    if (m0 == RCV_LOAD_GRAD_ENC || m0 == RCV_LOAD_GRAD_DEC) return false;
  }
  // Where this kernel is the faster one (op by op at the shapes of the 640 x 480 step, scripts/experiments/exp_r3_n3ops.sh): the layers
  // whose fp32 form is matrix-pipe bound -- 32 -> 32 (133 -> 86 us forward, 157 -> 107 data gradient), 16 -> 16 (136 -> 125, 186 -> 170),
  // the 32 -> 16 transposed conv with a two-tensor input (202 -> 178).  The HBM-bound forms (8 -> 16 stride 2, the forward transposed
  // convs: 102 -> 127, 115 -> 164 us) stream better through convs_mfma.hip, whose two workgroups per CU keep more loads in flight.
  const int Cin = op->i[RCV_I_CIN], m = op->i[RCV_I_INMODE];
  const bool two = m == RCV_LOAD_GRAD_ENC || m == RCV_LOAD_GRAD_DEC;
  if (kind == KIND_GATHER) return Cin == 32 || (Cin == 16 && op->i[RCV_I_STRIDE] == 1);
  return Cin == 32 && two;
}

bool convn_bf3_supported(const rcv_handle* h, const rcv_op* op, int kind) {
  const int aux = op->i[RCV_I_AUX0];
  return ((kind == KIND_GATHER && aux == 3) || (kind == KIND_TMERGED && aux == 4)) && op->i[RCV_I_CIN] <= 32;
}

int convn_bf3_plan(const rcv_handle* h, const rcv_op* op, int kind, ConvPlan* pl) {
  RCV_CHECK_ARG(n3_shape_ok(op, kind) && !(op->flags & RCV_F_MFMA_FP32), "split-bf16 narrow conv: shape %d -> %d (stride %d, dilation %d) not built",
                op->i[RCV_I_CIN], op->i[RCV_I_COUT], op->i[RCV_I_STRIDE], op->i[RCV_I_DIL]);
  N3Geom g;
  RCV_CHECK_ARG(n3_geometry(h, op, kind, &g), "split-bf16 narrow conv: no tile fits");
  const int Cout = op->i[RCV_I_COUT];
  pl->kind = kind; pl->narrow = 0; pl->dma = 0; pl->first = 0; pl->wino = 0; pl->small = 0; pl->bf3 = 2;
  pl->CK = op->i[RCV_I_CIN];
  pl->CoutV = kind == KIND_TMERGED ? 4 * Cout : Cout; pl->CoutP = round_up(pl->CoutV, 16);
  pl->WM = pl->CoutP / 16; pl->WN = g.WN;
  pl->R = g.R; pl->Wt = g.Wt; pl->IH = g.IH; pl->IW = g.IW; pl->tiles_x = g.tiles_x; pl->tiles_y = g.tiles_y;
  pl->n_co_tiles = 1; pl->n_phases = 1; pl->total_tiles = g.total; pl->grid = g.grid;
  pl->xl_floats = g.xl_bytes / 4; pl->wl_floats = 0; pl->lds = g.lds;
  return RCV_OK;
}

template <int CIN, int WM, int WN, int KIND, bool TWO>
static int n3_launch_inst(const ConvPlan& pl, const ConvArgs& a, hipStream_t s) {
  auto kern = convn_bf3_kernel<CIN, WM, WN, KIND, TWO>;
  static size_t configured[RCV_MAX_DEVICES];
  RCV_ENSURE_LDS(kern, pl.lds, pl.dev, configured);
  hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(512), pl.lds, s, a);
  RCV_HIP(hipGetLastError());
  return RCV_OK;
}
template <int CIN, int WM, int KIND>
static int n3_launch_wn(const ConvPlan& pl, const ConvArgs& a, bool two, hipStream_t s) {
  if (pl.WN == 5) return two ? n3_launch_inst<CIN, WM, 5, KIND, true>(pl, a, s) : n3_launch_inst<CIN, WM, 5, KIND, false>(pl, a, s);
  return two ? n3_launch_inst<CIN, WM, 3, KIND, true>(pl, a, s) : n3_launch_inst<CIN, WM, 3, KIND, false>(pl, a, s);
}

int convn_bf3_launch(const ConvPlan& pl, const ConvArgs& a, hipStream_t s) {
  const bool two = a.in_mode == RCV_LOAD_GRAD_ENC || a.in_mode == RCV_LOAD_GRAD_DEC;
  if (pl.kind == KIND_TMERGED) {
    if (a.Cin == 16) return n3_launch_wn<16, 2, KIND_TMERGED>(pl, a, two, s);
    return n3_launch_wn<32, 4, KIND_TMERGED>(pl, a, two, s);
  }
  switch (a.Cin * 10 + pl.WM) {
    case 81: return n3_launch_wn<8, 1, KIND_GATHER>(pl, a, two, s);
    case 82: return n3_launch_wn<8, 2, KIND_GATHER>(pl, a, two, s);
    case 161: return n3_launch_wn<16, 1, KIND_GATHER>(pl, a, two, s);
    case 162: return n3_launch_wn<16, 2, KIND_GATHER>(pl, a, two, s);
    case 321: return n3_launch_wn<32, 1, KIND_GATHER>(pl, a, two, s);
    default: return n3_launch_wn<32, 2, KIND_GATHER>(pl, a, two, s);
  }
}

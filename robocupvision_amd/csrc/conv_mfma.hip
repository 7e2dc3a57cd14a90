// 3x3 convolution as implicit GEMM on the gfx950 fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// One kernel family serves these roles of the ROBO-UNet step (reference call sites in include/rcv.h):
//   KIND_GATHER   (RCV_OP_CONV):  forward of Conv2d(k3, pad=dil, stride 1|2, dilation 1|2),
//                                 data gradient of a stride-1 conv (flipped/transposed filter),
//                                 data gradient of the stride-2 transposed conv;
//   KIND_TPHASE   (RCV_OP_TCONV): stride-2 transposed conv (ConvTranspose2d(k3,s2,p1,op1) forward, data
//                                 gradient of a stride-2 conv) as four output-parity phases with 1/2/2/4
//                                 taps -- no multiplications by inserted zeros (wide layers, MFMA bound);
//   KIND_TMERGED  (RCV_OP_TCONV): the same operator for narrow layers (Cout <= 32, HBM bound): a 2x2-tap
//                                 gather over the input with 4*Cout "virtual" output channels (one group
//                                 of Cout per output parity; 7 of the 16 tap/parity filter slices are
//                                 zero).  Every input pixel yields its whole 2x2 output block, so each
//                                 wave writes full contiguous output rows instead of every other pixel.
//
// GEMM mapping per workgroup:  D[co][pixel] += W[co][k] * X[k][pixel],  k = (tap, ci)
//   A operand  = packed filter [tap][ci][co] staged in LDS as [tap][ci][COT+16]  (co contiguous)
//   B operand  = input tile staged in LDS as [pixel][CK+1] (odd pixel pitch: the 16 pixels of an MFMA
//                block spread over the banks; the 4 k-lanes read 4 consecutive channels)
//   MFMA 16x16x4 lane map: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=4*(l>>4)+r][col=l&15]
//                => every lane ends with 4 consecutive output channels of one pixel: one 16-byte store.
// A wave owns WM co-blocks x WN pixel-blocks; a workgroup is WAVES_M x WAVES_N waves.
//
// Schedule: one (tile, co-tile[, parity]) per workgroup; several workgroups are resident per CU so that one
// stages (global -> registers -> LDS, 4 independent 16-byte loads per thread in flight) while others contract.
//
// While a tile is written to LDS the producer's BatchNorm (or the BN/ReLU backward of the consumer) is
// applied (RCV_LOAD_*), so normalised activations / pre-activation gradients never exist in HBM.  The
// epilogue adds bias / skip gradient, applies ReLU, stores, and emits the per-tile partial sums the
// following BatchNorm (forward or backward) needs -- fixed order, no atomics.
#include <stdlib.h>
#include <type_traits>
#include "conv_common.h"
#include "conv_epilogue.h"

// Stage the [IH*IW][CK] input tile of channel chunk c0 into xl ([pixel][CK+1]); UNR independent 16-byte
// loads per thread are issued before any is consumed (one load in flight per thread is latency bound).
template <int MODE, int CK, int NT>
__device__ __forceinline__ void stage_input(const ConvArgs& a, float* xl, const float* cl, const TileInfo& ti, int c0, int tid) {
  const int S = a.xpitch;
  constexpr int Q = CK / 4;
  constexpr int STEP = NT / Q;
  constexpr int UNR = 4;
  const int npix = a.IH * a.IW;
  if (MODE == RCV_LOAD_NCHW) {
    for (int pix0 = tid; pix0 < npix; pix0 += UNR * NT) {
      float v[UNR][CK];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int pix = pix0 + u * NT;
        const int iy = fd_div(pix, a.fdIW), ix = pix - iy * a.IW;
        const int gy = ti.oy0 + iy, gx = ti.ox0 + ix;
        const bool ok = pix < npix && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
#pragma unroll
        for (int j = 0; j < CK; ++j) {
          v[u][j] = 0.f;
          if (ok && c0 + j < a.Cin) v[u][j] = a.in[((size_t)(ti.n * a.Cin + c0 + j) * a.H + gy) * a.W + gx];
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int pix = pix0 + u * NT;
        if (pix < npix) {
#pragma unroll
          for (int j = 0; j < CK; ++j) xl[pix * S + j] = v[u][j];
        }
      }
    }
    return;
  }
  const int q = tid % Q;
  const int ch = c0 + 4 * q;
  float4 k[5];
  if (MODE != RCV_LOAD_PLAIN) {
#pragma unroll
    for (int j = 0; j < 5; ++j) k[j] = *reinterpret_cast<const float4*>(cl + j * a.Cin + ch);
  }
  for (int pix0 = tid / Q; pix0 < npix; pix0 += UNR * STEP) {
    float4 x[UNR], aux[UNR];
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int pix = pix0 + u * STEP;
      const int iy = fd_div(pix, a.fdIW), ix = pix - iy * a.IW;
      const int gy = ti.oy0 + iy, gx = ti.ox0 + ix;
      ok[u] = pix < npix && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      aux[u] = x[u];
      if (ok[u]) {
        const size_t off = ((size_t)(ti.n * a.H + gy) * a.W + gx) * a.Cin + ch;
        x[u] = ld4(a.in + off);
        if (MODE == RCV_LOAD_GRAD_ENC || MODE == RCV_LOAD_GRAD_DEC) aux[u] = ld4(a.in_aux + off);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int pix = pix0 + u * STEP;
      if (pix < npix) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok[u]) v = xform4<MODE>(x[u], aux[u], k);   // zero padding AFTER the transform (padded taps are 0)
        float* d = xl + pix * S + 4 * q;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    }
  }
}

template <int WM, int WN, int WAVES_M, int WAVES_N, int CK, int KIND>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void conv_mfma_kernel(const ConvArgs a) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int COT = WM * WAVES_M * 16;
  const int S = a.xpitch;
  constexpr int WS = COT + 16;
  constexpr int C4 = COT / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;
  float* xl = smem + a.wl_floats;
  float* cl = xl + a.xl_floats;            // [5][Cin] load constants
  float* red = cl + 5 * a.CinP + 16;       // [WAVES_N][2][COT]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
  const int l15 = lane & 15, l4 = lane >> 4;

  if (a.in_c && a.in_mode != RCV_LOAD_PLAIN && a.in_mode != RCV_LOAD_NCHW)
    for (int e = tid; e < 5 * a.Cin; e += NT) cl[e] = a.in_c[e];

  const TileInfo ti = decode_tile<KIND>(a, xcd_remap(blockIdx.x, a.total_tiles), COT);
  int nxt = 3, ntaps = 9;
  if (KIND == KIND_TPHASE) { nxt = 1 + ti.px; ntaps = (1 + ti.py) * nxt; }
  if (KIND == KIND_TMERGED) { nxt = 2; ntaps = 4; }
  const int IS = KIND == KIND_GATHER ? a.stride : 1;

  int pixoff[WN];
#pragma unroll
  for (int b = 0; b < WN; ++b) {
    const int p = (wave_n * WN + b) * 16 + l15;
    int ty = fd_div(p, a.fdWt), tx = p - ty * a.Wt;
    if (ty >= a.R) { ty = 0; tx = 0; }
    pixoff[b] = ((ty * IS) * a.IW + tx * IS) * S + l4;
  }
  const int aoff = l4 * WS + (wave_m * WM) * 16 + l15;

  f32x4 acc[WM][WN];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int b = 0; b < WN; ++b) acc[m][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int c0 = 0; c0 < a.CinP; c0 += CK) {
    __syncthreads();
    // ---- stage filter chunk: wl[j][ck][co]
    {
      const int total = ntaps * CK * C4;
      constexpr int WU = 4;
      for (int e0 = tid; e0 < total; e0 += WU * NT) {
        float4 v[WU];
#pragma unroll
        for (int u = 0; u < WU; ++u) {
          const int e = e0 + u * NT;
          v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (e < total) {
            const int row = e / C4, c4 = e % C4;
            const int j = row / CK, ck = row % CK;
            int t9 = j;
            if (KIND == KIND_TPHASE) {
              const int jy = j / nxt, jx = j - jy * nxt;
              const int ky = ti.py ? (jy ? 2 : 0) : 1;
              const int kx = ti.px ? (jx ? 2 : 0) : 1;
              t9 = ky * 3 + kx;
            }
            const int co = ti.co0 + 4 * c4;
            if (co < a.CoutP) v[u] = ld4(a.w + ((size_t)(t9 * a.CinP + c0 + ck) * a.CoutP + co));
          }
        }
#pragma unroll
        for (int u = 0; u < WU; ++u) {
          const int e = e0 + u * NT;
          if (e < total) {
            const int row = e / C4, c4 = e % C4;
            *reinterpret_cast<float4*>(wl + row * WS + 4 * c4) = v[u];
          }
        }
      }
    }
    // ---- stage input chunk
    if (!(a.flags & RCV_F_DBG_NOSTAGE)) {
      switch (a.in_mode) {
        case RCV_LOAD_PLAIN: stage_input<RCV_LOAD_PLAIN, CK, NT>(a, xl, cl, ti, c0, tid); break;
        case RCV_LOAD_AFFINE: stage_input<RCV_LOAD_AFFINE, CK, NT>(a, xl, cl, ti, c0, tid); break;
        case RCV_LOAD_AFFINE_RELU: stage_input<RCV_LOAD_AFFINE_RELU, CK, NT>(a, xl, cl, ti, c0, tid); break;
        case RCV_LOAD_GRAD_ENC: stage_input<RCV_LOAD_GRAD_ENC, CK, NT>(a, xl, cl, ti, c0, tid); break;
        case RCV_LOAD_GRAD_DEC: stage_input<RCV_LOAD_GRAD_DEC, CK, NT>(a, xl, cl, ti, c0, tid); break;
        default: stage_input<RCV_LOAD_NCHW, CK, NT>(a, xl, cl, ti, c0, tid); break;
      }
    }
    __syncthreads();
    if (a.flags & RCV_F_DBG_NOMFMA) continue;
    // ---- contraction over (tap, ci in chunk)
    auto tap = [&](int j, int dy, int dx) {
      const float* wj = wl + j * CK * WS + aoff;
      const float* xj = xl + (dy * a.IW + dx) * S;
#pragma unroll
      for (int kk = 0; kk < CK / 4; ++kk) {
        float av[WM], bv[WN];
#pragma unroll
        for (int m = 0; m < WM; ++m) av[m] = wj[kk * 4 * WS + m * 16];
#pragma unroll
        for (int b = 0; b < WN; ++b) bv[b] = xj[pixoff[b] + kk * 4];
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
          for (int b = 0; b < WN; ++b)
            acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[b], acc[m][b], 0, 0, 0);
      }
    };
    if (KIND == KIND_GATHER) {
#pragma unroll
      for (int j = 0; j < 9; ++j) tap(j, (j / 3) * a.dil, (j % 3) * a.dil);     // compile-time trip count: reads pipeline under MFMAs
    } else {
      for (int j = 0; j < ntaps; ++j) {
        const int jy = j / nxt, jx = j - jy * nxt;
        if (KIND == KIND_TPHASE) tap(j, ti.py ? (jy ? 0 : 1) : 0, ti.px ? (jx ? 0 : 1) : 0);
        else tap(j, jy, jx);
      }
    }
  }

  conv_epilogue<WM, WN, WAVES_M, WAVES_N, KIND>(a, ti, acc, red, tid);
}

// ---------------------------------------------------------------------------------------------------------
// Wide layers (COT >= 64): the filter chunk is ~90 % of the staged bytes and needs no transform, so it goes
// global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, asynchronous) into a DOUBLE buffer while
// the previous chunk is being contracted; the small input chunk rides along in a few registers.  One barrier
// per chunk, and the L2 latency of the filter stream is hidden behind ~90 MFMAs per wave.
//   LDS: wl[2][taps*4][COT] (lane-linear, as the DMA writes it), xl[2][IH*IW][5], constants, reduction scratch.
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* rcv_lds_ptr;
typedef const __attribute__((address_space(1))) void* rcv_glb_ptr;

#ifdef RCV_STAMPS
// diagnostic build only (make STAMPS=1): shader-clock stamps around the segments of the pipelined loop, summed per wave
#define RCV_STAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RCV_STAMP(t) do { } while (0)
#endif

template <int WM, int WN, int WAVES_M, int WAVES_N, int KIND, bool TWO, int XMAX = 4>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void conv_dma_kernel(const ConvArgs a) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int NW = NT / 64;
  constexpr int COT = WM * WAVES_M * 16;
  constexpr int CK = 4;                         // channels per pipeline step (one k-step of the MFMA)
  const int S = a.xpitch;
  constexpr int WS = COT;                       // unpadded rows: the DMA image is lane-linear
  constexpr int C4 = COT / 4;
  constexpr int AMAX = TWO ? XMAX : 1;          // XMAX: register slots of the staged input chunk ((pixel, quad) items per thread): 4, or 8 for the large stride-2 tiles
  constexpr int NTAPS_MAX = (KIND == KIND_GATHER || KIND == KIND_TALL) ? 9 : 4;
  constexpr int NPH = KIND == KIND_TALL ? 4 : 1;  // accumulator sets (output parities handled by this workgroup)
  constexpr int WBUF = NTAPS_MAX * CK * WS;     // floats per filter buffer
  constexpr int WU = (NTAPS_MAX * CK * C4 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;                             // [2][WBUF]
  float* xl = smem + 2 * WBUF;                  // [2][xl_floats]
  float* cl = xl + 2 * a.xl_floats;             // [5][Cin]
  float* red = cl + 5 * a.CinP + 16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int npix = a.IH * a.IW;
  // The input tile is staged XK (8 or 16) channels at a time, i.e. once per XSTEPS pipeline steps: a thread owns one 16-byte quad of
  // one pixel per slot, so a pixel's XK channels are one contiguous 32/64-byte piece of its NHWC record and the address arithmetic,
  // the load transform and its constants are paid once per XK/4 steps instead of every step.
  const int XK = a.xk, xq_shift = XK == 16 ? 2 : 1, XQ = 1 << xq_shift, XSTEPS = XK / CK;
  const int xtotal = npix << xq_shift;          // (pixel, quad) items of one input chunk; <= XMAX * NT, or <= 2 * XMAX * NT with XK == 16 (host)
  // A 16-channel chunk spans four pipeline steps, so its items may be staged in TWO phases that reuse the same register slots: phase 0
  // (items [0, XMAX*NT)) is loaded in the group's first step and written in its second, phase 1 (the rest) loaded in the third and
  // written in the fourth.  The large stride-2 tiles (5 x 81 input pixels) then read whole 64-byte halves of every pixel record in
  // two passes instead of 32-byte quarters in four: with two input tensors the four passes of the co-resident workgroups did not
  // survive in the 4 MiB L2 (round 2: 524 MB fetched per launch of the 32 -> 64 data gradient against 196 MB algorithmic).
  const bool two_phase = xtotal > XMAX * NT;

  const TileInfo ti = decode_tile<KIND>(a, xcd_remap(blockIdx.x, a.total_tiles), COT);
  int nxt = 3, ntaps = 9;
  if (KIND == KIND_TPHASE) { nxt = 1 + ti.px; ntaps = (1 + ti.py) * nxt; }
  if (KIND == KIND_TMERGED) { nxt = 2; ntaps = 4; }
  if (KIND == KIND_TALL) { nxt = 3; ntaps = 9; }
  const int IS = KIND == KIND_GATHER ? a.stride : 1;
  const int wtotal = ntaps * CK * C4;           // 16-byte pieces of one filter chunk (a multiple of 64 for COT >= 64)

  // ---- filter stream: per wave-instruction u one lane-constant source offset (floats, chunk 0) and one LDS destination.  The image
  // is lane-linear ([tap][k][COT], as the DMA writes it) but the SOURCE is free per lane: rows with odd k take their 16-channel
  // blocks pairwise swapped (quad index ^ 4), and the A-operand read below applies the same swap.  The rows k and k+1 that the two
  // k-lanes of a 32-lane LDS access group read then sit 16 banks apart instead of on the same 16 banks (2-way conflict on every A read).
  int woff[WU];
  int wdst[WU];
#pragma unroll
  for (int u = 0; u < WU; ++u) {
    int e0 = (u * NW + __builtin_amdgcn_readfirstlane(wave)) * 64;   // first piece of this wave-instruction (a scalar)
    if (e0 + 64 > wtotal) e0 = wtotal - 64;                 // past the end: repeat the last pieces (same bytes) -- every wave issues exactly WU
    const int e = e0 + lane;                                // instructions per step, branch-free
    const int row = e / C4, c4 = (e % C4) ^ ((row & 1) << 2);
    const int j = row / CK, ck = row % CK;
    int t9 = j;
    if (KIND == KIND_TPHASE) {
      const int jy = j / nxt, jx = j - jy * nxt;
      const int ky = ti.py ? (jy ? 2 : 0) : 1;
      const int kx = ti.px ? (jx ? 2 : 0) : 1;
      t9 = ky * 3 + kx;
    }
    woff[u] = (t9 * a.CinP + ck) * a.CoutP + ti.co0 + 4 * c4;
    wdst[u] = e0 * 4;
  }
  auto dma_one = [&](int buf, int step, int u) {
    const float* wsrc = a.w + (size_t)step * CK * a.CoutP;
    float* dst = wl + buf * WBUF + wdst[u];                 // + lane*16 B is added by the hardware
    __builtin_amdgcn_global_load_lds((rcv_glb_ptr)(wsrc + woff[u]), (rcv_lds_ptr)dst, 16, 0, 0);
  };
  auto dma_w = [&](int buf, int step) {
#pragma unroll
    for (int u = 0; u < WU; ++u) dma_one(buf, step, u);
  };

  // ---- input stream: registers -> (load transform) -> xl[g & 1] as [pixel][XK + pad]
  float4 px[XMAX], pa[AMAX];
  float4 kx[5];                                  // load constants of this thread's quad for the chunk in px (requested with it)
  int xsrc[2 * XMAX];                            // element offset of (pixel, quad) in the NHWC tensor, channel chunk 0; < 0: outside the plane
#pragma unroll
  for (int u = 0; u < 2 * XMAX; ++u) {
    const int idx = tid + u * NT;
    const int pix = idx >> xq_shift, q = idx & (XQ - 1);
    const int iy = fd_div(pix, a.fdIW), ix = pix - iy * a.IW;
    const int gy = ti.oy0 + iy, gx = ti.ox0 + ix;
    const bool ok = idx < xtotal && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    xsrc[u] = ok ? (((ti.n * a.H + gy) * a.W + gx) * a.Cin + 4 * q) : -1;      // (host: N*H*W*Cin < 2^31)
  }
  auto load_slot = [&](int g, int u, bool ph1 = false) {
    if ((u + (ph1 ? XMAX : 0)) * NT < xtotal) {  // workgroup uniform: whole slots without items are skipped
      // branch-free inside (an invalid item reads the chunk of element 0 and is masked in write_x)
      const int src = ph1 ? xsrc[XMAX + u] : xsrc[u];
      const size_t off = (size_t)(src >= 0 ? src : 0) + g * XK;
      px[u] = ld4(a.in + off);
      if (TWO) pa[TWO ? u : 0] = ld4(a.in_aux + off);
    }
  };
  const int nkx = a.in_mode == RCV_LOAD_PLAIN ? 0 : ((a.in_mode == RCV_LOAD_AFFINE || a.in_mode == RCV_LOAD_AFFINE_RELU) ? 2 : (a.in_mode == RCV_LOAD_GRAD_ENC ? 3 : 5));
  auto load_consts = [&](int g) {
    // straight from the (L2-resident) constant rows, not through an LDS copy: an LDS read behind an outstanding LDS-DMA request
    // makes the compiler wait for that request (possible alias), i.e. for the filter slab requested a tap earlier
#pragma unroll
    for (int j = 0; j < 5; ++j)
      if (j < nkx) kx[j] = ld4(a.in_c + j * a.Cin + g * XK + 4 * (tid & (XQ - 1)));
  };
  auto load_x = [&](int g, bool ph1 = false) {
#pragma unroll
    for (int u = 0; u < XMAX; ++u) load_slot(g, u, ph1);
    if (!ph1) load_consts(g);                    // the constants of the chunk stay in kx for its second phase
  };
  auto write_x_mode = [&](auto mode_c, int buf, int g, bool ph1) {
    constexpr int MODE = decltype(mode_c)::value;
    float* xb = xl + buf * a.xl_floats;
    const int q = tid & (XQ - 1);                // NT is a multiple of XQ: the quad of a thread is the same in every slot
    float4 k[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) k[j] = kx[j];
#pragma unroll
    for (int u = 0; u < XMAX; ++u) {
      const int v_ = u + (ph1 ? XMAX : 0);
      if (v_ * NT < xtotal) {
        const int idx = tid + v_ * NT;
        if (idx < xtotal) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if ((ph1 ? xsrc[XMAX + u] : xsrc[u]) >= 0) v = xform4<MODE>(px[u], pa[TWO ? u : 0], k);   // zero padding AFTER the transform
          float* d = xb + (idx >> xq_shift) * S + 4 * q;
          d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
      }
    }
  };
  auto write_x = [&](int buf, int g, bool ph1 = false) {
    if constexpr (TWO) {
      if (a.in_mode == RCV_LOAD_GRAD_ENC) write_x_mode(std::integral_constant<int, RCV_LOAD_GRAD_ENC>{}, buf, g, ph1);
      else write_x_mode(std::integral_constant<int, RCV_LOAD_GRAD_DEC>{}, buf, g, ph1);
    } else {
      if (a.in_mode == RCV_LOAD_AFFINE) write_x_mode(std::integral_constant<int, RCV_LOAD_AFFINE>{}, buf, g, ph1);
      else if (a.in_mode == RCV_LOAD_AFFINE_RELU) write_x_mode(std::integral_constant<int, RCV_LOAD_AFFINE_RELU>{}, buf, g, ph1);
      else write_x_mode(std::integral_constant<int, RCV_LOAD_PLAIN>{}, buf, g, ph1);
    }
  };

  int pixoff[WN];
#pragma unroll
  for (int b = 0; b < WN; ++b) {
    const int p = (wave_n * WN + b) * 16 + l15;
    int ty = fd_div(p, a.fdWt), tx = p - ty * a.Wt;
    if (ty >= a.R) { ty = 0; tx = 0; }
    pixoff[b] = ((ty * IS) * a.IW + tx * IS) * S + l4;
  }
  int aoff[WM];                                 // A operand: filter row k = l4, column block swapped on odd rows (see dma_w)
#pragma unroll
  for (int m = 0; m < WM; ++m) aoff[m] = l4 * WS + ((((wave_m * WM + m) * 16) + l15) ^ ((l4 & 1) << 4));

  f32x4 acc[NPH][WM][WN];
#pragma unroll
  for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int b = 0; b < WN; ++b) acc[ph][m][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nsteps = a.CinP / CK, ngroups = a.CinP / XK;
#ifdef RCV_STAMPS
  unsigned long long st_k0 = 0, st_a = 0, st_b = 0, st_seg[6] = {0, 0, 0, 0, 0, 0}, st_loop0 = 0, st_loop1 = 0, st_rt0 = __builtin_amdgcn_s_memrealtime();
  RCV_STAMP(st_k0);
#endif
  // Pipeline over the 4-channel steps.  F(i) = filter slab of step i (LDS-DMA into wl[i & 1], requested one step ahead);
  // X(g) = input chunk of group g = steps [g XSTEPS, (g+1) XSTEPS), in xl[g & 1]: loaded into registers in the first step of group g-1
  // and written (transformed) between the two MFMA halves of that group's second step, so neither the load latency nor the
  // transform sits between two contractions.  One barrier per step; everything a step waits for was requested a whole step earlier.
  // The operands of tap j+1 are read from LDS ahead of the MFMAs of tap j (two register sets).
  struct Ops { float av[WM]; float bv[WN]; };
  auto read_tap = [&](Ops& o, const float* wb, const float* xb, int j, int dy, int dx) {
    const float* wj = wb + j * CK * WS;
    const float* xj = xb + (dy * a.IW + dx) * S;
#pragma unroll
    for (int m = 0; m < WM; ++m) o.av[m] = wj[aoff[m]];
#pragma unroll
    for (int b = 0; b < WN; ++b) o.bv[b] = xj[pixoff[b]];
  };
  auto mfma_tap = [&](const Ops& o, int ph) {
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int b = 0; b < WN; ++b)
        acc[ph][m][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.av[m], o.bv[b], acc[ph][m][b], 0, 0, 0);
  };
  // tap j of this workgroup's schedule -> (dy, dx, accumulator set)
  auto tap_geom = [&](int j, int& dy, int& dx, int& ph) {
    ph = 0;
    if (KIND == KIND_GATHER) { dy = (j / 3) * a.dil; dx = (j % 3) * a.dil; }
    else if (KIND == KIND_TALL) {
      // filter tap (ky,kx) of ConvTranspose2d(k3,s2,p1,op1): output row 2y+py takes input row y+dy through ky:
      // py=0: (dy=0,ky=1); py=1: (dy=0,ky=2),(dy=1,ky=0) -- the same along x.  Nine taps, each into the accumulators of its parity.
      const int ky = j / 3, kx = j % 3;
      const int py = ky == 1 ? 0 : 1, pxx = kx == 1 ? 0 : 1;
      dy = ky == 0 ? 1 : 0; dx = kx == 0 ? 1 : 0;
      ph = NPH == 4 ? py * 2 + pxx : 0;
    } else {
      const int jy = j / nxt, jx = j - jy * nxt;
      if (KIND == KIND_TPHASE) { dy = ti.py ? (jy ? 0 : 1) : 0; dx = ti.px ? (jx ? 0 : 1) : 0; }
      else { dy = jy; dx = jx; }
    }
  };

  // ---- prologue: F(0), X(0) and the load constants are requested together (one memory round trip)
  dma_w(0, 0);
  load_x(0);
  write_x(0, 0);
  if (two_phase) { load_x(0, true); write_x(0, 0, true); }
  constexpr int JSPLIT = (KIND == KIND_GATHER || KIND == KIND_TALL) ? 5 : 2;
#ifdef RCV_STAMPS
  RCV_STAMP(st_loop0);
  st_a = st_loop0;
#endif
  for (int i = 0; i < nsteps; ++i) {
    const int buf = i & 1;
    const int g = i / XSTEPS, sub = i - g * XSTEPS;
    // F(i) (and X(g+1), if this is the second step of the group) have landed: they were requested a whole step ago
    // (the builtin, not inline asm: the compiler's own wait insertion then KNOWS that the input registers requested a step ago have
    // arrived, and does not park a vmcnt(0) in front of their use -- behind the LDS-DMA instructions issued in between)
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
    // xl[g & 1] complete, F(i) visible to every wave, buffers of step i-1 retired.  A bare barrier (the LDS writes of this wave are
    // waited for explicitly)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef RCV_STAMPS
    RCV_STAMP(st_b); st_seg[0] += st_b - st_a; st_a = st_b;
#endif
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef RCV_STAMPS
    RCV_STAMP(st_b); st_seg[1] += st_b - st_a; st_a = st_b;
#endif
    // F(i+1) and X(g+1) are requested from inside the tap loop, one instruction per tap (an LDS-DMA instruction holds the wave's
    // issue for 60..100 cycles: spread out, the matrix pipe keeps running behind each of them).  The last step re-requests its own
    // slab into the retired buffer instead of branching.
    const bool ph1 = sub >= 2;                     // second staging phase of the next chunk (two_phase: steps 2 and 3 of a 16-channel group)
    const bool ldx = (sub == 0 || (two_phase && sub == 2)) && g + 1 < ngroups;
    const int fstep = i + 1 < nsteps ? i + 1 : i;
    constexpr bool SPREAD = KIND == KIND_GATHER;   // (all-parity transposed conv, 5-MFMA taps: measured 10 % slower spread out)
    if (!SPREAD) {
      if (ldx) load_x(g + 1, ph1);
      dma_w(buf ^ 1, fstep);
    }
#ifdef RCV_STAMPS
    RCV_STAMP(st_b); st_seg[2] += st_b - st_a; st_a = st_b;
#endif
    const float* wb = wl + buf * WBUF;
    const float* xb = xl + (g & 1) * a.xl_floats + sub * CK;
    const bool wx = (sub == 1 || XSTEPS == 1 || (two_phase && sub == 3)) && g + 1 < ngroups;
    if (KIND == KIND_GATHER || KIND == KIND_TALL) {
      // Two operand register sets: the LDS reads of tap j+1 are issued one behind each of the first MFMAs of tap j (pinned with
      // sched_group_barrier: left alone, the compiler sinks them to the end of the tap and every tap starts with an exposed LDS
      // round trip: 84 % instead of ~97 % of the issue rate for a wave that has the matrix pipe to itself).
      constexpr int N_M = WM * WN, N_D = WM + WN, N_PAIR = N_M < N_D ? N_M : N_D;
      Ops o0, o1;
      int dy, dx, ph, ph_next = 0;
      tap_geom(0, dy, dx, ph);
      read_tap(o0, wb, xb, 0, dy, dx);
      if (N_M >= N_D + 2) __builtin_amdgcn_sched_group_barrier(0x100, N_D, 0);
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        Ops& cur = (j & 1) ? o1 : o0;
        Ops& nxt_ = (j & 1) ? o0 : o1;
        if (j == JSPLIT) {
#ifdef RCV_STAMPS
          RCV_STAMP(st_b); st_seg[3] += st_b - st_a; st_a = st_b;
#endif
          if (wx) write_x((g + 1) & 1, g + 1, ph1);
#ifdef RCV_STAMPS
          RCV_STAMP(st_b); st_seg[4] += st_b - st_a; st_a = st_b;
#endif
        }
        if (j + 1 < 9) { tap_geom(j + 1, dy, dx, ph_next); read_tap(nxt_, wb, xb, j + 1, dy, dx); }
        if (SPREAD && j < WU) dma_one(buf ^ 1, fstep, j < WU ? j : 0);           // F(i+1), one LDS-DMA instruction per tap
        if (SPREAD && j >= 5 && ldx) {                                           // X(g+1): registers, written to LDS in the next step
#pragma unroll
          for (int q = 0; q < XMAX / 4; ++q) load_slot(g + 1, (j - 5) * (XMAX / 4) + q, ph1);
          if (j == 8 && !ph1) load_consts(g + 1);
        }
        mfma_tap(cur, ph);
        ph = ph_next;
        if (N_M < N_D + 2) {
          // (5-MFMA taps of the all-parity transposed conv: pinning measured 8..14 % slower than the compiler's own order)
        } else if (j + 1 < 9) {
          if (N_D > N_PAIR) __builtin_amdgcn_sched_group_barrier(0x100, N_D - N_PAIR, 0);
#pragma unroll
          for (int k = 0; k < N_PAIR; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          if (N_M > N_PAIR) __builtin_amdgcn_sched_group_barrier(0x008, N_M - N_PAIR, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, N_M, 0);
        }
      }
    } else {
      Ops o;
      int dy, dx, ph;
      for (int j = 0; j < ntaps; ++j) {
        if (j == JSPLIT && wx) write_x((g + 1) & 1, g + 1, ph1);
        tap_geom(j, dy, dx, ph);
        read_tap(o, wb, xb, j, dy, dx);
        mfma_tap(o, 0);
      }
      if (ntaps <= JSPLIT && wx) write_x((g + 1) & 1, g + 1, ph1);
    }
#ifdef RCV_STAMPS
    RCV_STAMP(st_b); st_seg[5] += st_b - st_a; st_a = st_b;
#endif
  }
#ifdef RCV_STAMPS
  RCV_STAMP(st_loop1);
#endif
  if (KIND == KIND_TALL) {
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
      TileInfo tp = ti;
      tp.py = ph >> 1; tp.px = ph & 1;
      if (ph) __syncthreads();                      // `red` of the previous parity has been consumed
      conv_epilogue<WM, WN, WAVES_M, WAVES_N, KIND_TPHASE, 2>(a, tp, acc[ph], red, tid);   // 160 accumulator registers are live: small batches
    }
  } else {
    // data-gradient launches (TWO) carry a residual and a BN-backward operand per output element: requested in one batch
    conv_epilogue<WM, WN, WAVES_M, WAVES_N, KIND, (TWO || WM == 1) ? WN : 0>(a, ti, acc[0], red, tid);
  }
#ifdef RCV_STAMPS
  if (a.stamps) {
    unsigned long long st_end;
    RCV_STAMP(st_end);
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      unsigned long long* o = a.stamps + ((size_t)blockIdx.x * (NT / 64) + wave) * 12;
      o[0] = st_loop0 - st_k0; o[1] = st_loop1 - st_loop0; o[2] = st_end - st_loop1;
      for (int k = 0; k < 6; ++k) o[3 + k] = st_seg[k];
      o[9] = rt1 - st_rt0; o[10] = st_end - st_k0; o[11] = st_rt0;
    }
  }
#endif
}

// --------------------------------------------------------------------------------------------
// Host side: tiling choice and launch
// --------------------------------------------------------------------------------------------
template <int WM, int WN, int WAVES_M, int WAVES_N, int KIND, int XMAX = 4>
static int launch_dma(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s, int dev) {
  const bool two = a.in_mode == RCV_LOAD_GRAD_ENC || a.in_mode == RCV_LOAD_GRAD_DEC;
  if (two) {
    auto kern = conv_dma_kernel<WM, WN, WAVES_M, WAVES_N, KIND, true, XMAX>;
    static size_t configured[RCV_MAX_DEVICES];
    RCV_ENSURE_LDS(kern, lds, dev, configured);
    hipLaunchKernelGGL(kern, grid, dim3(WAVES_M * WAVES_N * 64), lds, s, a);
  } else {
    auto kern = conv_dma_kernel<WM, WN, WAVES_M, WAVES_N, KIND, false, XMAX>;
    static size_t configured[RCV_MAX_DEVICES];
    RCV_ENSURE_LDS(kern, lds, dev, configured);
    hipLaunchKernelGGL(kern, grid, dim3(WAVES_M * WAVES_N * 64), lds, s, a);
  }
  RCV_HIP(hipGetLastError());
  return RCV_OK;
}

template <int KIND>
static int launch_dma_tile(int tile, const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s, int dev) {
  switch (tile) {
    case 0: return launch_dma<2, 5, 4, 1, KIND>(a, grid, lds, s, dev);
    case 1: return launch_dma<2, 5, 2, 2, KIND>(a, grid, lds, s, dev);
    default: return launch_dma<1, 5, 4, 1, KIND>(a, grid, lds, s, dev);
  }
}

struct TileCfg {
  int WM, WN, WAVES_M, WAVES_N;
  int cot() const { return WM * WAVES_M * 16; }
  int pix() const { return WN * WAVES_N * 16; }
  int nt() const { return WAVES_M * WAVES_N * 64; }
};
static const TileCfg kTiles[] = {
    {2, 5, 4, 1},  // 0: COT 128, PIX  80
    {2, 5, 2, 2},  // 1: COT  64, PIX 160
    {2, 5, 1, 4},  // 2: COT  32, PIX 320
    {1, 5, 1, 4},  // 3: COT  16, PIX 320
    {1, 5, 4, 1},  // 4: COT  64, PIX  80
    {1, 5, 2, 2},  // 5: COT  32, PIX 160
    {1, 5, 1, 2},  // 6: COT  16, PIX 160 (128 threads)
};
static const int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

template <int WM, int WN, int WAVES_M, int WAVES_N, int CK, int KIND>
static int launch_inst(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s, int dev) {
  auto kern = conv_mfma_kernel<WM, WN, WAVES_M, WAVES_N, CK, KIND>;
  static size_t configured[RCV_MAX_DEVICES];   // raise the dynamic-LDS limit once per instantiation and device (host-side state only)
  RCV_ENSURE_LDS(kern, lds, dev, configured);
  hipLaunchKernelGGL(kern, grid, dim3(WAVES_M * WAVES_N * 64), lds, s, a);
  RCV_HIP(hipGetLastError());
  return RCV_OK;
}

template <int CK, int KIND>
static int launch_tile(int tile, const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s, int dev) {
  switch (tile) {
    case 0: return launch_inst<2, 5, 4, 1, CK, KIND>(a, grid, lds, s, dev);
    case 1: return launch_inst<2, 5, 2, 2, CK, KIND>(a, grid, lds, s, dev);
    case 2: return launch_inst<2, 5, 1, 4, CK, KIND>(a, grid, lds, s, dev);
    case 3: return launch_inst<1, 5, 1, 4, CK, KIND>(a, grid, lds, s, dev);
    case 4: return launch_inst<1, 5, 4, 1, CK, KIND>(a, grid, lds, s, dev);
    case 5: return launch_inst<1, 5, 2, 2, CK, KIND>(a, grid, lds, s, dev);
    default: return launch_inst<1, 5, 1, 2, CK, KIND>(a, grid, lds, s, dev);
  }
}

// R rows x Wt cols with R*Wt <= PIX and the staged input tile within `cap` pixels.
static bool plan_tile(int kind, int TH, int TW, int PIX, int s, int d, int cap, int* R, int* Wt, int* tiles_x, int* tiles_y) {
  // Tiles of the LDS-DMA kernel carry a cap on their staged input pixels: the widest row segment may then only fit with few rows
  // (a stride-2 tile of 1 x 80 outputs stages 3 x 161 inputs, 2 x 40 stages 5 x 81).  Among the segment widths that need the
  // fewest tiles, take the one that stages the fewest input pixels; without a cap the widest segment is taken as is.
  const bool search = cap < 65535;
  long best_tiles = -1, best_staged = 0;
  int nx = ceil_div(TW, PIX < TW ? PIX : TW);
  for (; nx <= TW; ++nx) {
    const int wt = ceil_div(TW, nx);
    if (search && best_tiles >= 0 && wt < 8) break;
    int r = PIX / wt;
    if (r > TH) r = TH;
    if (r < 1) continue;
    int ih = 0, iw = 0;
    for (; r >= 1; --r) {
      tile_halo(kind, r, wt, s, d, &ih, &iw);
      if (ih * iw <= cap) break;
    }
    if (r < 1) continue;
    r = ceil_div(TH, ceil_div(TH, r));   // balance rows over the tiles of a column
    tile_halo(kind, r, wt, s, d, &ih, &iw);
    const long tiles = (long)ceil_div(TW, wt) * ceil_div(TH, r), staged = tiles * ih * iw;
    if (best_tiles < 0 || tiles < best_tiles || (tiles == best_tiles && staged < best_staged)) {
      best_tiles = tiles; best_staged = staged;
      *R = r; *Wt = wt;
      *tiles_x = ceil_div(TW, wt);
      *tiles_y = ceil_div(TH, r);
    }
    if (!search) return true;
  }
  return best_tiles >= 0;
}

static int make_plan(const rcv_handle* h, const rcv_op* op, ConvPlan* pl) {
  const bool transposed = op->kind == RCV_OP_TCONV;
  const int N = op->i[RCV_I_N], H = op->i[RCV_I_H], W = op->i[RCV_I_W];
  const int Cin = op->i[RCV_I_CIN], Cout = op->i[RCV_I_COUT];
  const int Ho = op->i[RCV_I_HO], Wo = op->i[RCV_I_WO];
  const int s = op->i[RCV_I_STRIDE], d = op->i[RCV_I_DIL];
  RCV_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv: empty shape N=%d H=%d W=%d Cin=%d Cout=%d", N, H, W, Cin, Cout);
  RCV_CHECK_ARG(Cout % 4 == 0, "conv: Cout=%d must be a multiple of 4", Cout);
  const int mode = op->i[RCV_I_INMODE];
  if (mode == RCV_LOAD_NCHW) RCV_CHECK_ARG(Cin <= 4, "conv: NCHW input supports Cin<=4 (got %d)", Cin);
  else RCV_CHECK_ARG(Cin % 4 == 0, "conv: Cin=%d must be a multiple of 4 for NHWC operands", Cin);
  if (transposed) {
    RCV_CHECK_ARG(Ho == 2 * H && Wo == 2 * W, "tconv: output must be 2x input (%dx%d -> %dx%d)", H, W, Ho, Wo);
    pl->kind = op->i[RCV_I_AUX0] ? KIND_TMERGED : KIND_TPHASE;     // AUX0 = 1 (or 4: split-bf16): filter is packed in the merged layout
    RCV_CHECK_ARG(pl->kind == KIND_TPHASE || 4 * Cout <= 128, "tconv: merged layout needs Cout <= 32 (got %d)", Cout);
  } else {
    RCV_CHECK_ARG((s == 1 || s == 2) && (d == 1 || d == 2), "conv: stride %d dilation %d unsupported", s, d);
    RCV_CHECK_ARG(Ho == (H - 1) / s + 1 && Wo == (W - 1) / s + 1, "conv: bad output size %dx%d for input %dx%d stride %d", Ho, Wo, H, W, s);
    pl->kind = KIND_GATHER;
  }
  const int CinP = round_up(Cin, 4);
  pl->narrow = 0;
  pl->dma = 0;
  pl->first = 0;
  pl->wino = 0;
  pl->small = 0;
  pl->bf3 = 0;
  if (convn_bf3_supported(h, op, pl->kind)) return convn_bf3_plan(h, op, pl->kind, pl);
  if (conv_bf3_supported(h, op, pl->kind)) return conv_bf3_plan(h, op, pl);
  RCV_CHECK_ARG(op->i[RCV_I_AUX0] < 3 || op->i[RCV_I_AUX0] > 5, "conv: a filter packed in a split-bf16 layout needs a record one of the split-bf16 kernels runs");
  if (conv_wino_supported(h, op, pl->kind)) return conv_wino_plan(h, op, pl);
  RCV_CHECK_ARG(transposed || op->i[RCV_I_AUX0] != 2, "conv: a filter packed in the Winograd layout needs stride 1 / dilation 1");
  if (conv_first_supported(op, pl->kind)) return conv_first_plan(h, op, pl);
  if (convs_supported(h, op, pl->kind, CinP, pl->kind == KIND_TMERGED ? 4 * Cout : Cout)) return convs_make_plan(h, op, pl->kind, pl);
  if (conv_small_supported(h, op, pl->kind)) return conv_small_plan(h, op, pl);
  pl->CK = (CinP % 8 == 0) ? 8 : 4;
  const int Q = pl->CK / 4;
  pl->CoutV = pl->kind == KIND_TMERGED ? 4 * Cout : Cout;
  pl->CoutP = round_up(pl->CoutV, 16);
  const int TH = transposed ? H : Ho, TW = transposed ? W : Wo;
  // candidates by (virtual) output-channel count; minimise padded work, prefer the larger tile on ties
  // LDS-DMA filter streaming pays where the filter dominates the staged bytes (wide layers); with few input
  // channels its 4-channel chunks fragment the HBM-bound input reads instead
  // (Cin = 32 with more than 32 output channels: the 32 -> 64 stride-2 conv and its twin; staged 16 channels at a time they read
  // 64 of the 128 bytes of a pixel record per pass instead of the 32 of the general kernel's 8-channel chunks)
  const bool use_dma = mode != RCV_LOAD_NCHW && Cin >= 32 && CinP % 8 == 0 && (long long)N * H * W * Cin < (1ll << 31) && !RCV_ENV("RCV_NO_DMA");
  (void)Q;
  long best = -1;
  bool best_dma = false;
  pl->tile = -1;
  for (int t = 0; t < kNumTiles; ++t) {
    const int cot = kTiles[t].cot();
    if (cot > pl->CoutP && cot != 16) continue;          // never pad co beyond one MFMA block
    if (pl->kind == KIND_TMERGED && cot < pl->CoutP) continue;   // merged layout: all parities in one workgroup
    if (pl->CoutP >= 128 && cot < 64) continue;
    if (pl->CoutP >= 64 && cot < 32) continue;
    const bool tall5 = use_dma && t == 5 && pl->kind == KIND_TPHASE && !RCV_ENV("RCV_NO_TALL");     // 32-channel tile: only as KIND_TALL
    const bool dma_tile = (use_dma && (t == 0 || t == 1 || t == 4)) || tall5;
    const int cap = dma_tile ? kTiles[t].nt() * 2 : 65535;      // DMA variant: the input chunk (>= 8 channels = 2 quads per pixel) rides in 4 register slots per thread
    int R, Wt, tx, ty;
    if (!plan_tile(pl->kind, TH, TW, kTiles[t].pix(), s, d, cap, &R, &Wt, &tx, &ty)) continue;
    const long work = (long)tx * ty * kTiles[t].pix() * round_up(pl->CoutP, cot);
    const bool bigger = pl->tile >= 0 && kTiles[t].pix() * cot > kTiles[pl->tile].pix() * kTiles[pl->tile].cot();
    if (best < 0 || work < best || (work == best && ((dma_tile && !best_dma) || (dma_tile == best_dma && bigger)))) {
      best = work; best_dma = dma_tile; pl->tile = t; pl->R = R; pl->Wt = Wt; pl->tiles_x = tx; pl->tiles_y = ty;
    }
  }
  RCV_CHECK_ARG(pl->tile >= 0, "conv: no tile configuration for Cout=%d", Cout);
  if (pl->kind == KIND_TPHASE && use_dma && pl->tile == 2 && !RCV_ENV("RCV_NO_TALL")) {
    // 32-channel transposed conv with >= 64 input channels: the half-size tile runs as KIND_TALL on the LDS-DMA kernel
    // (64 -> 32 at 32x60x80: 0.100 instead of 0.140 ms)
    int R, Wt, tx, ty;
    if (plan_tile(pl->kind, TH, TW, kTiles[5].pix(), s, d, kTiles[5].nt() * 2, &R, &Wt, &tx, &ty)) {
      pl->tile = 5; pl->R = R; pl->Wt = Wt; pl->tiles_x = tx; pl->tiles_y = ty;
    }
  }
  {   // a grid that leaves compute units idle: the sibling tile with the same channel count and half the pixels doubles the
      // workgroups (small planes: 128->64 at 64x15x20 runs 0.068 instead of 0.090 ms)
    const int sibling = (pl->tile == 0 || pl->tile == 1) ? 4 : (pl->tile == 2 ? 5 : (pl->tile == 3 ? 6 : -1));     // 0 -> 4 halves the channels instead
    const long nwg = (long)N * pl->tiles_x * pl->tiles_y * ceil_div(pl->CoutP, kTiles[pl->tile].cot()) * (pl->kind == KIND_TPHASE ? 4 : 1);
    if (sibling >= 0 && nwg < h->num_cus && !(pl->kind == KIND_TMERGED && kTiles[sibling].cot() < pl->CoutP)) {
      const bool dma_tile = use_dma && sibling == 4;
      int R, Wt, tx, ty;
      if (plan_tile(pl->kind, TH, TW, kTiles[sibling].pix(), s, d, dma_tile ? kTiles[sibling].nt() * 2 : 65535, &R, &Wt, &tx, &ty)) {
        pl->tile = sibling; pl->R = R; pl->Wt = Wt; pl->tiles_x = tx; pl->tiles_y = ty;
      }
    }
  }
  if (const char* ev = RCV_ENV("RCV_CONV_TILE")) {      // experiment override: "tile,R,Wt"
    int t = -1, r = 0, wt = 0;
    if (sscanf(ev, "%d,%d,%d", &t, &r, &wt) == 3 && t >= 0 && t < kNumTiles && r > 0 && wt > 0 && r * wt <= kTiles[t].pix() &&
        kTiles[t].cot() <= round_up(pl->CoutP, 16) + 0) {
      pl->tile = t; pl->R = r; pl->Wt = wt; pl->tiles_x = ceil_div(TW, wt); pl->tiles_y = ceil_div(TH, r);
    }
  }
  const TileCfg& tc = kTiles[pl->tile];
  tile_halo(pl->kind, pl->R, pl->Wt, s, d, &pl->IH, &pl->IW);
  const int ntaps = pl->kind == KIND_GATHER ? 9 : 4;
  const bool tall5 = pl->tile == 5 && pl->kind == KIND_TPHASE && !RCV_ENV("RCV_NO_TALL");
  pl->dma = use_dma && (pl->tile == 0 || pl->tile == 1 || pl->tile == 4 || tall5) && pl->IH * pl->IW <= tc.nt() * 2;
  pl->xk = 0;
  size_t floats;
  if (pl->dma) {
    pl->CK = 4;
    const int taps_dma = (pl->kind == KIND_TPHASE && (pl->tile == 4 || pl->tile == 5) && !RCV_ENV("RCV_NO_TALL")) ? 9 : ntaps;   // KIND_TALL below
    pl->wl_floats = ntaps * 4 * tc.cot();
    // input chunk: 16 channels per staging pass when the tile fits the register slots and leaves room for two workgroups per CU
    // (the pipeline overlaps one workgroup's staging with the other's contraction), else 8
    const int npix = pl->IH * pl->IW, xs = pl->kind == KIND_GATHER ? s : 1;
    auto lds_floats = [&](int xk) {
      return 2 * (size_t)taps_dma * 4 * tc.cot() + 2 * (size_t)round_up(npix * conv_xpitch(xk, xs), 4) + 5 * CinP + 16 + (size_t)tc.WAVES_N * 2 * tc.cot();
    };
    const size_t lds_room = (size_t)h->max_lds / 2;
    const int slots = 4;                                   // (pixel, quad) register slots per thread (an 8-slot variant of the kernel for the large stride-2 tiles measured 2.6x slower)
    // (16-channel chunks may be staged in two phases through the same slots: twice the items, see conv_dma_kernel)
    // -- for the two-tensor (gradient) loads only: measured on the 32 -> 64 stride-2 pair at 32x120x160, the data gradient went
    // 0.127 -> 0.115 ms with its fetched bytes at the algorithmic 190 MB, the one-tensor forward launch (whose four passes did survive in L2)
    // 0.090 -> 0.101 ms because the larger input buffers cost it the third resident workgroup
    const bool two = mode == RCV_LOAD_GRAD_ENC || mode == RCV_LOAD_GRAD_DEC;
    pl->xk = (CinP % 16 == 0 && npix * 4 <= tc.nt() * slots * (two ? 2 : 1) && lds_floats(16) * sizeof(float) <= lds_room) ? 16 : 8;
    pl->xl_floats = round_up(npix * conv_xpitch(pl->xk, xs), 4);
    floats = 2 * (size_t)pl->wl_floats + 2 * (size_t)pl->xl_floats + 5 * CinP + 16 + (size_t)tc.WAVES_N * 2 * tc.cot();
  } else {
    pl->wl_floats = round_up(ntaps * pl->CK * (tc.cot() + 16), 4);
    pl->xl_floats = round_up(pl->IH * pl->IW * conv_xpitch(pl->CK, pl->kind == KIND_GATHER ? s : 1), 4);
    floats = (size_t)pl->wl_floats + pl->xl_floats + 5 * CinP + 16 + (size_t)tc.WAVES_N * 2 * tc.cot();
  }
  pl->lds = floats * sizeof(float);
  RCV_CHECK_ARG(pl->lds <= (size_t)h->max_lds, "conv: tile needs %zu B of LDS (limit %d)", pl->lds, h->max_lds);
  pl->n_co_tiles = ceil_div(pl->CoutP, tc.cot());
  pl->n_phases = pl->kind == KIND_TPHASE ? 4 : 1;
  const int n_pix_tiles = N * pl->tiles_x * pl->tiles_y;
  pl->total_tiles = n_pix_tiles * pl->n_co_tiles * pl->n_phases;
  if (pl->kind == KIND_TPHASE && pl->dma && (pl->tile == 4 || pl->tile == 5) && !RCV_ENV("RCV_NO_TALL")) {
    // all four output parities in one workgroup (KIND_TALL): one staging of the input and nine taps per chunk instead of four
    // workgroups with 1/2/2/4 taps each paying the whole per-chunk overhead; n_phases stays 4: the statistics rows are per parity
    pl->kind = KIND_TALL;
    pl->wl_floats = 9 * 4 * tc.cot();
    floats = 2 * (size_t)pl->wl_floats + 2 * (size_t)pl->xl_floats + 5 * CinP + 16 + (size_t)tc.WAVES_N * 2 * tc.cot();
    pl->lds = floats * sizeof(float);
    pl->total_tiles = n_pix_tiles * pl->n_co_tiles;
  }
  pl->grid = pl->total_tiles;
  return RCV_OK;
}

int rcv_launch_conv(const rcv_handle* h, const rcv_op* op, hipStream_t s, OpQuery* query) {
  ConvPlan pl;
  if (!rcv_plan_get(h, op, &pl)) {
    memset(&pl, 0, sizeof(pl));
    const int rc = make_plan(h, op, &pl);
    if (rc) return rc;
    pl.dev = h->device;
    rcv_plan_put(h, op, pl);
  }
  const int Cout = op->i[RCV_I_COUT];
  const int n_pix_tiles = op->i[RCV_I_N] * pl.tiles_x * pl.tiles_y;
  const int n_part = (pl.narrow || pl.first || pl.bf3 == 2) ? pl.grid : n_pix_tiles * pl.n_phases;
  if (query) {
    static const char* kn[] = {"conv", "tconv", "tconvm", "tconva"};
    if (pl.bf3 == 2) {
      snprintf(query->label, sizeof(query->label), "%sn_bf3<%d,%d,%d>", pl.kind == KIND_TMERGED ? "tconv" : "conv", pl.CK, pl.WM, pl.WN);
    } else if (pl.bf3) {
      snprintf(query->label, sizeof(query->label), "conv%s_bf3<64,%d>", op->i[RCV_I_STRIDE] == 2 ? "2" : "", 32 * pl.WN);
    } else if (pl.wino) {
      snprintf(query->label, sizeof(query->label), "conv_wino<64,%d>", 16 * pl.WN);
    } else if (pl.small) {
      snprintf(query->label, sizeof(query->label), "conv_small<%d>", op->i[RCV_I_DIL]);
    } else if (pl.first) {
      snprintf(query->label, sizeof(query->label), "conv_first<%d>", op->i[RCV_I_DIL]);
    } else if (pl.narrow) {
      snprintf(query->label, sizeof(query->label), "%ss_mfma<%d,%d,%d>", kn[pl.kind], pl.WM, pl.WN, pl.CK);
    } else {
      const TileCfg& tc = kTiles[pl.tile];
      snprintf(query->label, sizeof(query->label), "%s_%s<%d,%d,%d,%d,%d>", kn[pl.kind], pl.dma ? "dma" : "mfma", tc.WM, tc.WN, tc.WAVES_M,
               tc.WAVES_N, pl.CK);
    }
    query->n_part = op->i[RCV_I_STATS] != RCV_STATS_NONE ? n_part : 0;
    query->n_split = 0;
    query->part_bytes = (size_t)query->n_part * 2 * Cout * sizeof(float);
    return RCV_OK;
  }
  ConvArgs a;
  a.in = (const float*)op->p[RCV_P_IN];
  a.in_aux = (const float*)op->p[RCV_P_IN_AUX];
  a.in_c = (const float*)op->p[RCV_P_IN_C];
  a.w = (const float*)op->p[RCV_P_W];
  a.bias = (const float*)op->p[RCV_P_BIAS];
  a.out = (float*)op->p[RCV_P_OUT];
  a.resid = (const float*)op->p[RCV_P_RESID];
  a.epi_aux = (const float*)op->p[RCV_P_EPI_AUX];
  a.epi_c = (const float*)op->p[RCV_P_EPI_C];
  a.part = (float*)op->p[RCV_P_PART];
  a.N = op->i[RCV_I_N]; a.H = op->i[RCV_I_H]; a.W = op->i[RCV_I_W];
  a.Cin = op->i[RCV_I_CIN]; a.Cout = Cout; a.Ho = op->i[RCV_I_HO]; a.Wo = op->i[RCV_I_WO];
  a.CinP = round_up(a.Cin, 4); a.CoutP = pl.CoutP; a.CoutV = pl.CoutV;
  a.stride = op->i[RCV_I_STRIDE]; a.dil = op->i[RCV_I_DIL];
  a.R = pl.R; a.Wt = pl.Wt; a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y; a.IH = pl.IH; a.IW = pl.IW;
  a.n_pix_tiles = n_pix_tiles; a.n_co_tiles = pl.n_co_tiles; a.n_sub = pl.n_co_tiles * (pl.kind == KIND_TALL ? 1 : pl.n_phases);
  a.total_tiles = pl.total_tiles; a.nchunks = a.CinP / pl.CK;      // (conv_bf3: CK = 32, Cin % 32 == 0)
  a.in_mode = op->i[RCV_I_INMODE]; a.stats = op->i[RCV_I_STATS]; a.flags = op->flags;
  a.wl_floats = pl.wl_floats; a.xl_floats = pl.xl_floats;
  a.xk = pl.xk;
  a.xpitch = conv_xpitch((pl.dma || pl.wino) ? pl.xk : pl.CK, pl.kind == KIND_GATHER ? a.stride : 1);
  a.fdWt = make_fastdiv(pl.Wt); a.fdIW = make_fastdiv(pl.IW);
  a.fdTX = make_fastdiv(pl.tiles_x); a.fdTY = make_fastdiv(pl.tiles_y);
#ifdef RCV_STAMPS
  a.stamps = (unsigned long long*)op->p[RCV_P_X5];
#endif
  RCV_CHECK_ARG(a.in && a.w && a.out, "conv: null operand");
  RCV_CHECK_ARG(a.in_mode == RCV_LOAD_PLAIN || a.in_mode == RCV_LOAD_NCHW || a.in_c, "conv: load mode %d needs constants", a.in_mode);
  RCV_CHECK_ARG(!(a.in_mode == RCV_LOAD_GRAD_ENC || a.in_mode == RCV_LOAD_GRAD_DEC) || a.in_aux, "conv: gradient load needs aux tensor");
  RCV_CHECK_ARG(!(a.flags & RCV_F_BIAS) || a.bias, "conv: bias flag without bias");
  RCV_CHECK_ARG(!(a.flags & RCV_F_RESID) || a.resid, "conv: resid flag without tensor");
  RCV_CHECK_ARG(a.stats == RCV_STATS_NONE || a.part, "conv: statistics requested without workspace");
  RCV_CHECK_ARG(a.stats == RCV_STATS_NONE || op->i[RCV_I_NPART] == n_part, "conv: workspace rows %d != %d", op->i[RCV_I_NPART], n_part);
  RCV_CHECK_ARG(!(a.stats == RCV_STATS_BWD_ENC || a.stats == RCV_STATS_BWD_DEC) || a.epi_aux, "conv: backward statistics need epi_aux");
  RCV_CHECK_ARG(!(a.stats == RCV_STATS_BWD_ENC || a.stats == RCV_STATS_BWD_DEC) || a.epi_c, "conv: backward statistics need epi_c (scale, shift, mean)");
  if (pl.bf3 == 2) return convn_bf3_launch(pl, a, s);
  if (pl.bf3) return conv_bf3_launch(pl, a, s);
  if (pl.wino) return conv_wino_launch(pl, a, s);
  if (pl.first) return conv_first_launch(pl, a, s);
  if (pl.small) return conv_small_launch(pl, a, s);
  if (pl.narrow) return convs_launch(pl, a, a.in_mode == RCV_LOAD_GRAD_ENC || a.in_mode == RCV_LOAD_GRAD_DEC, s);
  const dim3 grid(pl.grid);
  if (pl.dma) {
    if (pl.kind == KIND_TALL)
      return pl.tile == 4 ? launch_dma<1, 5, 4, 1, KIND_TALL>(a, grid, pl.lds, s, h->device) : launch_dma<1, 5, 2, 2, KIND_TALL>(a, grid, pl.lds, s, h->device);
    if (pl.kind == KIND_TPHASE) return launch_dma_tile<KIND_TPHASE>(pl.tile, a, grid, pl.lds, s, h->device);
    if (pl.kind == KIND_TMERGED) return launch_dma_tile<KIND_TMERGED>(pl.tile, a, grid, pl.lds, s, h->device);
    return launch_dma_tile<KIND_GATHER>(pl.tile, a, grid, pl.lds, s, h->device);
  }
  if (pl.kind == KIND_TPHASE)
    return pl.CK == 8 ? launch_tile<8, KIND_TPHASE>(pl.tile, a, grid, pl.lds, s, h->device) : launch_tile<4, KIND_TPHASE>(pl.tile, a, grid, pl.lds, s, h->device);
  if (pl.kind == KIND_TMERGED)
    return pl.CK == 8 ? launch_tile<8, KIND_TMERGED>(pl.tile, a, grid, pl.lds, s, h->device) : launch_tile<4, KIND_TMERGED>(pl.tile, a, grid, pl.lds, s, h->device);
  return pl.CK == 8 ? launch_tile<8, KIND_GATHER>(pl.tile, a, grid, pl.lds, s, h->device) : launch_tile<4, KIND_GATHER>(pl.tile, a, grid, pl.lds, s, h->device);
}

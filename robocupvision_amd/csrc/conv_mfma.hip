// 3x3 convolution as implicit GEMM on the gfx950 fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// One kernel serves four roles of the ROBO-UNet step (reference call sites in include/rcv.h):
//   gather conv   (RCV_OP_CONV):  forward of Conv2d(k3, pad=dil, stride 1|2, dilation 1|2),
//                                 data gradient of a stride-1 conv (flipped/transposed filter),
//                                 data gradient of the stride-2 transposed conv;
//   transposed    (RCV_OP_TCONV): forward of ConvTranspose2d(k3,s2,p1,op1) and data gradient of a
//                                 stride-2 conv, as four output-parity phases (blockIdx.z) with
//                                 1/2/2/4 taps each -- no multiplications by inserted zeros.
//
// GEMM mapping per workgroup:  D[co][pixel] += W[co][k] * X[k][pixel],  k = (tap, ci)
//   A operand  = packed filter [tap][ci][co] staged in LDS as [tap][ci][COT+16]  (co contiguous)
//   B operand  = input tile staged in LDS as [pixel][CK+1] (odd pixel pitch => the 16 pixels of an
//                MFMA block hit 16 different banks; the 4 k-lanes read 4 consecutive channels)
//   MFMA 16x16x4 lane map (cdna guide section 3): A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
//                D[row=4*(l>>4)+r][col=l&15]  => every lane ends with 4 consecutive output channels
//                of one pixel: one 16-byte NHWC store.
// A wave owns WM co-blocks x WN pixel-blocks (accumulators WM*WN*4 VGPRs); a workgroup is
// WAVES_M x WAVES_N waves and walks Cin in chunks of CK channels.  The tile is R rows x Wt columns
// of the output (gather) / input (transposed) plane, flattened to <= PIX pixels.
//
// While a tile is staged, the producer's BatchNorm (or the BN/ReLU backward of the consumer) is
// applied to the operand (RCV_LOAD_*), so normalised activations and pre-activation gradients are
// never materialised in HBM.  The epilogue adds bias / skip gradient, applies ReLU, stores, and
// emits the per-workgroup partial sums the following BatchNorm (forward or backward) needs.
#include "rcv_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float* in;
  const float* in_aux;
  const float* in_c;   // [5][Cin]
  const float* w;      // [9][CinP][CoutP]
  const float* bias;
  float* out;
  const float* resid;
  const float* epi_aux;
  const float* epi_c;  // [2][Cout]
  float* part;         // [n_part][2][Cout]
  int N, H, W, Cin, Cout, Ho, Wo;
  int CinP, CoutP;
  int stride, dil;
  int R, Wt, tiles_x, tiles_y;
  int IH, IW;
  int in_mode, stats;
  uint32_t flags;
  int wl_floats;       // floats reserved for the filter tile (input tile follows)
  FastDiv fdWt, fdIW;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Operand transform applied while staging (see RCV_LOAD_* in rcv.h).
template <int MODE>
__device__ __forceinline__ float4 xform4(float4 x, float4 a, const float4 (&k)[5]) {
  float4 v;
  if (MODE == RCV_LOAD_PLAIN) {
    v = x;
  } else if (MODE == RCV_LOAD_AFFINE) {
    v.x = fmaf(x.x, k[0].x, k[1].x); v.y = fmaf(x.y, k[0].y, k[1].y);
    v.z = fmaf(x.z, k[0].z, k[1].z); v.w = fmaf(x.w, k[0].w, k[1].w);
  } else if (MODE == RCV_LOAD_AFFINE_RELU) {
    v.x = fmaxf(fmaf(x.x, k[0].x, k[1].x), 0.f); v.y = fmaxf(fmaf(x.y, k[0].y, k[1].y), 0.f);
    v.z = fmaxf(fmaf(x.z, k[0].z, k[1].z), 0.f); v.w = fmaxf(fmaf(x.w, k[0].w, k[1].w), 0.f);
  } else if (MODE == RCV_LOAD_GRAD_ENC) {
    v.x = a.x > 0.f ? fmaf(k[0].x, x.x, fmaf(k[2].x, a.x, k[1].x)) : 0.f;
    v.y = a.y > 0.f ? fmaf(k[0].y, x.y, fmaf(k[2].y, a.y, k[1].y)) : 0.f;
    v.z = a.z > 0.f ? fmaf(k[0].z, x.z, fmaf(k[2].z, a.z, k[1].z)) : 0.f;
    v.w = a.w > 0.f ? fmaf(k[0].w, x.w, fmaf(k[2].w, a.w, k[1].w)) : 0.f;
  } else {  // RCV_LOAD_GRAD_DEC
    v.x = fmaf(k[0].x, (fmaf(a.x, k[3].x, k[4].x) > 0.f ? x.x : 0.f), fmaf(k[2].x, a.x, k[1].x));
    v.y = fmaf(k[0].y, (fmaf(a.y, k[3].y, k[4].y) > 0.f ? x.y : 0.f), fmaf(k[2].y, a.y, k[1].y));
    v.z = fmaf(k[0].z, (fmaf(a.z, k[3].z, k[4].z) > 0.f ? x.z : 0.f), fmaf(k[2].z, a.z, k[1].z));
    v.w = fmaf(k[0].w, (fmaf(a.w, k[3].w, k[4].w) > 0.f ? x.w : 0.f), fmaf(k[2].w, a.w, k[1].w));
  }
  return v;
}

// Stage the [IH*IW][CK] input tile of channel chunk c0 into xl ([pixel][CK+1]).
template <int MODE, int CK, int NT>
__device__ __forceinline__ void stage_input(const ConvArgs& a, float* xl, int n, int oy0, int ox0, int c0, int tid) {
  constexpr int S = CK + 1;
  constexpr int Q = CK / 4;
  const int npix = a.IH * a.IW;
  if (MODE == RCV_LOAD_NCHW) {
    // network input image [N][Cin][H][W], Cin <= 4: one pixel per thread, planes read coalesced
    for (int pix = tid; pix < npix; pix += NT) {
      const int iy = fd_div(pix, a.fdIW), ix = pix - iy * a.IW;
      const int gy = oy0 + iy, gx = ox0 + ix;
      const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
#pragma unroll
      for (int j = 0; j < CK; ++j) {
        float v = 0.f;
        if (ok && c0 + j < a.Cin) v = a.in[((size_t)(n * a.Cin + c0 + j) * a.H + gy) * a.W + gx];
        xl[pix * S + j] = v;
      }
    }
    return;
  }
  const int q = tid % Q;
  const int ch = c0 + 4 * q;
  float4 k[5];
  if (MODE != RCV_LOAD_PLAIN) {
#pragma unroll
    for (int j = 0; j < 5; ++j) k[j] = ld4(a.in_c + (size_t)j * a.Cin + ch);
  }
  for (int pix = tid / Q; pix < npix; pix += NT / Q) {
    const int iy = fd_div(pix, a.fdIW), ix = pix - iy * a.IW;
    const int gy = oy0 + iy, gx = ox0 + ix;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) {
      const size_t off = ((size_t)(n * a.H + gy) * a.W + gx) * a.Cin + ch;
      const float4 x = ld4(a.in + off);
      float4 aux = x;
      if (MODE == RCV_LOAD_GRAD_ENC || MODE == RCV_LOAD_GRAD_DEC) aux = ld4(a.in_aux + off);
      v = xform4<MODE>(x, aux, k);   // zero padding is applied AFTER the transform (padded taps are 0)
    }
    float* d = xl + pix * S + 4 * q;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

template <int WM, int WN, int WAVES_M, int WAVES_N, int CK, bool TRANSPOSED>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void conv_mfma_kernel(const ConvArgs a) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int COT = WM * WAVES_M * 16;
  constexpr int S = CK + 1;
  constexpr int WS = COT + 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;
  float* xl = smem + a.wl_floats;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
  const int l15 = lane & 15, l4 = lane >> 4;

  int t = blockIdx.x;
  const int tx_i = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty_i = t % a.tiles_y;
  const int n = t / a.tiles_y;
  const int y0 = ty_i * a.R, x0 = tx_i * a.Wt;
  const int co0 = blockIdx.y * COT;

  // taps of this workgroup
  int py = 0, px = 0, ntaps = 9, nxt = 3;
  int oy0, ox0, IS;
  if (TRANSPOSED) {
    py = blockIdx.z >> 1;
    px = blockIdx.z & 1;
    nxt = 1 + px;
    ntaps = (1 + py) * nxt;
    oy0 = y0; ox0 = x0; IS = 1;
  } else {
    oy0 = y0 * a.stride - a.dil; ox0 = x0 * a.stride - a.dil; IS = a.stride;
  }

  // per-lane pixel of each B block
  int pixoff[WN];
  int p_ty[WN], p_tx[WN];
  bool p_ok[WN];
#pragma unroll
  for (int b = 0; b < WN; ++b) {
    const int p = (wave_n * WN + b) * 16 + l15;
    int ty = fd_div(p, a.fdWt), tx = p - ty * a.Wt;
    const bool ok = ty < a.R;
    if (!ok) { ty = 0; tx = 0; }
    p_ty[b] = ty; p_tx[b] = tx; p_ok[b] = ok;
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
      constexpr int C4 = COT / 4;
      const int total = ntaps * CK * C4;
      for (int e = tid; e < total; e += NT) {
        const int row = e / C4, c4 = e % C4;
        const int j = row / CK, ck = row % CK;
        int t9 = j;
        if (TRANSPOSED) {
          const int jy = j / nxt, jx = j - jy * nxt;
          const int ky = py ? (jy ? 2 : 0) : 1;
          const int kx = px ? (jx ? 2 : 0) : 1;
          t9 = ky * 3 + kx;
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int co = co0 + 4 * c4;
        if (co < a.CoutP) v = ld4(a.w + ((size_t)(t9 * a.CinP + c0 + ck) * a.CoutP + co));
        *reinterpret_cast<float4*>(wl + row * WS + 4 * c4) = v;
      }
    }
    // ---- stage input chunk
    switch (a.in_mode) {
      case RCV_LOAD_PLAIN: stage_input<RCV_LOAD_PLAIN, CK, NT>(a, xl, n, oy0, ox0, c0, tid); break;
      case RCV_LOAD_AFFINE: stage_input<RCV_LOAD_AFFINE, CK, NT>(a, xl, n, oy0, ox0, c0, tid); break;
      case RCV_LOAD_AFFINE_RELU: stage_input<RCV_LOAD_AFFINE_RELU, CK, NT>(a, xl, n, oy0, ox0, c0, tid); break;
      case RCV_LOAD_GRAD_ENC: stage_input<RCV_LOAD_GRAD_ENC, CK, NT>(a, xl, n, oy0, ox0, c0, tid); break;
      case RCV_LOAD_GRAD_DEC: stage_input<RCV_LOAD_GRAD_DEC, CK, NT>(a, xl, n, oy0, ox0, c0, tid); break;
      default: stage_input<RCV_LOAD_NCHW, CK, NT>(a, xl, n, oy0, ox0, c0, tid); break;
    }
    __syncthreads();
    // ---- contraction over (tap, ci in chunk)
    for (int j = 0; j < ntaps; ++j) {
      int dy, dx;
      if (TRANSPOSED) {
        const int jy = j / nxt, jx = j - jy * nxt;
        dy = py ? (jy ? 0 : 1) : 0;
        dx = px ? (jx ? 0 : 1) : 0;
      } else {
        const int ky = j / 3, kx = j - ky * 3;
        dy = ky * a.dil; dx = kx * a.dil;
      }
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
    }
  }

  // ---------------------------------- epilogue ----------------------------------
  float s1[WM][4], s2[WM][4];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[m][r] = 0.f; s2[m][r] = 0.f; }

#pragma unroll
  for (int m = 0; m < WM; ++m) {
    const int co = co0 + (wave_m * WM + m) * 16 + 4 * l4;
    const bool co_ok = co < a.Cout;
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 e0 = bias, e1 = bias;
    if (co_ok) {
      if (a.flags & RCV_F_BIAS) bias = ld4(a.bias + co);
      if (a.stats == RCV_STATS_BWD_DEC) { e0 = ld4(a.epi_c + co); e1 = ld4(a.epi_c + a.Cout + co); }
    }
#pragma unroll
    for (int b = 0; b < WN; ++b) {
      int oy = y0 + p_ty[b], ox = x0 + p_tx[b];
      bool ok = p_ok[b] && co_ok;
      if (TRANSPOSED) {
        ok = ok && oy < a.H && ox < a.W;
        oy = 2 * oy + py; ox = 2 * ox + px;
      } else {
        ok = ok && oy < a.Ho && ox < a.Wo;
      }
      if (!ok) continue;
      const size_t off = ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.Cout + co;
      float4 v = make_float4(acc[m][b][0] + bias.x, acc[m][b][1] + bias.y, acc[m][b][2] + bias.z, acc[m][b][3] + bias.w);
      if (a.flags & RCV_F_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (a.flags & RCV_F_RESID) { const float4 rr = ld4(a.resid + off); v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
      *reinterpret_cast<float4*>(a.out + off) = v;
      if (a.stats == RCV_STATS_FWD) {
        s1[m][0] += v.x; s1[m][1] += v.y; s1[m][2] += v.z; s1[m][3] += v.w;
        s2[m][0] = fmaf(v.x, v.x, s2[m][0]); s2[m][1] = fmaf(v.y, v.y, s2[m][1]);
        s2[m][2] = fmaf(v.z, v.z, s2[m][2]); s2[m][3] = fmaf(v.w, v.w, s2[m][3]);
      } else if (a.stats == RCV_STATS_BWD_ENC) {
        const float4 e = ld4(a.epi_aux + off);
        s1[m][0] += v.x; s1[m][1] += v.y; s1[m][2] += v.z; s1[m][3] += v.w;
        s2[m][0] = fmaf(v.x, e.x, s2[m][0]); s2[m][1] = fmaf(v.y, e.y, s2[m][1]);
        s2[m][2] = fmaf(v.z, e.z, s2[m][2]); s2[m][3] = fmaf(v.w, e.w, s2[m][3]);
      } else if (a.stats == RCV_STATS_BWD_DEC) {
        const float4 e = ld4(a.epi_aux + off);
        const float gx = fmaf(e.x, e0.x, e1.x) > 0.f ? v.x : 0.f;
        const float gy = fmaf(e.y, e0.y, e1.y) > 0.f ? v.y : 0.f;
        const float gz = fmaf(e.z, e0.z, e1.z) > 0.f ? v.z : 0.f;
        const float gw = fmaf(e.w, e0.w, e1.w) > 0.f ? v.w : 0.f;
        s1[m][0] += gx; s1[m][1] += gy; s1[m][2] += gz; s1[m][3] += gw;
        s2[m][0] = fmaf(gx, e.x, s2[m][0]); s2[m][1] = fmaf(gy, e.y, s2[m][1]);
        s2[m][2] = fmaf(gz, e.z, s2[m][2]); s2[m][3] = fmaf(gw, e.w, s2[m][3]);
      }
    }
  }

  if (a.stats != RCV_STATS_NONE) {
    // wave: sum over the 16 pixel lanes (xor 1,2,4,8 stays inside a 16-lane group); fixed order
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float u = s1[m][r], v = s2[m][r];
#pragma unroll
        for (int sh = 1; sh < 16; sh <<= 1) { u += __shfl_xor(u, sh); v += __shfl_xor(v, sh); }
        s1[m][r] = u; s2[m][r] = v;
      }
    __syncthreads();                      // everyone is done reading the tiles: reuse LDS
    float* red = smem;                    // [WAVES_N][2][COT]
    if (l15 == 0) {
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cl = (wave_m * WM + m) * 16 + 4 * l4 + r;
          red[(wave_n * 2 + 0) * COT + cl] = s1[m][r];
          red[(wave_n * 2 + 1) * COT + cl] = s2[m][r];
        }
    }
    __syncthreads();
    for (int e = tid; e < 2 * COT; e += NT) {
      const int which = e / COT, cl = e % COT;
      const int co = co0 + cl;
      if (co < a.Cout) {
        float u = 0.f;
#pragma unroll
        for (int wn = 0; wn < WAVES_N; ++wn) u += red[(wn * 2 + which) * COT + cl];
        const size_t row = (size_t)blockIdx.z * gridDim.x + blockIdx.x;
        a.part[(row * 2 + which) * a.Cout + co] = u;
      }
    }
  }
}

// --------------------------------------------------------------------------------------------
// Host side: tiling choice and launch
// --------------------------------------------------------------------------------------------
struct TileCfg {
  int WM, WN, WAVES_M, WAVES_N;
  int cot() const { return WM * WAVES_M * 16; }
  int pix() const { return WN * WAVES_N * 16; }
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

template <int WM, int WN, int WAVES_M, int WAVES_N, int CK, bool TR>
static int launch_inst(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  auto kern = conv_mfma_kernel<WM, WN, WAVES_M, WAVES_N, CK, TR>;
  static size_t configured = 0;
  if (lds > configured) {   // raise the dynamic-LDS limit once per instantiation (host-side state only)
    RCV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured = lds;
  }
  hipLaunchKernelGGL(kern, grid, dim3(WAVES_M * WAVES_N * 64), lds, s, a);
  RCV_HIP(hipGetLastError());
  return RCV_OK;
}

template <int CK, bool TR>
static int launch_tile(int tile, const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  switch (tile) {
    case 0: return launch_inst<2, 5, 4, 1, CK, TR>(a, grid, lds, s);
    case 1: return launch_inst<2, 5, 2, 2, CK, TR>(a, grid, lds, s);
    case 2: return launch_inst<2, 5, 1, 4, CK, TR>(a, grid, lds, s);
    case 3: return launch_inst<1, 5, 1, 4, CK, TR>(a, grid, lds, s);
    case 4: return launch_inst<1, 5, 4, 1, CK, TR>(a, grid, lds, s);
    case 5: return launch_inst<1, 5, 2, 2, CK, TR>(a, grid, lds, s);
    default: return launch_inst<1, 5, 1, 2, CK, TR>(a, grid, lds, s);
  }
}

struct ConvPlan {
  int tile, CK, R, Wt, tiles_x, tiles_y, IH, IW;
  size_t lds;
  int wl_floats;
  dim3 grid;
};

static void plan_tile(int TH, int TW, int PIX, int* R, int* Wt, int* tiles_x, int* tiles_y) {
  int wt, r;
  if (TW <= PIX) {
    wt = TW;
    r = PIX / wt;
    if (r > TH) r = TH;
    r = ceil_div(TH, ceil_div(TH, r));   // balance rows over the tiles of a column
  } else {
    const int nx = ceil_div(TW, PIX);
    wt = ceil_div(TW, nx);
    r = 1;
  }
  *R = r; *Wt = wt;
  *tiles_x = ceil_div(TW, wt);
  *tiles_y = ceil_div(TH, r);
}

static int make_plan(const rcv_handle* h, const rcv_op* op, bool transposed, ConvPlan* pl) {
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
  } else {
    RCV_CHECK_ARG((s == 1 || s == 2) && (d == 1 || d == 2), "conv: stride %d dilation %d unsupported", s, d);
    RCV_CHECK_ARG(Ho == (H - 1) / s + 1 && Wo == (W - 1) / s + 1, "conv: bad output size %dx%d for input %dx%d stride %d", Ho, Wo, H, W, s);
  }
  const int CinP = round_up(Cin, 4);
  pl->CK = (CinP % 8 == 0) ? 8 : 4;
  const int TH = transposed ? H : Ho, TW = transposed ? W : Wo;
  // candidates by output-channel count; minimise padded work, prefer the larger tile on ties
  long best = -1;
  for (int t = 0; t < kNumTiles; ++t) {
    const int cot = kTiles[t].cot();
    const int CoutP16 = round_up(Cout, 16);
    if (cot > CoutP16 && !(cot == 16)) continue;          // never pad co beyond one MFMA block
    if (CoutP16 >= 128 && cot < 64) continue;
    if (CoutP16 >= 64 && cot < 32) continue;
    int R, Wt, tx, ty;
    plan_tile(TH, TW, kTiles[t].pix(), &R, &Wt, &tx, &ty);
    const long work = (long)tx * ty * kTiles[t].pix() * round_up(CoutP16, cot);
    if (best < 0 || work < best || (work == best && kTiles[t].pix() * cot > kTiles[pl->tile].pix() * kTiles[pl->tile].cot())) {
      best = work; pl->tile = t; pl->R = R; pl->Wt = Wt; pl->tiles_x = tx; pl->tiles_y = ty;
    }
  }
  RCV_CHECK_ARG(best >= 0, "conv: no tile configuration for Cout=%d", Cout);
  const TileCfg& tc = kTiles[pl->tile];
  if (transposed) { pl->IH = pl->R + 1; pl->IW = pl->Wt + 1; }
  else { pl->IH = (pl->R - 1) * s + 2 * d + 1; pl->IW = (pl->Wt - 1) * s + 2 * d + 1; }
  RCV_CHECK_ARG(pl->IH * pl->IW < 65536, "conv: input tile too large");
  const int ntaps = transposed ? 4 : 9;
  pl->wl_floats = round_up(ntaps * pl->CK * (tc.cot() + 16), 4);
  size_t floats = (size_t)pl->wl_floats + (size_t)pl->IH * pl->IW * (pl->CK + 1);
  const size_t red = (size_t)tc.WAVES_N * 2 * tc.cot();
  if (floats < red) floats = red;
  pl->lds = floats * sizeof(float);
  RCV_CHECK_ARG(pl->lds <= (size_t)h->max_lds, "conv: tile needs %zu B of LDS (limit %d)", pl->lds, h->max_lds);
  pl->grid = dim3(N * pl->tiles_x * pl->tiles_y, ceil_div(round_up(Cout, 16), tc.cot()), transposed ? 4 : 1);
  return RCV_OK;
}

int rcv_launch_conv(const rcv_handle* h, const rcv_op* op, hipStream_t s, OpQuery* query) {
  const bool transposed = op->kind == RCV_OP_TCONV;
  ConvPlan pl;
  int rc = make_plan(h, op, transposed, &pl);
  if (rc) return rc;
  const int Cout = op->i[RCV_I_COUT];
  const int n_part = pl.grid.x * pl.grid.z;
  if (query) {
    const TileCfg& tc = kTiles[pl.tile];
    snprintf(query->label, sizeof(query->label), "%s_mfma<%d,%d,%d,%d,%d>", transposed ? "tconv" : "conv", tc.WM, tc.WN, tc.WAVES_M,
             tc.WAVES_N, pl.CK);
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
  a.CinP = round_up(a.Cin, 4); a.CoutP = round_up(Cout, 16);
  a.stride = op->i[RCV_I_STRIDE]; a.dil = op->i[RCV_I_DIL];
  a.R = pl.R; a.Wt = pl.Wt; a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y; a.IH = pl.IH; a.IW = pl.IW;
  a.in_mode = op->i[RCV_I_INMODE]; a.stats = op->i[RCV_I_STATS]; a.flags = op->flags;
  a.wl_floats = pl.wl_floats;
  a.fdWt = make_fastdiv(pl.Wt); a.fdIW = make_fastdiv(pl.IW);
  RCV_CHECK_ARG(a.in && a.w && a.out, "conv: null operand");
  RCV_CHECK_ARG(a.in_mode == RCV_LOAD_PLAIN || a.in_mode == RCV_LOAD_NCHW || a.in_c, "conv: load mode %d needs constants", a.in_mode);
  RCV_CHECK_ARG(!(a.in_mode == RCV_LOAD_GRAD_ENC || a.in_mode == RCV_LOAD_GRAD_DEC) || a.in_aux, "conv: gradient load needs aux tensor");
  RCV_CHECK_ARG(!(a.flags & RCV_F_BIAS) || a.bias, "conv: bias flag without bias");
  RCV_CHECK_ARG(!(a.flags & RCV_F_RESID) || a.resid, "conv: resid flag without tensor");
  RCV_CHECK_ARG(a.stats == RCV_STATS_NONE || a.part, "conv: statistics requested without workspace");
  RCV_CHECK_ARG(a.stats == RCV_STATS_NONE || op->i[RCV_I_NPART] == n_part, "conv: workspace rows %d != %d", op->i[RCV_I_NPART], n_part);
  RCV_CHECK_ARG(!(a.stats == RCV_STATS_BWD_ENC || a.stats == RCV_STATS_BWD_DEC) || a.epi_aux, "conv: backward statistics need epi_aux");
  RCV_CHECK_ARG(a.stats != RCV_STATS_BWD_DEC || a.epi_c, "conv: decoder statistics need epi_c");
  if (transposed) {
    return pl.CK == 8 ? launch_tile<8, true>(pl.tile, a, pl.grid, pl.lds, s) : launch_tile<4, true>(pl.tile, a, pl.grid, pl.lds, s);
  }
  return pl.CK == 8 ? launch_tile<8, false>(pl.tile, a, pl.grid, pl.lds, s) : launch_tile<4, false>(pl.tile, a, pl.grid, pl.lds, s);
}

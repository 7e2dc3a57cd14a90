// Shared declarations of the convolution kernels (conv_mfma.hip: general; convs_mfma.hip: narrow layers).
#pragma once
#include "rcv_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { KIND_GATHER = 0, KIND_TPHASE = 1, KIND_TMERGED = 2, KIND_TALL = 3 };   // TALL: all four output parities of a transposed conv in one workgroup

struct ConvArgs {
  const float* in;
  const float* in_aux;
  const float* in_c;   // [5][Cin]
  const float* w;      // [taps][CinP][CoutP]
  const float* bias;
  float* out;
  const float* resid;
  const float* epi_aux;
  const float* epi_c;  // [2][Cout]
  float* part;         // [n_part][2][Cout]
  int N, H, W, Cin, Cout, Ho, Wo;
  int CinP, CoutP, CoutV;      // padded rows / cols of the packed filter; number of (virtual) output channels
  int stride, dil;
  int R, Wt, tiles_x, tiles_y, IH, IW;
  int n_pix_tiles, n_co_tiles, n_sub, total_tiles, nchunks;
  int in_mode, stats;
  uint32_t flags;
  int wl_floats, xl_floats;    // LDS carve: filter tile, input tile (then constants, then reduction scratch)
  int xpitch;                  // floats per staged input pixel (see conv_xpitch)
  int xk;                      // conv_dma_kernel: channels per staged input chunk (8 or 16)
  FastDiv fdWt, fdIW;
  FastDiv fdTX, fdTY;          // tiles_x, tiles_y (narrow kernel: tile decode on the scalar unit)
#ifdef RCV_STAMPS
  unsigned long long* stamps;  // diagnostic build: per wave [12] cycle sums (conv_dma_kernel)
#endif
};

struct TileInfo {
  int n, y0, x0, co0, py, px, oy0, ox0, pt;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int KIND>
__device__ __forceinline__ TileInfo decode_tile(const ConvArgs& a, int t, int COT) {
  TileInfo ti;
  const int sub = t % a.n_sub;
  int pt = t / a.n_sub;
  ti.pt = pt;
  const int co_tile = sub % a.n_co_tiles, phase = sub / a.n_co_tiles;
  const int tx_i = pt % a.tiles_x;
  pt /= a.tiles_x;
  const int ty_i = pt % a.tiles_y;
  ti.n = pt / a.tiles_y;
  ti.y0 = ty_i * a.R;
  ti.x0 = tx_i * a.Wt;
  ti.co0 = co_tile * COT;
  ti.py = phase >> 1;
  ti.px = phase & 1;
  if (KIND == KIND_GATHER) { ti.oy0 = ti.y0 * a.stride - a.dil; ti.ox0 = ti.x0 * a.stride - a.dil; }
  else { ti.oy0 = ti.y0; ti.ox0 = ti.x0; }
  return ti;
}

// Operand transform applied while the tile is written to LDS (see RCV_LOAD_* in rcv.h).
template <int MODE>
__device__ __forceinline__ float4 xform4(float4 x, float4 a, const float4 (&k)[5]) {
  float4 v;
  if (MODE == RCV_LOAD_PLAIN || MODE == RCV_LOAD_NCHW) {
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


struct ConvPlan {
  int kind, tile, CK, R, Wt, tiles_x, tiles_y, IH, IW;
  int CoutV, CoutP, n_co_tiles, n_phases, total_tiles, grid;
  size_t lds;
  int wl_floats, xl_floats;
  int narrow;            // 1: convs_mfma.hip (persistent, filter resident in LDS)
  int dma;               // 1: conv_dma_kernel (filter chunks by LDS-DMA, double buffered)
  int xk;                // conv_dma_kernel: channels per staged input chunk
  int WM, WN, XMAX;      // narrow kernel template selection
  int first;             // 1: conv_first.hip (3 -> 8 channel first layer on the vector ALU)
  int dev;               // device of the handle the plan belongs to (per-device kernel attributes)
  int wino;              // 1: conv_wino.hip (Winograd F(2x2,3x3) for the wide stride-1 layers)
  int bf3;               // 1: conv_bf3.hip (wide stride-1 layers: fp32 products on the bf16 matrix pipe, filter in the split layout); 2: convn_bf3.hip (narrow layers)
  int small;             // 1: conv_small.hip (inference on planes of a few hundred pixels: one MFMA block per workgroup, K split over its waves)
};

// LDS pitch of a staged input pixel holding CK channels.  The B operand of v_mfma_f32_16x16x4_f32 is read with ds_read_b32 at
// (pixel(l15) * IS * pitch + l4): 16 pixels x 2 channel lanes per 32-lane half must fall on 32 different banks.  Unit-stride pixel
// blocks: pitch = CK + 2 (pitch/2 odd => 16 distinct even banks, + l4 the odd ones); stride-2 blocks: the lane stride is 2*pitch,
// so pitch = CK + 1 (odd).  (CK + 1 everywhere cost 46 % of all LDS cycles in bank conflicts on the unit-stride layers.)
static inline int conv_xpitch(int CK, int pixel_stride) {
  if (const char* ev = RCV_ENV("RCV_XPITCH_PAD")) return CK + atoi(ev);      // experiments build only
  return pixel_stride == 1 ? CK + 2 : CK + 1;
}

static inline void tile_halo(int kind, int R, int Wt, int s, int d, int* IH, int* IW) {
  if (kind == KIND_GATHER) { *IH = (R - 1) * s + 2 * d + 1; *IW = (Wt - 1) * s + 2 * d + 1; }
  else { *IH = R + 1; *IW = Wt + 1; }
}

// Winograd kernel (conv_wino.hip)
bool conv_wino_supported(const rcv_handle* h, const rcv_op* op, int kind);
bool conv_wino_wanted(const rcv_handle* h, const rcv_op* op, bool force);
int conv_wino_plan(const rcv_handle* h, const rcv_op* op, ConvPlan* pl);
int conv_wino_launch(const ConvPlan& pl, const ConvArgs& a, hipStream_t s);

// split-bf16 kernel (conv_bf3.hip)
bool conv_bf3_wanted(const rcv_handle* h, const rcv_op* op);
bool conv_bf3_supported(const rcv_handle* h, const rcv_op* op, int kind);
int conv_bf3_plan(const rcv_handle* h, const rcv_op* op, ConvPlan* pl);
int conv_bf3_launch(const ConvPlan& pl, const ConvArgs& a, hipStream_t s);

// split-bf16 narrow-layer kernel (convn_bf3.hip)
bool convn_bf3_wanted(const rcv_handle* h, const rcv_op* op);
bool convn_bf3_supported(const rcv_handle* h, const rcv_op* op, int kind);
int convn_bf3_plan(const rcv_handle* h, const rcv_op* op, int kind, ConvPlan* pl);
int convn_bf3_launch(const ConvPlan& pl, const ConvArgs& a, hipStream_t s);

// tiny-plane inference kernel (conv_small.hip)
bool conv_small_supported(const rcv_handle* h, const rcv_op* op, int kind);
int conv_small_plan(const rcv_handle* h, const rcv_op* op, ConvPlan* pl);
int conv_small_launch(const ConvPlan& pl, const ConvArgs& a, hipStream_t s);

// first-layer kernel (conv_first.hip)
bool conv_first_supported(const rcv_op* op, int kind);
int conv_first_plan(const rcv_handle* h, const rcv_op* op, ConvPlan* pl);
int conv_first_launch(const ConvPlan& pl, const ConvArgs& a, hipStream_t s);

// narrow-layer kernel (convs_mfma.hip)
bool convs_supported(const rcv_handle* h, const rcv_op* op, int kind, int CinP, int CoutV);
int convs_make_plan(const rcv_handle* h, const rcv_op* op, int kind, ConvPlan* pl);
int convs_launch(const ConvPlan& pl, const ConvArgs& a, bool two, hipStream_t s);

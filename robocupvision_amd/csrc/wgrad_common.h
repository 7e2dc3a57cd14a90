// Shared declarations of the filter-gradient kernels (wgrad_mfma.hip: matrix cores; wgrad_first.hip: first layer on the vector ALU).
#pragma once
#include "rcv_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WgradArgs {
  const float* g; const float* g_aux; const float* g_c;
  const float* p; const float* p_aux; const float* p_c;
  float* part;        // [nsplit][9][CBP][CAP]
  float* part_bias;   // [nsplit][CBP] or null
  int g_mode, p_mode;
  int N, H, W, Hp, Wp, CA, CB, CAP, CBP;
  int stride, dil;
  int R, Wt, Wt4, tiles_x, tiles_y, ntiles, IH, IW, SP, SG;
  int nsplit, nctiles;
  uint32_t dbg;
  int pl_floats, gl_floats;      // LDS carve: P tile, G tile (then the load constants)
  FastDiv fdWt4, fdIW;
};

__device__ __forceinline__ float4 wld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int MODE>
__device__ __forceinline__ float4 wxform4(float4 x, float4 a, const float4 (&k)[5]) {
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
  } else {
    v.x = fmaf(k[0].x, (fmaf(a.x, k[3].x, k[4].x) > 0.f ? x.x : 0.f), fmaf(k[2].x, a.x, k[1].x));
    v.y = fmaf(k[0].y, (fmaf(a.y, k[3].y, k[4].y) > 0.f ? x.y : 0.f), fmaf(k[2].y, a.y, k[1].y));
    v.z = fmaf(k[0].z, (fmaf(a.z, k[3].z, k[4].z) > 0.f ? x.z : 0.f), fmaf(k[2].z, a.z, k[1].z));
    v.w = fmaf(k[0].w, (fmaf(a.w, k[3].w, k[4].w) > 0.f ? x.w : 0.f), fmaf(k[2].w, a.w, k[1].w));
  }
  return v;
}

__device__ __forceinline__ float4 wxform_rt(int mode, float4 x, float4 a, const float4 (&k)[5]) {
  switch (mode) {
    case RCV_LOAD_PLAIN: case RCV_LOAD_NCHW: return x;
    case RCV_LOAD_AFFINE: return wxform4<RCV_LOAD_AFFINE>(x, a, k);
    case RCV_LOAD_AFFINE_RELU: return wxform4<RCV_LOAD_AFFINE_RELU>(x, a, k);
    case RCV_LOAD_GRAD_ENC: return wxform4<RCV_LOAD_GRAD_ENC>(x, a, k);
    default: return wxform4<RCV_LOAD_GRAD_DEC>(x, a, k);
  }
}

// padded gathered-channel count of the partial-filter layout [split][tap][CBP][CAP] (shared by the kernels and RCV_OP_WGRAD_REDUCE)
static inline int wgrad_cap(int CA) { return CA <= 4 ? 4 : round_up(CA, 16); }

// first-layer kernel (wgrad_first.hip)
bool wgrad_first_supported(const rcv_op* op);
int wgrad_first_nsplit(const rcv_handle* h, const rcv_op* op);
int wgrad_first_launch(const rcv_handle* h, const WgradArgs& a, hipStream_t s);

// wide stride-1 layers on the bf16 matrix pipe (wgrad_bf3.hip)
bool wgrad_bf3_supported(const rcv_handle* h, const rcv_op* op);
void wgrad_bf3_geometry(const rcv_handle* h, const rcv_op* op, int* tw, int* tiles_x, int* tiles_y, int* nsplit, int* nctiles);
int wgrad_bf3_launch(const rcv_handle* h, const WgradArgs& a, int tw, hipStream_t s);

// narrow layers on the bf16 matrix pipe (wgradn_bf3.hip)
bool wgradn_bf3_supported(const rcv_handle* h, const rcv_op* op);
void wgradn_bf3_geometry(const rcv_handle* h, const rcv_op* op, int* th, int* tiles_x, int* tiles_y, int* ngroups);
int wgradn_bf3_launch(const rcv_handle* h, const WgradArgs& a, int th, int ngroups, hipStream_t s);

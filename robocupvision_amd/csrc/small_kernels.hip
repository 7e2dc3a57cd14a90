// Memory-bound and bookkeeping kernels of the ROBO-UNet step (everything that is not a 3x3
// contraction).  All of them are HBM-bound streaming kernels or tiny per-channel reductions:
// 16-byte NHWC vector accesses, plane-coalesced NCHW accesses for the image / logits, wave
// shuffles + fixed-order LDS trees for the reductions (no float atomics).
#include "rcv_internal.h"

__device__ __forceinline__ float4 sld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void sst4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// streaming (nontemporal) accesses for tensors a kernel touches exactly once and nobody reads soon after: measured on this part
// (scripts/micro/stream_bw.hip) a 2-reads-1-write stream gains 4-5 % and a read-only one 8 % over plain accesses
typedef float sv4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 sld4_nt(const float* p) {
  const sv4f v = __builtin_nontemporal_load(reinterpret_cast<const sv4f*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int sh = 32; sh >= 1; sh >>= 1) v += __shfl_xor(v, sh);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int sh = 32; sh >= 1; sh >>= 1) v += __shfl_xor(v, sh);
  return v;
}

// ------------------------------------------------------------------------------------------
// RCV_OP_PACK: parameter [D0][D1][3][3] -> [9][rows_pad][cols_pad] (zero padded), table driven
// ------------------------------------------------------------------------------------------
__global__ void pack_kernel(const rcv_pack_job* __restrict__ jobs) {
  const rcv_pack_job jb = jobs[blockIdx.y];
  const int per_tap = jb.rows_pad * jb.cols_pad;
  if (jb.merged == 2) {
    // Winograd F(2x2,3x3) filter transform U = G g G^T (conv_wino.hip): dst [16 xi = (a,b)][rows_pad][cols_pad],
    //   G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]
    const int rows = jb.rows_from_d1 ? jb.D1 : jb.D0;
    const int cols = jb.rows_from_d1 ? jb.D0 : jb.D1;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < per_tap; e += gridDim.x * blockDim.x) {
      const int row = e / jb.cols_pad, col = e - row * jb.cols_pad;
      float g[3][3];
      const bool ok = row < rows && col < cols;
      const int d0 = jb.rows_from_d1 ? col : row, d1 = jb.rows_from_d1 ? row : col;
      const float sc = (ok && jb.scale) ? jb.scale[col] : 1.f;       // (G g G^T is linear in g: the per-channel factor commutes)
#pragma unroll
      for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = ok ? sc * jb.src[((size_t)d0 * jb.D1 + d1) * 9 + (jb.flip ? 8 - t : t)] : 0.f;
      float tg[4][3];      // G g
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        tg[0][q] = g[0][q];
        tg[1][q] = 0.5f * (g[0][q] + g[1][q] + g[2][q]);
        tg[2][q] = 0.5f * (g[0][q] - g[1][q] + g[2][q]);
        tg[3][q] = g[2][q];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        jb.dst[(size_t)(a * 4 + 0) * per_tap + e] = tg[a][0];
        jb.dst[(size_t)(a * 4 + 1) * per_tap + e] = 0.5f * (tg[a][0] + tg[a][1] + tg[a][2]);
        jb.dst[(size_t)(a * 4 + 2) * per_tap + e] = 0.5f * (tg[a][0] - tg[a][1] + tg[a][2]);
        jb.dst[(size_t)(a * 4 + 3) * per_tap + e] = tg[a][2];
      }
    }
    return;
  }
  const bool par = jb.merged == 1 || jb.merged == 4;        // merged-parity layout of a transposed conv (4: the same, split into bf16)
  const int total = (par ? 4 : 9) * per_tap;
  const int rows = jb.rows_from_d1 ? jb.D1 : jb.D0;
  const int cols = jb.rows_from_d1 ? jb.D0 : jb.D1;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int t = e / per_tap;
    const int rc = e - t * per_tap;
    // plain layouts: the column is the fastest index of the destination; split layouts ([..][col][32 k]): the row is -- consecutive
    // threads then write consecutive bf16 (with the column fastest every 2-byte store of a wave hit its own 64-byte line: 41 us per step)
    const bool row_fast = jb.merged >= 3;
    const int row = row_fast ? rc % jb.rows_pad : rc / jb.cols_pad;
    int col = row_fast ? rc / jb.rows_pad : rc - row * jb.cols_pad;
    const int vcol = col;               // (virtual) column of the packed layout
    float v = 0.f;
    int ts = jb.flip ? 8 - t : t;
    bool ok = row < rows && col < cols;
    if (par) {
      // virtual column = parity*cols + col; tap t = (dy,dx) of the 2x2 input window.
      // output row 2y+py takes input row y+dy through filter row ky: py=0: (dy=0,ky=1); py=1: (dy=0,ky=2),(dy=1,ky=0)
      const int ph = col / cols;
      col -= ph * cols;
      const int py = ph >> 1, px = ph & 1, dy = t >> 1, dx = t & 1;
      const int ky = py ? (dy ? 0 : 2) : (dy ? -1 : 1);
      const int kx = px ? (dx ? 0 : 2) : (dx ? -1 : 1);
      ok = row < rows && ph < 4 && ky >= 0 && kx >= 0;
      ts = ky * 3 + kx;
    }
    if (ok) {
      const int d0 = jb.rows_from_d1 ? col : row;
      const int d1 = jb.rows_from_d1 ? row : col;
      v = jb.src[((size_t)d0 * jb.D1 + d1) * 9 + ts];
      if (jb.scale) v *= jb.scale[col];
    }
    if (jb.merged >= 3) {
      // split-bf16 layouts of conv_bf3.hip / convn_bf3.hip: v = h + m + l exactly (three round-to-nearest bf16 steps, every remainder
      // exact); with k = tap * rows_pad + row, dst = [plane][k / 32][col][k % 32] bf16 (k-steps padded to 32, zero beyond the last tap)
      // -- the eight consecutive k a lane of v_mfma_f32_16x16x32_bf16 holds are 16 contiguous bytes
      __bf16* d16 = reinterpret_cast<__bf16*>(jb.dst);
      // (layout 5, the stride-2 wide convs: 16-channel chunks, k = tap * 16 + row % 16 inside chunk row / 16, five k-steps per chunk)
      const bool ch16 = jb.merged == 5;
      const int kk = ch16 ? t * 16 + (row & 15) : t * jb.rows_pad + row;
      const int kstep = ch16 ? (row >> 4) * 5 + (kk >> 5) : (kk >> 5);
      const size_t plane = (size_t)(ch16 ? (jb.rows_pad >> 4) * 5 : ((par ? 4 : 9) * jb.rows_pad + 31) >> 5) * jb.cols_pad * 32;
      const size_t o = ((size_t)kstep * jb.cols_pad + vcol) * 32 + (kk & 31);
      const __bf16 hh = (__bf16)v;
      const float r1 = v - (float)hh;
      const __bf16 mm = (__bf16)r1;
      const __bf16 ll = (__bf16)(r1 - (float)mm);
      d16[o] = hh; d16[plane + o] = mm; d16[2 * plane + o] = ll;
    } else {
      jb.dst[e] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// BatchNorm bookkeeping (one workgroup per channel; double accumulation over the partial rows)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_sum2_d(double& a, double& b) {
  __shared__ double sh[2][8];
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[0][w] = a; sh[1][w] = b; }
  __syncthreads();
  a = 0.0; b = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { a += sh[0][i]; b += sh[1][i]; }
}

__global__ void bn_finalize_kernel(const float* __restrict__ part, int n_part, int C, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* running_mean, float* running_var, float momentum,
                                   float eps, int training, float* consts, float* save_mean, float* save_istd) {
  const int c = blockIdx.x;
  // the channel's parameters are requested before the partial rows (uniform addresses: scalar loads), not after the reduction: they
  // miss in L2 once per step and the kernel is nothing but a chain of memory round trips
  const float g_c = gamma[c], b_c = beta[c];
  const bool upd = training && running_mean;
  const float rm_c = upd ? running_mean[c] : 0.f, rv_c = upd ? running_var[c] : 0.f;
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < n_part; i += blockDim.x) {
    s1 += (double)part[((size_t)i * 2 + 0) * C + c];
    s2 += (double)part[((size_t)i * 2 + 1) * C + c];
  }
  block_sum2_d(s1, s2);
  if (threadIdx.x == 0) {
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float istd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = g_c * istd;
    consts[0 * C + c] = sc;
    consts[1 * C + c] = b_c - (float)mean * sc;
    consts[2 * C + c] = (float)mean;          // row 2: batch mean (the backward sums are taken about it)
    consts[3 * C + c] = 0.f; consts[4 * C + c] = 0.f;
    save_mean[c] = (float)mean;
    save_istd[c] = istd;
    if (upd) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (1.f - momentum) * rm_c + momentum * (float)mean;
      running_var[c] = (1.f - momentum) * rv_c + momentum * (float)unbiased;
    }
  }
}

__global__ void bn_eval_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ rm, const float* __restrict__ rv, float eps, float* consts,
                               const float* __restrict__ conv_bias) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float istd = 1.f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * istd;
    const float sh = beta[c] - rm[c] * sc;
    consts[0 * C + c] = sc;
    consts[1 * C + c] = sh;
    consts[2 * C + c] = 0.f;
    consts[3 * C + c] = fmaf(conv_bias ? conv_bias[c] : 0.f, sc, sh);      // the conv's bias seen through the BatchNorm (inference folding)
    consts[4 * C + c] = 0.f;
  }
}

// dr = A*g + B + C*r  with  A = gamma*istd,  C = -A*istd*Sgx/M,  B = -A*Sg/M - C*mean,
// Sgx = istd*(Sgr - mean*Sg);  dgamma = Sgx, dbeta = Sg.   (aten::native_batch_norm_backward)
__global__ void bn_bwd_kernel(const float* __restrict__ part, int n_part, int C, double count, const float* __restrict__ gamma,
                              const float* __restrict__ save_mean, const float* __restrict__ save_istd,
                              const float* __restrict__ fwd_consts, float* consts, float* dgamma, float* dbeta) {
  const int c = blockIdx.x;
  // (the channel's scalars first, see bn_finalize_kernel)
  const float mean_c = save_mean[c], istd_c = save_istd[c], g_c = gamma[c];
  const float f0_c = fwd_consts[0 * C + c], f1_c = fwd_consts[1 * C + c];
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < n_part; i += blockDim.x) {
    s1 += (double)part[((size_t)i * 2 + 0) * C + c];
    s2 += (double)part[((size_t)i * 2 + 1) * C + c];
  }
  block_sum2_d(s1, s2);
  if (threadIdx.x == 0) {
    // s2 arrives centred: sum g*(r - mean) (accumulated about the batch mean to avoid cancellation)
    const double mean = mean_c, istd = istd_c;
    const double sgx = istd * s2;
    const double A = (double)g_c * istd;
    const double Cc = -A * istd * sgx / count;
    const double B = -A * s1 / count - Cc * mean;
    consts[0 * C + c] = (float)A;
    consts[1 * C + c] = (float)B;
    consts[2 * C + c] = (float)Cc;
    consts[3 * C + c] = f0_c;
    consts[4 * C + c] = f1_c;
    if (dgamma) dgamma[c] = (float)sgx;
    if (dbeta) dbeta[c] = (float)s1;
  }
}

// ------------------------------------------------------------------------------------------
// RCV_OP_COMBINE: up = relu(t*c0+c1) + f(r)   f = affine / affine+relu / identity  (model.py:509)
// ------------------------------------------------------------------------------------------
template <int MODE2>
__global__ void combine_kernel(const float* __restrict__ t, const float* __restrict__ tc, const float* __restrict__ r,
                               const float* __restrict__ rc, float* __restrict__ out, size_t n4, int C) {
  const int C4 = C / 4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(e % C4) * 4;
    const float4 a = sld4(t + e * 4), s = sld4(tc + ch), h = sld4(tc + C + ch);
    float4 v;
    v.x = fmaxf(fmaf(a.x, s.x, h.x), 0.f); v.y = fmaxf(fmaf(a.y, s.y, h.y), 0.f);
    v.z = fmaxf(fmaf(a.z, s.z, h.z), 0.f); v.w = fmaxf(fmaf(a.w, s.w, h.w), 0.f);
    float4 b = sld4(r + e * 4);
    if (MODE2 != RCV_LOAD_PLAIN) {
      const float4 s2 = sld4(rc + ch), h2 = sld4(rc + C + ch);
      b.x = fmaf(b.x, s2.x, h2.x); b.y = fmaf(b.y, s2.y, h2.y); b.z = fmaf(b.z, s2.z, h2.z); b.w = fmaf(b.w, s2.w, h2.w);
      if (MODE2 == RCV_LOAD_AFFINE_RELU) { b.x = fmaxf(b.x, 0.f); b.y = fmaxf(b.y, 0.f); b.z = fmaxf(b.z, 0.f); b.w = fmaxf(b.w, 0.f); }
    }
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    sst4(out + e * 4, v);
  }
}

// RCV_F_CONCAT: up[.., 0:C] = relu(t*c0+c1), up[.., C:2C] = f(r)   (torch.cat([layer(up), skip], 1), model.py:507)
__global__ void concat_kernel(const float* __restrict__ t, const float* __restrict__ tc, const float* __restrict__ r,
                              const float* __restrict__ rc, float* __restrict__ out, size_t n4, int C, int mode2) {
  const int C4 = C / 4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(e % C4) * 4;
    const size_t pix = e / C4;
    const float4 a = sld4(t + e * 4), s = sld4(tc + ch), h = sld4(tc + C + ch);
    float4 v;
    v.x = fmaxf(fmaf(a.x, s.x, h.x), 0.f); v.y = fmaxf(fmaf(a.y, s.y, h.y), 0.f);
    v.z = fmaxf(fmaf(a.z, s.z, h.z), 0.f); v.w = fmaxf(fmaf(a.w, s.w, h.w), 0.f);
    float4 b = sld4(r + e * 4);
    if (mode2 != RCV_LOAD_PLAIN) {
      const float4 s2 = sld4(rc + ch), h2 = sld4(rc + C + ch);
      b.x = fmaf(b.x, s2.x, h2.x); b.y = fmaf(b.y, s2.y, h2.y); b.z = fmaf(b.z, s2.z, h2.z); b.w = fmaf(b.w, s2.w, h2.w);
      if (mode2 == RCV_LOAD_AFFINE_RELU) { b.x = fmaxf(b.x, 0.f); b.y = fmaxf(b.y, 0.f); b.z = fmaxf(b.z, 0.f); b.w = fmaxf(b.w, 0.f); }
    }
    sst4(out + pix * 2 * C + ch, v);
    sst4(out + pix * 2 * C + C + ch, b);
  }
}

// ------------------------------------------------------------------------------------------
// 1x1 classifier (model.py:411): NHWC [.,CIN] -> NCHW logits [N][COUT][H][W]
// ------------------------------------------------------------------------------------------
#define CLS_MAX_OUT 8
#define CE_MAX_C 8
// FUSED: the classifier input up = relu(t*c0+c1) + f(r) (decoder block output + skip, model.py:509) is formed here from the
// block's raw tensors instead of being materialised by RCV_OP_COMBINE (saves one tensor write and one read at full resolution).
// (raw loads and the arithmetic are separate so that the streaming loops can request pixel i+1 before they work on pixel i)
template <int CIN, bool FUSED>
struct ClsRaw { float4 a[CIN / 4]; float4 b[FUSED ? CIN / 4 : 1]; };

// rch: channels per pixel of the skip tensor r (CIN for the decoder's skip add; fewer for LabelProp's `x[:, 0:8] += top`, model.py:565:
// the skip then reaches only the first rch input channels)
template <int CIN, bool FUSED>
__device__ __forceinline__ void cls_load_raw(ClsRaw<CIN, FUSED>& o, const float* __restrict__ x, const float* __restrict__ r, size_t p, int rch = CIN) {
#pragma unroll
  for (int q = 0; q < CIN / 4; ++q) {
    o.a[q] = sld4_nt(x + p * CIN + 4 * q);
    if (FUSED) o.b[FUSED ? q : 0] = 4 * q < rch ? sld4_nt(r + p * rch + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

template <int CIN, bool FUSED>
__device__ __forceinline__ void cls_form_up(float (&v)[CIN], const ClsRaw<CIN, FUSED>& raw, const float* __restrict__ tc,
                                            const float* __restrict__ rc, int mode2, int rch = CIN) {
#pragma unroll
  for (int q = 0; q < CIN / 4; ++q) {
    float4 a = raw.a[q];
    if (FUSED) {
      const float4 s = sld4(tc + 4 * q), h = sld4(tc + CIN + 4 * q);
      a.x = fmaxf(fmaf(a.x, s.x, h.x), 0.f); a.y = fmaxf(fmaf(a.y, s.y, h.y), 0.f);
      a.z = fmaxf(fmaf(a.z, s.z, h.z), 0.f); a.w = fmaxf(fmaf(a.w, s.w, h.w), 0.f);
      float4 b = raw.b[FUSED ? q : 0];
      if (4 * q >= rch) b = make_float4(0.f, 0.f, 0.f, 0.f);                 // input channels the skip does not reach
      else if (mode2 != RCV_LOAD_PLAIN) {
        const float4 s2 = sld4(rc + 4 * q), h2 = sld4(rc + rch + 4 * q);
        b.x = fmaf(b.x, s2.x, h2.x); b.y = fmaf(b.y, s2.y, h2.y); b.z = fmaf(b.z, s2.z, h2.z); b.w = fmaf(b.w, s2.w, h2.w);
        if (mode2 == RCV_LOAD_AFFINE_RELU) { b.x = fmaxf(b.x, 0.f); b.y = fmaxf(b.y, 0.f); b.z = fmaxf(b.z, 0.f); b.w = fmaxf(b.w, 0.f); }
      }
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
  }
}

template <int CIN, bool FUSED>
__device__ __forceinline__ void cls_load_up(float (&v)[CIN], const float* __restrict__ x, const float* __restrict__ tc,
                                            const float* __restrict__ r, const float* __restrict__ rc, int mode2, size_t p, int rch = CIN) {
  ClsRaw<CIN, FUSED> raw;
  cls_load_raw<CIN, FUSED>(raw, x, r, p, rch);
  cls_form_up<CIN, FUSED>(v, raw, tc, rc, mode2, rch);
}

// CE: the weighted cross-entropy partial sums, the arg-max mask and the pixel-accuracy count of RCV_OP_CE_FWD are taken from the
// logits while they are still in registers (same pixel -> thread assignment and summation order as ce_fwd_kernel: bit-identical loss).
template <int CIN, bool FUSED, bool CE>
__global__ void cls_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                               float* __restrict__ out, int N, int HW, int COUT, const float* __restrict__ tc,
                               const float* __restrict__ r, const float* __restrict__ rc, int mode2,
                               const int64_t* __restrict__ target, const float* __restrict__ cw, float* __restrict__ part,
                               uint8_t* __restrict__ argmax, int rch) {
  __shared__ float ws[CLS_MAX_OUT * CIN + CLS_MAX_OUT];
  __shared__ double sh[3][4];
  for (int e = threadIdx.x; e < COUT * CIN; e += blockDim.x) ws[e] = w[e];
  for (int e = threadIdx.x; e < COUT; e += blockDim.x) ws[CLS_MAX_OUT * CIN + e] = bias ? bias[e] : 0.f;
  __syncthreads();
  double a_nll = 0.0, a_w = 0.0, a_ok = 0.0;
  const size_t total = (size_t)N * HW;
  // (no software prefetch here: measured 0.205 vs 0.194 ms -- this kernel already streams at 4.7 TB/s; the backward one gained 30 %)
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
    float v[CIN];
    cls_load_up<CIN, FUSED>(v, x, tc, r, rc, mode2, p, rch);
    const size_t n = p / HW, hw = p % HW;
    float lg[CLS_MAX_OUT];
#pragma unroll
    for (int c = 0; c < CLS_MAX_OUT; ++c) {
      if (c < COUT) {
        float u = ws[CLS_MAX_OUT * CIN + c];
#pragma unroll
        for (int k = 0; k < CIN; ++k) u = fmaf(v[k], ws[c * CIN + k], u);
        __builtin_nontemporal_store(u, out + (n * COUT + c) * HW + hw);
        lg[c] = u;
      }
    }
    if (CE) {
      float mx = -INFINITY;
      int am = 0;
#pragma unroll
      for (int c = 0; c < CLS_MAX_OUT; ++c) if (c < COUT && lg[c] > mx) { mx = lg[c]; am = c; }
      float se = 0.f;
#pragma unroll
      for (int c = 0; c < CLS_MAX_OUT; ++c) if (c < COUT) se += expf(lg[c] - mx);
      const int tg = (int)target[p];
      float vt = 0.f;
#pragma unroll
      for (int c = 0; c < CLS_MAX_OUT; ++c) if (c == tg) vt = lg[c];
      const float wt = (unsigned)tg < (unsigned)COUT ? (cw ? cw[tg] : 1.f) : 0.f;   // label outside [0, C) (e.g. -100): ignored, like NLLLoss's ignore_index
      const float nll = (mx - vt) + logf(se);
      a_nll += (double)(wt * nll);
      a_w += (double)wt;
      a_ok += (am == tg) ? 1.0 : 0.0;
      if (argmax) argmax[p] = (uint8_t)am;
    }
  }
  if (CE) {
    a_nll = wave_sum_d(a_nll); a_w = wave_sum_d(a_w); a_ok = wave_sum_d(a_ok);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][wv] = a_nll; sh[1][wv] = a_w; sh[2][wv] = a_ok; }
    __syncthreads();
    if (threadIdx.x < 3) {
      double t = 0.0;
      for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[threadIdx.x][i];
      part[(size_t)blockIdx.x * 3 + threadIdx.x] = (float)t;
    }
  }
}

// 16-channel classifier (LabelProp, model.py:567): FOUR lanes per pixel, one 16-byte channel quad each, so that every load instruction
// of a wave reads 1 KB of consecutive memory (with a lane per pixel each of its four loads touched 64 different lines: 1.5 TB/s).  The
// partial dot products meet through two butterfly steps; lane q of a pixel then stores classes q and q + 4.
template <bool FUSED>
__global__ __launch_bounds__(256) void cls_fwd16_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ out, int N, int HW, int COUT, const float* __restrict__ tc,
                                                        const float* __restrict__ r, const float* __restrict__ rc, int mode2, int rch) {
  constexpr int CIN = 16;
  __shared__ float ws[CLS_MAX_OUT * CIN + CLS_MAX_OUT];
  for (int e = threadIdx.x; e < CLS_MAX_OUT * CIN; e += blockDim.x) ws[e] = e < COUT * CIN ? w[e] : 0.f;
  for (int e = threadIdx.x; e < CLS_MAX_OUT; e += blockDim.x) ws[CLS_MAX_OUT * CIN + e] = (bias && e < COUT) ? bias[e] : 0.f;
  __syncthreads();
  const int q = threadIdx.x & 3;
  float4 s = make_float4(1.f, 1.f, 1.f, 1.f), h = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s, h2 = h;
  if (FUSED) {
    s = sld4(tc + 4 * q); h = sld4(tc + CIN + 4 * q);
    if (4 * q < rch && mode2 != RCV_LOAD_PLAIN) { s2 = sld4(rc + 4 * q); h2 = sld4(rc + rch + 4 * q); }
  }
  const bool has_skip = FUSED && 4 * q < rch;
  const size_t total = (size_t)N * HW;
  for (size_t p = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; p < total; p += ((size_t)gridDim.x * blockDim.x) >> 2) {
    float4 a = sld4(x + p * CIN + 4 * q);
    if (FUSED) {
      a.x = fmaxf(fmaf(a.x, s.x, h.x), 0.f); a.y = fmaxf(fmaf(a.y, s.y, h.y), 0.f);
      a.z = fmaxf(fmaf(a.z, s.z, h.z), 0.f); a.w = fmaxf(fmaf(a.w, s.w, h.w), 0.f);
      if (has_skip) {
        float4 b = sld4(r + p * rch + 4 * q);
        if (mode2 != RCV_LOAD_PLAIN) {
          b.x = fmaf(b.x, s2.x, h2.x); b.y = fmaf(b.y, s2.y, h2.y); b.z = fmaf(b.z, s2.z, h2.z); b.w = fmaf(b.w, s2.w, h2.w);
          if (mode2 == RCV_LOAD_AFFINE_RELU) { b.x = fmaxf(b.x, 0.f); b.y = fmaxf(b.y, 0.f); b.z = fmaxf(b.z, 0.f); b.w = fmaxf(b.w, 0.f); }
        }
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
    }
    float lg[CLS_MAX_OUT];
#pragma unroll
    for (int c = 0; c < CLS_MAX_OUT; ++c) {
      const float* wc = ws + c * CIN + 4 * q;
      float u = fmaf(a.x, wc[0], fmaf(a.y, wc[1], fmaf(a.z, wc[2], a.w * wc[3])));
      u += __shfl_xor(u, 1);
      u += __shfl_xor(u, 2);
      lg[c] = u + ws[CLS_MAX_OUT * CIN + c];
    }
    const size_t n = p / HW, hw = p % HW;
#pragma unroll
    for (int c = 0; c < CLS_MAX_OUT; ++c)
      if ((c & 3) == q && c < COUT) out[(n * COUT + c) * HW + hw] = lg[c];
  }
}

// classifier backward: d_up[p][k] = sum_c dl[c][p] W[c][k];  dW[c][k] = sum_p dl[c][p] up[p][k];
// db[c] = sum_p dl[c][p];  optional decoder BN-backward statistics of d_up against t.
// CE: d loss / d logits is recomputed from (t, r, W, b, target) instead of being read: RCV_OP_CE_BWD and its tensor disappear (the
// logits are re-formed with the forward's FMA order, so the gradients are bit-identical to the unfused path).
// Built for every class count 1..8 (trainer.py:126-132 / train.py:301: numClass = 5 - nb - ng - nr - nl; model.py:462 nClass) and
// for 8 or 16 input channels; a thread keeps COUT x CIN filter-gradient accumulators, so the wide variants run one workgroup per SIMD set.
template <int CIN, int COUT, bool FUSED, bool CE>
__global__ __launch_bounds__(256, (CIN == 8 ? 2 : 1)) void cls_bwd_kernel(const float* __restrict__ up, const float* __restrict__ dl, const float* __restrict__ w,
                               float* __restrict__ dup, const float* __restrict__ t, const float* __restrict__ tc,
                               float* __restrict__ stat_part, float* __restrict__ w_part, int N, int HW, int stats,
                               const float* __restrict__ r, const float* __restrict__ rc, int mode2,
                               const int64_t* __restrict__ target, const float* __restrict__ cw, const float* __restrict__ bias,
                               const float* __restrict__ loss_out, const float* __restrict__ grad_out) {
  __shared__ float ws[COUT * CIN];
  __shared__ float red[4][COUT * CIN + COUT + 2 * CIN];
  for (int e = threadIdx.x; e < COUT * CIN; e += blockDim.x) ws[e] = w[e];
  __syncthreads();
  float dw[COUT][CIN], db[COUT], s1[CIN], s2[CIN];
#pragma unroll
  for (int c = 0; c < COUT; ++c) { db[c] = 0.f;
#pragma unroll
    for (int k = 0; k < CIN; ++k) dw[c][k] = 0.f; }
#pragma unroll
  for (int k = 0; k < CIN; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
  const size_t total = (size_t)N * HW;
  // pixel i+1 (and its label) is requested before pixel i is worked on: unconditionally, the last iteration re-requests its own pixel
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  ClsRaw<CIN, FUSED> cur;
  int64_t tcur = 0;
  if (p < total) { cls_load_raw<CIN, FUSED>(cur, FUSED ? t : up, r, p); if (CE) tcur = target[p]; }
  for (; p < total; p += stride) {
    const size_t pn = p + stride < total ? p + stride : p;
    ClsRaw<CIN, FUSED> nxt;
    cls_load_raw<CIN, FUSED>(nxt, FUSED ? t : up, r, pn);
    int64_t tnxt = 0;
    if (CE) tnxt = target[pn];
    const size_t n = p / HW, hw = p % HW;
    float g[COUT], u[CIN], d[CIN];
    cls_form_up<CIN, FUSED>(u, cur, tc, rc, mode2);                       // FUSED: up is re-formed from t and the skip tensor
    if (CE) {
      float mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < COUT; ++c) {
        float lgc = bias ? bias[c] : 0.f;
#pragma unroll
        for (int k = 0; k < CIN; ++k) lgc = fmaf(u[k], ws[c * CIN + k], lgc);
        g[c] = lgc;
        mx = fmaxf(mx, lgc);
      }
      float se = 0.f;
#pragma unroll
      for (int c = 0; c < COUT; ++c) { g[c] = expf(g[c] - mx); se += g[c]; }
      const int tg = (int)tcur;
      const float kf = (grad_out[0] / loss_out[1]) * ((unsigned)tg < (unsigned)COUT ? (cw ? cw[tg] : 1.f) : 0.f);
      const float inv = 1.f / se;
#pragma unroll
      for (int c = 0; c < COUT; ++c) g[c] = __fmul_rn(kf, fmaf(g[c], inv, c == tg ? -1.f : 0.f));      // explicit: same rounding as ce_bwd_kernel
    } else {
#pragma unroll
      for (int c = 0; c < COUT; ++c) g[c] = dl[(n * COUT + c) * HW + hw];
    }
#pragma unroll
    for (int k = 0; k < CIN; ++k) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < COUT; ++c) acc = fmaf(g[c], ws[c * CIN + k], acc);
      d[k] = acc;
    }
    if (CIN == 8 && (total & 63) == 0) {
      // a lane holds its pixel's 8 floats (32 B).  Store s (0|1) writes the 1 KiB of pixels 32s..32s+31 of the wave:
      // lane l sends float4 (l&1) of pixel 32s + (l>>1), fetched from that lane by shuffles => whole-line stores.
      const int lane = threadIdx.x & 63;
      const size_t wave_p0 = p - lane;
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        const int src = sidx * 32 + (lane >> 1);
        float4 lo, hi;
        lo.x = __shfl(d[0], src); lo.y = __shfl(d[1], src); lo.z = __shfl(d[2], src); lo.w = __shfl(d[3], src);
        hi.x = __shfl(d[4], src); hi.y = __shfl(d[5], src); hi.z = __shfl(d[6], src); hi.w = __shfl(d[7], src);
        sst4(dup + (wave_p0 + sidx * 32) * CIN + lane * 4, (lane & 1) ? hi : lo);
      }
    } else {
#pragma unroll
      for (int q = 0; q < CIN / 4; ++q) sst4(dup + p * CIN + 4 * q, make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]));
    }
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
      db[c] += g[c];
#pragma unroll
      for (int k = 0; k < CIN; ++k) dw[c][k] = fmaf(g[c], u[k], dw[c][k]);
    }
    if (stats == RCV_STATS_BWD_DEC) {
#pragma unroll
      for (int q = 0; q < CIN / 4; ++q) {
        const float4 a = FUSED ? cur.a[q] : sld4(t + p * CIN + 4 * q);      // FUSED: the raw decoder output is already in registers
        const float tv[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = 4 * q + j;
          const float gm = fmaf(tv[j], tc[k], tc[CIN + k]) > 0.f ? d[k] : 0.f;
          s1[k] += gm;
          s2[k] = fmaf(gm, tv[j] - tc[2 * CIN + k], s2[k]);
        }
      }
    }
    cur = nxt; tcur = tnxt;
  }
  // block reduction: wave shuffles, then the 4 waves through LDS in fixed order
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < COUT; ++c) {
#pragma unroll
    for (int k = 0; k < CIN; ++k) { const float v = wave_sum(dw[c][k]); if (lane == 0) red[wv][c * CIN + k] = v; }
    const float v = wave_sum(db[c]);
    if (lane == 0) red[wv][COUT * CIN + c] = v;
  }
#pragma unroll
  for (int k = 0; k < CIN; ++k) {
    const float v1 = wave_sum(s1[k]), v2 = wave_sum(s2[k]);
    if (lane == 0) { red[wv][COUT * CIN + COUT + k] = v1; red[wv][COUT * CIN + COUT + CIN + k] = v2; }
  }
  __syncthreads();
  const int nw = blockDim.x >> 6;
  for (int e = threadIdx.x; e < COUT * CIN + COUT + 2 * CIN; e += blockDim.x) {
    float v = 0.f;
    for (int i = 0; i < nw; ++i) v += red[i][e];
    if (e < COUT * CIN + COUT) w_part[(size_t)blockIdx.x * (COUT * CIN + COUT) + e] = v;
    else if (stats == RCV_STATS_BWD_DEC) stat_part[(size_t)blockIdx.x * 2 * CIN + (e - COUT * CIN - COUT)] = v;
  }
}

__global__ void rows_reduce_kernel(const float* __restrict__ part, int n_rows, int width, float* __restrict__ out0, int n0,
                                   float* __restrict__ out1) {
  // out0[e] (e < n0) and out1[e-n0] = sum over rows, fixed order, double accumulation
  const int e = blockIdx.x;
  double s = 0.0, dummy = 0.0;
  for (int i = threadIdx.x; i < n_rows; i += blockDim.x) s += (double)part[(size_t)i * width + e];
  block_sum2_d(s, dummy);
  if (threadIdx.x == 0) {
    if (e < n0) out0[e] = (float)s;
    else if (out1) out1[e - n0] = (float)s;
  }
}

// ------------------------------------------------------------------------------------------
// CrossEntropyLoss2d (model.py:76-82) + argmax / pixel accuracy (train.py:70-71)
// ------------------------------------------------------------------------------------------
__global__ void ce_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, const float* __restrict__ cw,
                              int N, int C, int HW, float* __restrict__ part, uint8_t* __restrict__ argmax) {
  __shared__ double sh[3][4];
  double a_nll = 0.0, a_w = 0.0, a_ok = 0.0;
  const size_t total = (size_t)N * HW;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
    const size_t n = p / HW, hw = p % HW;
    float v[CE_MAX_C];
    float mx = -INFINITY;
    int am = 0;
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c) {
      if (c < C) {
        v[c] = logits[(n * C + c) * HW + hw];
        if (v[c] > mx) { mx = v[c]; am = c; }   // strict '>' : first maximum wins (torch.max semantics)
      }
    }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c) if (c < C) se += expf(v[c] - mx);
    const int tg = (int)target[p];
    float vt = 0.f;
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c) if (c == tg) vt = v[c];
    const float w = (unsigned)tg < (unsigned)C ? (cw ? cw[tg] : 1.f) : 0.f;   // label outside [0, C) (e.g. -100): ignored, like NLLLoss's ignore_index
    const float nll = (mx - vt) + logf(se);
    a_nll += (double)(w * nll);
    a_w += (double)w;
    a_ok += (am == tg) ? 1.0 : 0.0;
    if (argmax) argmax[p] = (uint8_t)am;
  }
  a_nll = wave_sum_d(a_nll); a_w = wave_sum_d(a_w); a_ok = wave_sum_d(a_ok);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[0][wv] = a_nll; sh[1][wv] = a_w; sh[2][wv] = a_ok; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[threadIdx.x][i];
    part[(size_t)blockIdx.x * 3 + threadIdx.x] = (float)s;
  }
}

__global__ void ce_finalize_kernel(const float* __restrict__ part, int n_part, float* __restrict__ loss_out) {
  __shared__ double sh[3][4];
  double s[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < n_part; i += blockDim.x) {
    s[0] += (double)part[(size_t)i * 3 + 0]; s[1] += (double)part[(size_t)i * 3 + 1]; s[2] += (double)part[(size_t)i * 3 + 2];
  }
  const int wv = threadIdx.x >> 6;
  for (int j = 0; j < 3; ++j) { s[j] = wave_sum_d(s[j]); if ((threadIdx.x & 63) == 0) sh[j][wv] = s[j]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[3] = {0.0, 0.0, 0.0};
    for (int j = 0; j < 3; ++j) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t[j] += sh[j][i];
    loss_out[0] = (float)(t[0] / t[1]);
    loss_out[1] = (float)t[1];
    loss_out[2] = (float)t[2];
    loss_out[3] = (float)t[0];
  }
}

__global__ void ce_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, const float* __restrict__ cw,
                              const float* __restrict__ loss_out, const float* __restrict__ grad_out, int N, int C, int HW,
                              float* __restrict__ dlogits) {
  const float scale = grad_out[0] / loss_out[1];
  const size_t total = (size_t)N * HW;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
    const size_t n = p / HW, hw = p % HW;
    float v[CE_MAX_C];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c) if (c < C) { v[c] = logits[(n * C + c) * HW + hw]; mx = fmaxf(mx, v[c]); }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c) if (c < C) { v[c] = expf(v[c] - mx); se += v[c]; }
    const int tg = (int)target[p];
    const float k = scale * ((unsigned)tg < (unsigned)C ? (cw ? cw[tg] : 1.f) : 0.f);
    const float inv = 1.f / se;
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c)
      if (c < C) dlogits[(n * C + c) * HW + hw] = __fmul_rn(k, fmaf(v[c], inv, c == tg ? -1.f : 0.f));   // explicit: same rounding as cls_bwd_kernel<CE>
  }
}

// ------------------------------------------------------------------------------------------
// DiceLoss (model.py:5-43, multi-class branch): loss = 1 - mean_c( 2 w_c I_c / (S_c + N_c + eps) ),
//   I_c = sum_p P[p][c] [t_p == c],  S_c = sum_p P[p][c],  N_c = #[t_p == c],  P = softmax over channels.
// Forward writes out[0] = loss, out[2] = #pixels whose arg-max equals the target, and the per-class backward
// coefficients out[4+c] = A_c, out[12+c] = B_c with d loss / d P[p][c] = A_c [t_p == c] + B_c.
// ------------------------------------------------------------------------------------------
#define DICE_ROW (3 * CE_MAX_C + 1)
__global__ void dice_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, int N, int C, int HW,
                                float* __restrict__ part, uint8_t* __restrict__ argmax) {
  __shared__ double sh[4][DICE_ROW];
  double acc[DICE_ROW];
#pragma unroll
  for (int j = 0; j < DICE_ROW; ++j) acc[j] = 0.0;
  const size_t total = (size_t)N * HW;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
    const size_t n = p / HW, hw = p % HW;
    float v[CE_MAX_C];
    float mx = -INFINITY;
    int am = 0;
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c) {
      if (c < C) {
        v[c] = logits[(n * C + c) * HW + hw];
        if (v[c] > mx) { mx = v[c]; am = c; }
      }
    }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c) if (c < C) { v[c] = expf(v[c] - mx); se += v[c]; }
    const float inv = 1.f / se;
    const int tg = (int)target[p];
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c) {
      if (c < C) {
        const double pc = (double)(v[c] * inv);
        acc[CE_MAX_C + c] += pc;
        if (c == tg) { acc[c] += pc; acc[2 * CE_MAX_C + c] += 1.0; }
      }
    }
    acc[3 * CE_MAX_C] += (am == tg) ? 1.0 : 0.0;
    if (argmax) argmax[p] = (uint8_t)am;
  }
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < DICE_ROW; ++j) {
    const double s = wave_sum_d(acc[j]);
    if ((threadIdx.x & 63) == 0) sh[wv][j] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < DICE_ROW) {
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i][threadIdx.x];
    part[(size_t)blockIdx.x * DICE_ROW + threadIdx.x] = (float)s;
  }
}

__global__ void dice_finalize_kernel(const float* __restrict__ part, int n_part, const float* __restrict__ cw, int C, float eps,
                                     float* __restrict__ out) {
  __shared__ double tot[DICE_ROW];
  __shared__ double sh[4];
  // one column of the partial rows per pass, fixed order
  for (int j = 0; j < DICE_ROW; ++j) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n_part; i += blockDim.x) s += (double)part[(size_t)i * DICE_ROW + j];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i]; tot[j] = t; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double dice = 0.0;
    for (int c = 0; c < CE_MAX_C; ++c) {
      double A = 0.0, B = 0.0;
      if (c < C) {
        const double w = cw ? (double)cw[c] : 1.0;
        const double I = tot[c], K = tot[CE_MAX_C + c] + tot[2 * CE_MAX_C + c] + (double)eps;
        dice += 2.0 * w * I / K;
        A = -(2.0 * w / C) / K;
        B = (2.0 * w / C) * I / (K * K);
      }
      out[4 + c] = (float)A;
      out[4 + CE_MAX_C + c] = (float)B;
    }
    out[0] = (float)(1.0 - dice / C);
    out[1] = 0.f;
    out[2] = (float)tot[3 * CE_MAX_C];
    out[3] = 0.f;
  }
}

__global__ void dice_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, const float* __restrict__ fwd_out,
                                const float* __restrict__ grad_out, int N, int C, int HW, float* __restrict__ dlogits) {
  const float go = grad_out[0];
  float A[CE_MAX_C], B[CE_MAX_C];
#pragma unroll
  for (int c = 0; c < CE_MAX_C; ++c) { A[c] = fwd_out[4 + c]; B[c] = fwd_out[4 + CE_MAX_C + c]; }
  const size_t total = (size_t)N * HW;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
    const size_t n = p / HW, hw = p % HW;
    float v[CE_MAX_C];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c) if (c < C) { v[c] = logits[(n * C + c) * HW + hw]; mx = fmaxf(mx, v[c]); }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c) if (c < C) { v[c] = expf(v[c] - mx); se += v[c]; }
    const float inv = 1.f / se;
    const int tg = (int)target[p];
    float dot = 0.f;
    float gq[CE_MAX_C];
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c) {
      if (c < C) {
        v[c] *= inv;
        gq[c] = B[c] + (c == tg ? A[c] : 0.f);
        dot = fmaf(v[c], gq[c], dot);
      }
    }
#pragma unroll
    for (int c = 0; c < CE_MAX_C; ++c)
      if (c < C) dlogits[(n * C + c) * HW + hw] = go * v[c] * (gq[c] - dot);
  }
}

// ------------------------------------------------------------------------------------------
// Layout changes around the 3x3 classifier of the v2 net (model.py:411 with size=3, model.py:493):
// the convolution runs NHWC with its 5 output channels padded to 8; the logits the caller sees are NCHW.
// ------------------------------------------------------------------------------------------
template <int CP>
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ out, int N, int HW,
                                    int C) {
  const size_t total = (size_t)N * HW;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
    const size_t n = p / HW, hw = p % HW;
    float v[CP];
#pragma unroll
    for (int q = 0; q < CP / 4; ++q) {
      const float4 u = sld4(x + p * CP + 4 * q);
      v[4 * q] = u.x; v[4 * q + 1] = u.y; v[4 * q + 2] = u.z; v[4 * q + 3] = u.w;
    }
#pragma unroll
    for (int c = 0; c < CP; ++c) if (c < C) out[(n * C + c) * HW + hw] = v[c] + (bias ? bias[c] : 0.f);
  }
}

template <int CP>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int HW, int C) {
  const size_t total = (size_t)N * HW;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
    const size_t n = p / HW, hw = p % HW;
    float v[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) v[c] = c < C ? x[(n * C + c) * HW + hw] : 0.f;
#pragma unroll
    for (int q = 0; q < CP / 4; ++q) sst4(out + p * CP + 4 * q, make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]));
  }
}

// ------------------------------------------------------------------------------------------
// MaxPool2d(2,2) (model.py:97-100) of y = r*c0+c1, and its backward fused with the skip-gradient
// add and the BatchNorm-backward reductions of the producer.
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ void pool_fwd_kernel(const float* __restrict__ r, const float* __restrict__ cst, float* __restrict__ out, int N, int H,
                                int W, int C) {
  const int C4 = C / 4, Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)N * Ho * Wo * C4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(e % C4);
    size_t pp = e / C4;
    const int ox = (int)(pp % Wo); pp /= Wo;
    const int oy = (int)(pp % Ho);
    const int n = (int)(pp / Ho);
    float4 s = make_float4(1.f, 1.f, 1.f, 1.f), h = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE != RCV_LOAD_PLAIN) { s = sld4(cst + 4 * q); h = sld4(cst + C + 4 * q); }
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float4 v = sld4(r + (((size_t)n * H + 2 * oy + dy) * W + 2 * ox + dx) * C + 4 * q);
        v.x = fmaf(v.x, s.x, h.x); v.y = fmaf(v.y, s.y, h.y); v.z = fmaf(v.z, s.z, h.z); v.w = fmaf(v.w, s.w, h.w);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    sst4(out + e * 4, m);
  }
}

// dy (full res) = scatter(dp -> first arg-max of the 2x2 window) [+ skip grad]; partial rows of
// (sum dy, sum dy*r) per workgroup.  blockDim.x must be a multiple of C/4.
template <int MODE>
__global__ void pool_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ r, const float* __restrict__ cst,
                                const float* __restrict__ resid, float* __restrict__ dy, float* __restrict__ part, int N, int H,
                                int W, int C, int stats) {
  extern __shared__ float4 sh4[];
  const int C4 = C / 4, Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)N * Ho * Wo * C4;
  const int q = threadIdx.x % C4;
  float4 s = make_float4(1.f, 1.f, 1.f, 1.f), h = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 mu = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE != RCV_LOAD_PLAIN) { s = sld4(cst + 4 * q); h = sld4(cst + C + 4 * q); if (stats != RCV_STATS_NONE) mu = sld4(cst + 2 * C + 4 * q); }
  float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    size_t pp = e / C4;
    const int ox = (int)(pp % Wo); pp /= Wo;
    const int oy = (int)(pp % Ho);
    const int n = (int)(pp / Ho);
    const float4 g = sld4(dp + e * 4);
    float4 rv[4];
    float bx = -INFINITY, by = -INFINITY, bz = -INFINITY, bw = -INFINITY;
    int ix = 0, iy = 0, iz = 0, iw = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rv[j] = sld4(r + (((size_t)n * H + 2 * oy + (j >> 1)) * W + 2 * ox + (j & 1)) * C + 4 * q);
      const float vx = fmaf(rv[j].x, s.x, h.x), vy = fmaf(rv[j].y, s.y, h.y), vz = fmaf(rv[j].z, s.z, h.z), vw = fmaf(rv[j].w, s.w, h.w);
      if (vx > bx) { bx = vx; ix = j; }
      if (vy > by) { by = vy; iy = j; }
      if (vz > bz) { bz = vz; iz = j; }
      if (vw > bw) { bw = vw; iw = j; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t off = (((size_t)n * H + 2 * oy + (j >> 1)) * W + 2 * ox + (j & 1)) * C + 4 * q;
      float4 v = make_float4(ix == j ? g.x : 0.f, iy == j ? g.y : 0.f, iz == j ? g.z : 0.f, iw == j ? g.w : 0.f);
      if (resid) { const float4 rr = sld4(resid + off); v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
      sst4(dy + off, v);
      a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w;
      a2.x = fmaf(v.x, rv[j].x - mu.x, a2.x); a2.y = fmaf(v.y, rv[j].y - mu.y, a2.y);
      a2.z = fmaf(v.z, rv[j].z - mu.z, a2.z); a2.w = fmaf(v.w, rv[j].w - mu.w, a2.w);
    }
  }
  if (stats != RCV_STATS_NONE) {
    sh4[threadIdx.x] = a1;
    sh4[blockDim.x + threadIdx.x] = a2;
    __syncthreads();
    if ((int)threadIdx.x < 2 * C4) {
      const int which = threadIdx.x / C4, qq = threadIdx.x % C4;
      float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int e = qq; e < (int)blockDim.x; e += C4) { const float4 v = sh4[which * blockDim.x + e]; u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w; }
      sst4(part + ((size_t)blockIdx.x * 2 + which) * C + 4 * qq, u);
    }
  }
}

// The same, with lanes mapped to the INPUT rows: a thread owns one 16-byte piece (pixel column, channel quad) of the two input rows of
// an output row, so that every load and store instruction of a wave covers 1 KB of consecutive memory (in the kernel above a thread owns
// an output pixel and each of its four window loads touches half of every 128-byte line: 3.4 TB/s on the U-Net configuration).  The two
// columns of a window sit C/4 lanes apart: one cross-lane exchange of (row maximum, row index) decides the window.  First maximum in the
// window order (0,0), (0,1), (1,0), (1,1) wins ties, as in the kernel above and in aten::max_pool2d_with_indices.
// Needs C/4 a power of two < 64 and W * C/4 a multiple of 64 (a window's two lanes then lie in one wave); blockDim.x a multiple of C/4.
template <int MODE>
__global__ void pool_bwd_rows_kernel(const float* __restrict__ dp, const float* __restrict__ r, const float* __restrict__ cst,
                                     const float* __restrict__ resid, float* __restrict__ dy, float* __restrict__ part, int N, int H,
                                     int W, int C, int stats) {
  extern __shared__ float4 sh4[];
  const int C4 = C / 4, Ho = H / 2, Wo = W / 2, RW = W * C4;       // RW: 16-byte pieces of an input row
  const size_t total = (size_t)N * Ho * RW;
  const int q = threadIdx.x % C4;
  float4 s = make_float4(1.f, 1.f, 1.f, 1.f), h = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 mu = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE != RCV_LOAD_PLAIN) { s = sld4(cst + 4 * q); h = sld4(cst + C + 4 * q); if (stats != RCV_STATS_NONE) mu = sld4(cst + 2 * C + 4 * q); }
  float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
  const size_t nloop = (total + (size_t)gridDim.x * blockDim.x - 1) / ((size_t)gridDim.x * blockDim.x);      // uniform trip count: the exchange below needs every lane
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t it = 0; it < nloop; ++it, e += (size_t)gridDim.x * blockDim.x) {
    const bool live = e < total;
    const size_t ec = live ? e : total - 1;
    const size_t orow = ec / RW;                          // n * Ho + oy
    const int i = (int)(ec - orow * RW);                  // piece of the row: column i / C4, quad i % C4 (== q)
    const int col = i / C4, jx = col & 1;
    const size_t in0 = (orow * 2) * (size_t)RW * 4 + (size_t)i * 4;       // element offset of this piece in input row 2 oy
    const size_t in1 = in0 + (size_t)RW * 4;
    const float4 r0 = sld4(r + in0), r1 = sld4(r + in1);
    const float4 g = sld4(dp + (orow * Wo + (col >> 1)) * C + 4 * q);
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
    if (resid) { q0 = sld4(resid + in0); q1 = sld4(resid + in1); }
    const float v0[4] = {fmaf(r0.x, s.x, h.x), fmaf(r0.y, s.y, h.y), fmaf(r0.z, s.z, h.z), fmaf(r0.w, s.w, h.w)};
    const float v1[4] = {fmaf(r1.x, s.x, h.x), fmaf(r1.y, s.y, h.y), fmaf(r1.z, s.z, h.z), fmaf(r1.w, s.w, h.w)};
    const float gg[4] = {g.x, g.y, g.z, g.w};
    float d0[4], d1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // this column: rows 0 / 1 are window positions jx / 2 + jx; the other column's pair arrives from C4 lanes away
      const bool low = v1[c] > v0[c];
      const float best = low ? v1[c] : v0[c];
      const int idx = (low ? 2 : 0) + jx;
      const float pbest = __shfl_xor(best, C4);
      const int pidx = __shfl_xor(idx, C4);
      const bool mine = best > pbest || (best == pbest && idx < pidx);
      d0[c] = (mine && !low) ? gg[c] : 0.f;
      d1[c] = (mine && low) ? gg[c] : 0.f;
    }
    if (live) {
      const float4 o0 = make_float4(d0[0] + q0.x, d0[1] + q0.y, d0[2] + q0.z, d0[3] + q0.w);
      const float4 o1 = make_float4(d1[0] + q1.x, d1[1] + q1.y, d1[2] + q1.z, d1[3] + q1.w);
      sst4(dy + in0, o0);
      sst4(dy + in1, o1);
      a1.x += o0.x + o1.x; a1.y += o0.y + o1.y; a1.z += o0.z + o1.z; a1.w += o0.w + o1.w;
      a2.x = fmaf(o0.x, r0.x - mu.x, a2.x); a2.y = fmaf(o0.y, r0.y - mu.y, a2.y); a2.z = fmaf(o0.z, r0.z - mu.z, a2.z); a2.w = fmaf(o0.w, r0.w - mu.w, a2.w);
      a2.x = fmaf(o1.x, r1.x - mu.x, a2.x); a2.y = fmaf(o1.y, r1.y - mu.y, a2.y); a2.z = fmaf(o1.z, r1.z - mu.z, a2.z); a2.w = fmaf(o1.w, r1.w - mu.w, a2.w);
    }
  }
  if (stats != RCV_STATS_NONE) {
    sh4[threadIdx.x] = a1;
    sh4[blockDim.x + threadIdx.x] = a2;
    __syncthreads();
    if ((int)threadIdx.x < 2 * C4) {
      const int which = threadIdx.x / C4, qq = threadIdx.x % C4;
      float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = qq; k < (int)blockDim.x; k += C4) { const float4 v = sh4[which * blockDim.x + k]; u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w; }
      sst4(part + ((size_t)blockIdx.x * 2 + which) * C + 4 * qq, u);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Fused optimizer step over one flat fp32 buffer: g = grad*grad_scale + decay*sign(p) (the
// gradient of decay*sum|p|, train.py:23-27,53-55) followed by torch.optim.Adam's update
// (train.py:67, defaults betas .9/.999 eps 1e-8, no weight decay, no amsgrad).
// ------------------------------------------------------------------------------------------
// MET: the launch also books the step's metrics (train.py:52-53,69-73: loss = CE + decay*sum|p|, reg, #correct pixels): every
// workgroup leaves its sum of |p| (taken BEFORE the update, as the reference's l1reg(model) is) in `part`, the workgroup that
// finishes last adds the rows in index order and updates metrics[4] = {sum loss, sum reg, sum #correct, steps} in double.
// part = double[gridDim.x] followed by one uint32 ticket counter (zero before the first launch; left zero by every launch).
template <bool MET>
__global__ void adam_l1_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                               const float* __restrict__ lr_elem, size_t n, float lr, float b1, float b2, float eps, float decay,
                               float grad_scale, int step_host, const int* __restrict__ step_dev, double* part,
                               const float* __restrict__ loss_stats, double* metrics, const uint8_t* __restrict__ prune, int vec) {
  // bias corrections of step t (1-based): t comes from the record, or -- when the whole training step is replayed as a captured
  // graph and the host cannot change launch arguments -- from a device counter the caller advances ahead of this launch
  __shared__ float s_bc[2];
  if (threadIdx.x == 0) {
    const int t = step_dev ? *step_dev : step_host;
    s_bc[0] = (float)(1.0 - pow((double)b1, (double)t));
    s_bc[1] = (float)sqrt(1.0 - pow((double)b2, (double)t));
  }
  __syncthreads();
  const float bc1 = s_bc[0], bc2_sqrt = s_bc[1];
  float asum = 0.f;
  auto update = [&](float pv, float gv, float& mv, float& vv, float lre, bool pruned) -> float {
    if (MET) asum += fabsf(pv);                       // l1reg(model) runs over every parameter, stepped or not
    if (lre == 0.f) return pv;                        // parameter without a gradient (outside the graph / the groups): untouched, as under torch.optim.Adam
    const float sg = pv > 0.f ? 1.f : (pv < 0.f ? -1.f : 0.f);
    float gr = fmaf(decay, sg, gv * grad_scale);
    if (pruned) gr = 0.f;                             // train.py:59-65: param.grad[indices] = 0 after backward (the L1 part included)
    mv = b1 * mv + (1.f - b1) * gr;
    vv = b2 * vv + (1.f - b2) * gr * gr;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    const float step = lre / bc1;
    return pv - step * (mv / denom);
  };
  // four elements per thread and iteration (the flat buffers come from one allocation each: 16-byte aligned), scalar tail
  const size_t n4 = vec ? n / 4 : 0;
  const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsize = (size_t)gridDim.x * blockDim.x;
  for (size_t e = gtid; e < n4; e += gsize) {
    float4 pv = reinterpret_cast<const float4*>(p)[e];
    const float4 gv = reinterpret_cast<const float4*>(g)[e];
    float4 mv = reinterpret_cast<const float4*>(m)[e], vv = reinterpret_cast<const float4*>(v)[e];
    const float4 le = lr_elem ? reinterpret_cast<const float4*>(lr_elem)[e] : make_float4(lr, lr, lr, lr);
    const uchar4 pr = prune ? reinterpret_cast<const uchar4*>(prune)[e] : make_uchar4(0, 0, 0, 0);
    pv.x = update(pv.x, gv.x, mv.x, vv.x, le.x, pr.x != 0);
    pv.y = update(pv.y, gv.y, mv.y, vv.y, le.y, pr.y != 0);
    pv.z = update(pv.z, gv.z, mv.z, vv.z, le.z, pr.z != 0);
    pv.w = update(pv.w, gv.w, mv.w, vv.w, le.w, pr.w != 0);
    reinterpret_cast<float4*>(p)[e] = pv;
    reinterpret_cast<float4*>(m)[e] = mv;
    reinterpret_cast<float4*>(v)[e] = vv;
  }
  for (size_t e = 4 * n4 + gtid; e < n; e += gsize) {
    float mv = m[e], vv = v[e];
    p[e] = update(p[e], g[e], mv, vv, lr_elem ? lr_elem[e] : lr, prune && prune[e]);
    m[e] = mv; v[e] = vv;
  }
  if (MET) {
    __shared__ int is_last;
    unsigned* ticket = reinterpret_cast<unsigned*>(part + gridDim.x);
    double a = (double)asum, dummy = 0.0;
    block_sum2_d(a, dummy);
    if (threadIdx.x == 0) {
      // The row and the ticket are both device-scope atomics, i.e. performed at the memory side of the (per-XCD, mutually
      // non-coherent) L2s, and the exchange has returned before the ticket is requested: the row is in place when its ticket is
      // counted.  (A __threadfence() here is a write-back of the XCD's whole L2 -- holding the 13 MB this launch just wrote -- per
      // workgroup: it cost most of the kernel's time.)
      (void)atomicExch(reinterpret_cast<unsigned long long*>(part + blockIdx.x), (unsigned long long)__double_as_longlong(a));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      is_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (is_last) {
      double s = 0.0;
      for (unsigned i = threadIdx.x; i < gridDim.x; i += blockDim.x)      // device-scope loads: never served from this XCD's L2
        s += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(part + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      dummy = 0.0;
      __syncthreads();                                   // block_sum2_d's scratch is reused
      block_sum2_d(s, dummy);
      if (threadIdx.x == 0) {
        const double reg = (double)decay * s;
        metrics[0] += (double)loss_stats[0] + reg;
        metrics[1] += reg;
        metrics[2] += (double)loss_stats[2];
        metrics[3] += 1.0;
        *ticket = 0u;
      }
    }
  }
}

// torch.optim.SGD with momentum and weight decay (trainer.py:176-178), dampening 0, no nesterov
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, const float* __restrict__ lr_elem,
                           size_t n, float lr, float momentum, float wd, float grad_scale, int first) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float step = lr_elem ? lr_elem[e] : lr;
    if (step == 0.f) continue;                       // parameter outside the graph (grad None in PyTorch): untouched
    const float pv = p[e];
    const float gr = fmaf(wd, pv, g[e] * grad_scale);
    const float b = first ? gr : fmaf(momentum, buf[e], gr);
    buf[e] = b;
    p[e] = pv - step * b;
  }
}

__global__ void memset_kernel(float* __restrict__ p, size_t n) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) p[e] = 0.f;
}

// x[..., 0:Ca] += f(a)   (LabelProp model.py:565; f = relu(affine) of the `pre` block)
__global__ void add_slice_kernel(float* __restrict__ x, int C, const float* __restrict__ a, const float* __restrict__ ac, int Ca,
                                 size_t npix, int mode) {
  const int C4 = Ca / 4;
  const size_t total = npix * C4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(e % C4);
    const size_t p = e / C4;
    float4 b = sld4(a + p * Ca + 4 * q);
    if (mode != RCV_LOAD_PLAIN) {
      const float4 s = sld4(ac + 4 * q), h = sld4(ac + Ca + 4 * q);
      b.x = fmaf(b.x, s.x, h.x); b.y = fmaf(b.y, s.y, h.y); b.z = fmaf(b.z, s.z, h.z); b.w = fmaf(b.w, s.w, h.w);
      if (mode == RCV_LOAD_AFFINE_RELU) { b.x = fmaxf(b.x, 0.f); b.y = fmaxf(b.y, 0.f); b.z = fmaxf(b.z, 0.f); b.w = fmaxf(b.w, 0.f); }
    }
    float4 v = sld4(x + p * C + 4 * q);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    sst4(x + p * C + 4 * q, v);
  }
}

// counts[n][pred][label] += 1: per-block LDS histogram (integer atomics: order independent), then one global add per bin
__global__ void confusion_kernel(const uint8_t* __restrict__ pred, const int64_t* __restrict__ target, int* __restrict__ counts,
                                 int HW, int C) {
  __shared__ int hist[64];
  if (threadIdx.x < 64) hist[threadIdx.x] = 0;
  __syncthreads();
  const int n = blockIdx.y;
  const uint8_t* pp = pred + (size_t)n * HW;
  const int64_t* tt = target + (size_t)n * HW;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    const int p = pp[i], t = (int)tt[i];
    if (p < C && t >= 0 && t < C) atomicAdd(&hist[p * C + t], 1);
  }
  __syncthreads();
  if ((int)threadIdx.x < C * C && hist[threadIdx.x]) atomicAdd(&counts[(size_t)n * C * C + threadIdx.x], hist[threadIdx.x]);
}

// out = f(in): a block's output as a plain NHWC tensor (block-level module calls, LabelProp tail)
__global__ void materialize_kernel(const float* __restrict__ x, const float* __restrict__ c, float* __restrict__ out, size_t n4, int C,
                                   int mode) {
  const int C4 = C / 4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
    float4 v = sld4(x + e * 4);
    if (mode != RCV_LOAD_PLAIN) {
      const int ch = (int)(e % C4) * 4;
      const float4 s = sld4(c + ch), h = sld4(c + C + ch);
      v.x = fmaf(v.x, s.x, h.x); v.y = fmaf(v.y, s.y, h.y); v.z = fmaf(v.z, s.z, h.z); v.w = fmaf(v.w, s.w, h.w);
      if (mode == RCV_LOAD_AFFINE_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    }
    sst4(out + e * 4, v);
  }
}

// out = g and per-workgroup partial rows (sum g[*m], sum g[*m]*e) -- the BatchNorm-backward sums when the
// gradient arrives from outside the engine (block-level module calls).  blockDim.x % (C/4) == 0.
__global__ void bwd_stats_kernel(const float* __restrict__ g, const float* __restrict__ ev, const float* __restrict__ ec,
                                 float* __restrict__ out, float* __restrict__ part, size_t n4, int C, int stats, int Csrc, int coff) {
  extern __shared__ float4 sh4[];
  const int C4 = C / 4;
  const int q = threadIdx.x % C4;
  if (stats == RCV_STATS_NONE) {      // plain channel-slice copy (skip half of a concatenated gradient)
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x)
      sst4(out + e * 4, sld4(g + (e / C4) * Csrc + coff + (e % C4) * 4));
    return;
  }
  float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
  if (stats == RCV_STATS_BWD_DEC) { c0 = sld4(ec + 4 * q); c1 = sld4(ec + C + 4 * q); }
  const float4 mu = sld4(ec + 2 * C + 4 * q);
  float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
    float4 v = sld4(g + (e / C4) * Csrc + coff + (e % C4) * 4);
    sst4(out + e * 4, v);
    const float4 x = sld4(ev + e * 4);
    if (stats == RCV_STATS_BWD_DEC) {
      v.x = fmaf(x.x, c0.x, c1.x) > 0.f ? v.x : 0.f; v.y = fmaf(x.y, c0.y, c1.y) > 0.f ? v.y : 0.f;
      v.z = fmaf(x.z, c0.z, c1.z) > 0.f ? v.z : 0.f; v.w = fmaf(x.w, c0.w, c1.w) > 0.f ? v.w : 0.f;
    }
    a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w;
    a2.x = fmaf(v.x, x.x - mu.x, a2.x); a2.y = fmaf(v.y, x.y - mu.y, a2.y);
    a2.z = fmaf(v.z, x.z - mu.z, a2.z); a2.w = fmaf(v.w, x.w - mu.w, a2.w);
  }
  sh4[threadIdx.x] = a1;
  sh4[blockDim.x + threadIdx.x] = a2;
  __syncthreads();
  if ((int)threadIdx.x < 2 * C4) {
    const int which = threadIdx.x / C4, qq = threadIdx.x % C4;
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = qq; e < (int)blockDim.x; e += C4) { const float4 v = sh4[which * blockDim.x + e]; u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w; }
    sst4(part + ((size_t)blockIdx.x * 2 + which) * C + 4 * qq, u);
  }
}

// --------------------------------------------------------------------------------------------
// launcher
// --------------------------------------------------------------------------------------------
static inline int stream_grid(const rcv_handle* h, size_t work_items, int block) {
  size_t g = (work_items + block - 1) / block;
  const size_t cap = (size_t)h->num_cus * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// optimizer kernels: four elements per thread and iteration, at most two workgroups per CU (the metrics variant ends with one
// device-scope ticket per workgroup: 2048 of them on one address cost more than the update itself)
static inline int adam_grid(const rcv_handle* h, size_t n) {
  size_t g = (n / 4 + 255) / 256;
  const size_t cap = (size_t)h->num_cus * 2;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// number of workgroups used by the reducing stream kernels (deterministic function of the shape)
static inline int reduce_grid(const rcv_handle* h, size_t work_items, int block) {
  size_t g = (work_items + block - 1) / block;
  const size_t cap = (size_t)h->num_cus * 4;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

typedef void (*cls_bwd_fn)(const float*, const float*, const float*, float*, const float*, const float*, float*, float*, int, int, int,
                           const float*, const float*, int, const int64_t*, const float*, const float*, const float*, const float*);

template <int CIN, bool FUSED, bool CE>
static cls_bwd_fn cls_bwd_pick(int cout) {
  switch (cout) {
    case 1: return cls_bwd_kernel<CIN, 1, FUSED, CE>;
    case 2: return cls_bwd_kernel<CIN, 2, FUSED, CE>;
    case 3: return cls_bwd_kernel<CIN, 3, FUSED, CE>;
    case 4: return cls_bwd_kernel<CIN, 4, FUSED, CE>;
    case 5: return cls_bwd_kernel<CIN, 5, FUSED, CE>;
    case 6: return cls_bwd_kernel<CIN, 6, FUSED, CE>;
    case 7: return cls_bwd_kernel<CIN, 7, FUSED, CE>;
    case 8: return cls_bwd_kernel<CIN, 8, FUSED, CE>;
  }
  return nullptr;
}

// Every refusal that depends on the SHAPE of a record (channel counts, load modes, flags) sits in front of the `query` return: what
// rcv_op_workspace / the planner accepts, the launch accepts.  Only operand pointers and workspace row counts are checked after it.
int rcv_launch_small(const rcv_handle* h, const rcv_op* op, hipStream_t s, OpQuery* query) {
  const int N = op->i[RCV_I_N], H = op->i[RCV_I_H], W = op->i[RCV_I_W];
  const int Cin = op->i[RCV_I_CIN], Cout = op->i[RCV_I_COUT];
  if (query) {
    static const char* names[] = {"?", "conv", "tconv", "wgrad", "wgrad_reduce", "pack", "bn_finalize", "bn_eval", "bn_bwd", "combine",
                                  "cls_fwd", "cls_bwd", "ce_fwd", "ce_bwd", "pool_fwd", "pool_bwd", "adam_l1", "memset", "conv1x1",
                                  "add_slice", "materialize", "bwd_stats", "confusion", "dice_fwd", "dice_bwd", "nhwc_to_nchw",
                                  "nchw_to_nhwc", "sgd"};
    query->n_part = 0; query->n_split = 0; query->part_bytes = 0;
    snprintf(query->label, sizeof(query->label), "%s", (op->kind > 0 && op->kind <= 27) ? names[op->kind] : "?");
  }
  switch (op->kind) {
    case RCV_OP_PACK: {
      if (query) return RCV_OK;
      const int count = op->i[RCV_I_COUNT], max_elems = op->i[RCV_I_AUX0];
      RCV_CHECK_ARG(op->p[RCV_P_IN] && count > 0 && max_elems > 0, "pack: bad job table");
      int gx = ceil_div(max_elems, 256);
      if (gx > 64) gx = 64;
      hipLaunchKernelGGL(pack_kernel, dim3(gx, count), dim3(256), 0, s, (const rcv_pack_job*)op->p[RCV_P_IN]);
      break;
    }
    case RCV_OP_BN_FINALIZE: {
      const double count = (double)N * op->i[RCV_I_HO] * op->i[RCV_I_WO];
      RCV_CHECK_ARG(op->i[RCV_I_NPART] > 0 && Cout > 0 && count > 0, "bn_finalize: empty");
      if (query) return RCV_OK;
      RCV_CHECK_ARG(op->p[RCV_P_PART] && op->p[RCV_P_OUT] && op->p[RCV_P_X0] && op->p[RCV_P_X1] && op->p[RCV_P_X4] && op->p[RCV_P_X5],
                    "bn_finalize: null operand");
      hipLaunchKernelGGL(bn_finalize_kernel, dim3(Cout), dim3(256), 0, s, (const float*)op->p[RCV_P_PART], op->i[RCV_I_NPART], Cout, count,
                         (const float*)op->p[RCV_P_X0], (const float*)op->p[RCV_P_X1], (float*)op->p[RCV_P_X2], (float*)op->p[RCV_P_X3],
                         op->f[0], op->f[1], (op->flags & RCV_F_TRAINING) ? 1 : 0, (float*)op->p[RCV_P_OUT], (float*)op->p[RCV_P_X4],
                         (float*)op->p[RCV_P_X5]);
      break;
    }
    case RCV_OP_BN_EVAL: {
      if (query) return RCV_OK;
      RCV_CHECK_ARG(op->p[RCV_P_OUT] && op->p[RCV_P_X0] && op->p[RCV_P_X1] && op->p[RCV_P_X2] && op->p[RCV_P_X3], "bn_eval: null operand");
      hipLaunchKernelGGL(bn_eval_kernel, dim3(ceil_div(Cout, 64)), dim3(64), 0, s, Cout, (const float*)op->p[RCV_P_X0],
                         (const float*)op->p[RCV_P_X1], (const float*)op->p[RCV_P_X2], (const float*)op->p[RCV_P_X3], op->f[1],
                         (float*)op->p[RCV_P_OUT], (const float*)op->p[RCV_P_X4]);
      break;
    }
    case RCV_OP_BN_BWD: {
      if (query) return RCV_OK;
      const double count = (double)N * op->i[RCV_I_HO] * op->i[RCV_I_WO];
      RCV_CHECK_ARG(op->p[RCV_P_PART] && op->p[RCV_P_OUT] && op->p[RCV_P_X0] && op->p[RCV_P_X4] && op->p[RCV_P_X5] && op->p[RCV_P_IN_C],
                    "bn_bwd: null operand");
      hipLaunchKernelGGL(bn_bwd_kernel, dim3(Cout), dim3(256), 0, s, (const float*)op->p[RCV_P_PART], op->i[RCV_I_NPART], Cout, count,
                         (const float*)op->p[RCV_P_X0], (const float*)op->p[RCV_P_X4], (const float*)op->p[RCV_P_X5],
                         (const float*)op->p[RCV_P_IN_C], (float*)op->p[RCV_P_OUT], (float*)op->p[RCV_P_X1], (float*)op->p[RCV_P_X2]);
      break;
    }
    case RCV_OP_COMBINE: {
      const int m2 = op->i[RCV_I_INMODE2];
      RCV_CHECK_ARG(Cout > 0 && Cout % 4 == 0, "combine: %d channels unsupported (multiple of 4)", Cout);
      RCV_CHECK_ARG(m2 == RCV_LOAD_PLAIN || m2 == RCV_LOAD_AFFINE || m2 == RCV_LOAD_AFFINE_RELU, "combine: skip load mode %d unsupported", m2);
      if (query) return RCV_OK;
      RCV_CHECK_ARG(op->p[RCV_P_IN] && op->p[RCV_P_IN_C] && op->p[RCV_P_IN2] && op->p[RCV_P_OUT], "combine: null operand");
      const size_t n4 = (size_t)N * H * W * Cout / 4;
      const int g = stream_grid(h, n4, 256);
      RCV_CHECK_ARG(m2 == RCV_LOAD_PLAIN || op->p[RCV_P_IN2_C], "combine: skip constants missing");
      const float* t = (const float*)op->p[RCV_P_IN]; const float* tc = (const float*)op->p[RCV_P_IN_C];
      const float* r = (const float*)op->p[RCV_P_IN2]; const float* rc = (const float*)op->p[RCV_P_IN2_C];
      float* out = (float*)op->p[RCV_P_OUT];
      if (op->flags & RCV_F_CONCAT) hipLaunchKernelGGL(concat_kernel, dim3(g), dim3(256), 0, s, t, tc, r, rc, out, n4, Cout, m2);
      else if (m2 == RCV_LOAD_PLAIN) hipLaunchKernelGGL(combine_kernel<RCV_LOAD_PLAIN>, dim3(g), dim3(256), 0, s, t, tc, r, rc, out, n4, Cout);
      else if (m2 == RCV_LOAD_AFFINE) hipLaunchKernelGGL(combine_kernel<RCV_LOAD_AFFINE>, dim3(g), dim3(256), 0, s, t, tc, r, rc, out, n4, Cout);
      else hipLaunchKernelGGL(combine_kernel<RCV_LOAD_AFFINE_RELU>, dim3(g), dim3(256), 0, s, t, tc, r, rc, out, n4, Cout);
      break;
    }
    case RCV_OP_CLS_FWD: {
      const bool fused = (op->flags & RCV_F_FUSED_UP) != 0, with_ce = (op->flags & RCV_F_FUSED_CE) != 0;
      const int gce = reduce_grid(h, (size_t)N * H * W, 256);      // with the loss: one partial row per workgroup, same grid as RCV_OP_CE_FWD
      RCV_CHECK_ARG((Cin == 8 || Cin == 16) && Cout >= 1 && Cout <= CLS_MAX_OUT,
                    "classifier: %d -> %d channels unsupported (8 or 16 input channels, 1..%d classes)", Cin, Cout, CLS_MAX_OUT);
      const int mode2 = op->i[RCV_I_AUX0];
      // i[RCV_I_AUX1]: channels of the skip tensor when it is narrower than the classifier input (0 = Cin)
      const int rch = (fused && op->i[RCV_I_AUX1] > 0) ? op->i[RCV_I_AUX1] : Cin;
      if (fused) {
        RCV_CHECK_ARG(rch % 4 == 0 && rch >= 4 && rch <= Cin && (Cin == 8 || !with_ce), "classifier (fused decoder output): %d skip channels for %d inputs", rch, Cin);
        RCV_CHECK_ARG(mode2 == RCV_LOAD_PLAIN || mode2 == RCV_LOAD_AFFINE || mode2 == RCV_LOAD_AFFINE_RELU, "classifier: skip load mode %d", mode2);
      }
      RCV_CHECK_ARG(!with_ce || (fused && Cout <= CE_MAX_C), "classifier + cross entropy: needs the fused decoder input and <= %d classes", CE_MAX_C);
      if (query) {
        if (with_ce) { query->n_part = gce; query->part_bytes = (size_t)gce * 3 * sizeof(float); }
        return RCV_OK;
      }
      RCV_CHECK_ARG(op->p[RCV_P_IN] && op->p[RCV_P_W] && op->p[RCV_P_OUT], "classifier: null operand");
      const int g = stream_grid(h, (size_t)N * H * W, 256);
      const float* tc = (const float*)op->p[RCV_P_IN_C]; const float* r = (const float*)op->p[RCV_P_X3]; const float* rc = (const float*)op->p[RCV_P_X4];
      const int64_t* tgt = (const int64_t*)op->p[RCV_P_IN2]; const float* cw = (const float*)op->p[RCV_P_X0];
      float* part = (float*)op->p[RCV_P_PART]; uint8_t* am = (uint8_t*)op->p[RCV_P_X2];
      if (fused) RCV_CHECK_ARG(tc && r && (mode2 == RCV_LOAD_PLAIN || rc), "classifier (fused decoder output): operands missing");
      if (with_ce) {
        RCV_CHECK_ARG(tgt && part && op->p[RCV_P_X1], "classifier + cross entropy: needs target, workspace, loss_out");
        RCV_CHECK_ARG(op->i[RCV_I_NPART] == gce, "classifier + cross entropy: workspace rows %d != %d", op->i[RCV_I_NPART], gce);
        hipLaunchKernelGGL((cls_fwd_kernel<8, true, true>), dim3(gce), dim3(256), 0, s, (const float*)op->p[RCV_P_IN], (const float*)op->p[RCV_P_W], (const float*)op->p[RCV_P_BIAS], (float*)op->p[RCV_P_OUT], N, H * W, Cout, tc, r, rc, mode2, tgt, cw, part, am, rch);
        RCV_HIP(hipGetLastError());
        hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, s, (const float*)part, gce, (float*)op->p[RCV_P_X1]);
      } else if (Cin == 16) {
        const int g16 = stream_grid(h, (size_t)N * H * W * 4, 256);        // four lanes per pixel
        if (fused) hipLaunchKernelGGL((cls_fwd16_kernel<true>), dim3(g16), dim3(256), 0, s, (const float*)op->p[RCV_P_IN], (const float*)op->p[RCV_P_W], (const float*)op->p[RCV_P_BIAS], (float*)op->p[RCV_P_OUT], N, H * W, Cout, tc, r, rc, mode2, rch);
        else hipLaunchKernelGGL((cls_fwd16_kernel<false>), dim3(g16), dim3(256), 0, s, (const float*)op->p[RCV_P_IN], (const float*)op->p[RCV_P_W], (const float*)op->p[RCV_P_BIAS], (float*)op->p[RCV_P_OUT], N, H * W, Cout, tc, r, rc, mode2, rch);
      } else if (fused) {
        hipLaunchKernelGGL((cls_fwd_kernel<8, true, false>), dim3(g), dim3(256), 0, s, (const float*)op->p[RCV_P_IN], (const float*)op->p[RCV_P_W], (const float*)op->p[RCV_P_BIAS], (float*)op->p[RCV_P_OUT], N, H * W, Cout, tc, r, rc, mode2, tgt, cw, part, am, rch);
      } else
        hipLaunchKernelGGL((cls_fwd_kernel<8, false, false>), dim3(g), dim3(256), 0, s, (const float*)op->p[RCV_P_IN], (const float*)op->p[RCV_P_W], (const float*)op->p[RCV_P_BIAS], (float*)op->p[RCV_P_OUT], N, H * W, Cout, tc, r, rc, mode2, tgt, cw, part, am, rch);
      break;
    }
    case RCV_OP_CLS_BWD: {
      // partial rows: stats [g][2][Cin] in p[PART]; filter+bias [g][Cout*Cin+Cout] in p[X0]; then reduced in-op
      const int g = reduce_grid(h, (size_t)N * H * W, 256);
      const size_t wrow = (size_t)Cout * Cin + Cout;
      const bool fused = (op->flags & RCV_F_FUSED_UP) != 0;
      RCV_CHECK_ARG((Cin == 8 || Cin == 16) && Cout >= 1 && Cout <= CLS_MAX_OUT,
                    "classifier backward: %d -> %d channels unsupported (8 or 16 input channels, 1..%d classes)", Cin, Cout, CLS_MAX_OUT);
      RCV_CHECK_ARG(Cin == 8 || !fused, "classifier backward (fused decoder output): 8 input channels only (got %d)", Cin);
      RCV_CHECK_ARG(!(op->flags & RCV_F_FUSED_CE) || (fused && op->i[RCV_I_STATS] == RCV_STATS_BWD_DEC),
                    "classifier + cross entropy backward: needs the fused decoder input with its BatchNorm-backward statistics");
      RCV_CHECK_ARG(!fused || op->i[RCV_I_STATS] == RCV_STATS_BWD_DEC, "classifier backward (fused decoder output): decoder statistics kind expected");
      RCV_CHECK_ARG(op->i[RCV_I_STATS] == RCV_STATS_NONE || op->i[RCV_I_STATS] == RCV_STATS_BWD_DEC, "classifier backward: statistics kind %d unsupported",
                    op->i[RCV_I_STATS]);
      if (query) {
        query->n_part = g;
        query->part_bytes = (size_t)g * (2 * Cin + wrow) * sizeof(float);
        return RCV_OK;
      }
      RCV_CHECK_ARG((fused || op->p[RCV_P_IN]) && op->p[RCV_P_IN2] && op->p[RCV_P_W] && op->p[RCV_P_OUT] && op->p[RCV_P_PART] && op->p[RCV_P_X1],
                    "classifier backward: null operand");
      RCV_CHECK_ARG(op->i[RCV_I_NPART] == g, "classifier backward: workspace rows %d != %d", op->i[RCV_I_NPART], g);
      const int stats = op->i[RCV_I_STATS];
      RCV_CHECK_ARG(stats == RCV_STATS_NONE || (op->p[RCV_P_EPI_AUX] && op->p[RCV_P_EPI_C]), "classifier backward: stats operands missing");
      float* stat_part = (float*)op->p[RCV_P_PART];
      float* w_part = stat_part + (size_t)g * 2 * Cin;
      const float* r = (const float*)op->p[RCV_P_X3]; const float* rc = (const float*)op->p[RCV_P_X4];
      const int mode2 = op->i[RCV_I_AUX0];
      const bool with_ce = (op->flags & RCV_F_FUSED_CE) != 0;
      const int64_t* tgt = (const int64_t*)op->p[RCV_P_IN2]; const float* cw = (const float*)op->p[RCV_P_X0];
      const float* bias = (const float*)op->p[RCV_P_BIAS]; const float* loss_out = (const float*)op->p[RCV_P_X5];
      const float* grad_out = (const float*)op->p[RCV_P_IN2_AUX];
      cls_bwd_fn kern;
      const float* dl = (const float*)op->p[RCV_P_IN2];
      if (with_ce) {
        RCV_CHECK_ARG(r && (mode2 == RCV_LOAD_PLAIN || rc) && tgt && loss_out && grad_out, "classifier + cross entropy backward: operands missing");
        kern = cls_bwd_pick<8, true, true>(Cout);
        dl = nullptr;
      } else if (fused) {
        RCV_CHECK_ARG(r && (mode2 == RCV_LOAD_PLAIN || rc), "classifier backward (fused decoder output): operands missing");
        kern = cls_bwd_pick<8, true, false>(Cout);
      } else {
        kern = Cin == 8 ? cls_bwd_pick<8, false, false>(Cout) : cls_bwd_pick<16, false, false>(Cout);
      }
      hipLaunchKernelGGL(kern, dim3(g), dim3(256), 0, s, (const float*)op->p[RCV_P_IN], dl, (const float*)op->p[RCV_P_W], (float*)op->p[RCV_P_OUT],
                         (const float*)op->p[RCV_P_EPI_AUX], (const float*)op->p[RCV_P_EPI_C], stat_part, w_part, N, H * W, stats, r, rc, mode2, tgt,
                         cw, bias, loss_out, grad_out);
      RCV_HIP(hipGetLastError());
      // dW -> p[X1] ([Cout][Cin]), db -> p[X2]
      hipLaunchKernelGGL(rows_reduce_kernel, dim3((int)wrow), dim3(256), 0, s, w_part, g, (int)wrow, (float*)op->p[RCV_P_X1], Cout * Cin,
                         (float*)op->p[RCV_P_X2]);
      break;
    }
    case RCV_OP_CE_FWD: {
      const int HW = H * W;
      const int g = reduce_grid(h, (size_t)N * HW, 256);
      RCV_CHECK_ARG(Cout >= 1 && Cout <= CE_MAX_C, "cross entropy: %d classes unsupported (max %d)", Cout, CE_MAX_C);
      if (query) { query->n_part = g; query->part_bytes = (size_t)g * 3 * sizeof(float); return RCV_OK; }
      RCV_CHECK_ARG(op->p[RCV_P_IN] && op->p[RCV_P_IN2] && op->p[RCV_P_PART] && op->p[RCV_P_OUT], "cross entropy: null operand");
      RCV_CHECK_ARG(op->i[RCV_I_NPART] == g, "cross entropy: workspace rows %d != %d", op->i[RCV_I_NPART], g);
      hipLaunchKernelGGL(ce_fwd_kernel, dim3(g), dim3(256), 0, s, (const float*)op->p[RCV_P_IN], (const int64_t*)op->p[RCV_P_IN2],
                         (const float*)op->p[RCV_P_W], N, Cout, HW, (float*)op->p[RCV_P_PART],
                         (op->flags & RCV_F_ARGMAX) ? (uint8_t*)op->p[RCV_P_X0] : nullptr);
      RCV_HIP(hipGetLastError());
      hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, s, (const float*)op->p[RCV_P_PART], g, (float*)op->p[RCV_P_OUT]);
      break;
    }
    case RCV_OP_CE_BWD: {
      RCV_CHECK_ARG(Cout >= 1 && Cout <= CE_MAX_C, "cross entropy: %d classes unsupported", Cout);
      if (query) return RCV_OK;
      RCV_CHECK_ARG(op->p[RCV_P_IN] && op->p[RCV_P_IN2] && op->p[RCV_P_X0] && op->p[RCV_P_X1] && op->p[RCV_P_OUT], "cross entropy backward: null operand");
      const int g = stream_grid(h, (size_t)N * H * W, 256);
      hipLaunchKernelGGL(ce_bwd_kernel, dim3(g), dim3(256), 0, s, (const float*)op->p[RCV_P_IN], (const int64_t*)op->p[RCV_P_IN2],
                         (const float*)op->p[RCV_P_W], (const float*)op->p[RCV_P_X0], (const float*)op->p[RCV_P_X1], N, Cout, H * W,
                         (float*)op->p[RCV_P_OUT]);
      break;
    }
    case RCV_OP_POOL_FWD: {
      RCV_CHECK_ARG(Cout > 0 && Cout % 4 == 0 && H % 2 == 0 && W % 2 == 0, "maxpool: C=%d H=%d W=%d unsupported", Cout, H, W);
      RCV_CHECK_ARG(op->i[RCV_I_INMODE] == RCV_LOAD_PLAIN || op->i[RCV_I_INMODE] == RCV_LOAD_AFFINE, "maxpool: load mode %d unsupported", op->i[RCV_I_INMODE]);
      if (query) return RCV_OK;
      RCV_CHECK_ARG(op->p[RCV_P_IN] && op->p[RCV_P_OUT], "maxpool: null operand");
      const int g = stream_grid(h, (size_t)N * (H / 2) * (W / 2) * (Cout / 4), 256);
      if (op->i[RCV_I_INMODE] == RCV_LOAD_PLAIN)
        hipLaunchKernelGGL(pool_fwd_kernel<RCV_LOAD_PLAIN>, dim3(g), dim3(256), 0, s, (const float*)op->p[RCV_P_IN], (const float*)op->p[RCV_P_IN_C], (float*)op->p[RCV_P_OUT], N, H, W, Cout);
      else {
        RCV_CHECK_ARG(op->p[RCV_P_IN_C], "maxpool: constants missing");
        hipLaunchKernelGGL(pool_fwd_kernel<RCV_LOAD_AFFINE>, dim3(g), dim3(256), 0, s, (const float*)op->p[RCV_P_IN], (const float*)op->p[RCV_P_IN_C], (float*)op->p[RCV_P_OUT], N, H, W, Cout);
      }
      break;
    }
    case RCV_OP_POOL_BWD: {
      RCV_CHECK_ARG(Cout % 4 == 0 && Cout <= 512 && 256 % (Cout / 4) == 0 && H % 2 == 0 && W % 2 == 0, "maxpool backward: C=%d H=%d W=%d unsupported", Cout, H, W);
      const int g = reduce_grid(h, (size_t)N * (H / 2) * (W / 2) * (Cout / 4), 256);
      const int stats = op->i[RCV_I_STATS];
      if (query) {
        query->n_part = stats != RCV_STATS_NONE ? g : 0;
        query->part_bytes = (size_t)query->n_part * 2 * Cout * sizeof(float);
        return RCV_OK;
      }
      RCV_CHECK_ARG(op->p[RCV_P_IN] && op->p[RCV_P_EPI_AUX] && op->p[RCV_P_OUT], "maxpool backward: null operand");
      RCV_CHECK_ARG(stats == RCV_STATS_NONE || (op->p[RCV_P_PART] && op->i[RCV_I_NPART] == g), "maxpool backward: workspace rows mismatch");
      const float* resid = (op->flags & RCV_F_RESID) ? (const float*)op->p[RCV_P_RESID] : nullptr;
      const size_t lds = 2 * 256 * sizeof(float4);
      const int C4 = Cout / 4;
      const bool rows = (C4 & (C4 - 1)) == 0 && C4 < 64 && ((long long)W * C4) % 64 == 0 && !RCV_ENV("RCV_NO_POOL_ROWS");      // see pool_bwd_rows_kernel
      if (rows) {
        if (op->i[RCV_I_INMODE] != RCV_LOAD_PLAIN) RCV_CHECK_ARG(op->p[RCV_P_IN_C], "maxpool backward: constants missing");
        auto kern = op->i[RCV_I_INMODE] == RCV_LOAD_PLAIN ? pool_bwd_rows_kernel<RCV_LOAD_PLAIN> : pool_bwd_rows_kernel<RCV_LOAD_AFFINE>;
        hipLaunchKernelGGL(kern, dim3(g), dim3(256), lds, s, (const float*)op->p[RCV_P_IN], (const float*)op->p[RCV_P_EPI_AUX], (const float*)op->p[RCV_P_IN_C], resid, (float*)op->p[RCV_P_OUT], (float*)op->p[RCV_P_PART], N, H, W, Cout, stats);
      } else if (op->i[RCV_I_INMODE] == RCV_LOAD_PLAIN)
        hipLaunchKernelGGL(pool_bwd_kernel<RCV_LOAD_PLAIN>, dim3(g), dim3(256), lds, s, (const float*)op->p[RCV_P_IN], (const float*)op->p[RCV_P_EPI_AUX], (const float*)op->p[RCV_P_IN_C], resid, (float*)op->p[RCV_P_OUT], (float*)op->p[RCV_P_PART], N, H, W, Cout, stats);
      else {
        RCV_CHECK_ARG(op->p[RCV_P_IN_C], "maxpool backward: constants missing");
        hipLaunchKernelGGL(pool_bwd_kernel<RCV_LOAD_AFFINE>, dim3(g), dim3(256), lds, s, (const float*)op->p[RCV_P_IN], (const float*)op->p[RCV_P_EPI_AUX], (const float*)op->p[RCV_P_IN_C], resid, (float*)op->p[RCV_P_OUT], (float*)op->p[RCV_P_PART], N, H, W, Cout, stats);
      }
      break;
    }
    case RCV_OP_ADAM_L1: {
      const size_t n = (size_t)(uint32_t)op->i[RCV_I_COUNT];
      if (query) {                                       // workspace of the metrics variant: one double per workgroup + the ticket
        query->n_part = adam_grid(h, n);
        query->part_bytes = ((size_t)query->n_part + 1) * sizeof(double);
        return RCV_OK;
      }
      const int step = op->i[RCV_I_AUX0];
      const int* step_dev = (const int*)op->p[RCV_P_IN_AUX];
      RCV_CHECK_ARG(n > 0 && (step >= 1 || step_dev) && op->p[RCV_P_IN] && op->p[RCV_P_IN2] && op->p[RCV_P_X0] && op->p[RCV_P_X1], "adam: bad operand");
      const float b1 = op->f[1], b2 = op->f[2];
      // 16-byte vector path when every buffer is aligned for it (flat buffers of whole allocations are); otherwise element by element
      const int vec = ((((uintptr_t)op->p[RCV_P_IN] | (uintptr_t)op->p[RCV_P_IN2] | (uintptr_t)op->p[RCV_P_X0] | (uintptr_t)op->p[RCV_P_X1] |
                         (uintptr_t)op->p[RCV_P_X2]) & 15) == 0 && ((uintptr_t)op->p[RCV_P_X5] & 3) == 0) ? 1 : 0;
      const int g = adam_grid(h, n);
      double* part = (double*)op->p[RCV_P_PART];
      if (op->p[RCV_P_X3]) {
        RCV_CHECK_ARG(part && op->p[RCV_P_X4] && op->i[RCV_I_NPART] == g, "adam + metrics: needs loss stats and a %d-row workspace (rcv_op_workspace)", g);
        hipLaunchKernelGGL(adam_l1_kernel<true>, dim3(g), dim3(256), 0, s, (float*)op->p[RCV_P_IN], (const float*)op->p[RCV_P_IN2],
                           (float*)op->p[RCV_P_X0], (float*)op->p[RCV_P_X1], (const float*)op->p[RCV_P_X2], n, op->f[0], b1, b2, op->f[3],
                           op->f[4], op->f[5], step, step_dev, part, (const float*)op->p[RCV_P_X4], (double*)op->p[RCV_P_X3],
                           (const uint8_t*)op->p[RCV_P_X5], vec);
      } else {
        hipLaunchKernelGGL(adam_l1_kernel<false>, dim3(g), dim3(256), 0, s, (float*)op->p[RCV_P_IN], (const float*)op->p[RCV_P_IN2],
                           (float*)op->p[RCV_P_X0], (float*)op->p[RCV_P_X1], (const float*)op->p[RCV_P_X2], n, op->f[0], b1, b2, op->f[3],
                           op->f[4], op->f[5], step, step_dev, (double*)nullptr, (const float*)nullptr, (double*)nullptr,
                           (const uint8_t*)op->p[RCV_P_X5], vec);
      }
      break;
    }
    case RCV_OP_SGD: {
      if (query) return RCV_OK;
      const size_t n = (size_t)(uint32_t)op->i[RCV_I_COUNT];
      RCV_CHECK_ARG(op->p[RCV_P_IN] && op->p[RCV_P_IN2] && op->p[RCV_P_X0] && n > 0 && op->i[RCV_I_AUX0] >= 1, "sgd: bad operand");
      hipLaunchKernelGGL(sgd_kernel, dim3(stream_grid(h, n, 256)), dim3(256), 0, s, (float*)op->p[RCV_P_IN], (const float*)op->p[RCV_P_IN2],
                         (float*)op->p[RCV_P_X0], (const float*)op->p[RCV_P_X2], n, op->f[0], op->f[1], op->f[2], op->f[5],
                         op->i[RCV_I_AUX0] == 1 ? 1 : 0);
      break;
    }
    case RCV_OP_MEMSET: {
      if (query) return RCV_OK;
      const size_t n = (size_t)(uint32_t)op->i[RCV_I_COUNT];
      RCV_CHECK_ARG(op->p[RCV_P_OUT] && n > 0, "memset: bad operand");
      hipLaunchKernelGGL(memset_kernel, dim3(stream_grid(h, n, 256)), dim3(256), 0, s, (float*)op->p[RCV_P_OUT], n);
      break;
    }
    case RCV_OP_ADD_SLICE: {
      const int Ca = Cin;
      RCV_CHECK_ARG(Ca > 0 && Ca % 4 == 0 && Cout % 4 == 0 && Ca <= Cout, "add_slice: %d channels into %d unsupported", Ca, Cout);
      if (query) return RCV_OK;
      RCV_CHECK_ARG(op->p[RCV_P_OUT] && op->p[RCV_P_IN], "add_slice: null operand");
      const size_t npix = (size_t)N * H * W;
      hipLaunchKernelGGL(add_slice_kernel, dim3(stream_grid(h, npix * (Ca / 4), 256)), dim3(256), 0, s, (float*)op->p[RCV_P_OUT], Cout,
                         (const float*)op->p[RCV_P_IN], (const float*)op->p[RCV_P_IN_C], Ca, npix, op->i[RCV_I_INMODE]);
      break;
    }
    case RCV_OP_CONFUSION: {
      RCV_CHECK_ARG(Cout >= 1 && Cout <= 8 && N > 0 && H > 0 && W > 0, "confusion: %d classes unsupported (max 8)", Cout);
      if (query) return RCV_OK;
      RCV_CHECK_ARG(op->p[RCV_P_IN] && op->p[RCV_P_IN2] && op->p[RCV_P_OUT], "confusion: null operand");
      int gx = ceil_div(H * W, 256 * 8);
      if (gx > 64) gx = 64;
      hipLaunchKernelGGL(confusion_kernel, dim3(gx, N), dim3(256), 0, s, (const uint8_t*)op->p[RCV_P_IN], (const int64_t*)op->p[RCV_P_IN2],
                         (int*)op->p[RCV_P_OUT], H * W, Cout);
      break;
    }
    case RCV_OP_MATERIALIZE: {
      RCV_CHECK_ARG(Cout > 0 && Cout % 4 == 0, "materialize: %d channels unsupported (multiple of 4)", Cout);
      if (query) return RCV_OK;
      RCV_CHECK_ARG(op->p[RCV_P_IN] && op->p[RCV_P_OUT], "materialize: null operand");
      RCV_CHECK_ARG(op->i[RCV_I_INMODE] == RCV_LOAD_PLAIN || op->p[RCV_P_IN_C], "materialize: constants missing");
      const size_t n4 = (size_t)N * H * W * Cout / 4;
      hipLaunchKernelGGL(materialize_kernel, dim3(stream_grid(h, n4, 256)), dim3(256), 0, s, (const float*)op->p[RCV_P_IN],
                         (const float*)op->p[RCV_P_IN_C], (float*)op->p[RCV_P_OUT], n4, Cout, op->i[RCV_I_INMODE]);
      break;
    }
    case RCV_OP_BWD_STATS: {
      RCV_CHECK_ARG(Cout % 4 == 0 && Cout <= 1024 && 256 % (Cout / 4) == 0, "bwd_stats: C=%d unsupported", Cout);
      const size_t n4 = (size_t)N * H * W * Cout / 4;
      const int stats = op->i[RCV_I_STATS];
      const int Csrc = Cin > 0 ? Cin : Cout, coff = op->i[RCV_I_AUX0];
      RCV_CHECK_ARG(Csrc % 4 == 0 && coff % 4 == 0 && coff >= 0 && coff + Cout <= Csrc, "bwd_stats: slice [%d,%d) of %d channels", coff,
                    coff + Cout, Csrc);
      if (stats == RCV_STATS_NONE) {
        if (query) return RCV_OK;
        RCV_CHECK_ARG(op->p[RCV_P_IN] && op->p[RCV_P_OUT], "bwd_stats: null operand");
        hipLaunchKernelGGL(bwd_stats_kernel, dim3(stream_grid(h, n4, 256)), dim3(256), 0, s, (const float*)op->p[RCV_P_IN], (const float*)nullptr,
                           (const float*)nullptr, (float*)op->p[RCV_P_OUT], (float*)nullptr, n4, Cout, stats, Csrc, coff);
        break;
      }
      const int g = reduce_grid(h, n4, 256);
      if (query) { query->n_part = g; query->part_bytes = (size_t)g * 2 * Cout * sizeof(float); return RCV_OK; }
      RCV_CHECK_ARG(op->p[RCV_P_IN] && op->p[RCV_P_EPI_AUX] && op->p[RCV_P_OUT] && op->p[RCV_P_PART], "bwd_stats: null operand");
      RCV_CHECK_ARG(op->p[RCV_P_EPI_C], "bwd_stats: constants (scale, shift, mean) missing");
      RCV_CHECK_ARG(op->i[RCV_I_NPART] == g, "bwd_stats: workspace rows %d != %d", op->i[RCV_I_NPART], g);
      hipLaunchKernelGGL(bwd_stats_kernel, dim3(g), dim3(256), 2 * 256 * sizeof(float4), s, (const float*)op->p[RCV_P_IN],
                         (const float*)op->p[RCV_P_EPI_AUX], (const float*)op->p[RCV_P_EPI_C], (float*)op->p[RCV_P_OUT],
                         (float*)op->p[RCV_P_PART], n4, Cout, stats, Csrc, coff);
      break;
    }
    case RCV_OP_DICE_FWD: {
      const int HW = H * W;
      const int g = reduce_grid(h, (size_t)N * HW, 256);
      RCV_CHECK_ARG(Cout >= 2 && Cout <= CE_MAX_C, "dice loss: %d classes unsupported (2..%d)", Cout, CE_MAX_C);
      if (query) { query->n_part = g; query->part_bytes = (size_t)g * DICE_ROW * sizeof(float); return RCV_OK; }
      RCV_CHECK_ARG(op->p[RCV_P_IN] && op->p[RCV_P_IN2] && op->p[RCV_P_PART] && op->p[RCV_P_OUT], "dice loss: null operand");
      RCV_CHECK_ARG(op->i[RCV_I_NPART] == g, "dice loss: workspace rows %d != %d", op->i[RCV_I_NPART], g);
      hipLaunchKernelGGL(dice_fwd_kernel, dim3(g), dim3(256), 0, s, (const float*)op->p[RCV_P_IN], (const int64_t*)op->p[RCV_P_IN2], N, Cout,
                         HW, (float*)op->p[RCV_P_PART], (op->flags & RCV_F_ARGMAX) ? (uint8_t*)op->p[RCV_P_X0] : nullptr);
      RCV_HIP(hipGetLastError());
      hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(256), 0, s, (const float*)op->p[RCV_P_PART], g, (const float*)op->p[RCV_P_W],
                         Cout, op->f[1], (float*)op->p[RCV_P_OUT]);
      break;
    }
    case RCV_OP_DICE_BWD: {
      RCV_CHECK_ARG(Cout >= 2 && Cout <= CE_MAX_C, "dice loss: %d classes unsupported", Cout);
      if (query) return RCV_OK;
      RCV_CHECK_ARG(op->p[RCV_P_IN] && op->p[RCV_P_IN2] && op->p[RCV_P_X0] && op->p[RCV_P_X1] && op->p[RCV_P_OUT], "dice loss backward: null operand");
      hipLaunchKernelGGL(dice_bwd_kernel, dim3(stream_grid(h, (size_t)N * H * W, 256)), dim3(256), 0, s, (const float*)op->p[RCV_P_IN],
                         (const int64_t*)op->p[RCV_P_IN2], (const float*)op->p[RCV_P_X0], (const float*)op->p[RCV_P_X1], N, Cout, H * W,
                         (float*)op->p[RCV_P_OUT]);
      break;
    }
    case RCV_OP_NHWC_TO_NCHW: {
      RCV_CHECK_ARG(Cin == 8 && Cout >= 1 && Cout <= 8, "nhwc_to_nchw: %d -> %d channels unsupported (8 padded channels)", Cin, Cout);
      if (query) return RCV_OK;
      RCV_CHECK_ARG(op->p[RCV_P_IN] && op->p[RCV_P_OUT], "nhwc_to_nchw: null operand");
      hipLaunchKernelGGL(nhwc_to_nchw_kernel<8>, dim3(stream_grid(h, (size_t)N * H * W, 256)), dim3(256), 0, s, (const float*)op->p[RCV_P_IN],
                         (const float*)op->p[RCV_P_BIAS], (float*)op->p[RCV_P_OUT], N, H * W, Cout);
      break;
    }
    case RCV_OP_NCHW_TO_NHWC: {
      RCV_CHECK_ARG(Cout == 8 && Cin >= 1 && Cin <= 8, "nchw_to_nhwc: %d -> %d channels unsupported (8 padded channels)", Cin, Cout);
      if (query) return RCV_OK;
      RCV_CHECK_ARG(op->p[RCV_P_IN] && op->p[RCV_P_OUT], "nchw_to_nhwc: null operand");
      hipLaunchKernelGGL(nchw_to_nhwc_kernel<8>, dim3(stream_grid(h, (size_t)N * H * W, 256)), dim3(256), 0, s, (const float*)op->p[RCV_P_IN],
                         (float*)op->p[RCV_P_OUT], N, H * W, Cin);
      break;
    }
    default:
      rcv_set_error("unknown op kind %d", op->kind);
      return RCV_E_ARG;
  }
  RCV_HIP(hipGetLastError());
  return RCV_OK;
}

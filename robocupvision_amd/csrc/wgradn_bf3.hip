// 3x3 filter gradient of the narrow (<= 32-channel) layers with fp32 products formed on the bf16 matrix pipe.
//
//   dW[cb][ca][tap] = sum_p  P[p][cb] * G[s*p + tap - 1][ca]      (operands, load modes and partial-filter layout of wgrad_mfma.hip)
//
// The fp32 kernels of these layers (wgrad_mfma.hip, shared-role tiles) run at 0.38-0.5 of either roofline: their staging and MFMA phases
// add up, and 16 -> 16 at 32 x 240 x 320 needs 72 us of v_mfma_f32_16x16x4_f32 at peak against 39 us of HBM traffic.  Here, as in
// wgrad_bf3.hip (which has the arithmetic: every operand split exactly into three bf16 values, six bf16 MFMA products per multiply-add,
// and the transposing LDS read that feeds the MFMA from [pixel][channel] images), with the tiles of a narrow layer:
//   * one workgroup = ALL channels of the layer (CBT x CAT = 16 | 32 each, 8-channel operands zero padded to 16) and a K split: its four
//     consumer waves take the 32-pixel k-steps of a tile round robin, each with the full 9 x MB x NB accumulator set; every wave writes
//     its own partial filter (split index = 4 * workgroup + wave), the RCV_OP_WGRAD_REDUCE pass sums them in a fixed order;
//   * pointwise tile TH x 16 pixels (TH = 8 | 16: 4 | 8 k-steps per barrier), gathered tile ((TH-1) s + 3) x (15 s + 3) pixels, stride
//     s = 1 | 2; four producer waves stage tile i + 1 while tile i is contracted;
//   * LDS images [plane h|m|l][pixel][channels] bf16.  16 channels: pixel pitch 32 B (eight consecutive pixels x 32 B tile the 256-byte
//     bank window).  32 channels: the gathered image has a 96-byte pitch (64 + 32 pad: 3 r mod 8 is a bijection, tap shifts stay
//     immediates), the pointwise image a 64-byte pitch with its two 32-byte channel blocks swapped on every second group of four
//     pixels (its reads start at multiples of eight pixels: a per-lane constant).
#include "wgrad_common.h"

typedef __bf16 w3_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 w3_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 w3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) w3_bf16x4 w3_lds_bf16x4;
typedef __attribute__((address_space(3))) char w3_lds_char;

constexpr int w3_gpitch(int ch) { return ch == 16 ? 32 : 96; }
constexpr int w3_ppitch(int ch) { return ch == 16 ? 32 : 64; }

template <int CBT, int CAT, int S, int TH>
struct W3Geom {
  static constexpr int TW = 16, IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, GPIX = IH * IW, PPIX = TH * TW;
  static constexpr int GP = w3_gpitch(CAT), PP = w3_ppitch(CBT);
  static constexpr int GPLANE = GPIX * GP, PPLANE = PPIX * PP;
  static constexpr int BUF = (3 * GPLANE + 3 * PPLANE + 15) / 16 * 16;
};

__device__ __forceinline__ uint32_t w3_pack(float a, float b) {
  const w3_bf16x2 v = {(__bf16)a, (__bf16)b};            // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(uint32_t, v);
}
struct W3Tri { uint32_t h, m, l; };
__device__ __forceinline__ W3Tri w3_split2(float x0, float x1) {
  W3Tri t;
  t.h = w3_pack(x0, x1);
  const float r0 = x0 - __uint_as_float(t.h << 16), r1 = x1 - __uint_as_float(t.h & 0xffff0000u);       // exact
  t.m = w3_pack(r0, r1);
  const float s0 = r0 - __uint_as_float(t.m << 16), s1 = r1 - __uint_as_float(t.m & 0xffff0000u);       // exact
  t.l = w3_pack(s0, s1);
  return t;
}

// One operand tile in two steps (all loads in flight, then transform + split + LDS writes).  CH = staged channels (16 | 32): CH / 4
// threads share a pixel; a quad beyond the tensor's C channels, or a pixel outside the plane, is stored as zero.
template <int NPIX, int TWP, int CH, bool TWO>
struct W3Regs {
  static constexpr int PP = 256 / (CH / 4), NU = (NPIX + PP - 1) / PP;
  float4 x[NU], ax[TWO ? NU : 1];
  bool ok[NU];
};
template <int NPIX, int TWP, int CH, bool TWO>
__device__ __forceinline__ void w3_load(W3Regs<NPIX, TWP, CH, TWO>& r, const float* __restrict__ src, const float* __restrict__ aux, bool two, int tid, int C,
                                        int row0, int oy, int ox, int PH, int PW) {
  using R = W3Regs<NPIX, TWP, CH, TWO>;
  const int q = tid % (CH / 4), lp = tid / (CH / 4);
  const bool ch_ok = 4 * q < C;
#pragma unroll
  for (int u = 0; u < R::NU; ++u) {
    const int pix = u * R::PP + lp;
    const int iy = pix / TWP, ix = pix - iy * TWP;                  // (compile-time divisor)
    r.ok[u] = ch_ok && pix < NPIX && (unsigned)(oy + iy) < (unsigned)PH && (unsigned)(ox + ix) < (unsigned)PW;
    const uint32_t o = r.ok[u] ? (uint32_t)(((row0 + oy + iy) * PW + ox + ix) * C + 4 * q) : 0u;
    r.x[u] = wld4(src + o);
    if (TWO) { if (two) r.ax[u] = wld4(aux + o); }
  }
}
// IS_P: the pointwise image (its pitch and, with 32 channels, its block swizzle)
template <int MODE, int NPIX, int TWP, int CH, bool TWO, bool SUM, bool IS_P = SUM>
__device__ __forceinline__ void w3_store(const W3Regs<NPIX, TWP, CH, TWO>& r, const float* __restrict__ consts, char* img, int tid, int C, float4& sum) {
  using R = W3Regs<NPIX, TWP, CH, TWO>;
  constexpr int PITCH = IS_P ? w3_ppitch(CH) : w3_gpitch(CH), PLANE = NPIX * PITCH;
  const int q = tid % (CH / 4), lp = tid / (CH / 4);
  const bool ch_ok = 4 * q < C;
  float4 k[5];
  if (MODE != RCV_LOAD_PLAIN) {
#pragma unroll
    for (int j = 0; j < 5; ++j) k[j] = ch_ok ? wld4(consts + (size_t)j * C + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int u = 0; u < R::NU; ++u) {
    const int pix = u * R::PP + lp;
    float4 v = wxform4<MODE>(r.x[u], r.ax[TWO ? u : 0], k);
    if (!r.ok[u]) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (SUM) { sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w; }
    if (pix < NPIX) {
      const W3Tri a = w3_split2(v.x, v.y), b = w3_split2(v.z, v.w);
      char* d = img + pix * PITCH + 8 * q;
      if (IS_P && CH == 32) d = img + pix * PITCH + ((((q >> 2) ^ ((pix >> 2) & 1))) << 5) + 8 * (q & 3);
      *reinterpret_cast<uint2*>(d) = make_uint2(a.h, b.h);
      *reinterpret_cast<uint2*>(d + PLANE) = make_uint2(a.m, b.m);
      *reinterpret_cast<uint2*>(d + 2 * PLANE) = make_uint2(a.l, b.l);
    }
  }
}

__device__ __forceinline__ w3_bf16x8 w3_read(const w3_lds_char* p, int off0, int off1) {
  const w3_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((w3_lds_bf16x4*)(p + off0));
  const w3_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((w3_lds_bf16x4*)(p + off1));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// GTWO: the gathered operand may be a two-tensor gradient load (then the pointwise one is not); otherwise the pointwise one may be.
template <int CBT, int CAT, int S, int TH, bool GTWO>
__global__ __launch_bounds__(512) void wgradn_bf3_kernel(const WgradArgs a) {
  using G = W3Geom<CBT, CAT, S, TH>;
  constexpr int MB = CBT / 16, NB = CAT / 16, NKS = TH / 2;          // k-steps (32 pixels = two tile rows) per tile
  extern __shared__ __attribute__((aligned(16))) char smem_w3[];
  const bool producer = threadIdx.x >= 256;
  const int tid = producer ? (int)threadIdx.x - 256 : (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);

  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (producer) {
    auto stage = [&](int tile, char* buf) {
      int t = tile;
      const int tx_i = t % a.tiles_x;
      t /= a.tiles_x;
      const int ty_i = t % a.tiles_y;
      const int n = t / a.tiles_y;
      const int y0 = ty_i * TH, x0 = tx_i * G::TW;
      float4 nosum = make_float4(0.f, 0.f, 0.f, 0.f);
      W3Regs<G::GPIX, G::IW, CAT, GTWO> rg;
      W3Regs<G::PPIX, G::TW, CBT, !GTWO> rp;
      const bool p_two = !GTWO && (a.p_mode == RCV_LOAD_GRAD_ENC || a.p_mode == RCV_LOAD_GRAD_DEC);
      w3_load(rg, a.g, a.g_aux, true, tid, a.CA, n * a.H, y0 * S - 1, x0 * S - 1, a.H, a.W);
      w3_load(rp, a.p, a.p_aux, p_two, tid, a.CB, n * a.Hp, y0, x0, a.Hp, a.Wp);
      char* gi = buf;
      char* pi = buf + 3 * G::GPLANE;
      if (GTWO) {
        if (a.g_mode == RCV_LOAD_GRAD_ENC) w3_store<RCV_LOAD_GRAD_ENC, G::GPIX, G::IW, CAT, GTWO, false>(rg, a.g_c, gi, tid, a.CA, nosum);
        else w3_store<RCV_LOAD_GRAD_DEC, G::GPIX, G::IW, CAT, GTWO, false>(rg, a.g_c, gi, tid, a.CA, nosum);
      } else {
        switch (a.g_mode) {
          case RCV_LOAD_PLAIN: w3_store<RCV_LOAD_PLAIN, G::GPIX, G::IW, CAT, GTWO, false>(rg, a.g_c, gi, tid, a.CA, nosum); break;
          case RCV_LOAD_AFFINE: w3_store<RCV_LOAD_AFFINE, G::GPIX, G::IW, CAT, GTWO, false>(rg, a.g_c, gi, tid, a.CA, nosum); break;
          default: w3_store<RCV_LOAD_AFFINE_RELU, G::GPIX, G::IW, CAT, GTWO, false>(rg, a.g_c, gi, tid, a.CA, nosum); break;
        }
      }
      switch (a.p_mode) {
        case RCV_LOAD_PLAIN: w3_store<RCV_LOAD_PLAIN, G::PPIX, G::TW, CBT, !GTWO, true>(rp, a.p_c, pi, tid, a.CB, bsum); break;
        case RCV_LOAD_AFFINE: w3_store<RCV_LOAD_AFFINE, G::PPIX, G::TW, CBT, !GTWO, true>(rp, a.p_c, pi, tid, a.CB, bsum); break;
        case RCV_LOAD_AFFINE_RELU: w3_store<RCV_LOAD_AFFINE_RELU, G::PPIX, G::TW, CBT, !GTWO, true>(rp, a.p_c, pi, tid, a.CB, bsum); break;
        case RCV_LOAD_GRAD_ENC: if constexpr (!GTWO) w3_store<RCV_LOAD_GRAD_ENC, G::PPIX, G::TW, CBT, !GTWO, true>(rp, a.p_c, pi, tid, a.CB, bsum); break;
        default: if constexpr (!GTWO) w3_store<RCV_LOAD_GRAD_DEC, G::PPIX, G::TW, CBT, !GTWO, true>(rp, a.p_c, pi, tid, a.CB, bsum); break;
      }
    };
    // barrier for barrier the consumer path: 1 + one per tile + the bias partial's two.
    // What bounds this kernel on the layers whose filter gradient is an HBM stream (16 channels on a side), by ablation at 16 -> 16
    // (scripts/experiments/exp_r3_w3abl.sh): 138 us as it is, 101 with every load replaced by one cached element (same instructions, no
    // HBM traffic), 88 with the MFMAs removed as well -- the staging instruction stream itself.  Load transform 4, split 5.5, bounds /
    // addresses / bias sum ~5 vector instructions per ELEMENT: 2 320 channel quads x 60 instructions per tile = 2 200 cycles of the CU's
    // four SIMDs at full rate, twice that as one wave per SIMD issues them.  Tried on that evidence and not kept: a second register set
    // of loads in flight (0.147 -> 0.144 ms), a straight-line producer loop with the modes resolved outside it (same), SHARED roles --
    // all eight waves stage and contract, loads of the next tile in flight during the contraction (0.140 -> 0.132 ms at 16 -> 16; its
    // no-HBM, no-MFMA floor is still 74 us).  In this arithmetic such a layer is vector-ALU bound at ~1.5 x its HBM time; the fp32
    // kernels stage with 4-5 instructions per element and stay the faster ones where the matrix pipe is not the limit.
    const bool do_stage = !(a.dbg & RCV_F_DBG_NOSTAGE);           // (ablation timings: scripts/bench_op.py --flags)
    if (wg < a.ntiles && do_stage) stage(wg, smem_w3);
    __syncthreads();
    int it = 0;
    for (int tile = wg; tile < a.ntiles; tile += gridDim.x, ++it) {
      const int next = tile + gridDim.x;
      if (next < a.ntiles && do_stage) stage(next, smem_w3 + ((it + 1) & 1) * G::BUF);
      __syncthreads();
    }
    // bias partial: sum over the staging threads that hold the same channel quad (fixed order); the workgroup's row is that of its
    // wave 0 (split 4 wg), the rows of the other three waves are zero
    if (a.part_bias) {
      __syncthreads();
      float4* sb = reinterpret_cast<float4*>(smem_w3);
      sb[tid] = bsum;
      __syncthreads();
      constexpr int QP = CBT / 4;
      if (tid < QP) {
        float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = tid; e < 256; e += QP) { const float4 v = sb[e]; u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w; }
        if (4 * tid < a.CBP) {
          *reinterpret_cast<float4*>(a.part_bias + (size_t)(4 * blockIdx.x) * a.CBP + 4 * tid) = u;
#pragma unroll
          for (int w = 1; w < 4; ++w) *reinterpret_cast<float4*>(a.part_bias + (size_t)(4 * blockIdx.x + w) * a.CBP + 4 * tid) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    return;
  }

  // ---------------- consumer waves: wave w takes the k-steps w, w + 4, ... of every tile ----------------
  f32x4 acc[9][MB][NB];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int nn = 0; nn < NB; ++nn) acc[t][m][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // Lane 4 q + p of a 16-lane group supplies the address of block row q, channels 4 p .. 4 p + 3; the lane's pixel inside a k-step is
  // 16 h + 4 g + q = tile row 2 j + h, column 4 g + q.
  const int q = l15 >> 2, p = l15 & 3;
  const int kp = 4 * l4 + q;
  int a_lane[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m) a_lane[m] = 3 * G::GPLANE + kp * G::PP + ((CBT == 32 ? (m ^ ((kp >> 2) & 1)) : m) << 5) + 8 * p;
  const int g_lane = (kp * S) * G::GP + 8 * p;
  const w3_lds_char* lds0 = (const w3_lds_char*)smem_w3;

  auto contract = [&](const w3_lds_char* buf) {
    for (int j = wave; j < NKS; j += 4) {
      const w3_lds_char* pa = buf + (32 * j) * G::PP;
      const w3_lds_char* pg = buf + g_lane + ((2 * j) * S * G::IW) * G::GP;
      w3_bf16x8 A[MB][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int m = 0; m < MB; ++m) A[m][pl] = w3_read(pa + a_lane[m], pl * G::PPLANE, pl * G::PPLANE + 16 * G::PP);
      auto load_b = [&](int t, w3_bf16x8 (&B)[NB][3]) {
        const int ky = t / 3, kx = t % 3;
        const int r0 = ky * G::IW + kx, r1 = (S + ky) * G::IW + kx;         // tile rows 2 j and 2 j + 1
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int nn = 0; nn < NB; ++nn) B[nn][pl] = w3_read(pg, pl * G::GPLANE + r0 * G::GP + nn * 32, pl * G::GPLANE + r1 * G::GP + nn * 32);
      };
      w3_bf16x8 Bb[2][NB][3];
      load_b(0, Bb[0]);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t < 8) load_b(t + 1, Bb[(t + 1) & 1]);
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};      // smallest products first
#pragma unroll
        for (int e = 0; e < 6; ++e)
#pragma unroll
          for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int nn = 0; nn < NB; ++nn) acc[t][m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[m][TA[e]], Bb[t & 1][nn][TB[e]], acc[t][m][nn], 0, 0, 0);
        if (t < 8) {
#pragma unroll
          for (int e = 0; e < 6 * NB; ++e) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  __syncthreads();
  {
    int it = 0;
    for (int tile = wg; tile < a.ntiles; tile += gridDim.x, ++it) {
      if (!(a.dbg & RCV_F_DBG_NOMFMA)) contract(lds0 + (it & 1) * G::BUF);
      __syncthreads();
    }
  }
  const size_t split = (size_t)4 * blockIdx.x + wave;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int nn = 0; nn < NB; ++nn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cb = m * 16 + 4 * l4 + r, ca = nn * 16 + l15;
          if (cb < a.CBP && ca < a.CAP) a.part[((split * 9 + t) * a.CBP + cb) * a.CAP + ca] = acc[t][m][nn][r];
        }
  if (a.part_bias) { __syncthreads(); __syncthreads(); }
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
static inline int w3_ct(int c) { return c <= 16 ? 16 : 32; }
static inline size_t w3_buf(int cbt, int cat, int s, int th) {
  const int ih = (th - 1) * s + 3, iw = 15 * s + 3;
  return ((size_t)3 * ih * iw * w3_gpitch(cat) + (size_t)3 * th * 16 * w3_ppitch(cbt) + 15) / 16 * 16;
}

bool wgradn_bf3_supported(const rcv_handle* h, const rcv_op* op) {
  if (RCV_ENV("RCV_NO_BF3") || RCV_ENV("RCV_NO_BF3W") || (op->flags & RCV_F_MFMA_FP32)) return false;
  const int CA = op->i[RCV_I_CIN], CB = op->i[RCV_I_COUT], s = op->i[RCV_I_STRIDE];
  if (op->i[RCV_I_DIL] != 1 || op->i[RCV_I_INMODE] == RCV_LOAD_NCHW || (s != 1 && s != 2)) return false;
  // (8-channel operands would be staged zero padded to 16: half of their staging threads idle and twice the LDS image -- measured 176 ->
  // 246 us on the 8 <-> 16 stride-2 layers, which are HBM bound; they stay on the folded fp32 tile.  Also measured, correct to 1e-7 and
  // not kept: the pixel-PAIR view of those layers -- the gathered [H][W][8] tensor read as [H][W/2][16], the 16 MFMA columns = (pixel
  // parity, channel); with stride 2 the taps kx = 1, 2 of an output pixel are exactly one pair and kx = 0 the odd half of the pair
  // before it, so six column blocks carry the nine taps with the natural bytes in LDS: 207 / 228 us against 181 / 192 for the folded
  // tile.  What these layers lack here is not MFMA time but streaming rate: one 512-thread workgroup per CU reads 3.0-3.5 TB/s.)
  if (CA < 16 || CA > 32 || CB < 16 || CB > 32 || CA % 4 || CB % 4) return false;
  if (op->i[RCV_I_WO] < 16 || op->i[RCV_I_HO] < 8) return false;
  const long tiles = (long)op->i[RCV_I_N] * ceil_div(op->i[RCV_I_HO], 8) * ceil_div(op->i[RCV_I_WO], 16);
  if (tiles < h->num_cus) return false;                                   // small planes stay on the other kernels
  return 2 * w3_buf(w3_ct(CB), w3_ct(CA), s, 8) <= (size_t)h->max_lds;
}

// tile rows (16 where two buffers of it fit, else 8), tile counts and the workgroup count; n_split = 4 per workgroup (one per consumer wave)
void wgradn_bf3_geometry(const rcv_handle* h, const rcv_op* op, int* th, int* tiles_x, int* tiles_y, int* ngroups) {
  const int CA = op->i[RCV_I_CIN], CB = op->i[RCV_I_COUT], s = op->i[RCV_I_STRIDE];
  int TH = (s == 1 && w3_ct(CB) == 16 && w3_ct(CA) == 16 && 2 * w3_buf(16, 16, 1, 16) <= (size_t)h->max_lds) ? 16 : 8;
  if (const char* ev = RCV_ENV("RCV_BF3W_TH")) { const int v = atoi(ev); if (v == 8) TH = 8; }
  *th = TH;
  *tiles_x = ceil_div(op->i[RCV_I_WO], 16); *tiles_y = ceil_div(op->i[RCV_I_HO], TH);
  const long tiles = (long)op->i[RCV_I_N] * *tiles_x * *tiles_y;
  int g = tiles < h->num_cus ? (int)tiles : h->num_cus;
  g = ceil_div((int)tiles, ceil_div((int)tiles, g));                      // equal tile counts per workgroup
  *ngroups = g;
}

template <int CBT, int CAT, int S, int TH, bool GTWO>
static int w3_launch_inst(const WgradArgs& a, int ngroups, hipStream_t s, int dev) {
  auto kern = wgradn_bf3_kernel<CBT, CAT, S, TH, GTWO>;
  const size_t lds = 2 * (size_t)W3Geom<CBT, CAT, S, TH>::BUF;
  static size_t configured[RCV_MAX_DEVICES];
  RCV_ENSURE_LDS(kern, lds, dev, configured);
  hipLaunchKernelGGL(kern, dim3(ngroups), dim3(512), lds, s, a);
  RCV_HIP(hipGetLastError());
  return RCV_OK;
}
template <int CBT, int CAT, int S>
static int w3_launch_th(const WgradArgs& a, int th, bool g_two, int ngroups, hipStream_t s, int dev) {
  if constexpr (S == 1 && CBT == 16 && CAT == 16)      // (the only shape whose 16-row tile fits the LDS twice)
    if (th == 16) return g_two ? w3_launch_inst<CBT, CAT, S, 16, true>(a, ngroups, s, dev) : w3_launch_inst<CBT, CAT, S, 16, false>(a, ngroups, s, dev);
  return g_two ? w3_launch_inst<CBT, CAT, S, 8, true>(a, ngroups, s, dev) : w3_launch_inst<CBT, CAT, S, 8, false>(a, ngroups, s, dev);
}
template <int S>
static int w3_launch_s(const WgradArgs& a, int th, bool g_two, int ngroups, hipStream_t s, int dev) {
  const int cbt = w3_ct(a.CB), cat = w3_ct(a.CA);
  if (cbt == 16 && cat == 16) return w3_launch_th<16, 16, S>(a, th, g_two, ngroups, s, dev);
  if (cbt == 32 && cat == 16) return w3_launch_th<32, 16, S>(a, th, g_two, ngroups, s, dev);
  if (cbt == 16 && cat == 32) return w3_launch_th<16, 32, S>(a, th, g_two, ngroups, s, dev);
  return w3_launch_th<32, 32, S>(a, th, g_two, ngroups, s, dev);
}

int wgradn_bf3_launch(const rcv_handle* h, const WgradArgs& a, int th, int ngroups, hipStream_t s) {
  const bool g_two = a.g_mode == RCV_LOAD_GRAD_ENC || a.g_mode == RCV_LOAD_GRAD_DEC;
  return a.stride == 1 ? w3_launch_s<1>(a, th, g_two, ngroups, s, h->device) : w3_launch_s<2>(a, th, g_two, ngroups, s, h->device);
}

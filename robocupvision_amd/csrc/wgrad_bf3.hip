// 3x3 filter gradient of the wide (64 / 128-channel) stride-1 layers with fp32 products formed on the bf16 matrix pipe.
//
//   dW[cb][ca][tap] = sum_p  P[p][cb] * G[p + tap - 1][ca]          (same operands, load modes and partial-filter layout as wgrad_mfma.hip)
//
// gfx950 issues v_mfma_f32_16x16x4_f32 at 1/16 of the bf16 MFMA rate: the fp32 form of this kernel (wgrad_mfma.hip, 64 x 64 producer /
// consumer tile) is matrix-pipe bound at 93 TFLOP/s.  An fp32 value splits EXACTLY into three bf16 values, x = h + m + l (8 + 8 + 8
// significand bits, round-to-nearest at each step, every remainder exact); a product a*b is then the sum of nine bf16 x bf16 products, each
// exact in the fp32 accumulator.  This kernel issues the six largest (hh, hm, mh, hl, lh, mm): what it leaves out (ml, lm, ll) is below
// 2^-24 |ab|, the size of ONE fp32 rounding of the product.  Measured against fp64 (scripts/micro/split_mfma.hip, K = 1152 and 9216,
// Gaussian and post-ReLU / wide-dynamic-range operands): max and rms error <= those of the fp32 MFMA chain in every case; six bf16 MFMAs
// (v_mfma_f32_16x16x32_bf16) per K = 32 against eight fp32 MFMAs of four times the cycles each -- 2.0-2.5 x the fp32 matrix-pipe rate.
//
// GEMM mapping (per tap): D[cb][ca] += A[cb][k] * B[k][ca], k = pixel, 32 pixels per MFMA.  Lane (i, g) of the bf16 MFMA holds EIGHT
// consecutive k of row / column i -- the contraction index is the pixel, the tensors are NHWC -- so both operands are read from
// [pixel][channel] LDS images with the transposing read ds_read_b64_tr_b16 (4 pixel rows x 16 channels per 16-lane group, each lane gets
// 4 pixels of its channel; two reads = one operand).  k -> pixel: k = 8 g + 4 h + q  <->  tile pixel 32 j + 16 h + 4 g + q (j: k-step,
// h: which of the two reads), so that one read instruction covers 16 CONSECUTIVE tile pixels = runs of 8 inside one image row per
// 32-lane half.
//   G image: [plane h|m|l][IH x IW pixels][64 ch] bf16, pixel pitch 160 B (8 consecutive rows x 32 B fall into 8 distinct slots of the
//            256-byte bank window, wherever the run starts: tap shifts are plain immediates);
//   P image: [plane][64 pixels][64 ch] bf16, pitch 128 B, 32-byte channel blocks XOR-swizzled by (pixel >> 1) & 3 (its reads start at
//            multiples of 8 pixels: the swizzle is a per-lane constant).
// Workgroup = 64 x 64 channel tile, 4 consumer waves (2 x 2, each 32 x 32 channels x 9 taps = 36 accumulator blocks, as in the fp32
// kernel) + 4 producer waves that stage tile i+1 (global -> load transform -> split -> three 8-byte LDS writes per channel quad) while
// tile i is contracted; pointwise tile 64 pixels (8 x 8 or 4 x 16) = 2 k-steps = 432 MFMAs per consumer wave per barrier.
#include "wgrad_common.h"

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((address_space(3))) char lds_char;

constexpr int B3_GPITCH = 160, B3_PPITCH = 128, B3_PPLANE = 64 * B3_PPITCH;
template <int TW>
struct B3Geom {
  static constexpr int TH = 64 / TW, IW = TW + 2, IH = TH + 2, GPIX = IH * IW;
  static constexpr int GPLANE = GPIX * B3_GPITCH;
  static constexpr int BUF = 3 * GPLANE + 3 * B3_PPLANE;
};

// x = h + m + l for two values at once; each output word holds the two bf16 of one plane (element 0 in the low half)
struct B3Tri { uint32_t h, m, l; };
__device__ __forceinline__ uint32_t b3_pack(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};            // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ B3Tri b3_split2(float x0, float x1) {
  B3Tri t;
  t.h = b3_pack(x0, x1);
  const float r0 = x0 - __uint_as_float(t.h << 16), r1 = x1 - __uint_as_float(t.h & 0xffff0000u);       // exact
  t.m = b3_pack(r0, r1);
  const float s0 = r0 - __uint_as_float(t.m << 16), s1 = r1 - __uint_as_float(t.m & 0xffff0000u);       // exact, <= 8 significant bits
  t.l = b3_pack(s0, s1);
  return t;
}

// One operand tile: global -> load transform -> three bf16 planes in LDS, in two steps so that the loads of BOTH operand tiles of a pixel
// tile are in flight before the first is consumed (staged one batch of four after the other, a tile cost three serialized HBM/L2 round
// trips: 26 us of staging per launch beside 40 us of contraction, and the two did not overlap).  16 threads share a pixel (one 16-byte
// channel quad each); a pixel outside the plane is stored as zero.
template <int NPIX, int TWP, bool TWO>
struct B3Regs {
  static constexpr int NU = (NPIX + 15) / 16;        // 256 staging threads: 16 pixels per pass
  float4 x[NU], ax[TWO ? NU : 1];
  bool ok[NU];
};
template <int NPIX, int TWP, bool TWO>
__device__ __forceinline__ void b3_load(B3Regs<NPIX, TWP, TWO>& r, const float* __restrict__ src, const float* __restrict__ aux, bool two, int tid, int ch0, int C,
                                        int row0, int oy, int ox, int PH, int PW) {
  const int q = tid & 15, lp = tid >> 4;
  const int ch = ch0 + 4 * q;
#pragma unroll
  for (int u = 0; u < B3Regs<NPIX, TWP, TWO>::NU; ++u) {
    const int pix = u * 16 + lp;
    const int iy = pix / TWP, ix = pix - iy * TWP;                  // (compile-time divisor)
    r.ok[u] = pix < NPIX && (unsigned)(oy + iy) < (unsigned)PH && (unsigned)(ox + ix) < (unsigned)PW;
    const uint32_t o = r.ok[u] ? (uint32_t)(((row0 + oy + iy) * PW + ox + ix) * C + ch) : 0u;
    r.x[u] = wld4(src + o);
    if (TWO) { if (two) r.ax[u] = wld4(aux + o); }
  }
}
// SWZ: the P image's swizzle.
template <int MODE, int NPIX, int TWP, bool TWO, bool SUM, bool SWZ>
__device__ __forceinline__ void b3_store(const B3Regs<NPIX, TWP, TWO>& r, const float* __restrict__ consts, char* img, int plane_bytes, int pitch, int tid,
                                         int ch0, int C, float4& sum) {
  const int q = tid & 15, lp = tid >> 4;
  const int ch = ch0 + 4 * q;
  float4 k[5];
  if (MODE != RCV_LOAD_PLAIN) {
#pragma unroll
    for (int j = 0; j < 5; ++j) k[j] = wld4(consts + (size_t)j * C + ch);
  }
#pragma unroll
  for (int u = 0; u < B3Regs<NPIX, TWP, TWO>::NU; ++u) {
    const int pix = u * 16 + lp;
    float4 v = wxform4<MODE>(r.x[u], r.ax[TWO ? u : 0], k);
    if (!r.ok[u]) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (SUM) { sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w; }
    if (pix < NPIX) {
      const B3Tri a = b3_split2(v.x, v.y), b = b3_split2(v.z, v.w);
      int off = pix * pitch + 8 * q;
      if (SWZ) off = pix * pitch + ((((q >> 2) ^ ((pix >> 1) & 3))) << 5) + 8 * (q & 3);
      *reinterpret_cast<uint2*>(img + off) = make_uint2(a.h, b.h);
      *reinterpret_cast<uint2*>(img + plane_bytes + off) = make_uint2(a.m, b.m);
      *reinterpret_cast<uint2*>(img + 2 * plane_bytes + off) = make_uint2(a.l, b.l);
    }
  }
}

__device__ __forceinline__ bf16x8 b3_read(const lds_char* p, int off0, int off1) {
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + off0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + off1));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// GTWO: the gathered operand may be a two-tensor gradient load (then the pointwise one is not); otherwise the pointwise one may be.
template <int TW, bool GTWO>
__global__ __launch_bounds__(512) void wgrad_bf3_kernel(const WgradArgs a) {
  using G = B3Geom<TW>;
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  const bool producer = threadIdx.x >= 256;
  const int tid = producer ? (int)threadIdx.x - 256 : (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_n = wave & 1, wave_m = wave >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;

  const int n_ca_tiles = a.CAP / 64;
  const int bl = xcd_remap(blockIdx.x, gridDim.x);
  const int ctile = bl % a.nctiles, split = bl / a.nctiles;
  const int cb0 = (ctile / n_ca_tiles) * 64, ca0 = (ctile % n_ca_tiles) * 64;

  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  auto stage = [&](int tile, char* buf) {
    int t = tile;
    const int tx_i = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty_i = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int y0 = ty_i * G::TH, x0 = tx_i * TW;
    float4 nosum = make_float4(0.f, 0.f, 0.f, 0.f);
    // every load of the tile first (GTWO: the gathered operand is the two-tensor one, otherwise the pointwise one may be)
    B3Regs<G::GPIX, G::IW, GTWO> rg;
    B3Regs<64, TW, !GTWO> rp;
    const bool p_two = !GTWO && (a.p_mode == RCV_LOAD_GRAD_ENC || a.p_mode == RCV_LOAD_GRAD_DEC);
    b3_load(rg, a.g, a.g_aux, true, tid, ca0, a.CA, n * a.H, y0 - 1, x0 - 1, a.H, a.W);
    b3_load(rp, a.p, a.p_aux, p_two, tid, cb0, a.CB, n * a.Hp, y0, x0, a.Hp, a.Wp);
    char* gi = buf;
    char* pi = buf + 3 * G::GPLANE;
    if (GTWO) {
      if (a.g_mode == RCV_LOAD_GRAD_ENC) b3_store<RCV_LOAD_GRAD_ENC, G::GPIX, G::IW, GTWO, false, false>(rg, a.g_c, gi, G::GPLANE, B3_GPITCH, tid, ca0, a.CA, nosum);
      else b3_store<RCV_LOAD_GRAD_DEC, G::GPIX, G::IW, GTWO, false, false>(rg, a.g_c, gi, G::GPLANE, B3_GPITCH, tid, ca0, a.CA, nosum);
    } else {
      switch (a.g_mode) {
        case RCV_LOAD_PLAIN: b3_store<RCV_LOAD_PLAIN, G::GPIX, G::IW, GTWO, false, false>(rg, a.g_c, gi, G::GPLANE, B3_GPITCH, tid, ca0, a.CA, nosum); break;
        case RCV_LOAD_AFFINE: b3_store<RCV_LOAD_AFFINE, G::GPIX, G::IW, GTWO, false, false>(rg, a.g_c, gi, G::GPLANE, B3_GPITCH, tid, ca0, a.CA, nosum); break;
        default: b3_store<RCV_LOAD_AFFINE_RELU, G::GPIX, G::IW, GTWO, false, false>(rg, a.g_c, gi, G::GPLANE, B3_GPITCH, tid, ca0, a.CA, nosum); break;
      }
    }
    switch (a.p_mode) {
      case RCV_LOAD_PLAIN: b3_store<RCV_LOAD_PLAIN, 64, TW, !GTWO, true, true>(rp, a.p_c, pi, B3_PPLANE, B3_PPITCH, tid, cb0, a.CB, bsum); break;
      case RCV_LOAD_AFFINE: b3_store<RCV_LOAD_AFFINE, 64, TW, !GTWO, true, true>(rp, a.p_c, pi, B3_PPLANE, B3_PPITCH, tid, cb0, a.CB, bsum); break;
      case RCV_LOAD_AFFINE_RELU: b3_store<RCV_LOAD_AFFINE_RELU, 64, TW, !GTWO, true, true>(rp, a.p_c, pi, B3_PPLANE, B3_PPITCH, tid, cb0, a.CB, bsum); break;
      case RCV_LOAD_GRAD_ENC: if constexpr (!GTWO) b3_store<RCV_LOAD_GRAD_ENC, 64, TW, !GTWO, true, true>(rp, a.p_c, pi, B3_PPLANE, B3_PPITCH, tid, cb0, a.CB, bsum); break;
      default: if constexpr (!GTWO) b3_store<RCV_LOAD_GRAD_DEC, 64, TW, !GTWO, true, true>(rp, a.p_c, pi, B3_PPLANE, B3_PPITCH, tid, cb0, a.CB, bsum); break;
    }
  };
  // bias partial: sum over the staging threads that hold the same channel quad (fixed order)
  auto bias_partial = [&]() {
    if (a.part_bias && ca0 == 0) {
      __syncthreads();
      float4* sb = reinterpret_cast<float4*>(smem3);
      if (producer) sb[tid] = bsum;
      __syncthreads();
      if (producer && tid < 16) {
        float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = tid; e < 256; e += 16) { const float4 v = sb[e]; u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w; }
        *reinterpret_cast<float4*>(a.part_bias + (size_t)split * a.CBP + cb0 + 4 * tid) = u;
      }
    }
  };
  const bool do_stage = !(a.dbg & RCV_F_DBG_NOSTAGE), do_mfma = !(a.dbg & RCV_F_DBG_NOMFMA);      // (ablation timings: scripts/bench_op.py --flags)
  if (producer) {
    // barrier for barrier the consumer path below: 1 + one per tile (+ the bias partial's)
    if (split < a.ntiles && do_stage) stage(split, smem3);
    __syncthreads();
    int it = 0;
    for (int tile = split; tile < a.ntiles; tile += a.nsplit, ++it) {
      const int next = tile + a.nsplit;
      if (next < a.ntiles && do_stage) stage(next, smem3 + ((it + 1) & 1) * G::BUF);
      __syncthreads();
    }
    bias_partial();
    return;
  }

  // ---------------- consumer waves ----------------
  f32x4 acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) acc[t][m][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // lane parts of the read addresses.  Lane 4 q + p of a 16-lane group supplies the address of block row q, channels 4 p .. 4 p + 3.
  const int q = l15 >> 2, p = l15 & 3;
  const int kp = 4 * l4 + q;                                   // tile pixel (mod 16) of this lane's row
  int a_off[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) a_off[m] = 3 * G::GPLANE + kp * B3_PPITCH + (((wave_m * 2 + m) ^ ((kp >> 1) & 3)) << 5) + 8 * p;
  const int g_row = TW == 8 ? (l4 >> 1) * G::IW + 4 * (l4 & 1) + q : kp;        // G image row of the lane's pixel relative to the read's first row
  const int g_off = g_row * B3_GPITCH + (wave_n * 2) * 32 + 8 * p;
  const lds_char* lds0 = (const lds_char*)smem3;

  auto contract = [&](const lds_char* buf) {
    const lds_char* pa0 = buf + a_off[0];
    const lds_char* pa1 = buf + a_off[1];
    const lds_char* pg = buf + g_off;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16x8 A[2][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        A[0][pl] = b3_read(pa0, pl * B3_PPLANE + (32 * j) * B3_PPITCH, pl * B3_PPLANE + (32 * j + 16) * B3_PPITCH);
        A[1][pl] = b3_read(pa1, pl * B3_PPLANE + (32 * j) * B3_PPITCH, pl * B3_PPLANE + (32 * j + 16) * B3_PPITCH);
      }
      // operands of tap t + 1 are requested before the MFMAs of tap t (two register sets; the scheduling barrier keeps the compiler from
      // hoisting more reads than that: it spilled accumulators for them)
      auto load_b = [&](int t, bf16x8 (&B)[2][3]) {
        const int ky = t / 3, kx = t % 3;
        // first G image row of the two reads of this k-step (tile rows 4 j + 2 h for the 8-wide tile, 2 j + h for the 16-wide one)
        const int r0 = ((TW == 8 ? 4 * j : 2 * j) + ky) * G::IW + kx;
        const int r1 = ((TW == 8 ? 4 * j + 2 : 2 * j + 1) + ky) * G::IW + kx;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int nn = 0; nn < 2; ++nn) B[nn][pl] = b3_read(pg, pl * G::GPLANE + r0 * B3_GPITCH + nn * 32, pl * G::GPLANE + r1 * B3_GPITCH + nn * 32);
      };
      bf16x8 Bb[2][2][3];
      load_b(0, Bb[0]);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t < 8) load_b(t + 1, Bb[(t + 1) & 1]);
        // six products per block, smallest first; the four blocks of a term back to back (independent accumulators)
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int e = 0; e < 6; ++e)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) acc[t][m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[m][TA[e]], Bb[t & 1][nn][TB[e]], acc[t][m][nn], 0, 0, 0);
        // one LDS read behind each MFMA (a burst of 24 reads in front of them leaves the matrix pipe idle while they issue; left alone
        // the compiler sinks the reads to the END of the tap and the next tap starts with their latency)
        if (t < 8) {
#pragma unroll
          for (int e = 0; e < 24; ++e) {
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
    for (int tile = split; tile < a.ntiles; tile += a.nsplit, ++it) {
      if (do_mfma) contract(lds0 + (it & 1) * G::BUF);
      __syncthreads();
    }
  }
  if (!(a.dbg & RCV_F_DBG_NOEPI)) {
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int nn = 0; nn < 2; ++nn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cb = cb0 + (wave_m * 2 + m) * 16 + 4 * l4 + r;
          const int ca = ca0 + (wave_n * 2 + nn) * 16 + l15;
          a.part[(((size_t)split * 9 + t) * a.CBP + cb) * a.CAP + ca] = acc[t][m][nn][r];
        }
  }
  bias_partial();
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
bool wgrad_bf3_supported(const rcv_handle* h, const rcv_op* op) {
  if (RCV_ENV("RCV_NO_BF3") || (op->flags & RCV_F_MFMA_FP32)) return false;
  const int CA = op->i[RCV_I_CIN], CB = op->i[RCV_I_COUT];
  return op->i[RCV_I_STRIDE] == 1 && op->i[RCV_I_DIL] == 1 && op->i[RCV_I_INMODE] != RCV_LOAD_NCHW && CA % 64 == 0 && CB % 64 == 0 &&
         B3Geom<16>::BUF * 2 <= h->max_lds;
}

// pixel tile (8 x 8 or 4 x 16: the one that wastes fewer pixel slots on this plane) and the pixel split
void wgrad_bf3_geometry(const rcv_handle* h, const rcv_op* op, int* tw, int* tiles_x, int* tiles_y, int* nsplit, int* nctiles) {
  const int N = op->i[RCV_I_N], Hp = op->i[RCV_I_HO], Wp = op->i[RCV_I_WO];
  const long s8 = (long)ceil_div(Hp, 8) * ceil_div(Wp, 8), s16 = (long)ceil_div(Hp, 4) * ceil_div(Wp, 16);
  int TW = s16 < s8 ? 16 : 8;
  if (const char* ev = RCV_ENV("RCV_BF3_TW")) { const int v = atoi(ev); if (v == 8 || v == 16) TW = v; }
  *tw = TW;
  *tiles_x = ceil_div(Wp, TW); *tiles_y = ceil_div(Hp, 64 / TW);
  const int ctiles = (op->i[RCV_I_COUT] / 64) * (op->i[RCV_I_CIN] / 64);
  const int ntiles = N * *tiles_x * *tiles_y;
  int ns = h->num_cus / ctiles;                      // two 70+ KB buffers: one workgroup per CU
  if (ns < 1) ns = 1;
  if (ns > ntiles) ns = ntiles;
  *nsplit = ceil_div(ntiles, ceil_div(ntiles, ns));  // equal tile counts per workgroup
  *nctiles = ctiles;
}

template <int TW, bool GTWO>
static int b3_launch_inst(const WgradArgs& a, dim3 grid, hipStream_t s, int dev) {
  auto kern = wgrad_bf3_kernel<TW, GTWO>;
  const size_t lds = 2 * (size_t)B3Geom<TW>::BUF;
  static size_t configured[RCV_MAX_DEVICES];
  RCV_ENSURE_LDS(kern, lds, dev, configured);
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, a);
  RCV_HIP(hipGetLastError());
  return RCV_OK;
}

int wgrad_bf3_launch(const rcv_handle* h, const WgradArgs& a, int tw, hipStream_t s) {
  const bool g_two = a.g_mode == RCV_LOAD_GRAD_ENC || a.g_mode == RCV_LOAD_GRAD_DEC;
  const dim3 grid(a.nsplit * a.nctiles);
  if (tw == 8) return g_two ? b3_launch_inst<8, true>(a, grid, s, h->device) : b3_launch_inst<8, false>(a, grid, s, h->device);
  return g_two ? b3_launch_inst<16, true>(a, grid, s, h->device) : b3_launch_inst<16, false>(a, grid, s, h->device);
}

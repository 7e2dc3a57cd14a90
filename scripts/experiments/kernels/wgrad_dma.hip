// 3x3 filter gradient of the NARROW layers (<= 32 x 32 channel tiles, and the <= 8-channel layers with their nine taps folded into
// the MFMA column dimension): the same contraction as wgrad_mfma.hip,
//
//   dW[cb][ca][tap] = sum_p  P[p][cb] * G[s*p + tap*d - d][ca]          (conv: model.py:112, convT: model.py:186)
//
// but staged for layers whose operands are read once from HBM and barely reused (16 -> 16 at 240 x 320: 0.47 GB for 11 GFLOP).
// With register staging (wgrad_mfma.hip) the bytes in flight per CU are bounded by the registers of the staging waves (32 KB/CU =>
// 3-4 TB/s) and staging and matrix time simply ADD, whichever way the waves are split between the two roles (measured).
// Here every one of the 8 waves is an MFMA wave and the operands travel by LDS-DMA (global_load_lds_dwordx4):
//
//   raw tile i+1:  global --DMA--> LDS (lane-linear, no registers, no vector ALU, up to 64 KB per wave in flight), issued before
//                  the MFMA phase of tile i and complete when it ends;
//   transform:     raw -> operand layout in LDS, all 512 threads: the load transform of the producer (BatchNorm apply / ReLU /
//                  BatchNorm+ReLU backward, wgrad_common.h), zeros outside the plane, the bias partial sums;
//   contract:      the 8 waves split the 4-pixel k-steps of the tile, operands prefetched one k-step ahead.
//
// LDS: tp[R*Wt4][SP] tg[IH*IW][SG] (operands, padded strides) | raw P (+ its second tensor) | raw G (+ its second tensor).
// Partial filters [split][tap][cb][ca] and the fixed-order reduction (RCV_OP_WGRAD_REDUCE) are those of wgrad_mfma.hip.
#include "wgrad_common.h"

typedef __attribute__((address_space(3))) void* wg_lds_ptr;
typedef const __attribute__((address_space(1))) void* wg_glb_ptr;

// One LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to LDS bytes [lds, lds + 1024) (wave-uniform base in M0).
// Issued through inline assembly on purpose: for the builtin the compiler's wait-count pass puts `s_waitcnt vmcnt(0)` in front of the
// first LDS read that follows (it cannot prove that the read does not alias the DMA's destination), i.e. in front of the MFMA phase
// the transfer is supposed to run behind.  Completion is awaited explicitly (vmcnt(0) + barrier at the top of the tile loop); a
// VMEM operation the compiler does not know about can only make its own vmcnt waits longer, never shorter (the counter retires in order).
__device__ __forceinline__ void wg_dma16(const float* gsrc, uint32_t lds_byte_address) {
  const uint32_t m = __builtin_amdgcn_readfirstlane(lds_byte_address);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(m) : "memory", "m0");
}

// raw (lane-linear) tile -> operand tile.  Slot e = pix * Q + q (one 16-byte channel quad of one pixel); a thread keeps its quad.
template <int MODE, bool SUM, bool SCALAR_STORE>
__device__ __forceinline__ void wdma_transform(const float* __restrict__ raw, const float* __restrict__ raw_aux, const float* __restrict__ consts,
                                               float* __restrict__ dst, int tid, int nslots, int qshift, int ch0, int C, FastDiv fdTW, int TW,
                                               int TWV, int oy, int ox, int PH, int PW, int S, float4& sum) {
  constexpr bool TWO = MODE == RCV_LOAD_GRAD_ENC || MODE == RCV_LOAD_GRAD_DEC;
  constexpr int NT = 512, UNR = 4;
  const int q = tid & ((1 << qshift) - 1);
  const int ch = ch0 + 4 * q;
  const bool ch_ok = ch < C;
  float4 k[5];
  if (MODE != RCV_LOAD_PLAIN) {
#pragma unroll
    for (int j = 0; j < 5; ++j) k[j] = ch_ok ? wld4(consts + (size_t)j * C + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float4* r4 = reinterpret_cast<const float4*>(raw);
  const float4* a4 = reinterpret_cast<const float4*>(raw_aux);
  for (int e0 = tid; e0 < nslots; e0 += UNR * NT) {
    float4 x[UNR], ax[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int e = e0 + u * NT < nslots ? e0 + u * NT : e0;      // past the end: a repeated (discarded) read
      x[u] = r4[e];
      ax[u] = TWO ? a4[e] : x[u];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int e = e0 + u * NT;
      const int pix = e >> qshift;
      const int iy = fd_div(pix, fdTW), ix = pix - iy * TW;
      const int gy = oy + iy, gx = ox + ix;
      const bool ok = ch_ok && ix < TWV && (unsigned)gy < (unsigned)PH && (unsigned)gx < (unsigned)PW;
      float4 v = wxform4<MODE>(x[u], ax[u], k);
      if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < nslots) {
        if (SUM) { sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w; }
        float* o = dst + pix * S + 4 * q;
        if (SCALAR_STORE) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
        else *reinterpret_cast<float4*>(o) = v;
      }
    }
  }
}

// NBF == 0: WN column blocks of 16 gathered channels, nine taps each;  NBF > 0 (WM == WN == 1): NBF column blocks over n = tap*CA + ca.
// GTWO: the gathered operand is the two-tensor gradient load (convT layer); otherwise the pointwise one may be.
template <int WM, int WN, int NBF, bool GTWO>
__global__ __launch_bounds__(512) void wgrad_dma_kernel(const WgradArgs a) {
  constexpr int NT = 512, NW = 8;
  constexpr int CBT = WM * 16, CAT = WN * 16;
  constexpr bool FOLD = NBF > 0;
  constexpr int NACC = FOLD ? NBF : 9 * WN;
  constexpr int NUNIT = NACC * WM;                  // accumulator tiles (16 x 16) of a wave
  constexpr int MAXD = 6;                           // DMA instructions per wave, operand tensor and tile (host: <= 6 * 512 slots)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const bool p_two = !GTWO && (a.p_mode == RCV_LOAD_GRAD_ENC || a.p_mode == RCV_LOAD_GRAD_DEC);
  float* const tp = smem;
  float* const tg = tp + a.pl_floats;
  float* const rp = tg + a.gl_floats;
  float* const rg = rp + a.rp_floats * (p_two ? 2 : 1);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int n_ca_tiles = FOLD ? 1 : (a.CAP + CAT - 1) / CAT;
  const int bl = xcd_remap(blockIdx.x, gridDim.x);
  const int ctile = bl % a.nctiles, split = bl / a.nctiles;
  const int cb_tile = ctile / n_ca_tiles, ca_tile = ctile % n_ca_tiles;
  const int cb0 = cb_tile * CBT, ca0 = ca_tile * CAT;
  int qp_shift = CBT == 32 ? 3 : 2;                 // 16-byte quads per pointwise pixel that hold real channels (a power of two)
  while (qp_shift > 0 && (2 << qp_shift) >= a.CB - cb0) --qp_shift;
  const int qg_shift = FOLD ? (a.CA > 4 ? 1 : 0) : (CAT == 32 ? 3 : 2);
  const int s = a.stride, d = a.dil;
  const int np_pix = a.R * a.Wt4, ng_pix = a.IH * a.IW;
  const int np_slots = np_pix << qp_shift, ng_slots = ng_pix << qg_shift;

  // ---------------- raw tile: global -> LDS by DMA.  Lane l of wave-instruction e0/64 fetches slot e0 + l; slots without a real pixel
  // fetch element 0 (the transform pass zeroes them) ----------------
  const uint32_t smem_base = (uint32_t)(uintptr_t)(wg_lds_ptr)smem;
  auto dma_operand = [&](const float* __restrict__ src, const float* rdst, int nslots, int qshift, int ch0, int C, FastDiv fdTW, int TW, int TWV,
                         int row0, int oy, int ox, int PH, int PW) {
#pragma unroll
    for (int it = 0; it < MAXD; ++it) {
      const int e0 = (it * NW + wave) * 64;
      if (e0 >= nslots) break;
      const int e = e0 + lane;
      const int pix = e >> qshift, q = e & ((1 << qshift) - 1);
      const int iy = fd_div(pix, fdTW), ix = pix - iy * TW;
      const int gy = oy + iy, gx = ox + ix, ch = ch0 + 4 * q;
      const bool ok = e < nslots && ch < C && ix < TWV && (unsigned)gy < (unsigned)PH && (unsigned)gx < (unsigned)PW;
      const int off = ok ? ((row0 + gy) * PW + gx) * C + ch : 0;
      wg_dma16(src + off, smem_base + (uint32_t)((rdst - smem) + e0 * 4) * 4u);
    }
  };
  auto tile_origin = [&](int tile, int& n, int& y0, int& x0) {
    int t = tile;
    const int tx_i = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty_i = t % a.tiles_y;
    n = t / a.tiles_y;
    y0 = ty_i * a.R; x0 = tx_i * a.Wt;
  };
  auto issue_dma = [&](int tile) {
    int n, y0, x0;
    tile_origin(tile, n, y0, x0);
    dma_operand(a.p, rp, np_slots, qp_shift, cb0, a.CB, a.fdWt4, a.Wt4, a.Wt, n * a.Hp, y0, x0, a.Hp, a.Wp);
    if (p_two) dma_operand(a.p_aux, rp + a.rp_floats, np_slots, qp_shift, cb0, a.CB, a.fdWt4, a.Wt4, a.Wt, n * a.Hp, y0, x0, a.Hp, a.Wp);
    const int oy = y0 * s - d, ox = x0 * s - d;
    dma_operand(a.g, rg, ng_slots, qg_shift, ca0, a.CA, a.fdIW, a.IW, a.IW, n * a.H, oy, ox, a.H, a.W);
    if (GTWO) dma_operand(a.g_aux, rg + a.rg_floats, ng_slots, qg_shift, ca0, a.CA, a.fdIW, a.IW, a.IW, n * a.H, oy, ox, a.H, a.W);
  };

  // ---------------- transform pass ----------------
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  auto transform = [&](int tile) {
    int n, y0, x0;
    tile_origin(tile, n, y0, x0);
#define RCV_WDMA_P(MODE) wdma_transform<MODE, true, false>(rp, rp + a.rp_floats, a.p_c, tp, tid, np_slots, qp_shift, cb0, a.CB, a.fdWt4, a.Wt4, a.Wt, y0, x0, a.Hp, a.Wp, a.SP, bsum)
    switch (GTWO && (a.p_mode == RCV_LOAD_GRAD_ENC || a.p_mode == RCV_LOAD_GRAD_DEC) ? RCV_LOAD_PLAIN : a.p_mode) {
      case RCV_LOAD_PLAIN: RCV_WDMA_P(RCV_LOAD_PLAIN); break;
      case RCV_LOAD_AFFINE: RCV_WDMA_P(RCV_LOAD_AFFINE); break;
      case RCV_LOAD_AFFINE_RELU: RCV_WDMA_P(RCV_LOAD_AFFINE_RELU); break;
      case RCV_LOAD_GRAD_ENC: if (!GTWO) RCV_WDMA_P(RCV_LOAD_GRAD_ENC); break;
      default: if (!GTWO) RCV_WDMA_P(RCV_LOAD_GRAD_DEC); break;
    }
#undef RCV_WDMA_P
    float4 nosum = make_float4(0.f, 0.f, 0.f, 0.f);
    const int oy = y0 * s - d, ox = x0 * s - d;
#define RCV_WDMA_G(MODE) wdma_transform<MODE, false, FOLD>(rg, rg + a.rg_floats, a.g_c, tg, tid, ng_slots, qg_shift, ca0, a.CA, a.fdIW, a.IW, a.IW, oy, ox, a.H, a.W, a.SG, nosum)
    if (GTWO) {
      if (a.g_mode == RCV_LOAD_GRAD_ENC) RCV_WDMA_G(RCV_LOAD_GRAD_ENC);
      else RCV_WDMA_G(RCV_LOAD_GRAD_DEC);
    } else {
      switch (a.g_mode) {
        case RCV_LOAD_PLAIN: RCV_WDMA_G(RCV_LOAD_PLAIN); break;
        case RCV_LOAD_AFFINE: RCV_WDMA_G(RCV_LOAD_AFFINE); break;
        default: RCV_WDMA_G(RCV_LOAD_AFFINE_RELU); break;
      }
    }
#undef RCV_WDMA_G
  };

  // ---------------- accumulators, lane offsets, the k-step ----------------
  f32x4 acc[NACC][WM];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int m = 0; m < WM; ++m) acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int loff[NACC];
  if (FOLD) {
#pragma unroll
    for (int nb = 0; nb < NACC; ++nb) {
      int nn = nb * 16 + l15;
      if (nn >= 9 * a.CA) nn = 0;                       // columns beyond 9*CA are discarded at the end
      const int tap = nn / a.CA, ca = nn - tap * a.CA;
      const int ky = tap / 3, kx = tap - ky * 3;
      loff[nb] = ((ky * d) * a.IW + kx * d) * a.SG + ca + (l4 * s) * a.SG;
    }
  } else {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int nn = 0; nn < WN; ++nn) {
        const int ky = t / 3, kx = t - ky * 3;
        loff[t * WN + nn] = ((ky * d) * a.IW + kx * d) * a.SG + (l4 * s) * a.SG + nn * 16 + l15;
      }
  }
  const int ksteps = np_pix / 4;
  const int a_lane = l4 * a.SP + l15;
  auto load_ops = [&](int j, float (&av)[WM], float (&bv)[NACC]) {
    const int p0 = 4 * j;
    const int ty = fd_div(p0, a.fdWt4), tx = p0 - ty * a.Wt4;
#pragma unroll
    for (int m = 0; m < WM; ++m) av[m] = tp[p0 * a.SP + a_lane + m * 16];
    const float* gj = tg + ((ty * s) * a.IW + tx * s) * a.SG;
#pragma unroll
    for (int t = 0; t < NACC; ++t) bv[t] = gj[loff[t]];
  };
  auto mfma_ops = [&](const float (&av)[WM], const float (&bv)[NACC]) {
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
      for (int m = 0; m < WM; ++m) acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[t], acc[t][m], 0, 0, 0);
  };
  // (the reads are unconditional: past the end they repeat the last k-step and are discarded -- behind a branch the compiler waits for
  // ALL outstanding LDS reads where the paths meet, the prefetched set included)
  auto contract = [&]() {
    float av0[WM], bv0[NACC], av1[WM], bv1[NACC];
    const int jlast = ksteps - 1;
    int j = wave;
    load_ops(j < jlast ? j : jlast, av0, bv0);
    for (; j < ksteps; j += 2 * NW) {
      const int j1 = j + NW, j2 = j + 2 * NW;
      load_ops(j1 < jlast ? j1 : jlast, av1, bv1);
      mfma_ops(av0, bv0);
      load_ops(j2 < jlast ? j2 : jlast, av0, bv0);
      if (j1 < ksteps) mfma_ops(av1, bv1);
    }
  };

  // ---------------- main loop ----------------
  const bool do_stage = !(a.dbg & RCV_F_DBG_NOSTAGE), do_mfma = !(a.dbg & RCV_F_DBG_NOMFMA);
  {   // operand columns no transform thread writes (channel quads beyond CB) must read as zero
    const int nfl = a.pl_floats + a.gl_floats;
    for (int e = tid; e < nfl; e += NT) smem[e] = 0.f;
  }
  if (split < a.ntiles && do_stage) issue_dma(split);
  for (int tile = split; tile < a.ntiles; tile += a.nsplit) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA of `tile` has landed (assembly: the compiler must not drop a wait it sees no reason for) ...
    __syncthreads();                                 // ... and everybody's; all waves are done reading the previous operand tile
    if (do_stage) transform(tile);
    __syncthreads();                                 // operands complete; the raw buffers are free again
    const int next = tile + a.nsplit;
    __builtin_amdgcn_s_waitcnt(0x0F70);              // (nothing of the transform pass is left in flight: keeps the compiler from flushing vmcnt ahead of the k-loop)
    if (next < a.ntiles && do_stage) issue_dma(next);
    if (do_mfma) contract();
  }

  // ---------------- reduce the 8 pixel slices (fixed order) and write the partial filter: accumulator tile u of every wave goes to
  // LDS, wave (u mod 8) sums the eight copies in wave order and stores ----------------
  __syncthreads();
  {
    float* sc = smem + wave * (NUNIT * 256);
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
      for (int m = 0; m < WM; ++m) *reinterpret_cast<f32x4*>(sc + ((t * WM + m) * 64 + lane) * 4) = acc[t][m];
  }
  __syncthreads();
  if (!(a.dbg & (1u << 23))) {
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
      for (int m = 0; m < WM; ++m) {
        const int u = t * WM + m;
        if ((u & (NW - 1)) != wave) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(smem + (u * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < NW; ++w) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(smem + w * (NUNIT * 256) + (u * 64 + lane) * 4);
          v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
        }
        int tap, ca;
        if (FOLD) {
          const int nn = t * 16 + l15;
          tap = nn / a.CA; ca = nn - tap * a.CA;
          if (nn >= 9 * a.CA) continue;
        } else {
          tap = t / WN;
          ca = ca0 + (t % WN) * 16 + l15;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cb = cb0 + m * 16 + 4 * l4 + r;
          if (cb < a.CBP && ca < a.CAP) a.part[(((size_t)split * 9 + tap) * a.CBP + cb) * a.CAP + ca] = v[r];
        }
      }
  }
  // bias partial: sum over the threads that transformed the same channel quad (fixed order)
  if (a.part_bias && ca_tile == 0) {
    __syncthreads();
    float4* sb = reinterpret_cast<float4*>(smem);
    sb[tid] = bsum;
    __syncthreads();
    const int QP = 1 << qp_shift;
    if (tid < QP) {
      float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int e = tid; e < NT; e += QP) { const float4 v = sb[e]; u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w; }
      const int cb = cb0 + 4 * tid;
      if (cb < a.CBP) *reinterpret_cast<float4*>(a.part_bias + (size_t)split * a.CBP + cb) = u;
    }
  }
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
template <int WM, int WN, int NBF>
static int wdma_inst(const WgradArgs& a, bool gtwo, dim3 grid, size_t lds, hipStream_t s, int dev) {
  if (gtwo) {
    auto kern = wgrad_dma_kernel<WM, WN, NBF, true>;
    static size_t configured[RCV_MAX_DEVICES];
    RCV_ENSURE_LDS(kern, lds, dev, configured);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, a);
  } else {
    auto kern = wgrad_dma_kernel<WM, WN, NBF, false>;
    static size_t configured[RCV_MAX_DEVICES];
    RCV_ENSURE_LDS(kern, lds, dev, configured);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, a);
  }
  RCV_HIP(hipGetLastError());
  return RCV_OK;
}

// accumulator tiles per wave of channel tile `shape` (WGRAD_DMA_*): the reduction scratch is 8 waves x units x 1 KB
int wgrad_dma_units(int shape) {
  switch (shape) {
    case WGRAD_DMA_32x16: case WGRAD_DMA_16x32: return 18;
    case WGRAD_DMA_16x16: return 9;
    case WGRAD_DMA_FOLD2: return 2;
    default: return 5;
  }
}

int wgrad_dma_launch(const rcv_handle* h, const WgradArgs& a, int shape, bool gtwo, dim3 grid, size_t lds, hipStream_t s) {
  switch (shape) {
    case WGRAD_DMA_32x16: return wdma_inst<2, 1, 0>(a, gtwo, grid, lds, s, h->device);
    case WGRAD_DMA_16x32: return wdma_inst<1, 2, 0>(a, gtwo, grid, lds, s, h->device);
    case WGRAD_DMA_16x16: return wdma_inst<1, 1, 0>(a, gtwo, grid, lds, s, h->device);
    case WGRAD_DMA_FOLD2: return wdma_inst<1, 1, 2>(a, gtwo, grid, lds, s, h->device);
    default: return wdma_inst<1, 1, 5>(a, gtwo, grid, lds, s, h->device);
  }
}

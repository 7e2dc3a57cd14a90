// 3x3 filter gradient on the gfx950 fp32 matrix cores (v_mfma_f32_16x16x4_f32), split over pixels.
//
//   dW[cb][ca][tap] = sum_p  P[p][cb] * G[s*p + tap*d - d][ca]
//
//   conv layer  (model.py:112):  G = layer input  (BN of the producer applied on load, zero padded),
//                                P = dz, the gradient at the conv output (BN/ReLU backward on load);
//                                dW is [Cout][Cin][3][3], db[cb] = sum_p P[p][cb].
//   convT layer (model.py:186):  G = dt, the gradient at the 2x output (zero padded), stride 2,
//                                P = layer input; dW is [Cin][Cout][3][3] -- the same index formula.
//
// GEMM mapping:  D[cb][n] += A[cb][k] * B[k][n],  k = pixel (4 consecutive pixels of a tile row per MFMA).
//   regular mode (>= 16 gathered channels): n = ca, one accumulator set per tap; a wave keeps all nine taps of
//     its (WM x WN) x 16x16 channel tile in registers (9*WM*WN*4 VGPRs);
//   folded mode (<= 8 gathered channels: the image, the 8-channel layers): n = (tap, ca) flattened, so the nine
//     taps of a 3- or 8-channel operand fill 2 or 5 MFMA column blocks instead of 9 mostly empty ones.
// Both tiles sit in LDS ([pixel][C + pad]); B is read through a per-lane offset table.
//
// Schedule: workgroups are persistent over a pixel split and WAVE-SPECIALISED: 4 consumer waves run the MFMA
// phase of tile i out of one LDS buffer while 4 producer waves stage tile i+1 (global -> registers, load
// transform, -> LDS) into the other; one barrier per tile.  (Without it the staging time simply added to the MFMA
// time: two co-resident workgroups run in lockstep and overlap nothing.)  Workgroups write partial filters
// [split][tap][cb][ca] that RCV_OP_WGRAD_REDUCE sums in a fixed order (no float atomics => bitwise reproducible).
#include <stdlib.h>
#include "wgrad_common.h"

// One operand tile, global -> (load transform) -> LDS.  Tile-local pixel `pix` = (iy, ix) with ix < TW; it is real when ix < TWV and
// (oy + iy, ox + ix) lies inside the PH x PW plane of image row block `row0` (= n * PH); everything else is stored as zero.  Q
// threads share a pixel (one 16-byte channel quad each), UNR independent loads per thread in flight.  Offsets are 32-bit from the
// tensor base (checked on the host: fewer than 2^31 elements) and the loads are unconditional (lanes without a real pixel read
// element 0 and discard it), so that the loop body is straight-line code for every load mode.
template <int MODE, int UNR, bool SUM, bool SCALAR_STORE>
__device__ __forceinline__ void wstage_tile(const float* __restrict__ src, const float* __restrict__ aux, const float* __restrict__ consts,
                                            float* __restrict__ lds, int tid, int nthreads, int Q, int ch0, int C, int npix, FastDiv fdTW, int TW,
                                            int TWV, int row0, int oy, int ox, int PH, int PW, int S, float4& sum) {
  constexpr bool TWO = MODE == RCV_LOAD_GRAD_ENC || MODE == RCV_LOAD_GRAD_DEC;
  const int q = tid % Q;
  const int ch = ch0 + 4 * q;
  const bool ch_ok = ch < C;
  float4 k[5];
  if (MODE != RCV_LOAD_PLAIN) {
#pragma unroll
    for (int j = 0; j < 5; ++j) k[j] = ch_ok ? wld4(consts + (size_t)j * C + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // A thread's slots are `step` tile pixels apart: its tile coordinates, its tensor offset and its LDS address advance by constants
  // (with one carry into the next tile row), instead of a division, two multiply-adds and a multiply per element; a tile that lies
  // inside the plane (the common case) needs no bounds tests either.  Next to MFMAs every vector instruction costs matrix-pipe time.
  const int lp = tid / Q;
  const int step = nthreads / Q;
  const int dqy = fd_div(step, fdTW), dqx = step - dqy * TW;             // (uniform)
  const int nrows = fd_div(npix, fdTW);                                  // npix is a whole number of tile rows
  const bool interior = TW == TWV && oy >= 0 && ox >= 0 && oy + nrows <= PH && ox + TW <= PW;
  int iy = fd_div(lp, fdTW), ix = lp - iy * TW;
  uint32_t off = (uint32_t)(((row0 + oy + iy) * PW + ox + ix) * C + ch);  // element offset of the slot (modulo 2^32 while the slot is outside the plane)
  const uint32_t d_off = (uint32_t)((dqy * PW + dqx) * C), w_off = (uint32_t)((PW - TW) * C);
  float* dst = lds + lp * S + 4 * q;
  const int d_dst = step * S;
  for (int pix0 = lp; pix0 < npix; pix0 += UNR * step) {
    float4 x[UNR], ax[UNR];
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int pix = pix0 + u * step;
      if (interior) ok[u] = ch_ok && pix < npix;
      else ok[u] = ch_ok && pix < npix && ix < TWV && (unsigned)(oy + iy) < (unsigned)PH && (unsigned)(ox + ix) < (unsigned)PW;
      const uint32_t o = ok[u] ? off : 0u;
      x[u] = wld4(src + o);
      if (TWO) ax[u] = wld4(aux + o);
      else ax[u] = x[u];
      ix += dqx; off += d_off;
      const bool carry = ix >= TW;
      ix -= carry ? TW : 0;
      off += carry ? w_off : 0u;
      if (!interior) iy += dqy + (carry ? 1 : 0);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int pix = pix0 + u * step;
      float4 v = wxform4<MODE>(x[u], ax[u], k);
      if (!ok[u]) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (SUM) { sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w; }
      if (pix < npix) {
        if (SCALAR_STORE) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
        else *reinterpret_cast<float4*>(dst) = v;
      }
      dst += d_dst;
    }
  }
}

// NBF == 0: regular mode (WN column blocks of 16 gathered channels, 9 taps each)
// NBF  > 0: folded mode (WN must be 1): NBF column blocks over n = tap*CA + ca
// GTWO: the gathered operand is a two-tensor gradient load (convT layer); otherwise the pointwise one may be.
// GNCHW: the gathered operand is the NCHW image (<= 4 channels: only the 2-block folded tile and the 16-channel-wide gathered tiles are
// instantiated with it; compiled into every instantiation the path cost the hot ones a large part of their SGPR budget)
// CT_S / CT_IW / CT_SG > 0 (wave-specialised 64 x 64 tile only): the stride, the width of the gathered LDS tile and its pixel pitch are
// compile-time values (dilation 1), so that the 18 gathered-operand reads of a k-step are ONE address register plus immediate offsets
// instead of one address add each (round 2 counted 22 vector instructions per 36-MFMA k-step in the consumer waves, 20 of them adds).
template <int WM, int WN, int WAVES_M, int WAVES_N, int WAVES_K, int NBF, bool SPEC, bool GTWO, bool GNCHW = false, int CT_S = 0, int CT_IW = 0,
          int CT_SG = 0>
__global__ __launch_bounds__(WAVES_M* WAVES_N* WAVES_K * 64 + (SPEC ? 256 : 0)) void wgrad_mfma_kernel(const WgradArgs a) {
  constexpr int NTC = WAVES_M * WAVES_N * WAVES_K * 64;   // MFMA (consumer) threads: waves 0..3
  constexpr int NT = NTC;                                 // staging threads: SPEC ? the next 4 (producer) waves : the same waves
  static_assert(NTC == 256, "wave layouts are 4 waves");
  constexpr int CBT = WM * WAVES_M * 16;
  constexpr int CAT = WN * WAVES_N * 16;
  constexpr bool FOLD = NBF > 0;
  constexpr int NACC = FOLD ? NBF : 9 * WN;
  constexpr int QPMAX = CBT / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bufsz = a.pl_floats + a.gl_floats;            // two (P, G) tile buffers

  const bool producer = SPEC && threadIdx.x >= NTC;       // SPEC: staging-only waves
  const bool consumer = !producer;                        // MFMA waves (SPEC: MFMA-only)
  const bool stager = !SPEC || producer;
  const int tid = producer ? (int)threadIdx.x - NTC : (int)threadIdx.x;     // index inside the role
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_k = wave % WAVES_K;
  const int wave_n = (wave / WAVES_K) % WAVES_N;
  const int wave_m = wave / (WAVES_K * WAVES_N);
  const int l15 = lane & 15, l4 = lane >> 4;

  // 1-D grid, channel tile fastest: the channel tiles of one pixel split run side by side on ONE XCD and share its L2
  const int n_ca_tiles = FOLD ? 1 : (a.CAP + CAT - 1) / CAT;
  const int bl = xcd_remap(blockIdx.x, gridDim.x);
  const int ctile = bl % a.nctiles, split = bl / a.nctiles;
  const int cb_tile = ctile / n_ca_tiles, ca_tile = ctile % n_ca_tiles;
  const int cb0 = cb_tile * CBT, ca0 = ca_tile * CAT;
  // 16-byte quads of the pointwise operand that hold real channels (a power of two, so that it divides the thread count): with 8
  // channels in a 16-wide tile only half the staging threads would otherwise have a load to issue
  int QP = QPMAX;
  while (QP > 1 && (QP / 2) * 4 >= a.CB - cb0) QP /= 2;
  const int s = a.stride, d = a.dil;
  const int QG = FOLD ? (a.CA + 3) / 4 : CAT / 4;        // 16-byte quads per staged gathered pixel
  const int np_pix = a.R * a.Wt4, ng_pix = a.IH * a.IW;


  // ---------------- staging: global -> (load transform) -> LDS, 4 independent 16-byte loads per thread in flight ----------------
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int UNR = (FOLD || (SPEC && WAVES_K != 2)) ? 8 : 4;     // folded (<= 8 channel) tiles are HBM bound: more loads in flight (8 on the 16 x 16 tile: measured, no gain; on the 32 x 16 tiles > 256 registers)
  auto stage = [&](int tile, float* pl, float* gl) {
    int t = tile;
    const int tx_i = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty_i = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int y0 = ty_i * a.R, x0 = tx_i * a.Wt;
    // pointwise tile: R x Wt4 pixels (columns >= Wt and anything outside the plane are zero)
    // (GTWO: the host guarantees that at most one operand is a two-tensor gradient load -- the other switch arms are never compiled)
    switch (GTWO && (a.p_mode == RCV_LOAD_GRAD_ENC || a.p_mode == RCV_LOAD_GRAD_DEC) ? RCV_LOAD_PLAIN : a.p_mode) {
      case RCV_LOAD_PLAIN: wstage_tile<RCV_LOAD_PLAIN, UNR, true, false>(a.p, a.p_aux, a.p_c, pl, tid, NT, QP, cb0, a.CB, np_pix, a.fdWt4, a.Wt4, a.Wt, n * a.Hp, y0, x0, a.Hp, a.Wp, a.SP, bsum); break;
      case RCV_LOAD_AFFINE: wstage_tile<RCV_LOAD_AFFINE, UNR, true, false>(a.p, a.p_aux, a.p_c, pl, tid, NT, QP, cb0, a.CB, np_pix, a.fdWt4, a.Wt4, a.Wt, n * a.Hp, y0, x0, a.Hp, a.Wp, a.SP, bsum); break;
      case RCV_LOAD_AFFINE_RELU: wstage_tile<RCV_LOAD_AFFINE_RELU, UNR, true, false>(a.p, a.p_aux, a.p_c, pl, tid, NT, QP, cb0, a.CB, np_pix, a.fdWt4, a.Wt4, a.Wt, n * a.Hp, y0, x0, a.Hp, a.Wp, a.SP, bsum); break;
      case RCV_LOAD_GRAD_ENC: if (!GTWO) wstage_tile<RCV_LOAD_GRAD_ENC, UNR, true, false>(a.p, a.p_aux, a.p_c, pl, tid, NT, QP, cb0, a.CB, np_pix, a.fdWt4, a.Wt4, a.Wt, n * a.Hp, y0, x0, a.Hp, a.Wp, a.SP, bsum); break;
      default: if (!GTWO) wstage_tile<RCV_LOAD_GRAD_DEC, UNR, true, false>(a.p, a.p_aux, a.p_c, pl, tid, NT, QP, cb0, a.CB, np_pix, a.fdWt4, a.Wt4, a.Wt, n * a.Hp, y0, x0, a.Hp, a.Wp, a.SP, bsum); break;
    }
    if (GNCHW) {   // gathered tile from the NCHW image: one pixel (<= 4 planes) per thread
      for (int pix0 = tid; pix0 < ng_pix; pix0 += UNR * NT) {
        float4 x[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int pix = pix0 + u * NT;
          x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          const int iy = fd_div(pix, a.fdIW), ix = pix - iy * a.IW;
          const int gy = y0 * s - d + iy, gx = x0 * s - d + ix;
          if (pix < ng_pix && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) {
            const size_t plane = (size_t)a.H * a.W;
            const size_t base = (size_t)n * a.CA * plane + (size_t)gy * a.W + gx;
            if (0 < a.CA) x[u].x = a.g[base];
            if (1 < a.CA) x[u].y = a.g[base + plane];
            if (2 < a.CA) x[u].z = a.g[base + 2 * plane];
            if (3 < a.CA) x[u].w = a.g[base + 3 * plane];
          }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int pix = pix0 + u * NT;
          if (pix < ng_pix) {
            float* dst = gl + pix * a.SG;
            dst[0] = x[u].x; dst[1] = x[u].y; dst[2] = x[u].z; dst[3] = x[u].w;
            if (!FOLD) {
#pragma unroll
              for (int j = 4; j < CAT; ++j) dst[j] = 0.f;
            }
          }
        }
      }
    } else {
      // gathered tile: IH x IW pixels around the pointwise tile (zero padded outside the plane)
      float4 nosum = make_float4(0.f, 0.f, 0.f, 0.f);
      const int oy = y0 * s - d, ox = x0 * s - d;
      if (GTWO) {
        if (a.g_mode == RCV_LOAD_GRAD_ENC) wstage_tile<RCV_LOAD_GRAD_ENC, UNR, false, FOLD>(a.g, a.g_aux, a.g_c, gl, tid, NT, QG, ca0, a.CA, ng_pix, a.fdIW, a.IW, a.IW, n * a.H, oy, ox, a.H, a.W, a.SG, nosum);
        else wstage_tile<RCV_LOAD_GRAD_DEC, UNR, false, FOLD>(a.g, a.g_aux, a.g_c, gl, tid, NT, QG, ca0, a.CA, ng_pix, a.fdIW, a.IW, a.IW, n * a.H, oy, ox, a.H, a.W, a.SG, nosum);
      } else {
        switch (a.g_mode) {
          case RCV_LOAD_PLAIN: wstage_tile<RCV_LOAD_PLAIN, UNR, false, FOLD>(a.g, a.g_aux, a.g_c, gl, tid, NT, QG, ca0, a.CA, ng_pix, a.fdIW, a.IW, a.IW, n * a.H, oy, ox, a.H, a.W, a.SG, nosum); break;
          case RCV_LOAD_AFFINE: wstage_tile<RCV_LOAD_AFFINE, UNR, false, FOLD>(a.g, a.g_aux, a.g_c, gl, tid, NT, QG, ca0, a.CA, ng_pix, a.fdIW, a.IW, a.IW, n * a.H, oy, ox, a.H, a.W, a.SG, nosum); break;
          default: wstage_tile<RCV_LOAD_AFFINE_RELU, UNR, false, FOLD>(a.g, a.g_aux, a.g_c, gl, tid, NT, QG, ca0, a.CA, ng_pix, a.fdIW, a.IW, a.IW, n * a.H, oy, ox, a.H, a.W, a.SG, nosum); break;
        }
      }
    }
  };

  const bool do_stage = !(a.dbg & RCV_F_DBG_NOSTAGE), do_mfma = !(a.dbg & RCV_F_DBG_NOMFMA);
  if (QP < QPMAX) {      // the channel columns no staging thread writes must read as zero
    const int nfl = (SPEC ? 2 : 1) * bufsz;
    for (int e = threadIdx.x; e < nfl; e += blockDim.x) smem[e] = 0.f;
    __syncthreads();
  }
  // bias partial: sum over the threads that staged the same channel quad (fixed order); runs after the main loop
  auto bias_partial = [&]() {
    if (a.part_bias && ca_tile == 0) {
      __syncthreads();
      float4* sb = reinterpret_cast<float4*>(smem);
      if (stager) sb[tid] = bsum;            // the staging threads summed the pointwise operand
      __syncthreads();
      if (stager && tid < QP) {
        float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = tid; e < NT; e += QP) { const float4 v = sb[e]; u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w; }
        const int cb = cb0 + 4 * tid;
        if (cb < a.CBP) *reinterpret_cast<float4*>(a.part_bias + (size_t)split * a.CBP + cb) = u;
      }
    }
  };
  if (SPEC && producer) {
    // Producer waves: their whole program, BEFORE the accumulators exist (the 144 accumulator registers of the consumer role
    // are then not live here, which is what lets 8 loads per thread stay in flight).  Barrier for barrier the same sequence as
    // the consumer path below: 1 + one per tile + 2 per extra pixel slice + the bias partial's.
    if (split < a.ntiles && do_stage) stage(split, smem, smem + a.pl_floats);
    __syncthreads();
    int it = 0;
    for (int tile = split; tile < a.ntiles; tile += a.nsplit, ++it) {
      const int next = tile + a.nsplit;
      if (next < a.ntiles && do_stage) { float* pn = smem + ((it + 1) & 1) * bufsz; stage(next, pn, pn + a.pl_floats); }
      __syncthreads();
    }
    for (int kk = 1; kk < WAVES_K; ++kk) { __syncthreads(); __syncthreads(); }
    bias_partial();
    return;
  }

  // ---------------- accumulators and lane offsets ----------------
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
        loff[t * WN + nn] = ((ky * d) * a.IW + kx * d) * a.SG + (l4 * s) * a.SG + (wave_n * WN + nn) * 16 + l15;
      }
  }
  const int ksteps = np_pix / 4;
  const int a_lane = l4 * a.SP + (wave_m * WM) * 16 + l15;
  constexpr bool CT = CT_IW > 0;
  const int g_lane = (l4 * (CT ? CT_S : s)) * (CT ? CT_SG : a.SG) + (wave_n * WN) * 16 + l15;      // CT: the lane part of every gathered read

  // The k-step slice of this wave (wave_k is the same for all its lanes: made a scalar so that the loop and its pixel arithmetic run
  // on the scalar unit).
  const int wave_k_s = __builtin_amdgcn_readfirstlane(wave_k);
  auto load_ops = [&](const float* pl, const float* gl, int j, float (&av)[WM], float (&bv)[NACC]) {
    const int p0 = 4 * j;
    const int ty = fd_div(p0, a.fdWt4), tx = p0 - ty * a.Wt4;
#pragma unroll
    for (int m = 0; m < WM; ++m) av[m] = pl[p0 * a.SP + a_lane + m * 16];
    if constexpr (CT && !FOLD) {
      const float* gb = gl + ((ty * CT_S) * CT_IW + tx * CT_S) * CT_SG + g_lane;
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nn = 0; nn < WN; ++nn) bv[t * WN + nn] = gb[((t / 3) * CT_IW + (t % 3)) * CT_SG + nn * 16];      // immediates
    } else {
      const float* gj = gl + ((ty * s) * a.IW + tx * s) * a.SG;
#pragma unroll
      for (int t = 0; t < NACC; ++t) bv[t] = gj[loff[t]];
    }
  };
  auto mfma_ops = [&](const float (&av)[WM], const float (&bv)[NACC]) {
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
      for (int m = 0; m < WM; ++m) acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[t], acc[t][m], 0, 0, 0);
  };
  // Tiles with few MFMAs per k-step (2..18: the <= 32-channel tiles, 64..576 cycles of matrix work) read the operands of k-step j+1
  // while the MFMAs of k-step j run: with the reads issued right in front of their MFMAs every k-step began with an exposed LDS
  // round trip and the matrix pipe of the 16 x 16 tile was 36 % busy.  (The 36-MFMA k-steps of the wave-specialised tiles hide it
  // by themselves: measured, no difference there.)
  constexpr bool PIPE = true;
  auto contract = [&](const float* pl, const float* gl) {
    if (PIPE) {
      // (reads are unconditional -- past the end they repeat the last k-step and are discarded: behind a branch the compiler
      // waits for ALL outstanding LDS reads where the paths meet, the prefetched set included)
      float av0[WM], bv0[NACC], av1[WM], bv1[NACC];
      const int jlast = ksteps - 1;
      int j = wave_k_s;
      load_ops(pl, gl, j < jlast ? j : jlast, av0, bv0);
      for (; j < ksteps; j += 2 * WAVES_K) {
        const int j1 = j + WAVES_K, j2 = j + 2 * WAVES_K;
        load_ops(pl, gl, j1 < jlast ? j1 : jlast, av1, bv1);
        mfma_ops(av0, bv0);
        load_ops(pl, gl, j2 < jlast ? j2 : jlast, av0, bv0);
        if (j1 < ksteps) mfma_ops(av1, bv1);
      }
    } else {
      for (int j = wave_k_s; j < ksteps; j += WAVES_K) {
        float av[WM], bv[NACC];
        load_ops(pl, gl, j, av, bv);
        mfma_ops(av, bv);
      }
    }
  };
  if (SPEC) {
    // the producer waves (above) stage tile i+1 into the other buffer while these consumer waves contract tile i: one barrier per tile
    __syncthreads();
    int it = 0;
    for (int tile = split; tile < a.ntiles; tile += a.nsplit, ++it) {
      float* pl = smem + (it & 1) * bufsz;
      if (do_mfma) contract(pl, pl + a.pl_floats);
      __syncthreads();
    }
  } else {
    // staging-heavy (narrow) tiles: every wave stages, then every wave contracts; two workgroups per CU interleave
    for (int tile = split; tile < a.ntiles; tile += a.nsplit) {
      __syncthreads();
      if (do_stage) stage(tile, smem, smem + a.pl_floats);
      __syncthreads();
      if (do_mfma) contract(smem, smem + a.pl_floats);
    }
  }

  // ---- reduce the WAVES_K pixel-slices of the workgroup (fixed order) ----
  if (WAVES_K > 1) {
    constexpr int PER_WAVE = NACC * WM * 4 * 64;
    for (int kk = 1; kk < WAVES_K; ++kk) {
      __syncthreads();
      float* sc = smem + (wave_m * WAVES_N + wave_n) * PER_WAVE;
      if (consumer && wave_k == kk) {
#pragma unroll
        for (int t = 0; t < NACC; ++t)
#pragma unroll
          for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[((t * WM + m) * 4 + r) * 64 + lane] = acc[t][m][r];
      }
      __syncthreads();
      if (consumer && wave_k == 0) {
#pragma unroll
        for (int t = 0; t < NACC; ++t)
#pragma unroll
          for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][m][r] += sc[((t * WM + m) * 4 + r) * 64 + lane];
      }
    }
  }
  if (consumer && wave_k == 0 && !(a.dbg & RCV_F_DBG_NOEPI)) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
      int tap, ca;
      if (FOLD) {
        const int nn = t * 16 + l15;
        tap = nn / a.CA; ca = nn - tap * a.CA;
        if (nn >= 9 * a.CA) continue;
      } else {
        tap = t / WN;
        ca = ca0 + (wave_n * WN + (t % WN)) * 16 + l15;
      }
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cb = cb0 + (wave_m * WM + m) * 16 + 4 * l4 + r;
          if (cb < a.CBP && ca < a.CAP) a.part[(((size_t)split * 9 + tap) * a.CBP + cb) * a.CAP + ca] = acc[t][m][r];
        }
    }
  }
  bias_partial();
}

// dW[cb][ca][tap] = sum_split part[split][tap][cb][ca];  db[cb] = sum_split part_bias[split][cb]
// 64 consecutive partial-layout elements per workgroup (block `blk` of this layer), 16 split groups of 16 threads: a thread sums one
// 16-byte piece over the splits sidx = group, group + 16, ... with four 16-byte loads in flight -- up to 64 splits are ONE round trip.
// (A thread per element with four split groups and 4-byte loads: 35 us per batched launch of the headline step against 31; 256-element
// blocks with 1 KB runs per split and eight loads in flight: 37 -- a quarter of the workgroups.)  The order of additions is a fixed
// function of (nsplit) => bitwise reproducible.
__device__ __forceinline__ void wgrad_reduce_block(const float* __restrict__ part, const float* __restrict__ part_bias, float* __restrict__ dw,
                                                   float* __restrict__ db, int nsplit, int CB, int CA, int CBP, int CAP, int blk,
                                                   double (&sh)[16][64]) {
  const int el4 = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int totalP = 9 * CBP * CAP;                     // a multiple of 64: a block lies entirely in the filter part or in the bias row
  const int e = blk * 64 + 4 * el4;
  double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};     // double: bias / BN-adjacent filters are cancellation-heavy sums
  const float* src = nullptr;
  size_t st = 0;
  if (e < totalP) { src = part + e; st = (size_t)totalP; }
  else if (part_bias && e < totalP + CBP) { src = part_bias + (e - totalP); st = (size_t)CBP; }
  if (src) {
    int sidx = grp;
    for (; sidx + 48 < nsplit; sidx += 64) {
      const float4 v0 = wld4(src + (size_t)sidx * st), v1 = wld4(src + (size_t)(sidx + 16) * st);
      const float4 v2 = wld4(src + (size_t)(sidx + 32) * st), v3 = wld4(src + (size_t)(sidx + 48) * st);
      a0[0] += (double)v0.x + (double)v2.x; a0[1] += (double)v0.y + (double)v2.y; a0[2] += (double)v0.z + (double)v2.z; a0[3] += (double)v0.w + (double)v2.w;
      a1[0] += (double)v1.x + (double)v3.x; a1[1] += (double)v1.y + (double)v3.y; a1[2] += (double)v1.z + (double)v3.z; a1[3] += (double)v1.w + (double)v3.w;
    }
    for (; sidx < nsplit; sidx += 16) {
      const float4 v = wld4(src + (size_t)sidx * st);
      a0[0] += (double)v.x; a0[1] += (double)v.y; a0[2] += (double)v.z; a0[3] += (double)v.w;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) sh[grp][4 * el4 + j] = a0[j] + a1[j];
  __syncthreads();
  const int el = threadIdx.x;
  if (el < 64) {
    const int e1 = blk * 64 + el;
    double u = 0.0;
#pragma unroll
    for (int g = 0; g < 16; g += 4) u += (sh[g][el] + sh[g + 1][el]) + (sh[g + 2][el] + sh[g + 3][el]);
    const float uf = (float)u;
    if (e1 < totalP) {
      const int ca = e1 % CAP;
      const int cb = (e1 / CAP) % CBP;
      const int t = e1 / (CAP * CBP);
      if (ca < CA && cb < CB) dw[((size_t)cb * CA + ca) * 9 + t] = uf;
    } else if (part_bias && e1 < totalP + CBP) {
      const int cb = e1 - totalP;
      if (db && cb < CB) db[cb] = uf;
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ part_bias,
                                                           float* __restrict__ dw, float* __restrict__ db, int nsplit, int CB, int CA,
                                                           int CBP, int CAP) {
  __shared__ double sh[16][64];
  wgrad_reduce_block(part, part_bias, dw, db, nsplit, CB, CA, CBP, CAP, blockIdx.x, sh);
}

// several layers in one launch (RCV_OP_WGRAD_REDUCE_BATCH): the workgroup finds its job in the (<= 64 rows) table
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const rcv_reduce_job* __restrict__ jobs, int njobs) {
  __shared__ double sh[16][64];
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].first_block) ++j;
  const rcv_reduce_job jb = jobs[j];
  if (jb.nsplit == 0) {      // zero-fill job: db[0..CB) = 0 (the bias ahead of a BatchNorm has an identically zero gradient)
    const int e = ((int)blockIdx.x - jb.first_block) * 256 + (int)threadIdx.x;
    if (e < jb.CB) jb.db[e] = 0.f;
    return;
  }
  const int CAP = jb.CA <= 4 ? 4 : (jb.CA + 15) / 16 * 16, CBP = (jb.CB + 15) / 16 * 16;      // wgrad_cap / round_up of the host side
  const float* pb = jb.db ? jb.part + (size_t)jb.nsplit * 9 * CBP * CAP : nullptr;
  wgrad_reduce_block(jb.part, pb, jb.dw, jb.db, jb.nsplit, jb.CB, jb.CA, CBP, CAP, (int)blockIdx.x - jb.first_block, sh);
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
struct WTile {
  int WM, WN, WAVES_M, WAVES_N, WAVES_K, NBF, SPEC;
  int cbt() const { return WM * WAVES_M * 16; }
  int cat() const { return WN * WAVES_N * 16; }
  int nt() const { return 256; }      // staging threads
};
static const WTile kWT[] = {
    {2, 2, 2, 2, 1, 0, 1},  // 0: 64 x 64, producer/consumer waves
    {2, 2, 2, 1, 2, 0, 1},  // 1: 64 x 32
    {2, 2, 1, 2, 2, 0, 1},  // 2: 32 x 64
    {2, 2, 1, 1, 4, 0, 1},  // 3: 32 x 32
    {2, 1, 1, 1, 4, 0, 0},  // 4: 32 x 16, shared roles (staging heavy)
    {1, 2, 1, 1, 4, 0, 0},  // 5: 16 x 32
    {1, 1, 1, 1, 4, 0, 0},  // 6: 16 x 16
    {1, 1, 1, 1, 4, 2, 0},  // 7: 16 x (9 taps x <=3 ch folded into 2 blocks)
    {1, 1, 1, 1, 4, 5, 0},  // 8: 16 x (9 taps x <=8 ch folded into 5 blocks)
};

// the 64 x 64 producer/consumer tile with compile-time gathered-tile geometry (the shapes the planner picks for the 64- and
// 128-channel layers of the BASELINE planes on a 256-CU part); any other geometry runs the run-time-pitch instantiation
template <bool GTWO, int CT_S, int CT_IW, int CT_SG>
static int wlaunch_ct(const WgradArgs& a, dim3 grid, size_t lds, hipStream_t s, int dev) {
  auto kern = wgrad_mfma_kernel<2, 2, 2, 2, 1, 0, true, GTWO, false, CT_S, CT_IW, CT_SG>;
  static size_t configured[RCV_MAX_DEVICES];
  RCV_ENSURE_LDS(kern, lds, dev, configured);
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, a);
  RCV_HIP(hipGetLastError());
  return RCV_OK;
}

template <int WM, int WN, int WAVES_M, int WAVES_N, int WAVES_K, int NBF, bool SPEC>
static int wlaunch_inst(const WgradArgs& a, bool gtwo, dim3 grid, size_t lds, hipStream_t s, int dev) {
  constexpr int NT = WAVES_M * WAVES_N * WAVES_K * 64 + (SPEC ? 256 : 0);     // consumer (+ producer) waves
  if constexpr (SPEC && WAVES_K == 1 && NBF == 0) {
    if (a.dil == 1 && a.g_mode != RCV_LOAD_NCHW && !RCV_ENV("RCV_WGRAD_NO_CT")) {
      if (a.stride == 1 && a.IW == 22 && a.SG == 80 && !gtwo) return wlaunch_ct<false, 1, 22, 80>(a, grid, lds, s, dev);
      if (a.stride == 2 && a.IW == 17 && a.SG == 72) return gtwo ? wlaunch_ct<true, 2, 17, 72>(a, grid, lds, s, dev) : wlaunch_ct<false, 2, 17, 72>(a, grid, lds, s, dev);
    }
  }
  if (gtwo) {
    auto kern = wgrad_mfma_kernel<WM, WN, WAVES_M, WAVES_N, WAVES_K, NBF, SPEC, true>;
    static size_t configured[RCV_MAX_DEVICES];
    RCV_ENSURE_LDS(kern, lds, dev, configured);
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, a);
  } else if (a.g_mode == RCV_LOAD_NCHW) {
    constexpr bool CAN = NBF == 2 || (NBF == 0 && WN * WAVES_N == 1);       // tiles the planner gives an NCHW image (wmake_plan)
    if constexpr (CAN) {
      auto kern = wgrad_mfma_kernel<WM, WN, WAVES_M, WAVES_N, WAVES_K, NBF, SPEC, false, true>;
      static size_t configured[RCV_MAX_DEVICES];
      RCV_ENSURE_LDS(kern, lds, dev, configured);
      hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, a);
    } else {
      rcv_set_error("wgrad: NCHW gathered operand on a tile without that path");
      return RCV_E_ARG;
    }
  } else {
    auto kern = wgrad_mfma_kernel<WM, WN, WAVES_M, WAVES_N, WAVES_K, NBF, SPEC, false>;
    static size_t configured[RCV_MAX_DEVICES];
    RCV_ENSURE_LDS(kern, lds, dev, configured);
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, a);
  }
  RCV_HIP(hipGetLastError());
  return RCV_OK;
}

struct WPlan { int first; int bf3; int nctiles; int tile; int R, Wt, Wt4, tiles_x, tiles_y, IH, IW, SP, SG, nsplit, pl_floats, gl_floats; size_t lds; dim3 grid; int CAP, CBP; };

static int wmake_plan(const rcv_handle* h, const rcv_op* op, WPlan* pl) {
  const int N = op->i[RCV_I_N], H = op->i[RCV_I_H], W = op->i[RCV_I_W];
  const int CA = op->i[RCV_I_CIN], CB = op->i[RCV_I_COUT];
  const int Hp = op->i[RCV_I_HO], Wp = op->i[RCV_I_WO];
  const int s = op->i[RCV_I_STRIDE], d = op->i[RCV_I_DIL];
  RCV_CHECK_ARG(N > 0 && H > 0 && W > 0 && CA > 0 && CB > 0, "wgrad: empty shape");
  RCV_CHECK_ARG((s == 1 || s == 2) && (d == 1 || d == 2), "wgrad: stride %d dilation %d unsupported", s, d);
  RCV_CHECK_ARG(Hp == (H - 1) / s + 1 && Wp == (W - 1) / s + 1, "wgrad: pointwise plane %dx%d does not match %dx%d / %d", Hp, Wp, H, W, s);
  RCV_CHECK_ARG(CB % 4 == 0, "wgrad: pointwise channels %d must be a multiple of 4", CB);
  RCV_CHECK_ARG((size_t)N * H * W * CA < (1ull << 31) && (size_t)N * Hp * Wp * CB < (1ull << 31), "wgrad: operand exceeds 2^31 elements (32-bit offsets)");
  const bool nchw = op->i[RCV_I_INMODE] == RCV_LOAD_NCHW;
  if (nchw) RCV_CHECK_ARG(CA <= 4, "wgrad: NCHW gathered operand supports <=4 channels");
  else RCV_CHECK_ARG(CA % 4 == 0, "wgrad: gathered channels %d must be a multiple of 4", CA);
  pl->CAP = wgrad_cap(CA); pl->CBP = round_up(CB, 16);
  int cbt_want = pl->CBP >= 64 ? 64 : (pl->CBP >= 32 ? 32 : 16);
  int cat_want = pl->CAP >= 64 ? 64 : (pl->CAP >= 32 ? 32 : 16);
  if (const char* ev = RCV_ENV("RCV_WGRAD_CT")) { const int c = atoi(ev); if (c == 16 || c == 32) { if (cbt_want > c) cbt_want = c; if (cat_want > c) cat_want = c; } }
  pl->tile = -1;
  pl->first = 0;
  if (wgrad_first_supported(op)) {      // first layer: vector-ALU kernel (wgrad_first.hip), one partial row per persistent workgroup
    pl->first = 1;
    pl->nsplit = wgrad_first_nsplit(h, op);
    pl->nctiles = 1; pl->lds = 0; pl->grid = dim3(pl->nsplit, 1, 1);
    pl->R = 8; pl->Wt = 64; pl->Wt4 = 64; pl->tiles_x = ceil_div(Wp, 64); pl->tiles_y = ceil_div(Hp, 8);
    pl->IH = 8 + 2 * d; pl->IW = 64 + 2 * d; pl->SP = 8; pl->SG = 1; pl->pl_floats = 0; pl->gl_floats = 0;
    return RCV_OK;
  }
  pl->bf3 = 0;
  if (wgrad_bf3_supported(h, op)) {     // wide stride-1 layers: fp32 products on the bf16 matrix pipe (wgrad_bf3.hip)
    wgrad_bf3_geometry(h, op, &pl->bf3, &pl->tiles_x, &pl->tiles_y, &pl->nsplit, &pl->nctiles);
    pl->lds = 0; pl->grid = dim3(pl->nsplit * pl->nctiles, 1, 1);
    pl->R = 64 / pl->bf3; pl->Wt = pl->bf3; pl->Wt4 = pl->bf3; pl->IH = pl->R + 2; pl->IW = pl->bf3 + 2; pl->SP = 0; pl->SG = 0; pl->pl_floats = 0; pl->gl_floats = 0;
    return RCV_OK;
  }
  if (wgradn_bf3_supported(h, op)) {    // narrow layers, the same arithmetic (wgradn_bf3.hip): bf3 = -(tile rows), four splits per workgroup
    int th, ng;
    wgradn_bf3_geometry(h, op, &th, &pl->tiles_x, &pl->tiles_y, &ng);
    pl->bf3 = -th; pl->nsplit = 4 * ng; pl->nctiles = 1; pl->lds = 0; pl->grid = dim3(ng, 1, 1);
    pl->R = th; pl->Wt = 16; pl->Wt4 = 16; pl->IH = (th - 1) * s + 3; pl->IW = 15 * s + 3; pl->SP = 0; pl->SG = 0; pl->pl_floats = 0; pl->gl_floats = 0;
    return RCV_OK;
  }
  // (an NCHW image with 4 channels does not fit the 2-block folded tile and the 5-block one carries no NCHW path: 16 x 16 tile)
  const bool fold = CA <= 8 && cbt_want == 16 && !(nchw && 9 * CA > 32) && !RCV_ENV("RCV_NO_FOLD");
  if (fold) pl->tile = 9 * CA <= 32 ? 7 : 8;
  else {
    for (int t = 0; t < 7; ++t) if (kWT[t].cbt() == cbt_want && kWT[t].cat() == cat_want) pl->tile = t;
    if (pl->tile < 0) {  // 64x16 / 16x64 combinations: the widest tile not exceeding both
      const int cb2 = cbt_want > 32 ? 32 : cbt_want, ca2 = cat_want > 32 ? 32 : cat_want;
      for (int t = 0; t < 7; ++t) if (kWT[t].cbt() == cb2 && kWT[t].cat() == ca2) pl->tile = t;
    }
  }
  RCV_CHECK_ARG(pl->tile >= 0, "wgrad: no tile for %d x %d channels", CB, CA);
  const WTile& wt = kWT[pl->tile];
  const int QG = fold ? ceil_div(CA, 4) : wt.cat() / 4;
  pl->SP = wt.cbt() % 32 == 0 ? wt.cbt() + 16 : wt.cbt();
  if (fold) pl->SG = 4 * QG + 1;
  else pl->SG = s == 1 ? (wt.cat() % 32 == 0 ? wt.cat() + 16 : wt.cat()) : wt.cat() + 8;
  // pixel tile: widest row segment, then as many rows as the prefetch registers and the LDS budget allow
  int per_cu = wt.SPEC ? 1 : 2;                     // SPEC: 512 threads, two LDS buffers => one workgroup per CU
  if (const char* ev = RCV_ENV("RCV_WGRAD_OCC")) { const int o = atoi(ev); if (o >= 1 && o <= 4 && !wt.SPEC) per_cu = o; }
  // SPEC: per buffer.  (Round 3: the sweep of scripts/experiments/sweep_tiles.py found the 5 x 20 tile of the 64 x 64 producer/consumer
  // kernel -- 79.4 KB per buffer, i.e. ALL of the CU's LDS -- 4 % faster than the 3 x 20 tile this budget allows, 118 -> 113.5 us on the
  // 128 -> 128 layers; inside the step it measured 0.6 % SLOWER: with 108 KB the kernel leaves room for a narrow data-gradient workgroup
  // of the other stream on the same CU, with 159 KB it does not.  The budget stays.)
  const size_t budget = (wt.SPEC ? 78 : 160 / per_cu) * 1024 / sizeof(float);
  // Pixel tile: among the (row segment, rows) shapes that fit the LDS budget, the one with the lowest estimated time
  //     t ~ (1 + 3 / k-steps per wave and tile) / fill  +  w * halo,
  //   fill = real pointwise pixels / MFMA pixel slots of the launch (row segments padded to 4 columns, ragged last tiles, workgroups
  //          that get one tile fewer than the others or none at all: 480 tiles on 64 workgroups run 8 rounds with 94 % of the slots used),
  //   halo = staged global bytes per pointwise pixel relative to reading every operand element once (the gathered tile carries a
  //          halo of 2*dil rows and columns: a 1 x 80 tile of a stride-1 layer reads its gathered operand 3.1 times, a 4 x 40 tile 1.6 times);
  //   w: the producer/consumer tiles hide their staging behind the MFMAs (small w), the shared-role tiles add it to them.
  // RCV_WGRAD_WIDE=1 (experiments) restores "widest row segment first".
  const int ctiles_plan = ceil_div(pl->CBP, wt.cbt()) * (fold ? 1 : ceil_div(pl->CAP, wt.cat()));
  const int wg_max = (per_cu * h->num_cus) / ctiles_plan > 0 ? (per_cu * h->num_cus) / ctiles_plan : 1;
  int bestR = 0, bestWt = 0;
  {
    const int m1 = op->i[RCV_I_INMODE], m2 = op->i[RCV_I_INMODE2];
    const double cG = (double)CA * ((m1 == RCV_LOAD_GRAD_ENC || m1 == RCV_LOAD_GRAD_DEC) ? 2 : 1);
    const double cP = (double)CB * ((m2 == RCV_LOAD_GRAD_ENC || m2 == RCV_LOAD_GRAD_DEC) ? 2 : 1);
    const bool wide_first = RCV_ENV("RCV_WGRAD_WIDE") != nullptr;
    const int max_px = wt.SPEC ? 1024 : 640;
    const double w_halo = wt.SPEC ? 0.15 : 1.0;
    double best_cost = 1e30;
    int prevWt = 0;
    for (int nx = 1; nx <= Wp; ++nx) {                 // every distinct segment width (~2 sqrt(Wp) of them)
      const int Wt = ceil_div(Wp, nx), Wt4 = round_up(Wt, 4);
      if (Wt == prevWt) { if (Wt <= 4) break; continue; }
      prevWt = Wt;
      const int IW = (Wt4 - 1) * s + 2 * d + 1;
      // largest row count within the pixel cap and the LDS budget:  R Wt4 SP + ((R-1) s + 2d+1) IW SG <= budget
      long rmax = max_px / Wt4;
      const long fixed = (long)(2 * d + 1 - s) * IW * pl->SG, per_row = (long)Wt4 * pl->SP + (long)s * IW * pl->SG;
      if ((long)budget - fixed < per_row) continue;
      if (((long)budget - fixed) / per_row < rmax) rmax = ((long)budget - fixed) / per_row;
      if (rmax > Hp) rmax = Hp;
      int prevRe = 0;
      for (int R = (int)rmax; R >= 1; --R) {
        const int IH = (R - 1) * s + 2 * d + 1;
        if (R * Wt4 > max_px || IH * IW >= 65536) continue;
        if ((size_t)R * Wt4 * pl->SP + (size_t)IH * IW * pl->SG > budget) continue;
        const int Re = ceil_div(Hp, ceil_div(Hp, R));            // rows actually used (equal row groups)
        if (Re == prevRe) continue;
        prevRe = Re;
        if (wide_first) { if (bestR == 0) { bestR = R; bestWt = Wt; } break; }
        const int IHe = (Re - 1) * s + 2 * d + 1;
        const long ntile = (long)N * ceil_div(Wp, Wt) * ceil_div(Hp, Re);   // ragged last tiles cost as much as full ones
        const long wgs = ntile < wg_max ? ntile : wg_max;
        const long rounds = ceil_div(ntile, wgs);
        // (a launch that cannot give every CU a workgroup is charged for the idle ones as well)
        const double fill = (double)N * Hp * Wp / ((double)rounds * wg_max * Re * Wt4);
        const double halo = ((double)Re * Wt4 * cP + (double)IHe * IW * cG) / ((double)Re * Wt * (cP + (double)s * s * cG));
        // (every tile also costs a barrier and a cold start of the operand pipeline: about three k-steps of a wave's matrix time)
        const double per_tile = 1.0 + 3.0 * wt.WAVES_K / (Re * Wt4 / 4);
        const double cost = per_tile / fill + w_halo * halo;
        if (cost < best_cost * 0.995 || (cost < best_cost * 1.005 && Re * Wt > bestR * bestWt)) {
          best_cost = cost < best_cost ? cost : best_cost; bestR = Re; bestWt = Wt;
        }
        if (Re * 3 < (int)rmax) break;                           // much smaller tiles only add halo and barriers
      }
      if (wide_first && bestR) break;
    }
  }
  if (const char* ev = RCV_ENV("RCV_WGRAD_TILE")) {      // experiments build: "R,Wt" (scripts/experiments/sweep_tiles.py); the LDS check below still applies
    int r = 0, wt = 0;
    if (sscanf(ev, "%d,%d", &r, &wt) == 2 && r >= 1 && r <= Hp && wt >= 1 && wt <= Wp) { bestR = r; bestWt = wt; }
  }
  RCV_CHECK_ARG(bestR > 0, "wgrad: no pixel tile fits (plane %dx%d)", Hp, Wp);
  const int R = ceil_div(Hp, ceil_div(Hp, bestR));
  const int Wt4 = round_up(bestWt, 4);
  pl->R = R; pl->Wt = bestWt; pl->Wt4 = Wt4; pl->IW = (Wt4 - 1) * s + 2 * d + 1; pl->IH = (R - 1) * s + 2 * d + 1;
  pl->tiles_x = ceil_div(Wp, bestWt); pl->tiles_y = ceil_div(Hp, R);
  pl->pl_floats = round_up(R * Wt4 * pl->SP, 4);
  pl->gl_floats = round_up(pl->IH * pl->IW * pl->SG, 4);
  size_t floats = (wt.SPEC ? 2 : 1) * ((size_t)pl->pl_floats + pl->gl_floats);
  const int nacc = wt.NBF ? wt.NBF : 9 * wt.WN;
  const size_t red = (size_t)wt.WAVES_M * wt.WAVES_N * nacc * wt.WM * 4 * 64;
  if (wt.WAVES_K > 1 && floats < red) floats = red;
  const size_t bias_scratch = (size_t)wt.nt() * 4;
  if (floats < bias_scratch) floats = bias_scratch;
  pl->lds = floats * sizeof(float);
  RCV_CHECK_ARG(pl->lds <= (size_t)h->max_lds, "wgrad: tile needs %zu B of LDS (limit %d)", pl->lds, h->max_lds);
  const int ctiles = ceil_div(pl->CBP, wt.cbt()) * (fold ? 1 : ceil_div(pl->CAP, wt.cat()));
  const int ntiles = N * pl->tiles_x * pl->tiles_y;
  int nsplit = (per_cu * h->num_cus) / ctiles;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > ntiles) nsplit = ntiles;
  nsplit = ceil_div(ntiles, ceil_div(ntiles, nsplit));      // equal tile counts per workgroup
  pl->nsplit = nsplit;
  pl->grid = dim3(nsplit * ctiles, 1, 1);
  pl->nctiles = ctiles;
  if (RCV_ENV("RCV_DEBUG_PLAN"))
    fprintf(stderr, "wgrad plan: tile %d  R=%d Wt=%d (%d px) IHxIW=%dx%d  ntiles=%d nsplit=%d ctiles=%d lds=%zu\n", pl->tile, pl->R, pl->Wt,
            pl->R * pl->Wt4, pl->IH, pl->IW, ntiles, nsplit, ctiles, pl->lds);
  return RCV_OK;
}

int rcv_launch_wgrad(const rcv_handle* h, const rcv_op* op, hipStream_t s, OpQuery* query) {
  if (op->kind == RCV_OP_WGRAD_REDUCE_BATCH) {
    if (query) { snprintf(query->label, sizeof(query->label), "wgrad_reduce_batch"); query->n_part = 0; query->n_split = 0; query->part_bytes = 0; return RCV_OK; }
    const int njobs = op->i[RCV_I_COUNT], blocks = op->i[RCV_I_NPART];
    RCV_CHECK_ARG(op->p[RCV_P_IN] && njobs >= 1 && njobs <= 64 && blocks >= 1, "wgrad_reduce_batch: needs a job table (1..64 rows) and the block count");
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(blocks), dim3(256), 0, s, (const rcv_reduce_job*)op->p[RCV_P_IN], njobs);
    RCV_HIP(hipGetLastError());
    return RCV_OK;
  }
  if (op->kind == RCV_OP_WGRAD_REDUCE) {
    if (query) { snprintf(query->label, sizeof(query->label), "wgrad_reduce"); query->n_part = 0; query->n_split = 0; query->part_bytes = 0; return RCV_OK; }
    const int CA = op->i[RCV_I_CIN], CB = op->i[RCV_I_COUT], nsplit = op->i[RCV_I_NSPLIT];
    const int CAP = wgrad_cap(CA), CBP = round_up(CB, 16);
    const float* part = (const float*)op->p[RCV_P_PART];
    float* dw = (float*)op->p[RCV_P_OUT];
    float* db = (float*)op->p[RCV_P_BIAS];
    RCV_CHECK_ARG(part && dw && nsplit > 0, "wgrad_reduce: null operand");
    const float* pb = db ? part + (size_t)nsplit * 9 * CBP * CAP : nullptr;
    const int total = 9 * CBP * CAP + (db ? CBP : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(total, 64)), dim3(256), 0, s, part, pb, dw, db, nsplit, CB, CA, CBP, CAP);
    RCV_HIP(hipGetLastError());
    return RCV_OK;
  }
  WPlan pl;
  if (!rcv_plan_get(h, op, &pl)) {
    memset(&pl, 0, sizeof(pl));
    const int rc = wmake_plan(h, op, &pl);
    if (rc) return rc;
    rcv_plan_put(h, op, pl);
  }
  {   // load-mode refusals belong to the shape of the record: in front of the query return (what the planner accepts, the launch accepts)
    const int gm = op->i[RCV_I_INMODE], pm = op->i[RCV_I_INMODE2];
    const bool g2 = gm == RCV_LOAD_GRAD_ENC || gm == RCV_LOAD_GRAD_DEC, p2 = pm == RCV_LOAD_GRAD_ENC || pm == RCV_LOAD_GRAD_DEC;
    RCV_CHECK_ARG(pm != RCV_LOAD_NCHW, "wgrad: pointwise operand cannot be NCHW");
    RCV_CHECK_ARG(!(g2 && p2), "wgrad: at most one operand may be a gradient (two-tensor) load");
  }
  if (query && pl.first) {
    snprintf(query->label, sizeof(query->label), "wgrad_first<%d>", op->i[RCV_I_DIL]);
    query->n_part = 0;
    query->n_split = pl.nsplit;
    query->part_bytes = (size_t)pl.nsplit * (9 * (size_t)pl.CBP * pl.CAP + pl.CBP) * sizeof(float);
    return RCV_OK;
  }
  if (query && pl.bf3) {
    if (pl.bf3 > 0) snprintf(query->label, sizeof(query->label), "wgrad_bf3<%d>", pl.bf3);
    else snprintf(query->label, sizeof(query->label), "wgradn_bf3<%d,%d>", -pl.bf3, op->i[RCV_I_STRIDE]);
    query->n_part = 0;
    query->n_split = pl.nsplit;
    query->part_bytes = (size_t)pl.nsplit * (9 * (size_t)pl.CBP * pl.CAP + pl.CBP) * sizeof(float);
    return RCV_OK;
  }
  if (query) {
    const WTile& wt = kWT[pl.tile];
    snprintf(query->label, sizeof(query->label), "wgrad_mfma<%d,%d,%d,%d,%d,f%d>", wt.WM, wt.WN, wt.WAVES_M, wt.WAVES_N, wt.WAVES_K, wt.NBF);
    query->n_part = 0;
    query->n_split = pl.nsplit;
    query->part_bytes = (size_t)pl.nsplit * (9 * (size_t)pl.CBP * pl.CAP + pl.CBP) * sizeof(float);
    return RCV_OK;
  }
  WgradArgs a;
  a.g = (const float*)op->p[RCV_P_IN]; a.g_aux = (const float*)op->p[RCV_P_IN_AUX]; a.g_c = (const float*)op->p[RCV_P_IN_C];
  a.p = (const float*)op->p[RCV_P_IN2]; a.p_aux = (const float*)op->p[RCV_P_IN2_AUX]; a.p_c = (const float*)op->p[RCV_P_IN2_C];
  a.part = (float*)op->p[RCV_P_PART];
  a.g_mode = op->i[RCV_I_INMODE]; a.p_mode = op->i[RCV_I_INMODE2];
  a.N = op->i[RCV_I_N]; a.H = op->i[RCV_I_H]; a.W = op->i[RCV_I_W]; a.Hp = op->i[RCV_I_HO]; a.Wp = op->i[RCV_I_WO];
  a.CA = op->i[RCV_I_CIN]; a.CB = op->i[RCV_I_COUT]; a.CAP = pl.CAP; a.CBP = pl.CBP;
  a.stride = op->i[RCV_I_STRIDE]; a.dil = op->i[RCV_I_DIL];
  a.R = pl.R; a.Wt = pl.Wt; a.Wt4 = pl.Wt4; a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y;
  a.ntiles = a.N * pl.tiles_x * pl.tiles_y; a.IH = pl.IH; a.IW = pl.IW; a.SP = pl.SP; a.SG = pl.SG;
  a.dbg = op->flags; a.nsplit = pl.nsplit; a.nctiles = pl.nctiles; a.pl_floats = pl.pl_floats; a.gl_floats = pl.gl_floats;
  a.fdWt4 = make_fastdiv(pl.Wt4); a.fdIW = make_fastdiv(pl.IW);
  RCV_CHECK_ARG(a.g && a.p && a.part, "wgrad: null operand");
  RCV_CHECK_ARG(op->i[RCV_I_NSPLIT] == pl.nsplit, "wgrad: workspace splits %d != %d", op->i[RCV_I_NSPLIT], pl.nsplit);
  RCV_CHECK_ARG(a.g_mode == RCV_LOAD_PLAIN || a.g_mode == RCV_LOAD_NCHW || a.g_c, "wgrad: gathered load mode %d needs constants", a.g_mode);
  RCV_CHECK_ARG(a.p_mode == RCV_LOAD_PLAIN || a.p_c, "wgrad: pointwise load mode %d needs constants", a.p_mode);
  const bool g_two = a.g_mode == RCV_LOAD_GRAD_ENC || a.g_mode == RCV_LOAD_GRAD_DEC;
  const bool p_two = a.p_mode == RCV_LOAD_GRAD_ENC || a.p_mode == RCV_LOAD_GRAD_DEC;
  RCV_CHECK_ARG(!g_two || a.g_aux, "wgrad: gathered gradient load needs aux");
  RCV_CHECK_ARG(!p_two || a.p_aux, "wgrad: pointwise gradient load needs aux");
  a.part_bias = (op->flags & RCV_F_BIAS) ? a.part + (size_t)pl.nsplit * 9 * pl.CBP * pl.CAP : nullptr;
  if (pl.first) return wgrad_first_launch(h, a, s);
  if (pl.bf3 < 0) return wgradn_bf3_launch(h, a, -pl.bf3, (int)pl.grid.x, s);
  if (pl.bf3) return wgrad_bf3_launch(h, a, pl.bf3, s);
  switch (pl.tile) {
    case 0: return wlaunch_inst<2, 2, 2, 2, 1, 0, true>(a, g_two, pl.grid, pl.lds, s, h->device);
    case 1: return wlaunch_inst<2, 2, 2, 1, 2, 0, true>(a, g_two, pl.grid, pl.lds, s, h->device);
    case 2: return wlaunch_inst<2, 2, 1, 2, 2, 0, true>(a, g_two, pl.grid, pl.lds, s, h->device);
    case 3: return wlaunch_inst<2, 2, 1, 1, 4, 0, true>(a, g_two, pl.grid, pl.lds, s, h->device);
    case 4: return wlaunch_inst<2, 1, 1, 1, 4, 0, false>(a, g_two, pl.grid, pl.lds, s, h->device);
    case 5: return wlaunch_inst<1, 2, 1, 1, 4, 0, false>(a, g_two, pl.grid, pl.lds, s, h->device);
    case 6: return wlaunch_inst<1, 1, 1, 1, 4, 0, false>(a, g_two, pl.grid, pl.lds, s, h->device);
    case 7: return wlaunch_inst<1, 1, 1, 1, 4, 2, false>(a, g_two, pl.grid, pl.lds, s, h->device);
    default: return wlaunch_inst<1, 1, 1, 1, 4, 5, false>(a, g_two, pl.grid, pl.lds, s, h->device);
  }
}

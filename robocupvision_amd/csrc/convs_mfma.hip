// Narrow-layer 3x3 convolution (few channels, large planes): the HBM-bound end of the ROBO-UNet step.
//
// Same implicit GEMM and MFMA lane mapping as conv_mfma.hip (D[co][pixel] += W[co][k] X[k][pixel] on
// v_mfma_f32_16x16x4_f32), different schedule, because here a tile has only 9..72 k-steps and the
// per-tile fixed costs of the general kernel (re-staging the filter, cross-lane statistics, barriers with
// nothing in flight) dominated the run time:
//   * workgroups are PERSISTENT and walk a strided list of pixel tiles;
//   * the whole packed filter (all taps, all input channels, all output channels) is staged into LDS ONCE;
//   * the input tile of the NEXT pixel tile is fetched into registers (up to XMAX 16-byte loads per thread,
//     twice that for the two-tensor gradient loads) before the MFMA phase of the current tile and written to
//     LDS after it, so every CU keeps >= 64 KiB of HBM reads in flight while it computes and stores;
//   * BatchNorm partial sums are accumulated in registers ACROSS tiles and reduced once per workgroup
//     (one partial row per workgroup instead of one per tile): fixed order, no atomics;
//   * per-tile vector-ALU work is kept minimal (next to MFMAs a wave's other vector instructions advance at about one per MFMA):
//     staging slots and output blocks carry tile-relative byte offsets and packed 16-bit tile coordinates as lane constants, a tile
//     adds scalar bases, interior tiles skip all bounds work, the epilogue variant (statistics x residual x ReLU) is chosen per tile.
// A workgroup is 4 waves side by side along the pixels; every wave covers all (virtual) output channels
// (WM blocks of 16) for its WN blocks of 16 pixels.  Kinds: KIND_GATHER (stride 1|2, dilation 1|2) and
// KIND_TMERGED (transposed conv, merged-parity layout, see conv_mfma.hip).
#include <stdlib.h>
#include <type_traits>
#include "conv_common.h"

// 16-byte accesses at a 32-bit byte offset from a uniform base: the address is (scalar base + vector offset), no 64-bit vector
// arithmetic per access (the host keeps every tensor this kernel touches below 4 GiB)
__device__ __forceinline__ float4 ld4b(const float* base, uint32_t byte_off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void st4b(float* base, uint32_t byte_off, float4 v) {
  *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

#ifdef RCV_STAMPS
// diagnostic build only (make STAMPS=1): shader-clock stamps around the phases of the tile loop, summed per wave
// (scripts/bench_op.py --stamps 1 prints them)
#define RCVS_STAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define RCVS_SEG(k) do { RCVS_STAMP(st_b); st_seg[k] += st_b - st_a; st_a = st_b; } while (0)
#else
#define RCVS_SEG(k) do { } while (0)
#endif

// Per-tile scalars of the persistent loop (all uniform: they live in SGPRs)
struct NarrowTile {
  int n, oy0, ox0;          // image, first staged input row / column (may be negative: padding)
  uint32_t obase;           // byte offset of the tile's first output pixel (channel 0)
  uint32_t lim;             // (output rows of the tile inside the plane - 1) << 16 | (output columns inside the plane - 1)
  bool interior;            // the whole staged input tile lies inside the plane: no padding, no per-pixel bounds work
};

template <int KIND>
__device__ __forceinline__ NarrowTile narrow_decode(const ConvArgs& a, int t) {
  NarrowTile ti;
  const int q1 = fd_div(t, a.fdTX), tx_i = t - q1 * a.tiles_x;
  ti.n = fd_div(q1, a.fdTY);
  const int ty_i = q1 - ti.n * a.tiles_y;
  const int y0 = ty_i * a.R, x0 = tx_i * a.Wt;
  if (KIND == KIND_GATHER) {
    ti.oy0 = y0 * a.stride - a.dil; ti.ox0 = x0 * a.stride - a.dil;
    ti.obase = (uint32_t)(((ti.n * a.Ho + y0) * a.Wo + x0) * a.Cout) * 4u;
    ti.lim = (uint32_t)(a.Ho - y0 - 1) << 16 | (uint32_t)(a.Wo - x0 - 1);
  } else {
    ti.oy0 = y0; ti.ox0 = x0;
    ti.obase = (uint32_t)(((ti.n * a.Ho + 2 * y0) * a.Wo + 2 * x0) * a.Cout) * 4u;
    ti.lim = (uint32_t)(a.H - y0 - 1) << 16 | (uint32_t)(a.W - x0 - 1);
  }
  ti.interior = ti.oy0 >= 0 && ti.ox0 >= 0 && ti.oy0 + a.IH <= a.H && ti.ox0 + a.IW <= a.W;
  return ti;
}

// PAD: the tile touches the border of the plane (slots outside it are written as zeros).  A template parameter and a uniform branch at
// the call, not a run-time flag in here: as a flag the compiler turns it into two more selects per element on EVERY tile (10 of the
// 14 vector instructions of an interior slot).
template <int MODE, int XMAX, int AMAX, int CK, bool PAD>
__device__ __forceinline__ void narrow_write_x(const ConvArgs& a, float* xl, const float* cl, const float4 (&px)[XMAX],
                                               const float4 (&pa)[AMAX], uint32_t okmask, int tid) {
  constexpr int Q = CK / 4, STEP = 256 / Q;
  const int S = a.xpitch;
  const int q = tid % Q, lpix = tid / Q;
  const int npix = a.IH * a.IW;
  float4 k[5];
  if (MODE != RCV_LOAD_PLAIN && MODE != RCV_LOAD_NCHW) {
#pragma unroll
    for (int j = 0; j < 5; ++j) k[j] = *reinterpret_cast<const float4*>(cl + j * a.Cin + 4 * q);
  }
  float* d0 = xl + lpix * S + 4 * q;
#pragma unroll
  for (int u = 0; u < XMAX; ++u) {
    if (lpix < npix - u * STEP) {                // (right side uniform)
      float4 v = xform4<MODE>(px[u], pa[AMAX == XMAX ? u : 0], k);
      if (PAD && !((okmask >> u) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);      // zero padding AFTER the transform
      float* d = d0 + u * STEP * S;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  }
}

template <int WM, int WN, int CK, int KIND, int XMAX, bool TWO>
__global__ __launch_bounds__(256, 2) void convs_mfma_kernel(const ConvArgs a) {
  constexpr int NT = 256, WAVES = 4;
  constexpr int COT = WM * 16;
  const int S = a.xpitch;
  constexpr int WS = COT + 16;
  constexpr int Q = CK / 4;
  constexpr int STEP = NT / Q;
  constexpr int C4 = COT / 4;
  constexpr int NTAPS = KIND == KIND_GATHER ? 9 : 4;
  constexpr int NXT = KIND == KIND_GATHER ? 3 : 2;
  constexpr int AMAX = TWO ? XMAX : 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;                        // [NTAPS][CK][WS]
  float* xl = smem + a.wl_floats;          // [IH*IW][S]
  float* cl = xl + a.xl_floats;            // [5][Cin]
  float* red = cl + 5 * a.CinP + 16;       // [WAVES][2][COT]
  float* ec = red + WAVES * 2 * COT;       // [4][COT]: bias, epilogue scale, epilogue shift, batch mean per virtual output channel

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int npix = a.IH * a.IW;
  // (an NCHW image has at most 4 channels here and is never a gradient operand: the other instantiations do not carry that path --
  // compiled into all of them it cost every one-tensor variant ~6 registers and 8 spilled SGPRs: 1-2 % of the whole step)
  const bool nchw = CK == 4 && !TWO && KIND == KIND_GATHER && a.in_mode == RCV_LOAD_NCHW;
  const bool need_e = a.stats == RCV_STATS_BWD_ENC || a.stats == RCV_STATS_BWD_DEC;

  // ---- once per workgroup: load constants, epilogue constants and the whole filter
  if (a.in_c && a.in_mode != RCV_LOAD_PLAIN && !nchw)
    for (int e = tid; e < 5 * a.Cin; e += NT) cl[e] = a.in_c[e];
  for (int e = tid; e < 4 * COT; e += NT) {
    const int which = e / COT, cov = e - which * COT;
    const int co = KIND == KIND_TMERGED ? cov % a.Cout : cov;
    float v = 0.f;
    if (cov < a.CoutV) {
      if (which == 0 && (a.flags & RCV_F_BIAS)) v = a.bias[co];
      if (which == 1 && a.stats == RCV_STATS_BWD_DEC) v = a.epi_c[co];
      if (which == 2 && a.stats == RCV_STATS_BWD_DEC) v = a.epi_c[a.Cout + co];
      if (which == 3 && need_e) v = a.epi_c[2 * a.Cout + co];
    }
    ec[e] = v;
  }
  for (int e = tid; e < NTAPS * CK * C4; e += NT) {
    const int row = e / C4, c4 = e % C4;
    const int j = row / CK, ck = row % CK;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (4 * c4 < a.CoutP && ck < a.CinP) v = ld4(a.w + ((size_t)(j * a.CinP + ck) * a.CoutP + 4 * c4));
    *reinterpret_cast<float4*>(wl + row * WS + 4 * c4) = v;
  }

  // ---- lane constants of the staging slots (the same for every tile): position inside the staged tile, byte offset relative to the
  // tile's first input pixel.  Slots past the tile sit at row 32767: never inside the plane
  // (the host keeps H below 16384 for this kernel).
  const int q = tid % Q, lpix = tid / Q;
  uint32_t sxy[XMAX];                      // row << 16 | column
  uint32_t rel4[XMAX];
  uint32_t slotmask = 0;
#pragma unroll
  for (int u = 0; u < XMAX; ++u) {
    const int pix = lpix + u * STEP;
    const int pixc = pix < npix ? pix : npix - 1;
    const int iy = fd_div(pixc, a.fdIW), ix = pixc - iy * a.IW;
    sxy[u] = (uint32_t)(pix < npix ? iy : 0x7fff) << 16 | (uint32_t)ix;
    rel4[u] = (uint32_t)((iy * a.W + ix) * a.Cin + 4 * q) * 4u;
    slotmask |= (pix < npix ? 1u : 0u) << u;
  }

  // ---- prefetch registers
  float4 px[XMAX], pa[AMAX];
  uint32_t okmask = 0;
  auto prefetch = [&](const NarrowTile& ti) {
    if (nchw) {
      // quad q of a pixel = planes 4q .. 4q+3 of the image (one quad: <= 4 channels).  Branch-free like the NHWC path: clamped coordinates and plane indices, validity in okmask / the channel masks.
      okmask = 0;
      const uint32_t plane = (uint32_t)(a.H * a.W);
      const int c0 = 4 * q;
      uint32_t cpl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) cpl[j] = (uint32_t)(ti.n * a.Cin + (c0 + j < a.Cin ? c0 + j : a.Cin - 1)) * plane;
#pragma unroll
      for (int u = 0; u < XMAX; ++u) {
        const int gy = ti.oy0 + (int)(sxy[u] >> 16), gx = ti.ox0 + (int)(sxy[u] & 0xffffu);
        const bool inside = Q <= 2 && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        okmask |= (inside ? 1u : 0u) << u;
        const uint32_t pos = inside ? (uint32_t)(gy * a.W + gx) : 0u;
        const float v0 = a.in[cpl[0] + pos], v1 = a.in[cpl[1] + pos], v2 = a.in[cpl[2] + pos], v3 = a.in[cpl[3] + pos];
        px[u] = make_float4(v0, c0 + 1 < a.Cin ? v1 : 0.f, c0 + 2 < a.Cin ? v2 : 0.f, c0 + 3 < a.Cin ? v3 : 0.f);
      }
    } else {
      // Branch-free loads (behind divergent branches the compiler waits for each load on the spot and nothing stays in flight across
      // the contraction): one address add per slot on interior tiles, bounds tests and a select on the tiles that touch the border.
      // A slot outside the plane reads byte 0 of the tensor and is zeroed when it is written to LDS.
      const uint32_t tb4 = (uint32_t)(((ti.n * a.H + ti.oy0) * a.W + ti.ox0) * a.Cin) * 4u;
      uint32_t bo[XMAX];
      if (ti.interior) {
        okmask = slotmask;
#pragma unroll
        for (int u = 0; u < XMAX; ++u) bo[u] = tb4 + rel4[u];
      } else {
        okmask = 0;
#pragma unroll
        for (int u = 0; u < XMAX; ++u) {
          const bool inside = (unsigned)(ti.oy0 + (int)(sxy[u] >> 16)) < (unsigned)a.H && (unsigned)(ti.ox0 + (int)(sxy[u] & 0xffffu)) < (unsigned)a.W;
          okmask |= (inside ? 1u : 0u) << u;
          bo[u] = inside ? tb4 + rel4[u] : 0u;
        }
      }
#pragma unroll
      for (int u = 0; u < XMAX; ++u) {
        px[u] = ld4b(a.in, bo[u]);
        if (TWO) pa[TWO ? u : 0] = ld4b(a.in_aux, bo[u]);
      }
    }
  };

  float s1[WM][4], s2[WM][4];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[m][r] = 0.f; s2[m][r] = 0.f; }

  const int IS = KIND == KIND_GATHER ? a.stride : 1;
  const bool skip16 = KIND == KIND_TMERGED && WM == 4 && a.Cout == 16 && !(a.flags & RCV_F_DBG_NOSKIP);
  const bool skip8 = KIND == KIND_TMERGED && WM == 2 && a.Cout == 8 && !(a.flags & RCV_F_DBG_NOSKIP);
  const int aoff = l4 * WS + l15;

  // ---- lane constants of the output pixel blocks: LDS offset of the block's pixel, its row / column inside the tile (row 65535 for the
  // slots past the tile: never stored) and its byte offset relative to the tile's first output pixel
  int pixoff[WN];
  uint32_t tyx[WN];                        // row << 16 | column
  uint32_t ob[WN];
#pragma unroll
  for (int b = 0; b < WN; ++b) {
    const int p = (wave * WN + b) * 16 + l15;
    int ty = fd_div(p, a.fdWt), tx = p - ty * a.Wt;
    const bool used = ty < a.R;
    if (!used) { ty = 0; tx = 0; }
    pixoff[b] = ((ty * IS) * a.IW + tx * IS) * S + l4;
    tyx[b] = (uint32_t)(used ? ty : 0xffff) << 16 | (uint32_t)tx;
    ob[b] = (uint32_t)((KIND == KIND_GATHER ? ty * a.Wo + tx : 2 * (ty * a.Wo + tx)) * a.Cout) * 4u;
  }
  uint32_t om[WM];
  bool cok[WM];
#pragma unroll
  for (int m = 0; m < WM; ++m) {
    const int cov = m * 16 + 4 * l4;
    cok[m] = cov < a.CoutV;
    if (KIND == KIND_TMERGED) {
      const int ph = cov / a.Cout, co = cov - ph * a.Cout;
      om[m] = (uint32_t)((((ph >> 1) * a.Wo + (ph & 1)) * a.Cout) + co) * 4u;
    } else {
      om[m] = (uint32_t)cov * 4u;
    }
  }

#ifdef RCV_STAMPS
  unsigned long long st_k0 = 0, st_a = 0, st_b = 0, st_seg[6] = {0, 0, 0, 0, 0, 0}, st_loop0 = 0, st_loop1 = 0;
  const unsigned long long st_rt0 = __builtin_amdgcn_s_memrealtime();
  RCVS_STAMP(st_k0);
#endif
  int tile = xcd_remap(blockIdx.x, gridDim.x);     // neighbouring tiles (shared halo rows) stay on one XCD
  NarrowTile cur = narrow_decode<KIND>(a, tile < a.total_tiles ? tile : 0);
  if (tile < a.total_tiles) prefetch(cur);

#ifdef RCV_STAMPS
  RCVS_STAMP(st_loop0); st_a = st_loop0;
#endif
  while (tile < a.total_tiles) {
    // every wave is done reading the previous tile (and the filter is in place).  Bare barriers in this loop: the fence of
    // __syncthreads() would drain vmcnt, i.e. wait for the prefetch (and for the previous tile's output stores).
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    RCVS_SEG(0);
    auto write_x = [&](auto pad_c) {
      constexpr bool PAD = decltype(pad_c)::value;
      if (TWO) {
        if (a.in_mode == RCV_LOAD_GRAD_ENC) narrow_write_x<RCV_LOAD_GRAD_ENC, XMAX, AMAX, CK, PAD>(a, xl, cl, px, pa, okmask, tid);
        else narrow_write_x<RCV_LOAD_GRAD_DEC, XMAX, AMAX, CK, PAD>(a, xl, cl, px, pa, okmask, tid);
      } else {
        switch (a.in_mode) {
          case RCV_LOAD_PLAIN: narrow_write_x<RCV_LOAD_PLAIN, XMAX, AMAX, CK, PAD>(a, xl, cl, px, pa, okmask, tid); break;
          case RCV_LOAD_AFFINE: narrow_write_x<RCV_LOAD_AFFINE, XMAX, AMAX, CK, PAD>(a, xl, cl, px, pa, okmask, tid); break;
          case RCV_LOAD_AFFINE_RELU: narrow_write_x<RCV_LOAD_AFFINE_RELU, XMAX, AMAX, CK, PAD>(a, xl, cl, px, pa, okmask, tid); break;
          default: narrow_write_x<RCV_LOAD_NCHW, XMAX, AMAX, CK, true>(a, xl, cl, px, pa, okmask, tid); break;
        }
      }
    };
    if (a.flags & RCV_F_DBG_NOSTAGE) {
      // profiling ablation: no LDS writes (the prefetch loads below are still issued: they are unconditional)
    } else if (cur.interior && !nchw) {
      write_x(std::integral_constant<bool, false>{});
    } else {
      write_x(std::integral_constant<bool, true>{});
    }
    RCVS_SEG(1);
    const int ntile = tile + gridDim.x;
    // next tile's loads: in flight during the contraction and the stores below.  Unconditional (the last iteration re-requests its
    // own tile, L2 hits): behind a branch the compiler parks a vmcnt(0) in front of the contraction where the two paths meet.
    const NarrowTile nxt = narrow_decode<KIND>(a, ntile < a.total_tiles ? ntile : tile);
    prefetch(nxt);
    RCVS_SEG(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    RCVS_SEG(3);

    // ---- contraction
    if (a.flags & RCV_F_DBG_NOMFMA) { tile = ntile; cur = nxt; continue; }      // profiling ablation
    // the tile's limits against the lane constants of the pixel blocks: which blocks this tile stores
    bool pv[WN];
#pragma unroll
    for (int b = 0; b < WN; ++b) {       // row < rlim && column < wlim as one packed 16-bit minimum and one compare
      const u16x2 t = __builtin_bit_cast(u16x2, tyx[b]);
      pv[b] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(t, __builtin_bit_cast(u16x2, cur.lim))) == tyx[b];
    }
    // The statistics operand of the gradient variants (one 16-byte piece of the saved activation per output pixel and lane) is
    // requested HERE, before the contraction, for the first EROWS channel blocks: requested in the epilogue it put a full HBM round
    // trip per block group on the critical path of every tile (two waves per SIMD cannot hide it), which made the gradient variants
    // 1.5-2x slower than the forward ones.  Branch-free: a block that is not stored reads byte 0.
    constexpr int EROWS = !TWO ? 0 : (WM * WN <= 6 ? WM : (WM * WN < 10 ? 1 : 0));      // (the widest tiles have no registers left for it)
    float4 ee_early[EROWS ? EROWS : 1][WN];
    if (TWO && need_e) {
#pragma unroll
      for (int m = 0; m < EROWS; ++m)
#pragma unroll
        for (int b = 0; b < WN; ++b) ee_early[m][b] = ld4b(a.epi_aux, (pv[b] && cok[m]) ? cur.obase + om[m] + ob[b] : 0u);
    }
    constexpr bool TIGHT = TWO || (KIND == KIND_GATHER && WM * WN >= 10);
    f32x4 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int b = 0; b < WN; ++b) acc[m][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // wave priority: the MFMA phase yields to the staging / epilogue phases of the other wave on the SIMD (whose vector instructions
    // otherwise queue behind the back-to-back MFMAs): +2...7 % on the layers with short contractions, neutral elsewhere
    __builtin_amdgcn_s_setprio(0);
    // SC: LDS pitch of a staged pixel as a compile-time constant (dilation 1): the tap's column offset and the k-step are then immediate
    // offsets of the LDS reads and a pixel block needs one address register per tap ROW instead of one per tap (45 -> 15 registers for a
    // 3x3 filter and five blocks); SC = 0: run-time pitch and dilation
    auto contract = [&](auto sc_c) {
      constexpr int SC = decltype(sc_c)::value;
      // (TIGHT variants: opaque copies -- the compiler would otherwise hoist every (tap, block) address of every variant of this loop
      // out of the tile loop, 45 + 15 + 15 registers; two dozen adds per tile are the cheaper side of that trade there)
      int po[WN];
#pragma unroll
      for (int b = 0; b < WN; ++b) { po[b] = pixoff[b]; if (TIGHT) asm volatile("" : "+v"(po[b])); }
#pragma unroll
      for (int j = 0; j < NTAPS; ++j) {
        const int jy = j / NXT, jx = j % NXT;
        const int dy = (KIND == KIND_GATHER && SC == 0) ? jy * a.dil : jy, dx = (KIND == KIND_GATHER && SC == 0) ? jx * a.dil : jx;
        const float* wj = wl + j * CK * WS + aoff;
        const float* xj = SC ? xl + dy * a.IW * SC + jx * SC : xl + (dy * a.IW + dx) * S;
#pragma unroll
        for (int kk = 0; kk < CK / 4; ++kk) {
          float av[WM], bv[WN];
#pragma unroll
          for (int m = 0; m < WM; ++m) av[m] = wj[kk * 4 * WS + m * 16];
#pragma unroll
          for (int b = 0; b < WN; ++b) bv[b] = xj[po[b] + kk * 4];
#pragma unroll
          for (int m = 0; m < WM; ++m) {
            // merged transposed conv: the filter block of output parity (py,px) is structurally zero for the window taps with
            // dy > py or dx > px (7 of the 16 (parity, tap) pairs).  Where a 16-channel block holds whole parities the MFMAs on
            // those blocks are skipped: Cout = 16 -> block m is parity m; Cout = 8 -> block m holds both px of py = m.
            if (KIND == KIND_TMERGED && ((skip16 && (jy > (m >> 1) || jx > (m & 1))) || (skip8 && jy > m))) continue;
#pragma unroll
            for (int b = 0; b < WN; ++b)
              acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[b], acc[m][b], 0, 0, 0);
          }
        }
      }
    };
    if constexpr (TIGHT) {
      // the gradient variants need the registers (both operand tensors of the next tile and the statistics operand are in flight),
      // and so do the 3x3 variants with ten accumulator blocks
      const bool unit_dil = KIND != KIND_GATHER || a.dil == 1;
      if (unit_dil && S == CK + 2) contract(std::integral_constant<int, CK + 2>{});
      else if (unit_dil && S == CK + 1) contract(std::integral_constant<int, CK + 1>{});
      else contract(std::integral_constant<int, 0>{});
    } else {
      // the other one-tensor variants: register room to spare, every (tap, block) address stays hoisted (no address arithmetic per tile)
      contract(std::integral_constant<int, 0>{});
    }

    __builtin_amdgcn_s_setprio(3);
    RCVS_SEG(4);
    // ---- stores + statistics of this tile (statistics stay in registers).  Per block: two compares against the tile's limits, one
    // address add; everything else about a block's position is a lane constant.  The statistics kind, the residual flag and the ReLU flag select one
    // of sixteen straight-line copies of the epilogue per TILE (uniform branch): as run-time tests inside the block loop they cost a
    // register shuffle per block where the variants met again.
    auto epilogue = [&](auto stats_c, auto resid_c, auto relu_c) {
      constexpr int STATS = decltype(stats_c)::value;
      constexpr bool RESID = decltype(resid_c)::value;
      constexpr bool RELU = decltype(relu_c)::value;
      constexpr bool NEED_E = STATS == RCV_STATS_BWD_ENC || STATS == RCV_STATS_BWD_DEC;
      auto row = [&](auto m_c) {
        constexpr int m = decltype(m_c)::value;
        constexpr bool EARLY = TWO && NEED_E && m < EROWS;        // statistics operand already requested before the contraction
        const float4 bias = *reinterpret_cast<const float4*>(ec + 0 * COT + m * 16 + 4 * l4);
        float4 e0 = bias, e1 = bias, mu = bias;
        if (STATS == RCV_STATS_BWD_DEC) {
          e0 = *reinterpret_cast<const float4*>(ec + 1 * COT + m * 16 + 4 * l4);
          e1 = *reinterpret_cast<const float4*>(ec + 2 * COT + m * 16 + 4 * l4);
        }
        if (NEED_E) mu = *reinterpret_cast<const float4*>(ec + 3 * COT + m * 16 + 4 * l4);
        const uint32_t omt = cur.obase + om[m];
        // first the loads of the skip gradient / statistics operand of (a group of) the pixel blocks of this co-block (independent, in
        // flight together), then the arithmetic and the stores.  Groups of at most BG blocks: with all five of the widest two-tensor
        // tiles in one batch the kernel needed > 256 registers and the compiler parked part of the PREFETCHED input tile in scratch,
        // i.e. waited for those loads right after issuing them
        constexpr int BG = (!TWO || EARLY) ? WN : (WN > 3 ? 3 : (WM >= 4 ? 2 : WN));
#pragma unroll
        for (int b0 = 0; b0 < WN; b0 += BG) {
          float4 rr[BG], ee[BG];
          if (RESID || (NEED_E && !EARLY)) {
#pragma unroll
            for (int bb = 0; bb < BG; ++bb) {
              const int b = b0 + bb;
              if (b < WN && pv[b] && cok[m]) {
                if (RESID) rr[bb] = ld4b(a.resid, omt + ob[b]);
                if (NEED_E && !EARLY) ee[bb] = ld4b(a.epi_aux, omt + ob[b]);
              }
            }
          }
#pragma unroll
          for (int bb = 0; bb < BG; ++bb) {
            const int b = b0 + bb;
            if (b >= WN || !(pv[b] && cok[m])) continue;
            float4 v = make_float4(acc[m][b][0] + bias.x, acc[m][b][1] + bias.y, acc[m][b][2] + bias.z, acc[m][b][3] + bias.w);
            if (RELU) {     // one v_med3_f32 per element (fmaxf costs a canonicalising v_max first); NaN -> 0 like fmaxf(NaN, 0)
              const float inf = __builtin_inff();
              v.x = __builtin_amdgcn_fmed3f(v.x, 0.f, inf); v.y = __builtin_amdgcn_fmed3f(v.y, 0.f, inf);
              v.z = __builtin_amdgcn_fmed3f(v.z, 0.f, inf); v.w = __builtin_amdgcn_fmed3f(v.w, 0.f, inf);
            }
            if (RESID) { v.x += rr[bb].x; v.y += rr[bb].y; v.z += rr[bb].z; v.w += rr[bb].w; }
            st4b(a.out, omt + ob[b], v);
            float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
            if (NEED_E) e = EARLY ? ee_early[EARLY ? m : 0][b] : ee[bb];
            if (STATS == RCV_STATS_FWD) {
              s1[m][0] += v.x; s1[m][1] += v.y; s1[m][2] += v.z; s1[m][3] += v.w;
              s2[m][0] = fmaf(v.x, v.x, s2[m][0]); s2[m][1] = fmaf(v.y, v.y, s2[m][1]);
              s2[m][2] = fmaf(v.z, v.z, s2[m][2]); s2[m][3] = fmaf(v.w, v.w, s2[m][3]);
            } else if (STATS == RCV_STATS_BWD_ENC) {
              s1[m][0] += v.x; s1[m][1] += v.y; s1[m][2] += v.z; s1[m][3] += v.w;
              s2[m][0] = fmaf(v.x, e.x - mu.x, s2[m][0]); s2[m][1] = fmaf(v.y, e.y - mu.y, s2[m][1]);
              s2[m][2] = fmaf(v.z, e.z - mu.z, s2[m][2]); s2[m][3] = fmaf(v.w, e.w - mu.w, s2[m][3]);
            } else if (STATS == RCV_STATS_BWD_DEC) {
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
      };
      row(std::integral_constant<int, 0>{});
      if constexpr (WM > 1) row(std::integral_constant<int, 1>{});
      if constexpr (WM > 2) { row(std::integral_constant<int, 2>{}); row(std::integral_constant<int, 3>{}); }
    };
    auto epilogue_f = [&](auto stats_c, auto resid_c) {
      if (a.flags & RCV_F_RELU) epilogue(stats_c, resid_c, std::integral_constant<bool, true>{});
      else epilogue(stats_c, resid_c, std::integral_constant<bool, false>{});
    };
    auto epilogue_r = [&](auto stats_c) {
      if (a.flags & RCV_F_RESID) epilogue_f(stats_c, std::integral_constant<bool, true>{});
      else epilogue_f(stats_c, std::integral_constant<bool, false>{});
    };
    switch (a.stats) {
      case RCV_STATS_FWD: epilogue_r(std::integral_constant<int, RCV_STATS_FWD>{}); break;
      case RCV_STATS_BWD_ENC: epilogue_r(std::integral_constant<int, RCV_STATS_BWD_ENC>{}); break;
      case RCV_STATS_BWD_DEC: epilogue_r(std::integral_constant<int, RCV_STATS_BWD_DEC>{}); break;
      default: epilogue_r(std::integral_constant<int, RCV_STATS_NONE>{}); break;
    }
    RCVS_SEG(5);
    tile = ntile;
    cur = nxt;
  }
#ifdef RCV_STAMPS
  RCVS_STAMP(st_loop1);
#endif

  // ---- one statistics row per workgroup
  if (a.stats != RCV_STATS_NONE) {
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float u = s1[m][r], v = s2[m][r];
#pragma unroll
        for (int sh = 1; sh < 16; sh <<= 1) { u += __shfl_xor(u, sh); v += __shfl_xor(v, sh); }
        s1[m][r] = u; s2[m][r] = v;
      }
    __syncthreads();
    if (l15 == 0) {
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cl_ = m * 16 + 4 * l4 + r;
          red[(wave * 2 + 0) * COT + cl_] = s1[m][r];
          red[(wave * 2 + 1) * COT + cl_] = s2[m][r];
        }
    }
    __syncthreads();
    for (int e = tid; e < 2 * a.Cout; e += NT) {
      const int which = e / a.Cout, co = e - which * a.Cout;
      float u = 0.f;
      for (int cv = co; cv < a.CoutV; cv += a.Cout) {   // merged layout: the four parity groups of a real channel
#pragma unroll
        for (int wn = 0; wn < WAVES; ++wn) u += red[(wn * 2 + which) * COT + cv];
      }
      a.part[((size_t)blockIdx.x * 2 + which) * a.Cout + co] = u;
    }
  }
#ifdef RCV_STAMPS
  if (a.stamps) {
    unsigned long long st_end;
    RCVS_STAMP(st_end);
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      unsigned long long* o = a.stamps + ((size_t)blockIdx.x * WAVES + wave) * 12;
      o[0] = st_loop0 - st_k0; o[1] = st_loop1 - st_loop0; o[2] = st_end - st_loop1;
      for (int k2 = 0; k2 < 6; ++k2) o[3 + k2] = st_seg[k2];
      o[9] = rt1 - st_rt0; o[10] = st_end - st_k0; o[11] = st_rt0;
    }
  }
#endif
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
static const int kNarrowXMAX = 8;
// pixel blocks per wave (WN) the kernel is instantiated for, per staged channel count: small WN for layers whose tile is
// capped by the prefetch registers (stride-2 and 32-channel inputs), so that no MFMA slot runs on padding pixels
static const int kNarrowWNs[] = {5, 3, 2};
static inline bool narrow_wn_built(int WN, int CK, int WM) {
  if (WM == 4) return WN == 3 && CK >= 16;       // merged transposed conv with 16 real output channels (WN = 5 measured slower: 80 accumulator registers)
  return WN == 5 || (WN == 3 && CK >= 8) || (WN == 2 && CK == 16);
}

bool convs_supported(const rcv_handle* h, const rcv_op* op, int kind, int CinP, int CoutV) {
  (void)h;
  if (RCV_ENV("RCV_NO_NARROW")) return false;
  if (op->i[RCV_I_H] >= 16384 || op->i[RCV_I_W] >= 16384 || op->i[RCV_I_HO] >= 16384 || op->i[RCV_I_WO] >= 16384) return false;   // packed 16-bit tile coordinates
  // 32-bit byte offsets from the tensor bases: input and output stay below 4 GiB
  if ((long long)op->i[RCV_I_N] * op->i[RCV_I_H] * op->i[RCV_I_W] * op->i[RCV_I_CIN] >= (1ll << 30)) return false;
  if ((long long)op->i[RCV_I_N] * op->i[RCV_I_HO] * op->i[RCV_I_WO] * op->i[RCV_I_COUT] >= (1ll << 30)) return false;
  if (kind == KIND_TPHASE) return false;
  if (!(CinP == 4 || CinP == 8 || CinP == 16 || CinP == 32)) return false;
  if (kind == KIND_TMERGED && (CinP == 16 || CinP == 32) && round_up(CoutV, 16) == 64 && !RCV_ENV("RCV_NO_NARROW4")) return true;   // 4 x 16 virtual channels
  return round_up(CoutV, 16) <= 32;
}

int convs_make_plan(const rcv_handle* h, const rcv_op* op, int kind, ConvPlan* pl) {
  const int N = op->i[RCV_I_N], H = op->i[RCV_I_H], W = op->i[RCV_I_W];
  const int Cin = op->i[RCV_I_CIN], Cout = op->i[RCV_I_COUT];
  const int Ho = op->i[RCV_I_HO], Wo = op->i[RCV_I_WO];
  const int s = op->i[RCV_I_STRIDE], d = op->i[RCV_I_DIL];
  const int CinP = round_up(Cin, 4);
  pl->kind = kind; pl->narrow = 1; pl->dma = 0; pl->first = 0;
  pl->CK = CinP;
  pl->CoutV = kind == KIND_TMERGED ? 4 * Cout : Cout;
  pl->CoutP = round_up(pl->CoutV, 16);
  pl->WM = pl->CoutP / 16; pl->XMAX = kNarrowXMAX;
  const int Q = CinP / 4;
  const int cap = (256 / Q) * pl->XMAX;
  const int TH = kind == KIND_GATHER ? Ho : H, TW = kind == KIND_GATHER ? Wo : W;
  const int ntaps = kind == KIND_GATHER ? 9 : 4;
  // tile: rows x cols within 64*WN pixel slots and the prefetch capacity.  Cost model per tile (SIMD cycles): the MFMA phase
  // (every slot costs, used or not), staging the halo, a fixed part (barriers, prefetch issue, epilogue address work).
  double best = -1.0;
  int ovR = 0, ovW = 0, ovWN = 0;
  if (const char* ev = RCV_ENV("RCV_CONVS_TILE")) sscanf(ev, "%d,%d,%d", &ovR, &ovW, &ovWN);
  for (int WN : kNarrowWNs) {
    if (!narrow_wn_built(WN, CinP, pl->WM) || (ovWN > 0 && WN != ovWN)) continue;
    // (Round 3, scripts/experiments/sweep_tiles.py: op by op the three-block tile measured 7 % faster than the five-block one on the merged
    // 16 -> 8 transposed conv at 32x240x320, forward and data gradient; inside the two-stream step the same change measured +1.1 % on
    // the STEP (6.35 -> 6.42 ms) -- what an isolated op wins it takes from the filter gradient running beside it.  Not applied.)
    const int PIX = 64 * WN;
    for (int nx = 1; nx <= TW; ++nx) {
      const int wt = ceil_div(TW, nx);
      if (wt > PIX) continue;
      if (nx > 1 && wt < 8) break;
      for (int r = PIX / wt < TH ? PIX / wt : TH; r >= 1; --r) {
        int ih, iw;
        tile_halo(kind, r, wt, s, d, &ih, &iw);
        if (ih * iw > cap) continue;
        if (ovR > 0 && !(r == ovR && wt == ovW)) continue;
        const int rb = ceil_div(TH, ceil_div(TH, r));
        tile_halo(kind, rb, wt, s, d, &ih, &iw);
        const double tiles = (double)ceil_div(TW, wt) * ceil_div(TH, rb);
        const double mfma = 32.0 * WN * pl->WM * ntaps * (CinP / 4);
        const double stage = 24.0 * ceil_div(ih * iw * Q, 256);
        const double score = tiles * (mfma + stage + 1200.0 + 60.0 * WN * pl->WM);
        if (best < 0 || score < best) { best = score; pl->R = rb; pl->Wt = wt; pl->WN = WN; }
        break;
      }
    }
  }
  RCV_CHECK_ARG(best >= 0, "conv (narrow): no tile fits %dx%d", TH, TW);
  pl->tiles_x = ceil_div(TW, pl->Wt); pl->tiles_y = ceil_div(TH, pl->R);
  tile_halo(kind, pl->R, pl->Wt, s, d, &pl->IH, &pl->IW);
  const int S = conv_xpitch(CinP, kind == KIND_GATHER ? s : 1);
  pl->wl_floats = round_up(ntaps * CinP * (pl->CoutP + 16), 4);
  pl->xl_floats = round_up(pl->IH * pl->IW * S, 4);
  const size_t floats = (size_t)pl->wl_floats + pl->xl_floats + 5 * CinP + 16 + (size_t)4 * 2 * pl->CoutP + (size_t)4 * pl->CoutP;
  pl->lds = floats * sizeof(float);
  RCV_CHECK_ARG(pl->lds <= (size_t)h->max_lds, "conv (narrow): tile needs %zu B of LDS", pl->lds);
  pl->n_co_tiles = 1; pl->n_phases = 1;
  pl->total_tiles = N * pl->tiles_x * pl->tiles_y;
  int occ = (int)((size_t)h->max_lds / pl->lds);
  if (occ > 2) occ = 2;                       // launch bounds: 2 waves per SIMD
  if (occ < 1) occ = 1;
  if (const char* ev = RCV_ENV("RCV_CONVS_OCC")) { const int o = atoi(ev); if (o >= 1 && o <= 4) occ = o; }
  const int resident = h->num_cus * occ;
  const int per = ceil_div(pl->total_tiles, resident);
  pl->grid = ceil_div(pl->total_tiles, per);
  return RCV_OK;
}

template <int WM, int WN, int CK, int KIND, bool TWO>
static int convs_launch_inst(const ConvPlan& pl, const ConvArgs& a, hipStream_t s) {
  auto kern = convs_mfma_kernel<WM, WN, CK, KIND, kNarrowXMAX, TWO>;
  static size_t configured[RCV_MAX_DEVICES];
  RCV_ENSURE_LDS(kern, pl.lds, pl.dev, configured);
  hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(256), pl.lds, s, a);
  RCV_HIP(hipGetLastError());
  return RCV_OK;
}

template <int WM, int WN, int CK, int KIND>
static int convs_launch_two(const ConvPlan& pl, const ConvArgs& a, bool two, hipStream_t s) {
  return two ? convs_launch_inst<WM, WN, CK, KIND, true>(pl, a, s) : convs_launch_inst<WM, WN, CK, KIND, false>(pl, a, s);
}

template <int WM, int KIND>
static int convs_launch_ck(const ConvPlan& pl, const ConvArgs& a, bool two, hipStream_t s) {
  // instantiated (WN, CK) pairs: keep in step with narrow_wn_built()
  switch (pl.WN * 100 + pl.CK) {
    case 504: return convs_launch_two<WM, 5, 4, KIND>(pl, a, two, s);
    case 508: return convs_launch_two<WM, 5, 8, KIND>(pl, a, two, s);
    case 516: return convs_launch_two<WM, 5, 16, KIND>(pl, a, two, s);
    case 532: return convs_launch_two<WM, 5, 32, KIND>(pl, a, two, s);
    case 308: return convs_launch_two<WM, 3, 8, KIND>(pl, a, two, s);
    case 316: return convs_launch_two<WM, 3, 16, KIND>(pl, a, two, s);
    case 332: return convs_launch_two<WM, 3, 32, KIND>(pl, a, two, s);
    case 216: return convs_launch_two<WM, 2, 16, KIND>(pl, a, two, s);
    default:
      rcv_set_error("conv (narrow): no kernel for WN=%d CK=%d", pl.WN, pl.CK);
      return RCV_E_ARG;
  }
}

static int convs_launch_wm4(const ConvPlan& pl, const ConvArgs& a, bool two, hipStream_t s) {
  switch (pl.WN * 100 + pl.CK) {
    case 316: return convs_launch_two<4, 3, 16, KIND_TMERGED>(pl, a, two, s);
    case 332: return convs_launch_two<4, 3, 32, KIND_TMERGED>(pl, a, two, s);
    default:
      rcv_set_error("conv (narrow): no 64-channel kernel for WN=%d CK=%d", pl.WN, pl.CK);
      return RCV_E_ARG;
  }
}

int convs_launch(const ConvPlan& pl, const ConvArgs& a, bool two, hipStream_t s) {
  if (pl.WM == 4) return convs_launch_wm4(pl, a, two, s);
  if (pl.kind == KIND_GATHER) return pl.WM == 1 ? convs_launch_ck<1, KIND_GATHER>(pl, a, two, s) : convs_launch_ck<2, KIND_GATHER>(pl, a, two, s);
  return pl.WM == 1 ? convs_launch_ck<1, KIND_TMERGED>(pl, a, two, s) : convs_launch_ck<2, KIND_TMERGED>(pl, a, two, s);
}

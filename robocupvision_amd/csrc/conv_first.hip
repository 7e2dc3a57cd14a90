// First-layer 3x3 convolution (model.py:475 Level0.Conv0: 3 -> 8 on the NCHW image; PB_FCN's conv0, dilation 2): direct convolution
// on the vector ALU.
//
// With 3 input and 8 output channels the MFMA formulation wastes most of the matrix unit (K = 4 per tap with one padded channel,
// M = 16 rows for 8 channels) and what limits the MFMA kernel there is its per-tile bookkeeping (442 vector instructions per wave
// and tile, 46 % LDS bank-conflict cycles), not arithmetic: the layer is 2.1 GFMA against 0.43 GB of HBM traffic.  On gfx950 the fp32
// vector rate equals the fp32 MFMA rate, so the layer runs as plain FMAs:
//   * a workgroup owns a 16 x 64 output tile; a thread computes 4 consecutive pixels x 8 channels (32 accumulators);
//   * the image tile (with halo, widened to whole 16-byte quads of the image rows) is staged into LDS with ONE batch of 16-byte
//     loads per thread; the 27 x 8 filter sits in LDS as [tap][ci][8] and is read as wave-wide broadcasts;
//   * per input channel and filter row a thread reads its 4 + 2*dil input values once (16-byte aligned) and reuses them for the
//     three taps of the row: 72 LDS reads for 864 FMAs;
//   * bias / ReLU / BatchNorm sums are fused as in the MFMA kernels; workgroups are persistent and write ONE partial row.
#include "conv_common.h"

__device__ __forceinline__ float cf_wave_sum(float v) {
#pragma unroll
  for (int sh = 32; sh >= 1; sh >>= 1) v += __shfl_xor(v, sh);
  return v;
}

template <int DIL>
__global__ __launch_bounds__(256) void conv_first_kernel(const ConvArgs a, int tiles_x, int tiles_y, int total_tiles) {
  constexpr int TY = 16, TX = 64, NT = 256;
  constexpr int IH = TY + 2 * DIL, IWV = TX + 2 * DIL;       // staged rows / valid columns
  constexpr int IWP = TX + 8, NQ = IWP / 4;                  // LDS row: columns x0-4 .. x0+TX+3 (16-byte aligned quads of the image row)
  constexpr int UV = 4;                                      // 16-byte loads in flight per thread (3 planes, dilation 1: one batch)
  constexpr int SEG = 4 + 2 * DIL;                           // input values a thread needs per row
  __shared__ __attribute__((aligned(16))) float xs[4 * IH * IWP];
  __shared__ __attribute__((aligned(16))) float ws[9 * 4 * 8];
  __shared__ float red[4][16];
  __shared__ float4 stg[4 * 256];                            // per wave: 2 output rows x 64 pixels x 8 channels (store transpose)
  const int tid = threadIdx.x;
  const int ty = tid >> 4, tx = (tid & 15) * 4;
  const int Cin = a.Cin;
  // whole image rows are 16-byte aligned quads: the tile is staged with ONE batch of 16-byte loads per thread (a quad lies entirely
  // inside or outside the plane); otherwise (ragged widths, offset views) with scalar loads
  const bool vec = (a.W & 3) == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0;

  // filter: packed [tap][CinP = 4][CoutP = 16] -> LDS [tap][ci][8]
  for (int e = tid; e < 9 * 4 * 8; e += NT) {
    const int co = e & 7, ci = (e >> 3) & 3, tap = e >> 5;
    ws[e] = ci < Cin ? a.w[((size_t)tap * a.CinP + ci) * a.CoutP + co] : 0.f;
  }
  float bias[8];
#pragma unroll
  for (int co = 0; co < 8; ++co) bias[co] = (a.flags & RCV_F_BIAS) ? a.bias[co] : 0.f;
  float s1[8], s2[8];
#pragma unroll
  for (int co = 0; co < 8; ++co) { s1[co] = 0.f; s2[co] = 0.f; }
  const size_t plane = (size_t)a.H * a.W;

  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    int t = tile;
    const int tx_i = t % tiles_x; t /= tiles_x;
    const int ty_i = t % tiles_y;
    const int n = t / tiles_y;
    const int y0 = ty_i * TY, x0 = tx_i * TX;
    // previous tile fully consumed (and the filter is in place).  A bare barrier: __syncthreads() would also drain vmcnt, i.e. make
    // every wave wait for the HBM round trip of the previous tile's output stores before it may request the next tile.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- stage Cin planes of IH rows with zero padding.  (Holding the whole next tile in registers across the arithmetic was
    // tried: 256 registers, one wave per SIMD, no faster.)
    if (vec) {
      const int total = Cin * IH * NQ;
      for (int e0 = tid; e0 < total; e0 += UV * NT) {
        float4 v[UV];
        int dst[UV];
#pragma unroll
        for (int u = 0; u < UV; ++u) {       // branch-free: clamped address, zero selected afterwards (a branch around a load makes the
          const int e = e0 + u * NT;         // compiler wait for it on the spot, one HBM round trip per load instead of one per batch)
          const int ec = e < total ? e : total - 1;
          const int c = ec / (IH * NQ), r = ec - c * (IH * NQ);
          const int iy = r / NQ, j4 = r - iy * NQ;
          const int gy = y0 - DIL + iy, gx = x0 - 4 + 4 * j4;
          const int gyc = gy < 0 ? 0 : (gy < a.H ? gy : a.H - 1), gxc = gx < 0 ? 0 : (gx < a.W ? gx : a.W - 4);
          dst[u] = e < total ? (c * IH + iy) * IWP + 4 * j4 : -1;
          const float4 q = ld4(a.in + ((size_t)n * Cin + c) * plane + (size_t)gyc * a.W + gxc);
          const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
          v[u] = make_float4(ok ? q.x : 0.f, ok ? q.y : 0.f, ok ? q.z : 0.f, ok ? q.w : 0.f);
        }
#pragma unroll
        for (int u = 0; u < UV; ++u) if (dst[u] >= 0) *reinterpret_cast<float4*>(xs + dst[u]) = v[u];
      }
    } else {
      const int per_plane = IH * IWV, total = Cin * per_plane;
      for (int e0 = tid; e0 < total; e0 += 4 * NT) {
        float v[4];
        int dst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = e0 + u * NT;
          v[u] = 0.f; dst[u] = -1;
          if (e < total) {
            const int c = e / per_plane, r = e - c * per_plane;
            const int iy = r / IWV, ix = r - iy * IWV;
            const int gy = y0 - DIL + iy, gx = x0 - DIL + ix;
            dst[u] = (c * IH + iy) * IWP + ix + 4 - DIL;
            if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) v[u] = a.in[((size_t)n * Cin + c) * plane + (size_t)gy * a.W + gx];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (dst[u] >= 0) xs[dst[u]] = v[u];
      }
    }
    __syncthreads();

    // ---- 4 pixels x 8 channels per thread
    float acc[4][8];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int co = 0; co < 8; ++co) acc[p][co] = 0.f;
    for (int c = 0; c < Cin; ++c) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        // image columns x0+tx-DIL .. x0+tx+3+DIL = LDS columns tx+4-DIL .. tx+7+DIL of the row
        const float* row = xs + (c * IH + ty + ky * DIL) * IWP + tx;
        float seg[SEG];
        const float4 q1 = *reinterpret_cast<const float4*>(row + 4);
        if (DIL == 1) {
          seg[0] = row[3];
          seg[1] = q1.x; seg[2] = q1.y; seg[3] = q1.z; seg[4] = q1.w;
          seg[5] = row[8];
        } else {
          const float2 q0 = *reinterpret_cast<const float2*>(row + 2), q2 = *reinterpret_cast<const float2*>(row + 8);
          seg[0] = q0.x; seg[1] = q0.y;
          seg[2] = q1.x; seg[3] = q1.y; seg[4] = q1.z; seg[5] = q1.w;
          seg[SEG - 2] = q2.x; seg[SEG - 1] = q2.y;
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float* wp = ws + ((ky * 3 + kx) * 4 + c) * 8;
          const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
          const float w8[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const float xv = seg[p + kx * DIL];
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[p][co] = fmaf(xv, w8[co], acc[p][co]);
          }
        }
      }
    }

    // ---- bias, ReLU, statistics; then the stores.  A thread holds 4 pixels x 8 channels = 128 contiguous bytes, so a direct store
    // instruction would touch 64 different 128-byte lines with 16 bytes each (measured: the kernel ran at 2.2 TB/s with the vector
    // ALU 44 % busy).  The wave transposes through a private 4 KB LDS window instead (two passes of two rows; quads rotated within
    // each 128-byte block against bank conflicts) and every store instruction writes 1 KB of consecutive memory.
    const int gy = y0 + ty;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const bool okp = gy < a.H && x0 + tx + p < a.W;
#pragma unroll
      for (int co = 0; co < 8; ++co) {
        float v = acc[p][co] + bias[co];
        if (a.flags & RCV_F_RELU) v = fmaxf(v, 0.f);
        acc[p][co] = v;
        if (okp) { s1[co] += v; s2[co] = fmaf(v, v, s2[co]); }
      }
    }
    {
      const int lane = tid & 63, wv = tid >> 6;
      float4* st = stg + wv * 256;                              // [2 rows][128 quads]
      const int lrow = lane >> 4, t4 = lane & 15;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        if ((lrow >> 1) == pass) {
#pragma unroll
          for (int j = 0; j < 8; ++j)                           // quad j = 2 p + h of this thread's 8; block = t4
            st[(lrow & 1) * 128 + 8 * t4 + ((j + t4) & 7)] = make_float4(acc[j >> 1][4 * (j & 1)], acc[j >> 1][4 * (j & 1) + 1],
                                                                         acc[j >> 1][4 * (j & 1) + 2], acc[j >> 1][4 * (j & 1) + 3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // same wave: LDS executes its instructions in order
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int f = lane + 64 * k;                          // quad index in the 2 x 128 window
          const int r = f >> 7, fq = f & 127, blk = fq >> 3;
          const float4 v = st[r * 128 + 8 * blk + (((fq & 7) + blk) & 7)];
          const int oy = y0 + wv * 4 + pass * 2 + r, ox = x0 + (fq >> 1);
          if (oy < a.H && ox < a.W) *reinterpret_cast<float4*>(a.out + ((size_t)(n * a.H + oy) * a.W + x0) * 8 + fq * 4) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the window is rewritten
      }
    }
  }

  // ---- one statistics row per workgroup: wave sums, then the 4 waves in fixed order
  if (a.stats == RCV_STATS_FWD) {
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int co = 0; co < 8; ++co) {
      const float u = cf_wave_sum(s1[co]), v = cf_wave_sum(s2[co]);
      if (lane == 0) { red[wv][co] = u; red[wv][8 + co] = v; }
    }
    __syncthreads();
    if (tid < 16) {
      const float u = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      a.part[(size_t)blockIdx.x * 16 + tid] = u;        // [2][8]: sums, then sums of squares
    }
  }
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
bool conv_first_supported(const rcv_op* op, int kind) {
  if (RCV_ENV("RCV_NO_CONV_FIRST")) return false;
  const int d = op->i[RCV_I_DIL], st = op->i[RCV_I_STATS];
  return kind == KIND_GATHER && op->i[RCV_I_INMODE] == RCV_LOAD_NCHW && op->i[RCV_I_CIN] <= 4 && op->i[RCV_I_COUT] == 8 &&
         op->i[RCV_I_STRIDE] == 1 && (d == 1 || d == 2) && (st == RCV_STATS_NONE || st == RCV_STATS_FWD) && !(op->flags & RCV_F_RESID);
}

int conv_first_plan(const rcv_handle* h, const rcv_op* op, ConvPlan* pl) {
  pl->kind = KIND_GATHER; pl->narrow = 0; pl->dma = 0; pl->first = 1;
  pl->CK = 4; pl->CoutV = 8; pl->CoutP = 16;
  pl->R = 16; pl->Wt = 64;
  pl->tiles_x = ceil_div(op->i[RCV_I_W], 64); pl->tiles_y = ceil_div(op->i[RCV_I_H], 16);
  pl->n_co_tiles = 1; pl->n_phases = 1;
  pl->total_tiles = op->i[RCV_I_N] * pl->tiles_x * pl->tiles_y;
  const int cap = h->num_cus * 4;                   // persistent: ~20 KB of LDS and < 128 registers, several workgroups per CU
  pl->grid = pl->total_tiles < cap ? pl->total_tiles : cap;
  pl->lds = 0;
  return RCV_OK;
}

int conv_first_launch(const ConvPlan& pl, const ConvArgs& a, hipStream_t s) {
  if (a.dil == 1) hipLaunchKernelGGL(conv_first_kernel<1>, dim3(pl.grid), dim3(256), 0, s, a, pl.tiles_x, pl.tiles_y, pl.total_tiles);
  else hipLaunchKernelGGL(conv_first_kernel<2>, dim3(pl.grid), dim3(256), 0, s, a, pl.tiles_x, pl.tiles_y, pl.total_tiles);
  RCV_HIP(hipGetLastError());
  return RCV_OK;
}

// Filter gradient of the first layer (model.py:475 Level0.Conv0, 3 -> 8 channels on the NCHW image; PB_FCN's dilated conv0) on the
// vector ALU.
//
//   dW[cb][ci][ky][kx] = sum_p dz[p][cb] * img[ci][p + (ky,kx)*d - d]        db[cb] = sum_p dz[p][cb]
//
// 27 x 8 outputs from 19 floats per pixel (3 image planes, r and dy of the 8-channel output): 0.75 GB of HBM traffic and 2.1 GFMA at
// 32 x 480 x 640.  In the MFMA formulation (k = pixels, n = (tap, ci) folded into two 16-column blocks) the matrix unit is 80 % idle
// and the kernel is a chain of staging latencies (1.96 TB/s measured).  Here the contraction runs as plain FMAs, with the fp32
// vector rate of gfx950 equal to its fp32 MFMA rate:
//   * a workgroup is six waves, one per (image channel, half of the output channels); a lane owns one pixel column of an 8 x 64 tile
//     and walks its 8 rows, keeping 9 taps x 4 output channels in 36 registers ACROSS all tiles of the (persistent) workgroup;
//   * dz = BN/ReLU backward applied while staging (RCV_LOAD_GRAD_*), written once to LDS as [pixel][8]; the image tile (with halo)
//     plane by plane; a lane reads 16 B of dz and 9 image values per pixel for 36 FMAs, every LDS access lane-contiguous;
//   * 24 KB of LDS, 106 registers: two workgroups (12 waves) per CU overlap their staging and arithmetic phases;
//   * at the end every wave sums its 36 accumulators over the lanes (fixed butterfly) and writes ONE partial row in the layout
//     RCV_OP_WGRAD_REDUCE sums ([split][tap][cb][ca]), the two waves of image channel 0 also the bias row -- no atomics, bitwise reproducible.
#include <stdlib.h>
#include "wgrad_common.h"

__device__ __forceinline__ float wf_wave_sum(float v) {
#pragma unroll
  for (int sh = 32; sh >= 1; sh >>= 1) v += __shfl_xor(v, sh);
  return v;
}

template <int DIL>
__global__ __launch_bounds__(384) void wgrad_first_kernel(const WgradArgs a, int tiles_x, int tiles_y, int total_tiles) {
  constexpr int TY = 8, TX = 64, NT = 384;
  constexpr int IH = TY + 2 * DIL, IW = TX + 2 * DIL;
  __shared__ float4 dzs[TY * TX * 2];          // [pixel][2 quads]
  __shared__ float xs[3 * IH * IW];            // [ci][row][col]
  __shared__ float4 kc[2 * 5];                 // load constants of the two channel quads
  const int tid = threadIdx.x, lane = tid & 63, ci = tid >> 7, half = (tid >> 6) & 1;
  const int split = blockIdx.x;
  const bool p_two = a.p_mode == RCV_LOAD_GRAD_ENC || a.p_mode == RCV_LOAD_GRAD_DEC;
  const size_t plane = (size_t)a.H * a.W;

  if (tid < 10) kc[tid] = a.p_mode != RCV_LOAD_PLAIN ? wld4(a.p_c + (size_t)(tid % 5) * a.CB + 4 * (tid / 5)) : make_float4(0.f, 0.f, 0.f, 0.f);
  float acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = 0.f;
  float bs[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) bs[c] = 0.f;

  // Staging slots of a thread (the same for every tile): image element e = tid + u * NT of the [ci][row][col] tile -> offset relative to
  // the tile's first halo pixel and packed (row, column); dz pixel (tid >> 1) + u * NT / 2, channel quad tid & 1.
  constexpr int UNR = 3;                                  // 3 * 192 = 576 >= 512 dz pixels
  constexpr int PER_PLANE = IH * IW, NU = (3 * PER_PLANE + NT - 1) / NT;
  const int total = a.CA * PER_PLANE;
  const int q = tid & 1;
  const float* p_aux = p_two ? a.p_aux : a.p;             // (one-tensor modes: the second load repeats the first, L1 hit)
  int irel[NU];
  uint32_t iyx[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int e = tid + u * NT;
    const int ec = e < total ? e : 0;
    const int c = ec / PER_PLANE, r = ec - c * PER_PLANE;
    const int iy = r / IW, ix = r - iy * IW;
    irel[u] = c * (int)plane + iy * a.W + ix;
    iyx[u] = (uint32_t)(e < total ? iy : 0x7fff) << 16 | (uint32_t)ix;      // slots past the tile: row 32767, never inside the plane
  }

  // (A register-prefetch pipeline -- loads of tile i+1 issued branch-free before the arithmetic of tile i -- was measured: 0.246 vs
  // 0.251 ms at 32x480x640 and slower on small planes -- and again on top of the one-batch staging below: 0.239 vs 0.179 ms; what bounds
  // the kernel is the number of bytes in flight per CU.  Round 2: ALL
  // loads of a tile -- the dz quads and the image values -- are requested in one branch-free batch: staged one after the other, and the
  // image in two passes, a tile cost three HBM round trips.)
  for (int tile = split; tile < total_tiles; tile += gridDim.x) {
    int t = tile;
    const int tx_i = t % tiles_x; t /= tiles_x;
    const int ty_i = t % tiles_y;
    const int n = t / tiles_y;
    const int y0 = ty_i * TY, x0 = tx_i * TX;
    __syncthreads();                                      // the previous tile is consumed
    float4 x[UNR], ax[UNR];
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int pix = (tid >> 1) + u * (NT / 2);
      const int iy = pix >> 6, ix = pix & 63;
      const int gy = y0 + iy, gx = x0 + ix;
      ok[u] = pix < TY * TX && gy < a.Hp && gx < a.Wp;
      const uint32_t off = ok[u] ? (uint32_t)(((n * a.Hp + gy) * a.Wp + gx) * a.CB + 4 * q) : 0u;      // (host: < 2^31 elements)
      x[u] = wld4(a.p + off);
      ax[u] = wld4(p_aux + off);
    }
    float vimg[NU];
    bool iok[NU];
    const int ibase = n * a.CA * (int)plane + (y0 - DIL) * a.W + (x0 - DIL);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int gy = y0 - DIL + (int)(iyx[u] >> 16), gx = x0 - DIL + (int)(iyx[u] & 0xffffu);
      iok[u] = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      vimg[u] = a.g[iok[u] ? (uint32_t)(ibase + irel[u]) : 0u];
    }
    {   // ---- dz tile: BN/ReLU backward while it goes to LDS
      float4 k[5];                                        // from LDS, behind the tile loads: not live while those are in flight
#pragma unroll
      for (int j = 0; j < 5; ++j) k[j] = kc[q * 5 + j];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int pix = (tid >> 1) + u * (NT / 2);
        if (pix < TY * TX) {
          float4 v = wxform_rt(a.p_mode, x[u], ax[u], k);
          if (!ok[u]) v = make_float4(0.f, 0.f, 0.f, 0.f);
          dzs[pix * 2 + q] = v;
        }
      }
    }
    // ---- image tile: CA planes of IH x IW with zero padding
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int e = tid + u * NT;
      if (e < total) xs[e] = iok[u] ? vimg[u] : 0.f;
    }
    __syncthreads();
    if (ci < a.CA) {
      const float* xc = xs + ci * IH * IW + lane;
#pragma unroll 2
      for (int j = 0; j < TY; ++j) {
        const float4 d0 = dzs[(j * TX + lane) * 2 + half];
        const float dz[4] = {d0.x, d0.y, d0.z, d0.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) bs[c] += dz[c];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float xv = xc[(j + ky * DIL) * IW + kx * DIL];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[ky * 3 + kx][c] = fmaf(xv, dz[c], acc[ky * 3 + kx][c]);
          }
      }
    }
  }

  // ---- one partial row per workgroup
  if (ci < a.CA) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float u = wf_wave_sum(acc[t][c]);
        if (lane == 0) a.part[(((size_t)split * 9 + t) * a.CBP + 4 * half + c) * a.CAP + ci] = u;
      }
  }
  if (ci == 0 && a.part_bias) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float u = wf_wave_sum(bs[c]);
      if (lane == 0) a.part_bias[(size_t)split * a.CBP + 4 * half + c] = u;
    }
  }
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
bool wgrad_first_supported(const rcv_op* op) {
  if (RCV_ENV("RCV_NO_WGRAD_FIRST")) return false;
  const int d = op->i[RCV_I_DIL];
  return op->i[RCV_I_INMODE] == RCV_LOAD_NCHW && op->i[RCV_I_CIN] <= 3 && op->i[RCV_I_COUT] == 8 && op->i[RCV_I_STRIDE] == 1 &&
         (d == 1 || d == 2) && op->i[RCV_I_INMODE2] != RCV_LOAD_NCHW;
}

static inline int wf_occ() { if (const char* e = RCV_ENV("RCV_WF_OCC")) { const int o = atoi(e); if (o >= 1 && o <= 6) return o; } return 2; }   // two workgroups of six waves per CU (106 registers; a third one needs <= 96 and spilled: slower)

static inline int wf_tiles(const rcv_op* op, int* tx, int* ty) {
  *tx = ceil_div(op->i[RCV_I_WO], 64); *ty = ceil_div(op->i[RCV_I_HO], 8);
  return op->i[RCV_I_N] * *tx * *ty;
}

int wgrad_first_nsplit(const rcv_handle* h, const rcv_op* op) {
  int tx, ty;
  const int ntiles = wf_tiles(op, &tx, &ty);
  int nsplit = h->num_cus * wf_occ();                // persistent workgroups of six waves (24 KB of LDS each)
  if (nsplit > ntiles) nsplit = ntiles;
  return ceil_div(ntiles, ceil_div(ntiles, nsplit)); // equal tile counts per workgroup
}

int wgrad_first_launch(const rcv_handle* h, const WgradArgs& a, hipStream_t s) {
  const int tiles_x = ceil_div(a.Wp, 64), tiles_y = ceil_div(a.Hp, 8);
  const int total = a.N * tiles_x * tiles_y;
  (void)h;
  const dim3 g(a.nsplit), b(384);
  if (a.dil == 1) hipLaunchKernelGGL((wgrad_first_kernel<1>), g, b, 0, s, a, tiles_x, tiles_y, total);
  else hipLaunchKernelGGL((wgrad_first_kernel<2>), g, b, 0, s, a, tiles_x, tiles_y, total);
  RCV_HIP(hipGetLastError());
  return RCV_OK;
}

// Wide 3x3 stride-1 convolutions (>= 64 output channels, >= 32 input channels) by Winograd F(2x2, 3x3) on the gfx950 fp32 matrix cores.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// 2.25x fewer multiplications than the direct form (conv_mfma.hip), same fp32 arithmetic (products of transformed operands, summed
// over the input channels by v_mfma_f32_16x16x4_f32); the measured error against an fp64 convolution is ~1.5x the direct kernel's
// (tests/test_gpu_blocks.py::test_winograd_error_budget), far inside the 1e-3 bar.  Serves the forward of Conv2d(k3, s1, p1) and the
// data gradient of the same (flipped / transposed filter), for the layers where the matrix pipe is the bound.
//
// The 16 transformed positions xi = (a, b) are 16 independent GEMMs  M_xi[co][tile] = sum_ci U_xi[ci][co] V_xi[ci][tile]:
//   U = G g G^T   computed once per step by RCV_OP_PACK (layout [xi][ci][co], rcv_pack_job.merged == 2);
//   V = B^T d B   computed in the kernel from the staged input tile (the producer's BatchNorm / the consumer's BN-ReLU backward is
//                 applied while the raw tile is written to LDS, exactly as in the direct kernels; zero padding after it).
// Workgroup = 8 waves = 64 output channels x 80 Winograd tiles (320 output pixels).  Wave w owns xi = 2w, 2w+1 for ALL 4 x 5 MFMA
// blocks of the workgroup tile (160 accumulator registers), so per xi it reads 4 A + 5 B operands for 20 MFMAs.  U streams
// global -> LDS by LDS-DMA one k-step (4 input channels, 16 KB) ahead; the input chunk (16 channels) rides in registers one group
// (4 k-steps) ahead; V of k-step j+1 is formed by the vector ALU while the matrix pipe runs k-step j.  One barrier per k-step.
// After the K loop the 16 M_xi of a (channel, tile) live in 8 different waves: they meet through LDS, one 16-tile block at a time,
// where the output transform, bias / ReLU / skip gradient, the NHWC stores and the BatchNorm partial sums are applied.
#include <type_traits>
#include "conv_common.h"

typedef __attribute__((address_space(3))) void* wino_lds_ptr;
typedef const __attribute__((address_space(1))) void* wino_glb_ptr;

namespace {
constexpr int W_NT = 512, W_NW = 8, W_COT = 64, W_CK = 4, W_XK = 16, W_XMAX = 4;
constexpr int W_UBUF = 16 * W_CK * W_COT;          // floats of one U k-step slab
// W_TB = 16-tile blocks of a workgroup: 5 (80 tiles, the large planes) or 3 (48 tiles: planes whose 80-tile grid would leave CUs idle).
// Row pitch of V ([xi][ci][tile]) = 16 W_TB = 16 mod 32 for both: the two k-lanes of a 32-lane LDS access group hit different banks.
constexpr int W_EXP = W_COT + 4;                   // pitch of the exchange image [xi][tile][co] (68: conflict-free 16-byte accesses)
}

template <bool TWO, int W_TB>
__global__ __launch_bounds__(W_NT) void conv_wino_kernel(const ConvArgs a) {
  constexpr int W_TILES = 16 * W_TB, W_VP = W_TILES, W_VBUF = 16 * W_CK * W_VP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ul = smem;                                // [2][W_UBUF]
  float* vl = ul + 2 * W_UBUF;                     // [2][W_VBUF]
  float* xl = vl + 2 * W_VBUF;                     // [2][xl_floats]: raw (load-transformed) input chunk, [pixel][16 + 2]
  float* cl = xl + 2 * a.xl_floats;                // [5][Cin] load constants (in LDS, not registers: the accumulators leave no room)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int S = a.xpitch;
  const int npix = a.IH * a.IW;
  const int xtotal = npix << 2;                    // (pixel, quad) items of one 16-channel chunk

  // ---- which tile block
  const int t = xcd_remap(blockIdx.x, a.total_tiles);
  const int co_tile = t % a.n_co_tiles;
  int pt = t / a.n_co_tiles;
  const int part_row = pt;
  const int bx = pt % a.tiles_x; pt /= a.tiles_x;
  const int by = pt % a.tiles_y;
  const int n = pt / a.tiles_y;
  const int ty0 = by * a.R, tx0 = bx * a.Wt;       // first Winograd tile (tile coordinates)
  const int co0 = co_tile * W_COT;
  const int oy0 = 2 * ty0 - 1, ox0 = 2 * tx0 - 1;  // image coordinates of raw pixel (0, 0)
  const int ntile = a.R * a.Wt;                    // tiles of a block (<= 80)

  // ---- U stream: two LDS-DMA instructions per wave and k-step; odd rows take their 16-channel blocks pairwise swapped (see conv_mfma.hip)
  int woff[2], wdst[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = (u * W_NW + wave) * 64 + lane;   // 16-byte piece: row = xi*4 + ci, 16 pieces per row
    const int row = e >> 4, c4 = (e & 15) ^ ((row & 1) << 2);
    woff[u] = (((row >> 2) * a.CinP + (row & 3)) * a.CoutP) + co0 + 4 * c4;
    wdst[u] = (u * W_NW + wave) * 256;
  }
  auto dma_u = [&](int buf, int step) {
    const float* src = a.w + (size_t)step * W_CK * a.CoutP;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_global_load_lds((wino_glb_ptr)(src + woff[u]), (wino_lds_ptr)(ul + buf * W_UBUF + wdst[u]), 16, 0, 0);
  };

  // ---- raw input chunk: registers -> (load transform) -> xl
  float4 px[W_XMAX], pa[TWO ? W_XMAX : 1];
  int xsrc[W_XMAX];
#pragma unroll
  for (int u = 0; u < W_XMAX; ++u) {
    const int idx = tid + u * W_NT;
    const int pix = idx >> 2, q = idx & 3;
    const int iy = fd_div(pix, a.fdIW), ix = pix - iy * a.IW;
    const int gy = oy0 + iy, gx = ox0 + ix;
    const bool ok = idx < xtotal && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    xsrc[u] = ok ? (((n * a.H + gy) * a.W + gx) * a.Cin + 4 * q) : -1;
  }
  auto load_x = [&](int g) {
#pragma unroll
    for (int u = 0; u < W_XMAX; ++u) {
      if (u * W_NT < xtotal) {
        const size_t off = (size_t)(xsrc[u] >= 0 ? xsrc[u] : 0) + g * W_XK;
        px[u] = ld4(a.in + off);
        if (TWO) pa[TWO ? u : 0] = ld4(a.in_aux + off);
      }
    }
  };
  auto write_x_mode = [&](auto mode_c, int buf, int g) {
    constexpr int MODE = decltype(mode_c)::value;
    float* xb = xl + buf * a.xl_floats;
    const int q = tid & 3;
    float4 kx[5];
    if (MODE != RCV_LOAD_PLAIN) {
      constexpr int NK = (MODE == RCV_LOAD_AFFINE || MODE == RCV_LOAD_AFFINE_RELU) ? 2 : (MODE == RCV_LOAD_GRAD_ENC ? 3 : 5);
#pragma unroll
      for (int j = 0; j < NK; ++j) kx[j] = *reinterpret_cast<const float4*>(cl + j * a.Cin + g * W_XK + 4 * q);
    }
#pragma unroll
    for (int u = 0; u < W_XMAX; ++u) {
      if (u * W_NT < xtotal) {
        const int idx = tid + u * W_NT;
        if (idx < xtotal) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (xsrc[u] >= 0) v = xform4<MODE>(px[u], pa[TWO ? u : 0], kx);      // zero padding AFTER the transform
          float* d = xb + (idx >> 2) * S + 4 * q;
          d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
      }
    }
  };
  auto write_x = [&](int buf, int g) {
    if constexpr (TWO) {
      if (a.in_mode == RCV_LOAD_GRAD_ENC) write_x_mode(std::integral_constant<int, RCV_LOAD_GRAD_ENC>{}, buf, g);
      else write_x_mode(std::integral_constant<int, RCV_LOAD_GRAD_DEC>{}, buf, g);
    } else {
      if (a.in_mode == RCV_LOAD_AFFINE) write_x_mode(std::integral_constant<int, RCV_LOAD_AFFINE>{}, buf, g);
      else if (a.in_mode == RCV_LOAD_AFFINE_RELU) write_x_mode(std::integral_constant<int, RCV_LOAD_AFFINE_RELU>{}, buf, g);
      else write_x_mode(std::integral_constant<int, RCV_LOAD_PLAIN>{}, buf, g);
    }
  };

  // ---- V = B^T d B of one k-step (4 channels): work item = (tile, channel); 320 items on waves 0..4
  //   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
  const int vt_tile = tid >> 2, vt_c = tid & 3;      // items tid < 320
  int vt_src = 0;
  {
    const int q = vt_tile < ntile ? vt_tile : 0;
    const int tr = fd_div(q, a.fdWt), tc = q - tr * a.Wt;
    vt_src = ((2 * tr) * a.IW + 2 * tc) * S + vt_c;
  }
  auto v_transform = [&](int vbuf, const float* xb_step) {
    if (tid < W_TILES * 4) {
      const float* d = xb_step + vt_src;
      float r[4][4];
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) r[p][q] = d[(p * a.IW + q) * S];
      float tq[4][4];      // rows: B^T d
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        tq[0][q] = r[0][q] - r[2][q];
        tq[1][q] = r[1][q] + r[2][q];
        tq[2][q] = r[2][q] - r[1][q];
        tq[3][q] = r[1][q] - r[3][q];
      }
      // V image [xi][ci][tile], row pitch = 16 mod 32 (the B reads below).  The 32 lanes of one write access are 8 tiles x 4 channels:
      // rows ci = 2, 3 would fall on the banks of rows 0, 1 (2-way conflict on all 16 writes of every item; with the raw-chunk and
      // exchange accesses 32 % of the LDS-active cycles of round 2's profile), so they store tile t at t ^ 8 -- the B read of k-lane
      // ci applies the same swap, every lane still gets ITS tile
      float* o = vl + vbuf * W_VBUF + vt_c * W_VP + (vt_tile ^ ((vt_c >> 1) << 3));
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        o[(p * 4 + 0) * 4 * W_VP] = tq[p][0] - tq[p][2];
        o[(p * 4 + 1) * 4 * W_VP] = tq[p][1] + tq[p][2];
        o[(p * 4 + 2) * 4 * W_VP] = tq[p][2] - tq[p][1];
        o[(p * 4 + 3) * 4 * W_VP] = tq[p][1] - tq[p][3];
      }
    }
  };

  // ---- accumulators: [xi of this wave][co block][tile block]
  f32x4 acc[2][4][W_TB];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int b = 0; b < W_TB; ++b) acc[x][m][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int aoff[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) aoff[m] = l4 * W_COT + ((m * 16 + l15) ^ ((l4 & 1) << 4));
  const int boff = l4 * W_VP + (l15 ^ ((l4 >> 1) << 3));        // (tile swap of rows 2, 3: see v_transform)

  const int nsteps = a.CinP / W_CK, ngroups = a.CinP / W_XK;
  // ---- prologue
  dma_u(0, 0);
  load_x(0);
  if (a.in_c && a.in_mode != RCV_LOAD_PLAIN)
    for (int e = tid; e < 5 * a.Cin; e += W_NT) cl[e] = a.in_c[e];
  __syncthreads();
  write_x(0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  v_transform(0, xl);

  for (int j = 0; j < nsteps; ++j) {
    const int buf = j & 1;
    const int g = j >> 2, sub = j & 3;
    __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): U(j) has landed (requested a k-step ago); so has the chunk in px
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (j + 1 < nsteps) dma_u(buf ^ 1, j + 1);       // U(j+1)
    if (sub == 0 && g + 1 < ngroups) load_x(g + 1);
    const float* ub = ul + buf * W_UBUF;
    const float* vb = vl + buf * W_VBUF;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int xi = 2 * wave + x;
      const float* ux = ub + xi * W_CK * W_COT;
      const float* vx = vb + xi * W_CK * W_VP + boff;
      float av[4], bv[W_TB];
#pragma unroll
      for (int m = 0; m < 4; ++m) av[m] = ux[aoff[m]];
#pragma unroll
      for (int b = 0; b < W_TB; ++b) bv[b] = vx[b * 16];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int b = 0; b < W_TB; ++b)
          acc[x][m][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[b], acc[x][m][b], 0, 0, 0);
      if (x == 0 && j + 1 < nsteps) {
        // V of the next k-step, formed while the matrix pipe runs this one.  Its raw chunk: the group of step j+1.  (Round 3: letting the
        // transform waves that share a SIMD -- 0 and 4, or 1 and 3 -- transform at different points of the k-step instead of together
        // measured 1-3 % SLOWER, exp_r3_wino.sh: the barrier-aligned schedule is not what idles the matrix pipe.)
        const int g1 = (j + 1) >> 2, sub1 = (j + 1) & 3;
        v_transform(buf ^ 1, xl + (g1 & 1) * a.xl_floats + sub1 * W_CK);
      }
    }
    if (sub == 1 && g + 1 < ngroups) write_x((g + 1) & 1, g + 1);     // X(g+1) -> LDS; first read by the V transform two k-steps later
  }

  // ---- output transform + epilogue, one 16-tile block at a time through LDS
  //   A^T = [1 1 1 0; 0 1 -1 -1]
  float* ex = smem;                                  // [16 xi][16 tiles][W_EXP]
  const int e_tile = tid & 15, e_quad = (tid >> 4) & 15, e_i = tid >> 8;      // reader: (tile, 4 channels, output row)
  const int co = co0 + 4 * e_quad;
  const bool co_ok = co < a.Cout;
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), mu = bias, e0 = bias, e1 = bias;
  if (co_ok) {
    if (a.flags & RCV_F_BIAS) bias = ld4(a.bias + co);
    if (a.stats == RCV_STATS_BWD_DEC) { e0 = ld4(a.epi_c + co); e1 = ld4(a.epi_c + a.Cout + co); }
    if (a.stats == RCV_STATS_BWD_DEC || a.stats == RCV_STATS_BWD_ENC) mu = ld4(a.epi_c + 2 * a.Cout + co);
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const bool bwd_stats = a.stats == RCV_STATS_BWD_ENC || a.stats == RCV_STATS_BWD_DEC;
#pragma unroll
  for (int tb = 0; tb < W_TB; ++tb) {
    // this thread's two output pixels of the block, and their skip-gradient / BatchNorm-backward operands: requested BEFORE the
    // exchange, so that their HBM round trip runs under the two barriers and the LDS traffic of the block
    const int q = tb * 16 + e_tile;
    const int tr = fd_div(q < ntile ? q : 0, a.fdWt), tc = (q < ntile ? q : 0) - tr * a.Wt;
    const int oy = 2 * (ty0 + tr) + e_i;
    const int oxb = 2 * (tx0 + tc);
    const bool row_ok = q < ntile && co_ok && oy < a.Ho;
    bool okp[2];
    size_t offp[2];
    float4 rr[2], ee[2];
#pragma unroll
    for (int jx = 0; jx < 2; ++jx) {
      okp[jx] = row_ok && oxb + jx < a.Wo;
      offp[jx] = okp[jx] ? ((size_t)(n * a.Ho + oy) * a.Wo + oxb + jx) * a.Cout + co : 0;
      rr[jx] = make_float4(0.f, 0.f, 0.f, 0.f); ee[jx] = rr[jx];
    }
    if (a.flags & RCV_F_RESID) {
#pragma unroll
      for (int jx = 0; jx < 2; ++jx) rr[jx] = ld4(a.resid + offp[jx]);
    }
    if (bwd_stats) {
#pragma unroll
      for (int jx = 0; jx < 2; ++jx) ee[jx] = ld4(a.epi_aux + offp[jx]);
    }
    __syncthreads();                                 // the K loop / the previous block is done with this LDS
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int xi = 2 * wave + x;
#pragma unroll
      for (int m = 0; m < 4; ++m)
        *reinterpret_cast<float4*>(ex + (xi * 16 + l15) * W_EXP + m * 16 + 4 * l4) =
            make_float4(acc[x][m][tb][0], acc[x][m][tb][1], acc[x][m][tb][2], acc[x][m][tb][3]);
    }
    __syncthreads();
    // rows a of M needed for output row i: i = 0: a = 0,1,2 (+ + +); i = 1: a = 1,2,3 (+ - -)
    float4 rs[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float4 m1 = *reinterpret_cast<const float4*>(ex + ((1 * 4 + b) * 16 + e_tile) * W_EXP + 4 * e_quad);
      const float4 m2 = *reinterpret_cast<const float4*>(ex + ((2 * 4 + b) * 16 + e_tile) * W_EXP + 4 * e_quad);
      const float4 m03 = *reinterpret_cast<const float4*>(ex + (((e_i ? 3 : 0) * 4 + b) * 16 + e_tile) * W_EXP + 4 * e_quad);
      if (e_i == 0) rs[b] = make_float4(m03.x + m1.x + m2.x, m03.y + m1.y + m2.y, m03.z + m1.z + m2.z, m03.w + m1.w + m2.w);
      else rs[b] = make_float4(m1.x - m2.x - m03.x, m1.y - m2.y - m03.y, m1.z - m2.z - m03.z, m1.w - m2.w - m03.w);
    }
    float4 o[2];
    o[0] = make_float4(rs[0].x + rs[1].x + rs[2].x, rs[0].y + rs[1].y + rs[2].y, rs[0].z + rs[1].z + rs[2].z, rs[0].w + rs[1].w + rs[2].w);
    o[1] = make_float4(rs[1].x - rs[2].x - rs[3].x, rs[1].y - rs[2].y - rs[3].y, rs[1].z - rs[2].z - rs[3].z, rs[1].w - rs[2].w - rs[3].w);
#pragma unroll
    for (int jx = 0; jx < 2; ++jx) {
      if (!okp[jx]) continue;
      float4 v = make_float4(o[jx].x + bias.x, o[jx].y + bias.y, o[jx].z + bias.z, o[jx].w + bias.w);
      if (a.flags & RCV_F_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (a.flags & RCV_F_RESID) { v.x += rr[jx].x; v.y += rr[jx].y; v.z += rr[jx].z; v.w += rr[jx].w; }
      *reinterpret_cast<float4*>(a.out + offp[jx]) = v;
      const float4 e = ee[jx];
      if (a.stats == RCV_STATS_FWD) {
        s1[0] += v.x; s1[1] += v.y; s1[2] += v.z; s1[3] += v.w;
        s2[0] = fmaf(v.x, v.x, s2[0]); s2[1] = fmaf(v.y, v.y, s2[1]); s2[2] = fmaf(v.z, v.z, s2[2]); s2[3] = fmaf(v.w, v.w, s2[3]);
      } else if (a.stats == RCV_STATS_BWD_ENC) {
        s1[0] += v.x; s1[1] += v.y; s1[2] += v.z; s1[3] += v.w;
        s2[0] = fmaf(v.x, e.x - mu.x, s2[0]); s2[1] = fmaf(v.y, e.y - mu.y, s2[1]);
        s2[2] = fmaf(v.z, e.z - mu.z, s2[2]); s2[3] = fmaf(v.w, e.w - mu.w, s2[3]);
      } else if (a.stats == RCV_STATS_BWD_DEC) {
        const float gx = fmaf(e.x, e0.x, e1.x) > 0.f ? v.x : 0.f;
        const float gy = fmaf(e.y, e0.y, e1.y) > 0.f ? v.y : 0.f;
        const float gz = fmaf(e.z, e0.z, e1.z) > 0.f ? v.z : 0.f;
        const float gw = fmaf(e.w, e0.w, e1.w) > 0.f ? v.w : 0.f;
        s1[0] += gx; s1[1] += gy; s1[2] += gz; s1[3] += gw;
        s2[0] = fmaf(gx, e.x - mu.x, s2[0]); s2[1] = fmaf(gy, e.y - mu.y, s2[1]);
        s2[2] = fmaf(gz, e.z - mu.z, s2[2]); s2[3] = fmaf(gw, e.w - mu.w, s2[3]);
      }
    }
  }
  if (a.stats != RCV_STATS_NONE) {
    // the 16 tile lanes of a (channel quad, output row) group: xor 1,2,4,8 stays inside the 16-lane group; then the two output rows
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float u = s1[r], v = s2[r];
#pragma unroll
      for (int sh = 1; sh < 16; sh <<= 1) { u += __shfl_xor(u, sh); v += __shfl_xor(v, sh); }
      s1[r] = u; s2[r] = v;
    }
    __syncthreads();
    float* red = smem;                               // [2 rows i][2][64]
    if (e_tile == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[(e_i * 2 + 0) * W_COT + 4 * e_quad + r] = s1[r];
        red[(e_i * 2 + 1) * W_COT + 4 * e_quad + r] = s2[r];
      }
    }
    __syncthreads();
    if (tid < 2 * W_COT) {
      const int which = tid / W_COT, c = tid % W_COT;
      if (co0 + c < a.Cout) a.part[((size_t)part_row * 2 + which) * a.Cout + co0 + c] = red[which * W_COT + c] + red[(2 + which) * W_COT + c];
    }
  }
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
bool conv_wino_supported(const rcv_handle* h, const rcv_op* op, int kind) {
  if (kind != KIND_GATHER || op->i[RCV_I_AUX0] != 2) return false;     // AUX0 == 2: the filter was packed in the Winograd layout
  (void)h;
  return true;
}

// Would this op run on the Winograd kernel if its filter were packed for it?  (The engine asks before it lays out the filter.)
bool conv_wino_wanted(const rcv_handle* h, const rcv_op* op, bool force) {
  const int N = op->i[RCV_I_N], H = op->i[RCV_I_H], W = op->i[RCV_I_W], Cin = op->i[RCV_I_CIN], Cout = op->i[RCV_I_COUT];
  if (op->kind != RCV_OP_CONV || op->i[RCV_I_STRIDE] != 1 || op->i[RCV_I_DIL] != 1) return false;
  if (op->i[RCV_I_INMODE] == RCV_LOAD_NCHW || Cin % 16 || Cin < 32 || Cout % 4 || Cout < 64) return false;     // (32 and 48 input channels: measured 20-26 % faster than the direct kernel)
  if ((long long)N * H * W * Cin >= (1ll << 31)) return false;
  if (force) return true;
  ConvPlan pl;
  memset(&pl, 0, sizeof(pl));
  rcv_op o2 = *op;
  o2.i[RCV_I_AUX0] = 2;
  if (conv_wino_plan(h, &o2, &pl) != RCV_OK) return false;
  // the grid must cover the chip (one 512-thread workgroup per CU): small planes stay on the direct kernel
  return (long)pl.grid * 4 >= (long)h->num_cus * 3;
}

int conv_wino_plan(const rcv_handle* h, const rcv_op* op, ConvPlan* pl) {
  const int N = op->i[RCV_I_N], Ho = op->i[RCV_I_HO], Wo = op->i[RCV_I_WO], Cin = op->i[RCV_I_CIN], Cout = op->i[RCV_I_COUT];
  RCV_CHECK_ARG(op->i[RCV_I_STRIDE] == 1 && op->i[RCV_I_DIL] == 1 && Cin % 16 == 0 && Cout % 4 == 0,
                "winograd conv: needs stride 1, dilation 1, Cin %% 16 == 0 (got s%d d%d Cin %d Cout %d)", op->i[RCV_I_STRIDE], op->i[RCV_I_DIL], Cin, Cout);
  const int TH = ceil_div(Ho, 2), TW = ceil_div(Wo, 2);       // Winograd tiles of a plane
  // tile block: widest row segment of tiles, then as many tile rows as fit in the workgroup's tiles and 512 staged pixels
  auto plan_blocks = [&](int tiles, int* bR, int* bW) {
    int best = -1;
    for (int nx = 1; nx <= TW; ++nx) {
      const int wt = ceil_div(TW, nx);
      if (wt > tiles) continue;
      int r = tiles / wt;
      if (r > TH) r = TH;
      while (r >= 1 && (2 * r + 2) * (2 * wt + 2) > 512) --r;
      if (r < 1) continue;
      r = ceil_div(TH, ceil_div(TH, r));
      const int blocks = ceil_div(TW, wt) * ceil_div(TH, r);
      if (best < 0 || blocks < best) { best = blocks; *bR = r; *bW = wt; }
      if (wt < 8) break;
    }
    return best;
  };
  int bR = 0, bW = 0;
  int best = plan_blocks(80, &bR, &bW);
  RCV_CHECK_ARG(best > 0, "winograd conv: no tile block for a %dx%d plane", Ho, Wo);
  pl->WN = 5;                                                // 16-tile blocks per workgroup
  const int n_co = ceil_div(round_up(Cout, 16), W_COT);
  if ((long)N * best * n_co * 4 < (long)h->num_cus * 3) {    // the 80-tile grid leaves CUs idle: 48-tile workgroups if THEY cover the chip
    int r3 = 0, w3 = 0;
    const int b3 = plan_blocks(48, &r3, &w3);
    if (b3 > 0 && (long)N * b3 * n_co * 4 >= (long)h->num_cus * 3) { best = b3; bR = r3; bW = w3; pl->WN = 3; }
  }
  pl->kind = KIND_GATHER; pl->tile = 0; pl->CK = 4; pl->narrow = 0; pl->dma = 0; pl->first = 0; pl->wino = 1;
  pl->R = bR; pl->Wt = bW; pl->tiles_x = ceil_div(TW, bW); pl->tiles_y = ceil_div(TH, bR);
  pl->IH = 2 * bR + 2; pl->IW = 2 * bW + 2;
  pl->CoutV = Cout; pl->CoutP = round_up(Cout, 16);
  pl->n_co_tiles = ceil_div(pl->CoutP, W_COT); pl->n_phases = 1;
  pl->xk = W_XK;
  pl->xl_floats = round_up(pl->IH * pl->IW * conv_xpitch(W_XK, 1), 4);
  pl->wl_floats = 0;
  size_t floats = 2 * (size_t)W_UBUF + 2 * (size_t)(16 * W_CK * 16 * pl->WN) + 2 * (size_t)pl->xl_floats + 5 * (size_t)round_up(Cin, 4);
  const size_t exch = (size_t)16 * 16 * W_EXP;
  if (floats < exch) floats = exch;
  pl->lds = floats * sizeof(float);
  RCV_CHECK_ARG(pl->lds <= (size_t)h->max_lds, "winograd conv: tile needs %zu B of LDS (limit %d)", pl->lds, h->max_lds);
  pl->total_tiles = N * pl->tiles_x * pl->tiles_y * pl->n_co_tiles;
  pl->grid = pl->total_tiles;
  return RCV_OK;
}

int conv_wino_launch(const ConvPlan& pl, const ConvArgs& a, hipStream_t s) {
  const bool two = a.in_mode == RCV_LOAD_GRAD_ENC || a.in_mode == RCV_LOAD_GRAD_DEC;
#define WINO_LAUNCH(TWO_, TB_)                                                         \
  do {                                                                                 \
    auto kern = conv_wino_kernel<TWO_, TB_>;                                           \
    static size_t configured[RCV_MAX_DEVICES];                                         \
    RCV_ENSURE_LDS(kern, pl.lds, pl.dev, configured);                                  \
    hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(W_NT), pl.lds, s, a);                 \
  } while (0)
  if (pl.WN == 3) { if (two) WINO_LAUNCH(true, 3); else WINO_LAUNCH(false, 3); }
  else { if (two) WINO_LAUNCH(true, 5); else WINO_LAUNCH(false, 5); }
#undef WINO_LAUNCH
  RCV_HIP(hipGetLastError());
  return RCV_OK;
}

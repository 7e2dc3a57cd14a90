// The epilogue shared by the convolution kernels whose lanes end with four consecutive output channels of one pixel (conv_mfma.hip,
// conv_bf3.hip).
#pragma once
#include "conv_common.h"

// Epilogue shared by the conv kernels: bias / ReLU / skip-gradient add, 16-byte NHWC stores, and the per-tile
// BatchNorm partial sums (forward or backward), reduced in a fixed order through `red` ([WAVES_N][2][COT]).
// Two parts, so that a persistent kernel can keep the statistics of all its tiles in registers and reduce them once:
//   conv_epilogue_tile : bias / ReLU / residual, stores, and the lane's share of the BatchNorm sums ADDED to s1 / s2;
//   conv_epilogue_stats: the fixed-order reduction of s1 / s2 over the workgroup and the partial row `row` of a.part.
template <int WM, int WN, int WAVES_M, int WAVES_N, int KIND, int EB = (WM == 1 ? WN : 0)>
__device__ __forceinline__ void conv_epilogue_tile(const ConvArgs& a, const TileInfo& ti, f32x4 (&acc)[WM][WN], float (&s1)[WM][4], float (&s2)[WM][4], int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int m = 0; m < WM; ++m) {
      const int cov = ti.co0 + (wave_m * WM + m) * 16 + 4 * l4;     // (virtual) output channel of this lane
      int co = cov, py = ti.py, pxx = ti.px;
      if (KIND == KIND_TMERGED) { const int ph = cov / a.Cout; co = cov - ph * a.Cout; py = ph >> 1; pxx = ph & 1; }
      const bool co_ok = cov < a.CoutV;
      float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 e0 = bias, e1 = bias, mu = bias;
      if (co_ok) {
        if (a.flags & RCV_F_BIAS) bias = ld4(a.bias + co);
        if (a.stats == RCV_STATS_BWD_DEC) { e0 = ld4(a.epi_c + co); e1 = ld4(a.epi_c + a.Cout + co); }
        if (a.stats == RCV_STATS_BWD_DEC || a.stats == RCV_STATS_BWD_ENC) mu = ld4(a.epi_c + 2 * a.Cout + co);
      }
      // The residual / BN-backward operands of all WN pixel blocks are requested in one batch, branch-free (a lane without an output
      // element reads element 0 and discards it): with each load inside its own `if (ok)` the compiler waits for it on the spot,
      // 2 * WN serialized HBM round trips per channel block in the data-gradient kernels.
      if constexpr (EB == 0) {
        // 32-channel-per-wave tiles (WM = 2): the batch below costs 12 registers and with them the fourth wave per SIMD, which the
        // forward launches of the same kernel (no epilogue operands) pay for: measured +2..7 % there against -1 % in the backward
        // launches.  These tiles load their epilogue operands where they are used.
#pragma unroll
        for (int b = 0; b < WN; ++b) {
          const int p = (wave_n * WN + b) * 16 + l15;
          const int ty = fd_div(p, a.fdWt), tx = p - ty * a.Wt;
          int oy = ti.y0 + ty, ox = ti.x0 + tx;
          bool ok = ty < a.R && co_ok;
          if (KIND != KIND_GATHER) {
            ok = ok && oy < a.H && ox < a.W;
            oy = 2 * oy + py; ox = 2 * ox + pxx;
          } else {
            ok = ok && oy < a.Ho && ox < a.Wo;
          }
          if (!ok) continue;
          const size_t off = ((size_t)(ti.n * a.Ho + oy) * a.Wo + ox) * a.Cout + co;
          float4 v = make_float4(acc[m][b][0] + bias.x, acc[m][b][1] + bias.y, acc[m][b][2] + bias.z, acc[m][b][3] + bias.w);
          if (a.flags & RCV_F_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (a.flags & RCV_F_RESID) { const float4 rr = ld4(a.resid + off); v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
          *reinterpret_cast<float4*>(a.out + off) = v;
          if (a.stats == RCV_STATS_FWD) {
            s1[m][0] += v.x; s1[m][1] += v.y; s1[m][2] += v.z; s1[m][3] += v.w;
            s2[m][0] = fmaf(v.x, v.x, s2[m][0]); s2[m][1] = fmaf(v.y, v.y, s2[m][1]);
            s2[m][2] = fmaf(v.z, v.z, s2[m][2]); s2[m][3] = fmaf(v.w, v.w, s2[m][3]);
          } else if (a.stats == RCV_STATS_BWD_ENC) {
            const float4 e = ld4(a.epi_aux + off);
            s1[m][0] += v.x; s1[m][1] += v.y; s1[m][2] += v.z; s1[m][3] += v.w;
            s2[m][0] = fmaf(v.x, e.x - mu.x, s2[m][0]); s2[m][1] = fmaf(v.y, e.y - mu.y, s2[m][1]);
            s2[m][2] = fmaf(v.z, e.z - mu.z, s2[m][2]); s2[m][3] = fmaf(v.w, e.w - mu.w, s2[m][3]);
          } else if (a.stats == RCV_STATS_BWD_DEC) {
            const float4 e = ld4(a.epi_aux + off);
            const float gx = fmaf(e.x, e0.x, e1.x) > 0.f ? v.x : 0.f;
            const float gy = fmaf(e.y, e0.y, e1.y) > 0.f ? v.y : 0.f;
            const float gz = fmaf(e.z, e0.z, e1.z) > 0.f ? v.z : 0.f;
            const float gw = fmaf(e.w, e0.w, e1.w) > 0.f ? v.w : 0.f;
            s1[m][0] += gx; s1[m][1] += gy; s1[m][2] += gz; s1[m][3] += gw;
            s2[m][0] = fmaf(gx, e.x - mu.x, s2[m][0]); s2[m][1] = fmaf(gy, e.y - mu.y, s2[m][1]);
            s2[m][2] = fmaf(gz, e.z - mu.z, s2[m][2]); s2[m][3] = fmaf(gw, e.w - mu.w, s2[m][3]);
          }
        }
      } else {
        const bool bwd_stats = a.stats == RCV_STATS_BWD_ENC || a.stats == RCV_STATS_BWD_DEC;
  #pragma unroll
        for (int b0 = 0; b0 < WN; b0 += (EB > 0 ? EB : 1)) {       // EB blocks per batch (register budget of the caller)
          bool okb[EB > 0 ? EB : 1];
          size_t offb[EB > 0 ? EB : 1];
  #pragma unroll
          for (int i = 0; i < EB; ++i) {
            const int b = b0 + i;
            okb[i] = false; offb[i] = 0;
            if (b < WN) {
              const int p = (wave_n * WN + b) * 16 + l15;
              const int ty = fd_div(p, a.fdWt), tx = p - ty * a.Wt;
              int oy = ti.y0 + ty, ox = ti.x0 + tx;
              bool ok = ty < a.R && co_ok;
              if (KIND != KIND_GATHER) {
                ok = ok && oy < a.H && ox < a.W;
                oy = 2 * oy + py; ox = 2 * ox + pxx;
              } else {
                ok = ok && oy < a.Ho && ox < a.Wo;
              }
              okb[i] = ok;
              offb[i] = ok ? ((size_t)(ti.n * a.Ho + oy) * a.Wo + ox) * a.Cout + co : 0;
            }
          }
          float4 rrb[EB > 0 ? EB : 1], eb[EB > 0 ? EB : 1];
          if (a.flags & RCV_F_RESID) {
  #pragma unroll
            for (int i = 0; i < EB; ++i) if (b0 + i < WN) rrb[i] = ld4(a.resid + offb[i]);
          }
          if (bwd_stats) {
  #pragma unroll
            for (int i = 0; i < EB; ++i) if (b0 + i < WN) eb[i] = ld4(a.epi_aux + offb[i]);
          }
  #pragma unroll
          for (int i = 0; i < EB; ++i) {
            const int b = b0 + i;
            if (b >= WN) continue;
            if (!okb[i]) continue;
            const size_t off = offb[i];
            float4 v = make_float4(acc[m][b][0] + bias.x, acc[m][b][1] + bias.y, acc[m][b][2] + bias.z, acc[m][b][3] + bias.w);
            if (a.flags & RCV_F_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (a.flags & RCV_F_RESID) { const float4 rr = rrb[i]; v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
            *reinterpret_cast<float4*>(a.out + off) = v;
            if (a.stats == RCV_STATS_FWD) {
              s1[m][0] += v.x; s1[m][1] += v.y; s1[m][2] += v.z; s1[m][3] += v.w;
              s2[m][0] = fmaf(v.x, v.x, s2[m][0]); s2[m][1] = fmaf(v.y, v.y, s2[m][1]);
              s2[m][2] = fmaf(v.z, v.z, s2[m][2]); s2[m][3] = fmaf(v.w, v.w, s2[m][3]);
            } else if (a.stats == RCV_STATS_BWD_ENC) {
              const float4 e = eb[i];
              s1[m][0] += v.x; s1[m][1] += v.y; s1[m][2] += v.z; s1[m][3] += v.w;
              s2[m][0] = fmaf(v.x, e.x - mu.x, s2[m][0]); s2[m][1] = fmaf(v.y, e.y - mu.y, s2[m][1]);
              s2[m][2] = fmaf(v.z, e.z - mu.z, s2[m][2]); s2[m][3] = fmaf(v.w, e.w - mu.w, s2[m][3]);
            } else if (a.stats == RCV_STATS_BWD_DEC) {
              const float4 e = eb[i];
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
      }
    }
}

template <int WM, int WAVES_M, int WAVES_N, int KIND>
__device__ __forceinline__ void conv_epilogue_stats(const ConvArgs& a, size_t row, int co0, float (&s1)[WM][4], float (&s2)[WM][4], float* red, int tid) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int COT = WM * WAVES_M * 16;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
  const int l15 = lane & 15, l4 = lane >> 4;
    {
      // wave: sum over the 16 pixel lanes (xor 1,2,4,8 stays inside a 16-lane group); fixed order
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float u = s1[m][r], v = s2[m][r];
#pragma unroll
          for (int sh = 1; sh < 16; sh <<= 1) { u += __shfl_xor(u, sh); v += __shfl_xor(v, sh); }
          s1[m][r] = u; s2[m][r] = v;
        }
      if (l15 == 0) {
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int cl_ = (wave_m * WM + m) * 16 + 4 * l4 + r;
            red[(wave_n * 2 + 0) * COT + cl_] = s1[m][r];
            red[(wave_n * 2 + 1) * COT + cl_] = s2[m][r];
          }
      }
      __syncthreads();
      if (KIND == KIND_TMERGED) {
        // real channel = virtual channel mod Cout: sum the (up to) four parity groups in fixed order
        for (int e = tid; e < 2 * a.Cout; e += NT) {
          const int which = e / a.Cout, co = e - which * a.Cout;
          float u = 0.f;
          for (int cv = co; cv < a.CoutV; cv += a.Cout) {
#pragma unroll
            for (int wn = 0; wn < WAVES_N; ++wn) u += red[(wn * 2 + which) * COT + cv];
          }
          a.part[(row * 2 + which) * a.Cout + co] = u;
        }
      } else {
        for (int e = tid; e < 2 * COT; e += NT) {
          const int which = e / COT, cl_ = e % COT;
          const int co = co0 + cl_;
          if (co < a.Cout) {
            float u = 0.f;
#pragma unroll
            for (int wn = 0; wn < WAVES_N; ++wn) u += red[(wn * 2 + which) * COT + cl_];
            a.part[(row * 2 + which) * a.Cout + co] = u;
          }
        }
      }
  }
}

template <int WM, int WN, int WAVES_M, int WAVES_N, int KIND, int EB = (WM == 1 ? WN : 0)>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, const TileInfo& ti, f32x4 (&acc)[WM][WN], float* red, int tid) {
  float s1[WM][4], s2[WM][4];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[m][r] = 0.f; s2[m][r] = 0.f; }
  conv_epilogue_tile<WM, WN, WAVES_M, WAVES_N, KIND, EB>(a, ti, acc, s1, s2, tid);
  if (a.stats != RCV_STATS_NONE) {
    const size_t row = (size_t)(KIND == KIND_TPHASE ? (ti.py * 2 + ti.px) * a.n_pix_tiles : 0) + ti.pt;
    conv_epilogue_stats<WM, WAVES_M, WAVES_N, KIND>(a, row, ti.co0, s1, s2, red, tid);
  }
}

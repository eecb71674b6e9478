// GroupNorm statistics handed from a convolution's epilogue to the consumer of its output (qstat_out tables, aldm_hip.h): the
// device-side summation shared by aldm_groupnorm_apply, aldm_gn_silu_conv3x3_small (norm.hip) and the halo convolution that
// normalises its input on the way into LDS (igemm_halo.hip).
#pragma once
#include "common.h"

struct GnSrc {
  const bf16* x; const float* tab;
  int C;        // channels of this source
  int bm;       // generic tiles: rows per M-tile (a tile may run into the next image: slot 1)
  int tpi;      // > 0: tiles are image-aligned, `tpi` per image, slot 0 only (halo kernels)
};

// (sum, sum of squares) of group g of image b from the producers' per-tile tables: lane j of the group's lpg lanes takes tiles
// t0 + j, t0 + j + lpg, ...  Four tiles per round with independent loads: a one-tile-per-iteration loop is a chain of L2 round
// trips (HW = 4000: 8 of them per lane, half the launch's time; a VAE image of 65536 pixels has 512 tiles).
__device__ __forceinline__ void gn_source_sums(const GnSrc& s, int b, int HW, int qa, int nq, int j, int lpg, float& a, float& q2) {
  const int Q = s.C >> 2;
  int t0, t1;
  if (s.tpi > 0) { t0 = b * s.tpi; t1 = t0 + s.tpi - 1; }
  else { t0 = (b * HW) / s.bm; t1 = ((b + 1) * HW - 1) / s.bm; }
  float a4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
  for (int t = t0 + j; t <= t1; t += 4 * lpg) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int tt = min(t + u * lpg, t1);                   // (clamped: always a valid address; the weight below drops repeats)
      const float w = (t + u * lpg <= t1) ? 1.f : 0.f;
      const int slot = (s.tpi > 0) ? 0 : (((tt * s.bm) / HW) != b);
      const float* row = s.tab + ((long long)(tt * 2 + slot) * Q + qa) * 2;
      for (int k = 0; k < nq; ++k) {
        const float2 v = *reinterpret_cast<const float2*>(row + 2 * k);
        a4[u] = fmaf(w, v.x, a4[u]);
        q4[u] = fmaf(w, v.y, q4[u]);
      }
    }
  }
  a += (a4[0] + a4[1]) + (a4[2] + a4[3]);
  q2 += (q4[0] + q4[1]) + (q4[2] + q4[3]);
}
// a group of the concatenation may take its first channels from source 1 and the rest from source 2 (256 + 128 channels in 32
// groups of 12: group 21 is channels 252 .. 263); the 4-channel quads never straddle (C1 % 4 == 0)
__device__ __forceinline__ void gn_group_sums(const GnSrc& s1, const GnSrc& s2, int b, int HW, int Cg, int g, int j, int lpg, float& a, float& q2) {
  const int c0 = g * Cg, c1 = c0 + Cg;
  if (c0 < s1.C) gn_source_sums(s1, b, HW, c0 >> 2, (min(c1, s1.C) - c0) >> 2, j, lpg, a, q2);
  if (c1 > s1.C) {
    const int lo = max(c0, s1.C);
    gn_source_sums(s2, b, HW, (lo - s1.C) >> 2, (c1 - lo) >> 2, j, lpg, a, q2);
  }
}


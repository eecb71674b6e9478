// GroupNorm(+SiLU) and LayerNorm for channels-last bf16 activations on gfx950.
// HBM/L2-bound kernels: 8-byte (4-channel) accesses for GroupNorm strips, 16-byte rows for LayerNorm,
// fp32 statistics, two-pass (mean, then centred variance) out of an LDS copy of the strip when it fits.
//
// GroupNorm serves F.group_norm in ResnetBlock2D.norm1/2, Transformer2DModel.norm, conv_norm_out and
// the VAE's group norms; LayerNorm serves BasicTransformerBlock.norm1/2/3 -- all under
// UNet2DConditionModel.forward [REF script/train/train_audioldm_lora.py:539-546] /
// AutoencoderKL.decode [REF script/inference/generate_audio.py:47-52].
#include <cstdlib>
#include "common.h"
#include "gn_stats.h"

namespace {

constexpr int GN_THREADS = 512;
constexpr int GN_MAXCG = 256;       // channels per group the register-strip kernels stage gamma / beta for (UNet: <= 80)
constexpr int GN_LDS_QUADS = 16384;  // 128 KiB of bf16x4 strip cache

__device__ __forceinline__ float block_sum(float v, float* red, int tid, int nthreads) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nthreads / 64; ++i) t += red[i];
  return t;
}

// two sums with ONE pair of barriers (red: 2 x nthreads / 64 floats)
__device__ __forceinline__ void block_sum_pair(float& a, float& b, float* red, int tid, int nthreads) {
  a = wave_sum(a);
  b = wave_sum(b);
  __syncthreads();
  const int nw = nthreads / 64;
  if ((tid & 63) == 0) { red[tid >> 6] = a; red[nw + (tid >> 6)] = b; }
  __syncthreads();
  float ta = 0.f, tb = 0.f;
  for (int i = 0; i < nw; ++i) { ta += red[i]; tb += red[nw + i]; }
  a = ta; b = tb;
}

// One workgroup per (batch, group).  A "quad" is 4 consecutive channels of one pixel (8 bytes); every
// group width used by the models is a multiple of 4 and so is the concat boundary C1.
__global__ __launch_bounds__(GN_THREADS) void groupnorm_kernel(const bf16* __restrict__ x, const bf16* __restrict__ x2,
                                                               int HW, int C1, int C2, int groups, float eps,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int act,
                                                               bf16* __restrict__ y, AldmDiv dqpp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x4* cache = reinterpret_cast<bf16x4*>(smem);
  float* red = reinterpret_cast<float*>(smem + (size_t)GN_LDS_QUADS * 8);

  const int C = C1 + C2;
  const int Cg = C / groups, qpp = Cg >> 2;  // quads per pixel in this group
  // batch-minor block order: the 32 group-workgroups of one image then share an XCD (blocks b, b+8, ... land on
  // the same L2), so each 128-byte line of the image is fetched into ONE L2 instead of eight.
  const int nb = gridDim.x / groups;
  const int g = blockIdx.x / nb, b = blockIdx.x - g * nb;
  const int c0 = g * Cg;
  const int nquads = HW * qpp;
  const bool cached = nquads <= GN_LDS_QUADS;
  const int tid = threadIdx.x;

  auto load_quad = [&](int q) -> bf16x4 {
    const int pix = aldm_div(q, dqpp), j = q - pix * qpp;
    const int c = c0 + 4 * j;
    if (c < C1) return *reinterpret_cast<const bf16x4*>(x + ((long long)b * HW + pix) * C1 + c);
    return *reinterpret_cast<const bf16x4*>(x2 + ((long long)b * HW + pix) * C2 + (c - C1));
  };

  float s = 0.f;
  for (int q = tid; q < nquads; q += GN_THREADS) {
    const bf16x4 v = load_quad(q);
    if (cached) cache[q] = v;
    s += (float)v[0] + (float)v[1] + (float)v[2] + (float)v[3];
  }
  const float n = (float)nquads * 4.f;
  const float mean = block_sum(s, red, tid, GN_THREADS) / n;
  float ss = 0.f;
  for (int q = tid; q < nquads; q += GN_THREADS) {
    const bf16x4 v = cached ? cache[q] : load_quad(q);
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float d = (float)v[k] - mean; ss += d * d; }
  }
  const float var = block_sum(ss, red, tid, GN_THREADS) / n;
  const float rstd = rsqrtf(var + eps);

  for (int q = tid; q < nquads; q += GN_THREADS) {
    const bf16x4 v = cached ? cache[q] : load_quad(q);
    const int pix = aldm_div(q, dqpp), j = q - pix * qpp;
    const int c = c0 + 4 * j;
    bf16x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gm = gamma[c + k] * rstd;
      float t = ((float)v[k] - mean) * gm + beta[c + k];
      if (act == ALDM_ACT_SILU) t = silu_f(t);
      o[k] = (bf16)t;
    }
    *reinterpret_cast<bf16x4*>(y + ((long long)b * HW + pix) * C + c) = o;
  }
}

// Register-resident variant for strips of <= 512*QPT quads (every UNet site): all loads are issued up front
// (QPT independent 8-byte loads in flight per thread), the strip never leaves registers, two block reductions.
template <int QPT>
__global__ __launch_bounds__(GN_THREADS) void groupnorm_reg_kernel(const bf16* __restrict__ x, const bf16* __restrict__ x2,
                                                                   int HW, int C1, int C2, int groups, float eps,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, int act,
                                                                   bf16* __restrict__ y, AldmDiv dqpp) {
  aldm_touch_kernargs<96>();                // 84 bytes of explicit arguments: both lines in one round (common.h)
#ifndef ALDM_NO_KA_PREFETCH
  aldm_prefetch_next_kernargs<96>(threadIdx.x);
#endif
  __shared__ float red[2 * GN_THREADS / 64];
  __shared__ __attribute__((aligned(16))) float sgm[GN_MAXCG], sbt[GN_MAXCG];
  const int C = C1 + C2;
  const int Cg = C / groups, qpp = Cg >> 2;
  const int nb = gridDim.x / groups;
  const int g = blockIdx.x / nb, b = blockIdx.x - g * nb;   // batch-minor: one image's groups share an XCD's L2
  const int c0 = g * Cg;
  const int nquads = HW * qpp;
  const int tid = threadIdx.x;
  // the group's gamma / beta go to LDS now, under the strip's loads (visible after the first reduction's barrier): fetched per
  // output quad they were a load -> wait -> compute -> store chain of QPT L2 round trips at the END of the kernel
  if (tid < Cg) { sgm[tid] = gamma[c0 + tid]; sbt[tid] = beta[c0 + tid]; }
  const float pivot = (float)((c0 < C1) ? x[(long long)b * HW * C1 + c0] : x2[(long long)b * HW * C2 + (c0 - C1)]);

  bf16x4 v[QPT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    const int q = tid + i * GN_THREADS;
    bf16x4 t = {0, 0, 0, 0};
    if (q < nquads) {
      const int pix = aldm_div(q, dqpp), j = q - pix * qpp;
      const int c = c0 + 4 * j;
      t = (c < C1) ? *reinterpret_cast<const bf16x4*>(x + ((long long)b * HW + pix) * C1 + c)
                   : *reinterpret_cast<const bf16x4*>(x2 + ((long long)b * HW + pix) * C2 + (c - C1));
    }
    v[i] = t;
  }
  // mean and variance from ONE reduction: sums of (x - p) and (x - p)^2 around a pivot p every thread knows without a barrier -- the
  // strip's first element (requested with the strip).  Shifted, the one-pass variance has no cancellation problem when |mean| >> std;
  // the second reduction (two more barriers, ~0.4 us of a 6 us launch) is gone.
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    if (tid + i * GN_THREADS < nquads) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float d = (float)v[i][k] - pivot; s += d; ss = fmaf(d, d, ss); }
    }
  }
  const float n = (float)nquads * 4.f;
  block_sum_pair(s, ss, red, tid, GN_THREADS);
  const float dm = s / n;
  const float mean = pivot + dm;
  const float rstd = rsqrtf(fmaxf(ss / n - dm * dm, 0.f) + eps);
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    const int q = tid + i * GN_THREADS;
    if (q < nquads) {
      const int pix = aldm_div(q, dqpp), j = q - pix * qpp;
      const int c = c0 + 4 * j;
      const f32x4 gm = *reinterpret_cast<const f32x4*>(sgm + 4 * j);
      const f32x4 bt = *reinterpret_cast<const f32x4*>(sbt + 4 * j);
      bf16x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = ((float)v[i][k] - mean) * (gm[k] * rstd) + bt[k];
        if (act == ALDM_ACT_SILU) t = silu_f(t);
        o[k] = (bf16)t;
      }
      *reinterpret_cast<bf16x4*>(y + ((long long)b * HW + pix) * C + c) = o;
    }
  }
}

// GroupNorm(+SiLU) over split-K partial tiles (aldm_groupnorm_partials): the strip is summed out of the fp32 workspace
// [splits][B*HW][C] in split order (as igemm_reduce_kernel sums), bias and the per-image row bias (the time-embedding
// projection of ResnetBlock2D) are added, and the fp32 strip stays in registers for the statistics.
template <int QPT>
__global__ __launch_bounds__(GN_THREADS) void groupnorm_partials_kernel(const float* __restrict__ ws, int splits, long long sstride,
                                                                        int HW, int C, int groups, float eps,
                                                                        const float* __restrict__ bias,
                                                                        const float* __restrict__ rowbias, int rowbias_ld,
                                                                        const bf16* __restrict__ res, bf16* __restrict__ sum_out,
                                                                        const bf16* __restrict__ x2, int C2,
                                                                        const float* __restrict__ gamma,
                                                                        const float* __restrict__ beta, int act,
                                                                        bf16* __restrict__ y, AldmDiv dqpp) {
  aldm_touch_kernargs<160>();               // 148 bytes of explicit arguments: all three lines in one round (common.h)
#ifndef ALDM_NO_KA_PREFETCH
  aldm_prefetch_next_kernargs<160>(threadIdx.x);
#endif
  // C = channels of the partial tiles (first source); C2 more channels come as plain bf16 from x2 (torch.cat([h, skip]) in
  // front of an up-block ResnetBlock2D's norm1).  A group lies wholly in one source (host-checked: C % group width == 0).
  __shared__ float red[2 * GN_THREADS / 64];
  __shared__ __attribute__((aligned(16))) float sgm[GN_MAXCG], sbt[GN_MAXCG], sadd[GN_MAXCG];
  const int Ct = C + C2;
  const int Cg = Ct / groups, qpp = Cg >> 2;
  const int nb = gridDim.x / groups;
  const int g = blockIdx.x / nb, b = blockIdx.x - g * nb;
  const int c0 = g * Cg;
  const int nquads = HW * qpp;
  const int tid = threadIdx.x;
  // per-channel constants of the group through LDS, fetched under the partial tiles' loads: gamma, beta, and bias + this image's
  // row bias as one addend (each used to be a load -> wait chain of its own behind the split-K sum)
  if (tid < Cg) {
    const int c = c0 + tid;
    sgm[tid] = gamma[c];
    sbt[tid] = beta[c];
    float a = 0.f;
    if (c < C) {
      if (bias) a = bias[c];
      if (rowbias) a += rowbias[(long long)b * rowbias_ld + c];
    }
    sadd[tid] = a;
  }

  f32x4 v[QPT];
  bf16x4 rv[QPT];
  long long off[QPT], yoff[QPT];
  int ch[QPT];
  const bool second = c0 >= C;                 // (workgroup-uniform) this group's channels come from x2
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    const int q = tid + i * GN_THREADS;
    const int qq = q < nquads ? q : 0;
    const int pix = aldm_div(qq, dqpp), j = qq - pix * qpp;
    ch[i] = c0 + 4 * j;
    off[i] = ((long long)b * HW + pix) * C + ch[i];
    yoff[i] = ((long long)b * HW + pix) * Ct + ch[i];
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    rv[i] = bf16x4{0, 0, 0, 0};
    if (second) {
      if (q < nquads) rv[i] = *reinterpret_cast<const bf16x4*>(x2 + ((long long)b * HW + pix) * C2 + (ch[i] - C));
    } else if (res && q < nquads) {
      rv[i] = *reinterpret_cast<const bf16x4*>(res + off[i]);   // in flight with the partial tiles
    }
  }
  if (second) splits = 0;
  int sp = 0;
  for (; sp + 4 <= splits; sp += 4) {
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
      if (tid + i * GN_THREADS < nquads) {
        const float* w0 = ws + off[i] + sp * sstride;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(w0), a1 = *reinterpret_cast<const f32x4*>(w0 + sstride);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(w0 + 2 * sstride), a3 = *reinterpret_cast<const f32x4*>(w0 + 3 * sstride);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[i][k] = (((v[i][k] + a0[k]) + a1[k]) + a2[k]) + a3[k];
      }
    }
  }
  if (sp < splits) {                           // 1 .. 3 partials left: requested together (one round trip), added in split order
    const int rem = splits - sp;
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
      if (tid + i * GN_THREADS < nquads) {
        const float* w0 = ws + off[i] + sp * sstride;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(w0);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(w0 + (rem > 1 ? 1 : 0) * sstride);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(w0 + (rem > 2 ? 2 : 0) * sstride);
        v[i] += a0;
        if (rem > 1) v[i] += a1;
        if (rem > 2) v[i] += a2;
      }
    }
  }
  __syncthreads();                              // sadd / sgm / sbt are in LDS
  const float pivot = sadd[0];
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    if (tid + i * GN_THREADS < nquads) {
      const int c = ch[i];
      if (!second) v[i] += *reinterpret_cast<const f32x4*>(sadd + (c - c0));   // bias + row bias
      if (second || res) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[i][k] += (float)rv[i][k];
      }
      if (sum_out && !second) {
        bf16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (bf16)v[i][k];
        *reinterpret_cast<bf16x4*>(sum_out + off[i]) = o;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float d = v[i][k] - pivot; s += d; ss = fmaf(d, d, ss); }
    }
  }
  // (one reduction for mean and variance, shifted by the group's first bias + row bias -- the part of the values that can be large
  //  against their spread; see groupnorm_reg_kernel)
  const float n = (float)nquads * 4.f;
  block_sum_pair(s, ss, red, tid, GN_THREADS);
  const float dm = s / n;
  const float mean = pivot + dm;
  const float rstd = rsqrtf(fmaxf(ss / n - dm * dm, 0.f) + eps);
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    if (tid + i * GN_THREADS < nquads) {
      const int c = ch[i];
      const f32x4 gm = *reinterpret_cast<const f32x4*>(sgm + (c - c0));
      const f32x4 bt = *reinterpret_cast<const f32x4*>(sbt + (c - c0));
      bf16x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = (v[i][k] - mean) * (gm[k] * rstd) + bt[k];
        if (act == ALDM_ACT_SILU) t = silu_f(t);
        o[k] = (bf16)t;
      }
      *reinterpret_cast<bf16x4*>(y + yoff[i]) = o;
    }
  }
}

// LayerNorm: LPR lanes per row (a full wave, or half a wave when the row has <= 32 16-byte chunks, i.e. C <= 256 -- the
// widest level of the UNet -- so that no lane idles), 16-byte chunks, C <= LPR*8*MAXC.
constexpr int LN_MAXC = 4;  // chunks per lane -> C <= 2048
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int LPR>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16* __restrict__ x, int M, int C,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        bf16* __restrict__ y) {
  constexpr int RPW = 64 / LPR;                          // rows per wave
  const int lane = threadIdx.x & (LPR - 1);
  const int row = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RPW + ((threadIdx.x & 63) / LPR);
  const bool live = row < M;                             // keep every lane in the shuffles
  const int nch = C >> 3;
  const bf16* xr = x + (long long)(live ? row : 0) * C;
  bf16x8 v[LN_MAXC];
  f32x4 g0[LN_MAXC], g1[LN_MAXC], b0[LN_MAXC], b1[LN_MAXC];   // gamma / beta requested WITH the row (they were a load -> wait -> store
  float s = 0.f;                                              // chain behind the statistics: one more L2 round trip per launch)
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int ch = lane + LPR * i;
    if (ch < nch) {
      v[i] = *reinterpret_cast<const bf16x8*>(xr + ch * 8);
      g0[i] = *reinterpret_cast<const f32x4*>(gamma + ch * 8); g1[i] = *reinterpret_cast<const f32x4*>(gamma + ch * 8 + 4);
      b0[i] = *reinterpret_cast<const f32x4*>(beta + ch * 8); b1[i] = *reinterpret_cast<const f32x4*>(beta + ch * 8 + 4);
    }
  }
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i)
    if (lane + LPR * i < nch) {
#pragma unroll
      for (int k = 0; k < 8; ++k) s += (float)v[i][k];
    }
  const float mean = row_sum<LPR>(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int ch = lane + LPR * i;
    if (ch < nch) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float d = (float)v[i][k] - mean; ss += d * d; }
    }
  }
  const float rstd = rsqrtf(row_sum<LPR>(ss) / (float)C + eps);
  if (!live) return;
  bf16* yr = y + (long long)row * C;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int ch = lane + LPR * i;
    if (ch < nch) {
      bf16x8 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[k] = (bf16)(((float)v[i][k] - mean) * rstd * g0[i][k] + b0[i][k]);
        o[4 + k] = (bf16)(((float)v[i][4 + k] - mean) * rstd * g1[i][k] + b1[i][k]);
      }
      *reinterpret_cast<bf16x8*>(yr + ch * 8) = o;
    }
  }
}

// RoBERTa-style embedding sum + LayerNorm for the CLAP text tower: one wave per token.
//   e = word[id] + type[0] + pos[pid],  pid = pad + (id != pad ? #{i <= j : ids[b][i] != pad} : 0)
// (fairseq make_positions, as transformers ClapTextEmbeddings.create_position_ids_from_input_ids).  fp32 tables,
// fp32 statistics, bf16 activations out.
__global__ __launch_bounds__(256) void embed_layernorm_kernel(const long long* __restrict__ ids, int M, int L, int C,
                                                              const float* __restrict__ word, int vocab,
                                                              const float* __restrict__ pos, int npos,
                                                              const float* __restrict__ type0,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, int pad,
                                                              bf16* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= M) return;
  const int b = row / L, j = row - b * L;
  const long long* idr = ids + (long long)b * L;
  int cnt = 0;
  for (int i = lane; i <= j; i += 64) cnt += idr[i] != pad;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  long long id = idr[j];
  int pid = pad + (id != pad ? cnt : 0);
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);          // ids are validated on the host; clamp keeps the read in bounds
  pid = pid >= npos ? npos - 1 : pid;
  const float* wr = word + id * C;
  const float* pr = pos + (long long)pid * C;
  const int nch = C >> 3;
  float v[LN_MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(wr + ch * 8 + 4 * h);
        const f32x4 p4 = *reinterpret_cast<const f32x4*>(pr + ch * 8 + 4 * h);
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(type0 + ch * 8 + 4 * h);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[i][4 * h + k] = a[k] + t4[k] + p4[k]; s += v[i][4 * h + k]; }
      }
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i)
    if (lane + 64 * i < nch) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float d = v[i][k] - mean; ss += d * d; }
    }
  const float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
  bf16* yr = y + (long long)row * C;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      bf16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (bf16)((v[i][k] - mean) * rstd * gamma[ch * 8 + k] + beta[ch * 8 + k]);
      *reinterpret_cast<bf16x8*>(yr + ch * 8) = o;
    }
  }
}


// ---- GroupNorm as ONE coalesced pass: the statistics were handed over by the producer(s) ---------------------------------------
// aldm_igemm (qstat_out) leaves, per M-tile and 4-channel quad, the partial (sum, sum of squares) of what it stored.  This kernel
// folds the partials of its image into per-group mean / rstd (a few KB from L2) and then streams its slab of pixels row by row
// with 16-byte accesses -- no strip-per-workgroup mapping (8 bytes out of every pixel row, 16 workgroups touching every line), no
// block reductions over the data.
constexpr int ALDM_GN_FINALIZE_MIN_TILES = 96;

// mean / rstd per (image, group) into ws [B][64][2], one workgroup per image: the stand-alone first phase of aldm_groupnorm_apply for
// images of many tiles (every apply workgroup summing every tile of its image costs more than the pixels it then normalises)
__global__ __launch_bounds__(256) void groupnorm_finalize_kernel(GnSrc s1, GnSrc s2, int HW, int groups, float eps, int lpg, float* __restrict__ ws) {
  const int tid = threadIdx.x, b = blockIdx.x;
  const int Cg = (s1.C + s2.C) / groups;
  const int g = tid / lpg, j = tid - g * lpg;
  float a = 0.f, q2 = 0.f;
  if (g < groups) gn_group_sums(s1, s2, b, HW, Cg, g, j, lpg, a, q2);
  for (int o = 1; o < lpg; o <<= 1) { a += __shfl_xor(a, o, 64); q2 += __shfl_xor(q2, o, 64); }
  if (g < groups && j == 0) {
    const float n = (float)HW * (float)Cg;
    const float mu = a / n;
    ws[(b * 64 + g) * 2] = mu;
    ws[(b * 64 + g) * 2 + 1] = rsqrtf(fmaxf(q2 / n - mu * mu, 0.f) + eps);
  }
}

__global__ __launch_bounds__(256) void groupnorm_apply_kernel(GnSrc s1, GnSrc s2, int HW, int groups, float eps,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                              bf16* __restrict__ y, int pxb, int lpg, AldmDiv dupp, const float* __restrict__ ws) {
  __shared__ float sm[64], sr[64];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int C = s1.C + s2.C, Cg = C / groups;
  if (ws) {                                                  // statistics already final (groupnorm_finalize_kernel)
    if (tid < groups) { sm[tid] = ws[(b * 64 + tid) * 2]; sr[tid] = ws[(b * 64 + tid) * 2 + 1]; }
  } else {
    // ---- statistics: lpg lanes per group, each sums a share of the image's tiles ----
    const int g = tid / lpg, j = tid - g * lpg;
    float a = 0.f, q2 = 0.f;
    if (g < groups) gn_group_sums(s1, s2, b, HW, Cg, g, j, lpg, a, q2);
    for (int o = 1; o < lpg; o <<= 1) { a += __shfl_xor(a, o, 64); q2 += __shfl_xor(q2, o, 64); }
    if (g < groups && j == 0) {
      const float n = (float)HW * (float)Cg;
      const float mu = a / n;
      sm[g] = mu;
      sr[g] = rsqrtf(fmaxf(q2 / n - mu * mu, 0.f) + eps);
    }
  }
  __syncthreads();
  // ---- apply: a thread keeps ONE 16-byte unit (8 channels) and walks down the slab's pixels, so gamma / beta and the two groups'
  //      mean / rstd become 16 registers of scale / shift, computed once (per-unit reloads of them were 4x the data traffic
  //      through L1: 20 us for a 33 MB tensor) ----
  const int upp = C >> 3;                                    // units per pixel
  const int ppp = 256 / upp;                                 // pixels per pass of the workgroup
  const int p0 = blockIdx.x * pxb, np = min(pxb, HW - p0);
  if (tid < ppp * upp) {
    const int pl0 = aldm_div(tid, dupp), cu = tid - pl0 * upp;
    const int c = cu << 3;
    const int gA = c / Cg, gB = (c + 4) / Cg;
    const float mA = sm[gA], rA = sr[gA], mB = sm[gB], rB = sr[gB];
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c), g1 = *reinterpret_cast<const f32x4*>(gamma + c + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c), b1 = *reinterpret_cast<const f32x4*>(beta + c + 4);
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sc[k] = g0[k] * rA; sh[k] = b0[k] - mA * sc[k];
      sc[4 + k] = g1[k] * rB; sh[4 + k] = b1[k] - mB * sc[4 + k];
    }
    const bool first = c < s1.C;
    const bf16* src = first ? s1.x + c : s2.x + (c - s1.C);
    const int Cs = first ? s1.C : s2.C;
    // four pixels per round, their loads issued together: a load / compute / store loop is one memory round trip per pixel, and
    // a 4000-pixel image gives a CU one workgroup (one wave per SIMD) to hide it with -- 2 TB/s
    const long long pix0 = (long long)b * HW + p0;
    for (int pl = pl0; pl < np; pl += 4 * ppp) {
      bf16x8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x8*>(src + (pix0 + min(pl + u * ppp, np - 1)) * Cs);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (pl + u * ppp < np) {
          bf16x8 o;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float t = fmaf((float)v[u][k], sc[k], sh[k]);
            if (act == ALDM_ACT_SILU) t = silu_f(t);
            o[k] = (bf16)t;
          }
          *reinterpret_cast<bf16x8*>(y + (pix0 + pl + u * ppp) * C + c) = o;
        }
      }
    }
  }
}


// ---- GroupNorm + SiLU + 3x3 convolution to a few channels, one launch (conv_norm_out -> SiLU -> conv_out at the UNet's exit) ----
// The apply pass of the norm wrote 8 MB that the 128 -> 8 channel convolution then read nine times through L2 on a 64-column tile
// with 8 live columns (9 + 14.5 us).  Here a workgroup takes RB image rows of one image: statistics from the producer's tables (as
// groupnorm_apply), the rows plus a one-pixel halo normalised + SiLU'd ONCE on their way into LDS (padding pixels are zeros, as the
// convolution pads the ACTIVATION), the weights (Cout x 9 C) next to them, and the nine taps as shifted LDS reads feeding
// v_mfma_f32_16x16x32_bf16: one image row of W = 16 pixels is one M-tile, the Cout <= 16 output channels its N.
constexpr int GC_W = 16, GC_RB = 8, GC_PIX = 272;            // 272-byte pixel stride (256 + 16): the 16 pixel lanes of a fragment read spread over all banks

template <int C>
__global__ __launch_bounds__(256) void gn_silu_conv3x3_small_kernel(GnSrc s1, int H, int groups, float eps, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, const bf16* __restrict__ w, int w_ld,
                                                                    const float* __restrict__ bias, int Cout, float* __restrict__ out, int lpg) {
  static_assert(C == 128, "tile sizes below are for 128 input channels");
  constexpr int CH = C / 8;                                   // 16-byte chunks per pixel
  constexpr int WROW = 9 * C * 2 + 16;                        // weight row stride in LDS (+16: rows on different banks)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xs = smem;                                            // [(RB + 2) x (W + 2) pixels][GC_PIX]
  char* Ws = smem + (GC_RB + 2) * (GC_W + 2) * GC_PIX;        // [16][WROW] (rows >= Cout are never read)
  __shared__ float sm[64], sr[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, r0 = blockIdx.x * GC_RB;
  const int HW = H * GC_W, Cg = C / groups;
  {   // statistics: lpg lanes per group
    const int g = tid / lpg, j = tid - g * lpg;
    float a = 0.f, q2 = 0.f;
    GnSrc none{nullptr, nullptr, 0, 1, 0};
    if (g < groups) gn_group_sums(s1, none, b, HW, Cg, g, j, lpg, a, q2);
    for (int o = 1; o < lpg; o <<= 1) { a += __shfl_xor(a, o, 64); q2 += __shfl_xor(q2, o, 64); }
    if (g < groups && j == 0) {
      const float n = (float)HW * (float)Cg;
      const float mu = a / n;
      sm[g] = mu;
      sr[g] = rsqrtf(fmaxf(q2 / n - mu * mu, 0.f) + eps);
    }
  }
  // weights -> LDS (independent of the statistics: issued before the barrier)
  for (int i = tid; i < Cout * (9 * C / 8); i += 256) {
    const int o = i / (9 * C / 8), ck = i - o * (9 * C / 8);
    *reinterpret_cast<bf16x8*>(Ws + o * WROW + ck * 16) = *reinterpret_cast<const bf16x8*>(w + (long long)o * w_ld + ck * 8);
  }
  // the strip's raw values: thread = one 16-byte channel chunk (tid & 15) of pixels (tid >> 4) + 16 j of the (RB + 2) x W strip
  constexpr int NPX = (GC_RB + 2) * GC_W, PPT = NPX / 16;     // 160 pixels, 10 per thread
  const int cu = tid & (CH - 1), c = cu * 8;
  bf16x8 v[PPT];
  bool inimg[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int pl = (tid >> 4) + 16 * j;
    const int row = r0 - 1 + pl / GC_W, col = pl % GC_W;
    inimg[j] = row >= 0 && row < H;
    const int rr = min(max(row, 0), H - 1);
    v[j] = *reinterpret_cast<const bf16x8*>(s1.x + ((long long)b * HW + (long long)rr * GC_W + col) * C + c);
  }
  // zero the two padding columns of every strip row
  for (int i = tid; i < (GC_RB + 2) * 2 * CH; i += 256) {
    const int rowi = i / (2 * CH), rem = i - rowi * 2 * CH;
    const int colp = (rem / CH) ? (GC_W + 1) : 0;
    *reinterpret_cast<uint4*>(Xs + (rowi * (GC_W + 2) + colp) * GC_PIX + (rem % CH) * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();                                            // statistics ready
  {
    const int gA = c / Cg, gB = (c + 4) / Cg;
    const float mA = sm[gA], rA = sr[gA], mB = sm[gB], rB = sr[gB];
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c), g1 = *reinterpret_cast<const f32x4*>(gamma + c + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c), b1 = *reinterpret_cast<const f32x4*>(beta + c + 4);
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sc[k] = g0[k] * rA; sh[k] = b0[k] - mA * sc[k];
      sc[4 + k] = g1[k] * rB; sh[4 + k] = b1[k] - mB * sc[4 + k];
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int pl = (tid >> 4) + 16 * j;
      const int rowi = pl / GC_W, col = pl % GC_W;
      bf16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = inimg[j] ? (bf16)silu_f(fmaf((float)v[j][k], sc[k], sh[k])) : (bf16)0.f;
      *reinterpret_cast<bf16x8*>(Xs + (rowi * (GC_W + 2) + col + 1) * GC_PIX + cu * 16) = o;
    }
  }
  __syncthreads();
  // ---- 9 taps x C / 32 k-steps; wave = 2 image rows (2 M-tiles), lane (m = lane & 15: pixel / output channel, q = lane >> 4: 8-wide k chunk)
  const int m = lane & 15, q = lane >> 4;
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3, dx = tap % 3;                     // strip row = image row - r0 + 1 + (dy - 1); column + 1 + (dx - 1)
#pragma unroll
    for (int ks = 0; ks < C / 32; ++ks) {
      const bf16x8 wf = m < Cout ? *reinterpret_cast<const bf16x8*>(Ws + m * WROW + (tap * C + ks * 32 + q * 8) * 2) : zero8;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int rowi = 2 * wave + t + dy;
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(Xs + (rowi * (GC_W + 2) + m + dx) * GC_PIX + (ks * 32 + q * 8) * 2);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, wf, acc[t], 0, 0, 0);
      }
    }
  }
  // D[pixel 4 q + i][channel m]
  if (m < Cout) {
    const float bb = bias ? bias[m] : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = r0 + 2 * wave + t;
      if (row < H) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          out[((long long)b * HW + (long long)row * GC_W + 4 * q + i) * Cout + m] = acc[t][i] + bb;
      }
    }
  }
}

}  // namespace

extern "C" int aldm_groupnorm(const void* x, const void* x2, int B, int HW, int C1, int C2, int groups, float eps,
                              const float* gamma, const float* beta, int act, void* y, void* stream) {
  ALDM_CHECK_ARG(x && y && gamma && beta, "groupnorm: null pointer");
  ALDM_CHECK_ARG(B > 0 && HW > 0 && C1 > 0 && C2 >= 0 && groups > 0, "groupnorm: bad dims");
  const int C = C1 + C2;
  ALDM_CHECK_ARG(C % groups == 0 && (C / groups) % 4 == 0 && C1 % 4 == 0, "groupnorm: group width %d / C1 %d must be multiples of 4", C / groups, C1);
  ALDM_CHECK_ARG(C2 == 0 || x2, "groupnorm: C2 without x2");
  const size_t lds = (size_t)GN_LDS_QUADS * 8 + 64;
  static unsigned long long attr_done = 0;   // per-device bit mask (aldm_set_max_lds)
  if (int rc = aldm_set_max_lds(reinterpret_cast<const void*>(groupnorm_kernel), (int)lds, &attr_done, "groupnorm")) return rc;
  const long long nquads = (long long)HW * (C / groups / 4);
  const AldmDiv dq = aldm_make_div((unsigned)(C / groups / 4));
#define ALDM_GN_REG(QPT)                                                                                               \
  hipLaunchKernelGGL(groupnorm_reg_kernel<QPT>, dim3(B * groups), dim3(GN_THREADS), 0, (hipStream_t)stream,            \
                     (const bf16*)x, (const bf16*)x2, HW, C1, C2, groups, eps, gamma, beta, act, (bf16*)y, dq)
  const bool reg_ok = C / groups <= GN_MAXCG;                 // (the register-strip kernels keep the group's gamma / beta in LDS)
  if (reg_ok && nquads <= 4 * GN_THREADS) ALDM_GN_REG(4);
  else if (reg_ok && nquads <= 8 * GN_THREADS) ALDM_GN_REG(8);
  else if (reg_ok && nquads <= 16 * GN_THREADS) ALDM_GN_REG(16);
  else if (reg_ok && nquads <= 32 * GN_THREADS) ALDM_GN_REG(32);
  else
    hipLaunchKernelGGL(groupnorm_kernel, dim3(B * groups), dim3(GN_THREADS), lds, (hipStream_t)stream, (const bf16*)x,
                       (const bf16*)x2, HW, C1, C2, groups, eps, gamma, beta, act, (bf16*)y, dq);
#undef ALDM_GN_REG
  return aldm_launch_status("groupnorm");
}

extern "C" int aldm_groupnorm_partials(const float* ws, int splits, int B, int HW, int C, const float* bias,
                                       const float* rowbias, int rowbias_ld, const void* res, void* sum_out,
                                       const void* x2, int C2, int groups, float eps, const float* gamma, const float* beta,
                                       int act, void* y, void* stream) {
  ALDM_CHECK_ARG(ws && y && gamma && beta && splits >= 1, "groupnorm_partials: null pointer / bad splits");
  ALDM_CHECK_ARG(B > 0 && HW > 0 && C > 0 && C2 >= 0 && groups > 0 && (C2 == 0 || x2), "groupnorm_partials: bad dims");
  const int Ct = C + C2;
  ALDM_CHECK_ARG(Ct % groups == 0 && (Ct / groups) % 4 == 0 && C % (Ct / groups) == 0,
                 "groupnorm_partials: group width %d must be a multiple of 4 and divide the first source's %d channels", Ct / groups, C);
  ALDM_CHECK_ARG(!rowbias || rowbias_ld >= C, "groupnorm_partials: rowbias_ld");
  ALDM_CHECK_ARG(Ct / groups <= GN_MAXCG, "groupnorm_partials: at most %d channels per group", GN_MAXCG);
  const long long nquads = (long long)HW * (Ct / groups / 4);
  ALDM_CHECK_ARG(nquads <= 8 * GN_THREADS, "groupnorm_partials: strip of %lld quads exceeds the register-resident limit %d", nquads, 8 * GN_THREADS);
  const AldmDiv dq = aldm_make_div((unsigned)(Ct / groups / 4));
  const long long sstride = (long long)B * HW * C;
#define ALDM_GNP(QPT)                                                                                                  \
  hipLaunchKernelGGL(groupnorm_partials_kernel<QPT>, dim3(B * groups), dim3(GN_THREADS), 0, (hipStream_t)stream, ws,   \
                     splits, sstride, HW, C, groups, eps, bias, rowbias, rowbias_ld, (const bf16*)res, (bf16*)sum_out,        \
                     (const bf16*)x2, C2, gamma, beta, act, (bf16*)y, dq)
  if (nquads <= GN_THREADS) ALDM_GNP(1);
  else if (nquads <= 2 * GN_THREADS) ALDM_GNP(2);
  else if (nquads <= 4 * GN_THREADS) ALDM_GNP(4);
  else ALDM_GNP(8);
#undef ALDM_GNP
  return aldm_launch_status("groupnorm_partials");
}

extern "C" int aldm_layernorm(const void* x, int M, int C, const float* gamma, const float* beta, float eps, void* y,
                              void* stream) {
  ALDM_CHECK_ARG(x && y && gamma && beta && M > 0, "layernorm: null pointer / bad M");
  ALDM_CHECK_ARG(C % 8 == 0 && C <= 64 * 8 * LN_MAXC, "layernorm: C=%d must be a multiple of 8 and <= %d", C, 64 * 8 * LN_MAXC);
  if (C <= 256)
    hipLaunchKernelGGL(layernorm_kernel<32>, dim3(cdiv(M, 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, M, C, gamma,
                       beta, eps, (bf16*)y);
  else
    hipLaunchKernelGGL(layernorm_kernel<64>, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, M, C, gamma,
                       beta, eps, (bf16*)y);
  return aldm_launch_status("layernorm");
}

extern "C" int aldm_embed_layernorm(const long long* ids, int B, int L, int C, const float* word, int vocab,
                                    const float* pos, int npos, const float* type0, const float* gamma,
                                    const float* beta, float eps, int pad_idx, void* y, void* stream) {
  ALDM_CHECK_ARG(ids && word && pos && type0 && gamma && beta && y && B > 0 && L > 0, "embed_layernorm: null pointer / bad dims");
  ALDM_CHECK_ARG(C % 8 == 0 && C <= 64 * 8 * LN_MAXC, "embed_layernorm: C=%d must be a multiple of 8 and <= %d", C, 64 * 8 * LN_MAXC);
  ALDM_CHECK_ARG(vocab > 0 && npos > pad_idx + 1 && L + pad_idx + 1 <= npos, "embed_layernorm: %d tokens do not fit %d positions (pad %d)", L, npos, pad_idx);
  const int M = B * L;
  hipLaunchKernelGGL(embed_layernorm_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, ids, M, L, C, word, vocab,
                     pos, npos, type0, gamma, beta, eps, pad_idx, (bf16*)y);
  return aldm_launch_status("embed_layernorm");
}

static int gn_apply_bytes() {
  static const int v = getenv("ALDM_GN_APPLY_BYTES") ? atoi(getenv("ALDM_GN_APPLY_BYTES")) : 32768;
  return v;
}

extern "C" int aldm_groupnorm_apply(const void* x, const float* qstat, int bm, int tpi, const void* x2, const float* qstat2, int bm2,
                                    int tpi2, int B, int HW, int C1, int C2, int groups, float eps, const float* gamma,
                                    const float* beta, int act, void* y, float* stat_ws, void* stream) {
  ALDM_CHECK_ARG(x && qstat && y && gamma && beta, "groupnorm_apply: null pointer");
  ALDM_CHECK_ARG(B > 0 && HW > 0 && C1 > 0 && C2 >= 0 && groups > 0 && groups <= 64 && (C2 == 0 || (x2 && qstat2)), "groupnorm_apply: bad dims");
  const int C = C1 + C2;
  ALDM_CHECK_ARG(C % groups == 0 && (C / groups) % 4 == 0 && C1 % 8 == 0 && C2 % 8 == 0,
                 "groupnorm_apply: group width %d must be a multiple of 4; C1 = %d, C2 = %d multiples of 8", C / groups, C1, C2);
  ALDM_CHECK_ARG(C / 8 <= 256, "groupnorm_apply: at most 2048 channels (a workgroup walks %d 16-byte units per pixel with 256 threads)", C / 8);
  ALDM_CHECK_ARG((tpi > 0 || (bm > 0 && bm <= HW)) && (C2 == 0 || tpi2 > 0 || (bm2 > 0 && bm2 <= HW)),
                 "groupnorm_apply: an M-tile of the producer may span at most two images (tile rows <= H*W)");
  int lpg = 1;
  while (lpg * 2 * groups <= 256 && lpg < 64) lpg *= 2;      // lanes per group (power of two, whole groups inside one wave)
  int pxb = 16;
  while (pxb < 256 && pxb * C * 2 < gn_apply_bytes()) pxb *= 2;   // ~32 KB of pixels per workgroup (ALDM_GN_APPLY_BYTES: tuning aid)
  GnSrc s1{(const bf16*)x, qstat, C1, bm, tpi}, s2{(const bf16*)x2, qstat2, C2, bm2, tpi2};
  // tiles per image the statistics phase walks; past ALDM_GN_FINALIZE_MIN_TILES a one-workgroup-per-image launch sums them once
  const int tiles = max(tpi > 0 ? tpi : HW / bm + 1, C2 ? (tpi2 > 0 ? tpi2 : HW / bm2 + 1) : 0);
  const bool two_phase = stat_ws && tiles >= ALDM_GN_FINALIZE_MIN_TILES;
  if (two_phase) {
    hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, s1, s2, HW, groups, eps, lpg, stat_ws);
    if (int rc = aldm_launch_status("groupnorm_finalize")) return rc;
  }
  hipLaunchKernelGGL(groupnorm_apply_kernel, dim3(cdiv(HW, pxb), B), dim3(256), 0, (hipStream_t)stream, s1, s2, HW, groups, eps, gamma,
                     beta, act, (bf16*)y, pxb, lpg, aldm_make_div((unsigned)(C >> 3)), two_phase ? (const float*)stat_ws : nullptr);
  return aldm_launch_status("groupnorm_apply");
}

extern "C" int aldm_gn_silu_conv3x3_small(const void* x, const float* qstat, int bm, int tpi, int B, int H, int W, int C, int groups,
                                          float eps, const float* gamma, const float* beta, const void* w, int w_ld, const float* bias,
                                          int Cout, float* out, void* stream) {
  ALDM_CHECK_ARG(x && qstat && gamma && beta && w && out, "gn_silu_conv3x3_small: null pointer");
  ALDM_CHECK_ARG(B > 0 && H > 0 && W == GC_W && C == 128 && Cout >= 1 && Cout <= 16 && w_ld >= 9 * C && w_ld % 8 == 0,
                 "gn_silu_conv3x3_small: built for W = 16, C = 128, Cout <= 16 (got W %d, C %d, Cout %d)", W, C, Cout);
  ALDM_CHECK_ARG(groups > 0 && groups <= 64 && C % groups == 0 && (C / groups) % 4 == 0, "gn_silu_conv3x3_small: bad groups");
  ALDM_CHECK_ARG(tpi > 0 || (bm > 0 && bm <= H * W), "gn_silu_conv3x3_small: an M-tile of the producer may span at most two images");
  int lpg = 1;
  while (lpg * 2 * groups <= 256 && lpg < 64) lpg *= 2;
  const int lds = (GC_RB + 2) * (GC_W + 2) * GC_PIX + 16 * (9 * 128 * 2 + 16);
  auto kern = gn_silu_conv3x3_small_kernel<128>;
  static unsigned long long attr_done = 0;
  if (int rc = aldm_set_max_lds(reinterpret_cast<const void*>(kern), lds, &attr_done, "gn_silu_conv3x3_small")) return rc;
  GnSrc s1{(const bf16*)x, qstat, C, bm, tpi};
  hipLaunchKernelGGL(kern, dim3(cdiv(H, GC_RB), B), dim3(256), lds, (hipStream_t)stream, s1, H, groups, eps, gamma, beta, (const bf16*)w, w_ld,
                     bias, Cout, out, lpg);
  return aldm_launch_status("gn_silu_conv3x3_small");
}

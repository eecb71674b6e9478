// Backward / training kernels of the LoRA fine-tune step on gfx950 (configs 3/4):
// the loop body [REF script/train/train_audioldm_lora.py:499-565]  add_noise -> UNet -> MSE -> backward -> AdamW,
// with the base model frozen [REF train:374-376]: only dX flows through the frozen ops and only the LoRA
// A/B matrices receive weight gradients [REF train:378-385].
//
//   groupnorm_bwd / layernorm_bwd   dX of F.group_norm(+SiLU) / F.layer_norm (statistics recomputed in registers)
//   geglu_fwd / geglu_bwd           GEGLU kept un-fused in training so the projection is available to the backward
//   upsample_nearest_bwd            adjoint of the nearest up-sampling folded into the up-sampler convs
//   tn_small                        dA = U^T X and dB = dY^T T: rank-r "reduce over tokens" products, scattered straight
//                                   into the flat fp32 LoRA gradient buffer that RCCL all-reduces
//   lora_pack                       flat fp32 LoRA parameters -> the bf16 packed operands of the fused GEMMs (one launch)
//   transpose_tokens                [B*N][C] row-major -> [B][C][Npad] token-major (attention operands)
//   mse_grad                        eps-prediction MSE loss value and its gradient
#include "common.h"

namespace {

constexpr int GN_THREADS = 512;

constexpr int GNB_MAXCG = 256;      // channels per group groupnorm_bwd stages gamma / beta for

template <int NT = GN_THREADS>
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red, int tid) {
  a = wave_sum(a);
  b = wave_sum(b);
  __syncthreads();
  if ((tid & 63) == 0) { red[tid >> 6] = a; red[16 + (tid >> 6)] = b; }
  __syncthreads();
  float ta = 0.f, tb = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) { ta += red[i]; tb += red[16 + i]; }
  a = ta; b = tb;
}

// y = act(gn(x)) ; given dy -> dx.  One workgroup per (image, group), strip in registers (<= NT*QPT quads).  The long strips
// (4096 pixels) run with NT = 1024 threads: half the registers per thread, 16 waves per workgroup to hide the strided 8-byte loads
// (a strip is Cg*2 bytes out of every pixel row) -- measured 121 -> 75 us at 4096 x 384 (34 -> 31, 25 -> 23 us for the two shorter long-strip shapes).
template <int QPT, int NT = GN_THREADS>
__global__ __launch_bounds__(NT) void groupnorm_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ x2,
                                                                   const bf16* __restrict__ dy, int HW, int C1, int C2,
                                                                   int groups, float eps, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, int act,
                                                                   bf16* __restrict__ dx, bf16* __restrict__ dx2,
                                                                   const bf16* dx_acc, const bf16* dx2_acc, AldmDiv dqpp,
                                                                   const float* __restrict__ dy_ws, int dy_splits, long long dy_sstride) {
  // dy_ws: dY still as the split-K partial tiles [dy_splits][B*HW][C] (fp32) of the dX convolution that produced it
  // (aldm_igemm defer_reduce): summed here in split order and rounded to bf16, exactly what igemm_reduce_kernel would store
  __shared__ float red[32];
  __shared__ __attribute__((aligned(16))) float sgm[GNB_MAXCG], sbt[GNB_MAXCG];
  const int C = C1 + C2;
  const int Cg = C / groups, qpp = Cg >> 2;
  const int nb = gridDim.x / groups;
  const int g = blockIdx.x / nb, b = blockIdx.x - g * nb;
  const int c0 = g * Cg;
  const int nquads = HW * qpp;
  const int tid = threadIdx.x;
  // the group's gamma / beta through LDS (visible after the first reduction's barrier): read per element from global they were
  // 4-byte loads with a wait each, in both sweeps
  if (tid < Cg) { sgm[tid] = gamma[c0 + tid]; sbt[tid] = beta[c0 + tid]; }
  const float pivot = (float)((c0 < C1) ? x[(long long)b * HW * C1 + c0] : x2[(long long)b * HW * C2 + (c0 - C1)]);

  bf16x4 v[QPT], d[QPT], pa[QPT];                       // pa: the gradient x / x2 already hold (dx_acc / dx2_acc), requested with the strip
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    const int q = tid + i * NT;
    bf16x4 t = {0, 0, 0, 0}, u = {0, 0, 0, 0};
    pa[i] = bf16x4{0, 0, 0, 0};
    if (q < nquads) {
      const int pix = aldm_div(q, dqpp), j = q - pix * qpp;
      const int c = c0 + 4 * j;
      t = (c < C1) ? *reinterpret_cast<const bf16x4*>(x + ((long long)b * HW + pix) * C1 + c)
                   : *reinterpret_cast<const bf16x4*>(x2 + ((long long)b * HW + pix) * C2 + (c - C1));
      if (c < C1) { if (dx_acc) pa[i] = *reinterpret_cast<const bf16x4*>(dx_acc + ((long long)b * HW + pix) * C1 + c); }
      else if (dx2 && dx2_acc) pa[i] = *reinterpret_cast<const bf16x4*>(dx2_acc + ((long long)b * HW + pix) * C2 + (c - C1));
      const long long doff = ((long long)b * HW + pix) * C + c;
      if (dy_ws) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        const float* w0 = dy_ws + doff;
        int sp = 0;
        for (; sp + 4 <= dy_splits; sp += 4) {               // four partials in flight per trip, summed in split order
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(w0 + sp * dy_sstride), a1 = *reinterpret_cast<const f32x4*>(w0 + (sp + 1) * dy_sstride);
          const f32x4 a2 = *reinterpret_cast<const f32x4*>(w0 + (sp + 2) * dy_sstride), a3 = *reinterpret_cast<const f32x4*>(w0 + (sp + 3) * dy_sstride);
#pragma unroll
          for (int k = 0; k < 4; ++k) a[k] = (((a[k] + a0[k]) + a1[k]) + a2[k]) + a3[k];
        }
        for (; sp < dy_splits; ++sp) a += *reinterpret_cast<const f32x4*>(w0 + sp * dy_sstride);
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = (bf16)a[k];
      } else {
        u = *reinterpret_cast<const bf16x4*>(dy + doff);
      }
    }
    v[i] = t;
    d[i] = u;
  }
  // mean and variance from one reduction, shifted by the strip's first element (see groupnorm_reg_kernel in norm.hip)
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < QPT; ++i)
    if (tid + i * NT < nquads) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float e = (float)v[i][k] - pivot; s += e; ss = fmaf(e, e, ss); }
    }
  const float n = (float)nquads * 4.f;
  block_sum2<NT>(s, ss, red, tid);
  const float dm = s / n;
  const float mean = pivot + dm;
  const float rstd = rsqrtf(fmaxf(ss / n - dm * dm, 0.f) + eps);

  // gd = dz * gamma: short strips keep it in fp32 registers for the second sweep (the SiLU derivative costs an exp and a
  // reciprocal per element); for the long ones that would double the footprint, so they recompute it
  constexpr bool KEEP = QPT <= 8;
  f32x4 gdk[KEEP ? QPT : 1];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    const int q = tid + i * NT;
    if constexpr (KEEP) gdk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q < nquads) {
      const int j = q - aldm_div(q, dqpp) * qpp;
      const f32x4 gm = *reinterpret_cast<const f32x4*>(sgm + 4 * j), bt = *reinterpret_cast<const f32x4*>(sbt + 4 * j);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = ((float)v[i][k] - mean) * rstd;
        float dz = (float)d[i][k];
        if (act == ALDM_ACT_SILU) {
          const float z = xh * gm[k] + bt[k];
          const float sg = sigmoid_f(z);
          dz *= sg * (1.f + z * (1.f - sg));
        }
        const float gd = dz * gm[k];
        if constexpr (KEEP) gdk[i][k] = gd;
        s1 += gd;
        s2 += gd * xh;
      }
    }
  }
  block_sum2<NT>(s1, s2, red, tid);
  s1 /= n;
  s2 /= n;
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    const int q = tid + i * NT;
    if (q < nquads) {
      const int pix = aldm_div(q, dqpp), j = q - pix * qpp;
      const int c = c0 + 4 * j;
      const f32x4 gm = *reinterpret_cast<const f32x4*>(sgm + 4 * j), bt = *reinterpret_cast<const f32x4*>(sbt + 4 * j);
      bf16x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = ((float)v[i][k] - mean) * rstd;
        float gd;
        if constexpr (KEEP) {
          gd = gdk[i][k];
        } else {
          float dz = (float)d[i][k];
          if (act == ALDM_ACT_SILU) {
            const float z = xh * gm[k] + bt[k];
            const float sg = sigmoid_f(z);
            dz *= sg * (1.f + z * (1.f - sg));
          }
          gd = dz * gm[k];
        }
        o[k] = (bf16)(rstd * (gd - s1 - xh * s2));
      }
      // dx_acc / dx2_acc: the gradient x / x2 already received from their other consumers (residual / skip joins) -- added
      // here instead of in a separate launch.  They may alias dx / dx2 (each element is read and written by one thread).
      if (c < C1) {
        const long long off = ((long long)b * HW + pix) * C1 + c;
        if (dx_acc) { for (int k = 0; k < 4; ++k) o[k] = (bf16)((float)o[k] + (float)pa[i][k]); }
        *reinterpret_cast<bf16x4*>(dx + off) = o;
      } else if (dx2) {
        const long long off = ((long long)b * HW + pix) * C2 + (c - C1);
        if (dx2_acc) { for (int k = 0; k < 4; ++k) o[k] = (bf16)((float)o[k] + (float)pa[i][k]); }
        *reinterpret_cast<bf16x4*>(dx2 + off) = o;
      }
    }
  }
}

constexpr int LN_MAXC = 4;
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// LPR lanes per row: a full wave, or half a wave for rows of <= 32 16-byte chunks (C <= 256), as in norm.hip
template <int LPR>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, int M,
                                                            int C, const float* __restrict__ gamma, float eps,
                                                            bf16* __restrict__ dx, const bf16* dx_acc) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & (LPR - 1);
  const int row = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RPW + ((threadIdx.x & 63) / LPR);
  const bool live = row < M;                       // every lane stays in the shuffles
  const int nch = C >> 3;
  const bf16* xr = x + (long long)(live ? row : 0) * C;
  const bf16* dr = dy + (long long)(live ? row : 0) * C;
  bf16x8 v[LN_MAXC], d[LN_MAXC], acc[LN_MAXC];
  float gm[LN_MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int ch = lane + LPR * i;
    if (ch < nch) {
      v[i] = *reinterpret_cast<const bf16x8*>(xr + ch * 8);
      d[i] = *reinterpret_cast<const bf16x8*>(dr + ch * 8);
      // (the gradient the residual stream already holds: requested with the row, not as a load -> wait -> store tail)
      if (dx_acc) acc[i] = *reinterpret_cast<const bf16x8*>(dx_acc + (long long)(live ? row : 0) * C + ch * 8);
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + ch * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + ch * 8 + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { gm[i][k] = g0[k]; gm[i][4 + k] = g1[k]; }
#pragma unroll
      for (int k = 0; k < 8; ++k) s += (float)v[i][k];
    }
  }
  const float mean = row_sum<LPR>(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i)
    if (lane + LPR * i < nch) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float e = (float)v[i][k] - mean; ss += e * e; }
    }
  const float rstd = rsqrtf(row_sum<LPR>(ss) / (float)C + eps);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    if (lane + LPR * i < nch) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = ((float)v[i][k] - mean) * rstd;
        const float gd = (float)d[i][k] * gm[i][k];
        s1 += gd;
        s2 += gd * xh;
      }
    }
  }
  s1 = row_sum<LPR>(s1) / (float)C;
  s2 = row_sum<LPR>(s2) / (float)C;
  if (!live) return;
  bf16* o = dx + (long long)row * C;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int ch = lane + LPR * i;
    if (ch < nch) {
      bf16x8 r;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = ((float)v[i][k] - mean) * rstd;
        r[k] = (bf16)(rstd * ((float)d[i][k] * gm[i][k] - s1 - xh * s2));
      }
      if (dx_acc) {                            // the residual stream's gradient from its other consumers (may alias dx: each element
#pragma unroll                                 //  is read and written by one thread)
        for (int k = 0; k < 8; ++k) r[k] = (bf16)((float)r[k] + (float)acc[i][k]);
      }
      *reinterpret_cast<bf16x8*>(o + ch * 8) = r;
    }
  }
}

// h [M][2I] in blocks of (16 value | 16 gate) (the packing of ops.pack_geglu) -> out [M][I] = value * gelu_erf(gate)
__global__ void geglu_fwd_kernel(const bf16* __restrict__ h, long long M, int I, bf16* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over M * I / 4
  const int q = I >> 2;
  if (idx >= M * q) return;
  const long long m = idx / q;
  const int n = (int)(idx - m * q) * 4;
  const int nv = ((n >> 4) << 5) + (n & 15);
  const bf16x4 a = *reinterpret_cast<const bf16x4*>(h + m * 2 * I + nv);
  const bf16x4 g = *reinterpret_cast<const bf16x4*>(h + m * 2 * I + nv + 16);
  bf16x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (bf16)((float)a[k] * gelu_erf_f((float)g[k]));
  *reinterpret_cast<bf16x4*>(out + m * I + n) = o;
}

__global__ void geglu_bwd_kernel(const bf16* __restrict__ h, const bf16* __restrict__ dout, long long M, int I,
                                 bf16* __restrict__ dh) {
  // 4 outputs per thread: the kernel is VALU-bound (erf + exp per element), and 8 per thread (16-byte accesses) measured 9.6 us
  // against 9.0 at M = 8192 -- fewer waves to overlap the transcendental chains
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = I >> 2;
  if (idx >= M * q) return;
  const long long m = idx / q;
  const int n = (int)(idx - m * q) * 4;
  const int nv = ((n >> 4) << 5) + (n & 15);
  const bf16x4 a = *reinterpret_cast<const bf16x4*>(h + m * 2 * I + nv);
  const bf16x4 g = *reinterpret_cast<const bf16x4*>(h + m * 2 * I + nv + 16);
  const bf16x4 d = *reinterpret_cast<const bf16x4*>(dout + m * I + n);
  bf16x4 da, dg;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float gv = (float)g[k], dv = (float)d[k];
    const float cdf = 0.5f * (1.f + erf_as_f(gv * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * gv * gv);
    da[k] = (bf16)(dv * gv * cdf);
    dg[k] = (bf16)(dv * (float)a[k] * (cdf + gv * pdf));
  }
  *reinterpret_cast<bf16x4*>(dh + m * 2 * I + nv) = da;
  *reinterpret_cast<bf16x4*>(dh + m * 2 * I + nv + 16) = dg;
}

__global__ void add_bf16_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, long long n, bf16* __restrict__ c) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i + 8 <= n) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(a + i), y = *reinterpret_cast<const bf16x8*>(b + i);
    bf16x8 z;
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = (bf16)((float)x[k] + (float)y[k]);
    *reinterpret_cast<bf16x8*>(c + i) = z;
  } else {
    for (long long j = i; j < n; ++j) c[j] = (bf16)((float)a[j] + (float)b[j]);
  }
}

// dx[b][ih][iw][c] = sum of dy over the nearest-neighbour preimage of (ih, iw)
__global__ void upsample_nearest_bwd_kernel(const bf16* __restrict__ dy, int B, int IH, int IW, int OH, int OW, int C,
                                            bf16* __restrict__ dx) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over B*IH*IW*C/8
  const int cq = C >> 3;
  if (idx >= (long long)B * IH * IW * cq) return;
  const int c = (int)(idx % cq) * 8;
  long long t = idx / cq;
  const int iw = (int)(t % IW); t /= IW;
  const int ih = (int)(t % IH);
  const int b = (int)(t / IH);
  const int oh0 = (ih * OH + IH - 1) / IH, oh1 = ((ih + 1) * OH + IH - 1) / IH;
  const int ow0 = (iw * OW + IW - 1) / IW, ow1 = ((iw + 1) * OW + IW - 1) / IW;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int oh = oh0; oh < oh1; ++oh)
    for (int ow = ow0; ow < ow1; ++ow) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(dy + (((long long)b * OH + oh) * OW + ow) * C + c);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += (float)v[k];
    }
  bf16x8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (bf16)acc[k];
  *reinterpret_cast<bf16x8*>(dx + (((long long)b * IH + ih) * IW + iw) * C + c) = o;
}

// out[p][q] += sum_m P[m][p] * Q[m][q]  (P: [M][Rp] bf16, Q: [M][ldq] bf16 columns q0..q0+Qc), rank rows p < Rp <= 64.
// Row p of the product is scattered by `rows[p]`: dst[(q - qlo) * qstride] += scale * value for q in [qlo, qhi).
struct TnRow { float* dst; int qlo, qhi, qstride; float scale; };
template <int RP>
__global__ __launch_bounds__(256) void tn_small_kernel(const bf16* __restrict__ P, const bf16* __restrict__ Q, int M, int ldq,
                                                       int Qc, const TnRow* __restrict__ rows, int m_per_block) {
  constexpr int PR = RP / 4;                 // rank rows per thread
  __shared__ float Ps[32][RP];
  __shared__ float Qs[32][64];
  const int tid = threadIdx.x, tq = tid & 63, tp = tid >> 6;
  const int q0 = blockIdx.x * 64;
  const int mb = blockIdx.y * m_per_block, me = min(M, mb + m_per_block);
  float acc[PR];
#pragma unroll
  for (int i = 0; i < PR; ++i) acc[i] = 0.f;
  for (int m0 = mb; m0 < me; m0 += 32) {
    __syncthreads();
    for (int i = tid; i < 32 * RP; i += 256) {
      const int mm = i / RP, pp = i - mm * RP;
      Ps[mm][pp] = (m0 + mm < me) ? (float)P[(long long)(m0 + mm) * RP + pp] : 0.f;
    }
    for (int i = tid; i < 32 * 64; i += 256) {
      const int mm = i >> 6, qq = i & 63;
      Qs[mm][qq] = (m0 + mm < me && q0 + qq < Qc) ? (float)Q[(long long)(m0 + mm) * ldq + q0 + qq] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int mm = 0; mm < 32; ++mm) {
      const float qv = Qs[mm][tq];
#pragma unroll
      for (int i = 0; i < PR; ++i) acc[i] += Ps[mm][tp * PR + i] * qv;
    }
  }
  const int q = q0 + tq;
  if (q < Qc) {
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      const TnRow r = rows[tp * PR + i];
      if (r.dst && q >= r.qlo && q < r.qhi) atomicAdd(r.dst + (long long)(q - r.qlo) * r.qstride, acc[i] * r.scale);
    }
  }
}

// MFMA form of the same product for the aligned case (ldq, Qc multiples of 8): out[p][q] = sum_m P[m][p] Q[m][q] reduces over
// the ROW index of two row-major matrices, so both MFMA operands are k-strided in memory.  Tiles are staged row-major in LDS
// as [32 m][32 col] slabs with 64-byte rows and consumed through gfx950's transposing LDS read (ds_read_b64_tr_b16: each
// 16-lane group fetches 4 rows x 16 columns and receives them column-major), which is conflict-free on 64-byte rows (a
// 32-lane half reads 4 x 64 contiguous bytes).  A workgroup owns 128 q-columns (one 32-column slab per wave) and an m-range;
// partial sums leave through the same fp32 atomics / row table as the scalar kernel.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 tr_frag(const char* slab, int lane, int kbase) {
  // 32x32x16 operand for row/column (lane & 31): k = kbase + 8*(lane>>5) .. +7
  const int g = lane >> 4, qrow = (lane >> 2) & 3, pp = lane & 3;
  const char* a = slab + (kbase + 8 * (g >> 1) + qrow) * 64 + (16 * (g & 1) + 4 * pp) * 2;
  typedef s16x4 __attribute__((address_space(3))) lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * 64));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int RP>
__device__ __forceinline__ void tn_mfma_body(const bf16* __restrict__ P, const bf16* __restrict__ Q, int M, int ldq, int Qc,
                                             const TnRow* __restrict__ rows, int m_per_block, int bx, int by, char* smem) {
  constexpr int PT = RP / 32;                    // 32-row p tiles
  constexpr int SLAB = 32 * 64;                  // bytes of one [32 m][32 col] bf16 slab
  constexpr int STAGE = (PT + 4) * SLAB;
  constexpr int PCH = PT * 128;                  // 16-byte chunks of the P tile per stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = bx * 128;
  const int mb = by * m_per_block, me = min(M, mb + m_per_block);
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8 qreg[2], preg;
  auto prefetch = [&](int m0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i, row = idx >> 4, ch = idx & 15;
      const int m = m0 + row, q = q0 + ch * 8;
      qreg[i] = (m < me && q < Qc) ? *reinterpret_cast<const bf16x8*>(Q + (long long)m * ldq + q) : zero8;
    }
    if (tid < PCH) {
      const int row = tid / (PT * 4), ch = tid - row * (PT * 4);
      preg = (m0 + row < me) ? *reinterpret_cast<const bf16x8*>(P + (long long)(m0 + row) * RP + ch * 8) : zero8;
    }
  };
  auto stage = [&](int buf) {
    char* st = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i, row = idx >> 4, ch = idx & 15;
      *reinterpret_cast<bf16x8*>(st + (PT + (ch >> 2)) * SLAB + row * 64 + (ch & 3) * 16) = qreg[i];
    }
    if (tid < PCH) {
      const int row = tid / (PT * 4), ch = tid - row * (PT * 4);
      *reinterpret_cast<bf16x8*>(st + (ch >> 2) * SLAB + row * 64 + (ch & 3) * 16) = preg;
    }
  };
  f32x16 acc[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const int nst = (me - mb + 31) / 32;
  prefetch(mb);
  stage(0);
  __syncthreads();
  for (int s = 0; s < nst; ++s) {
    if (s + 1 < nst) prefetch(mb + 32 * (s + 1));
    const char* st = smem + (s & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 bq = tr_frag(st + (PT + wave) * SLAB, lane, 16 * ks);
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        const bf16x8 ap = tr_frag(st + t * SLAB, lane, 16 * ks);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap, bq, acc[t], 0, 0, 0);
      }
    }
    if (s + 1 < nst) stage((s + 1) & 1);
    __syncthreads();
  }
  const int q = q0 + wave * 32 + (lane & 31), hh = lane >> 5;
  if (q < Qc) {
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const TnRow r = rows[t * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh];
        if (r.dst && q >= r.qlo && q < r.qhi) atomicAdd(r.dst + (long long)(q - r.qlo) * r.qstride, acc[t][i] * r.scale);
      }
  }
}

template <int RP>
__global__ __launch_bounds__(256) void tn_mfma_kernel(const bf16* __restrict__ P, const bf16* __restrict__ Q, int M, int ldq,
                                                      int Qc, const TnRow* __restrict__ rows, int m_per_block) {
  __shared__ __attribute__((aligned(16))) char smem[2 * (RP / 32 + 4) * 32 * 64];
  tn_mfma_body<RP>(P, Q, M, ldq, Qc, rows, m_per_block, blockIdx.x, blockIdx.y, smem);
}

// All LoRA-gradient products of one training step in ONE launch: the tape defers its 2 x 64 rank-r products to the end of
// the backward pass and hands them over as a job table; a workgroup finds its job by binary search over the jobs' first
// workgroup ids.  One launch instead of 128 (each ~2.5 us of launch floor plus ramp-up for ~1 us of HBM traffic), and the
// whole GPU streams X / dY at once.
struct TnJob { const bf16* P; const bf16* Q; const TnRow* rows; int M, ldq, Qc, qt, mpb, wg0; };
template <int RP>
__global__ __launch_bounds__(256) void tn_mfma_batched_kernel(const TnJob* __restrict__ jobs, int njobs) {
  __shared__ __attribute__((aligned(16))) char smem[2 * (RP / 32 + 4) * 32 * 64];
  const int wg = blockIdx.x;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {                                  // last job whose first workgroup id is <= wg (uniform across the block)
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].wg0 <= wg) lo = mid; else hi = mid - 1;
  }
  const TnJob j = jobs[lo];
  const int local = wg - j.wg0;
  const int by = local / j.qt, bx = local - by * j.qt;
  tn_mfma_body<RP>(j.P, j.Q, j.M, j.ldq, j.Qc, j.rows, j.mpb, bx, by, smem);
}

struct PackJob { const float* src; bf16* dst; int rows, cols, src_ld, dst_ld, transpose; float scale; };
__global__ void lora_pack_kernel(const PackJob* __restrict__ jobs) {
  const PackJob j = jobs[blockIdx.x];
  const int n = j.rows * j.cols;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int r = i / j.cols, c = i - r * j.cols;
    const float v = j.src[(long long)r * j.src_ld + c] * j.scale;
    if (j.transpose) j.dst[(long long)c * j.dst_ld + r] = (bf16)v;
    else j.dst[(long long)r * j.dst_ld + c] = (bf16)v;
  }
}

// in: rows [B*N][ld_in], columns c0..c0+C  ->  out [B][C][Npad] (token-contiguous), 64x64 LDS tiles
__global__ __launch_bounds__(256) void transpose_tokens_kernel(const bf16* __restrict__ in, int ld_in, int N, int C, int Npad,
                                                               bf16* __restrict__ out) {
  __shared__ bf16 tile[64][66];
  const int b = blockIdx.z, n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  for (int i = tid; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (n0 + r < N && c0 + c < C) ? in[((long long)b * N + n0 + r) * ld_in + c0 + c] : (bf16)0.f;
  }
  __syncthreads();
  for (int i = tid; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < C && n0 + r < Npad) out[((long long)b * C + c0 + c) * Npad + n0 + r] = tile[r][c];
  }
}

// Fast path of the same transpose (ld_in, C multiples of 8; 16-byte aligned input): a wave moves a [64 tokens][32 channels]
// block -- coalesced 16-byte row loads into a 64-byte-row LDS slab, then the transposing LDS read (ds_read_b64_tr_b16, see
// tr_frag) hands every lane 8 consecutive TOKENS of one channel = one 16-byte token-major store.  4 waves = 128 channels.
__global__ __launch_bounds__(256) void transpose_tokens_tr_kernel(const bf16* __restrict__ in, int ld_in, int N, int C, int Npad,
                                                                  bf16* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) char slab[4][64 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, n0 = blockIdx.x * 64, c0 = blockIdx.y * 128 + wave * 32;
  char* sl = slab[wave];
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; ++i) {                       // 64 rows x 4 chunks of 16 B = 256 chunks, 4 per lane
    const int idx = lane + 64 * i, row = idx >> 2, ch = idx & 3;
    const int n = n0 + row, cc = c0 + ch * 8;
    bf16x8 v = zero8;
    if (n < N && cc < C) v = *reinterpret_cast<const bf16x8*>(in + ((long long)b * N + n) * ld_in + cc);
    *reinterpret_cast<bf16x8*>(sl + row * 64 + ch * 16) = v;
  }
  __syncthreads();                                    // (wave-private slab; the barrier keeps EXEC full for the tr reads)
  const int c = c0 + (lane & 31), hh = lane >> 5;
#pragma unroll
  for (int kb = 0; kb < 64; kb += 16) {
    const bf16x8 v = tr_frag(sl, lane, kb);
    const int n = n0 + kb + 8 * hh;
    if (c < C && n < Npad) *reinterpret_cast<bf16x8*>(out + ((long long)b * C + c) * Npad + n) = v;
  }
}

__global__ __launch_bounds__(256) void mse_grad_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                       long long n, float gscale, bf16* __restrict__ dpred,
                                                       float* __restrict__ loss) {
  __shared__ float red[4];
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float e = 0.f;
  if (i < n) {
    e = pred[i] - target[i];
    dpred[i] = (bf16)(2.f * e / (float)n * gscale);
  }
  float s = wave_sum(e * e);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) / (float)n);
}

inline unsigned nblk(long long n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

static int groupnorm_bwd_impl(const void* x, const void* x2, const void* dy, const float* dy_ws, int dy_splits, int B, int HW,
                              int C1, int C2, int groups, float eps, const float* gamma, const float* beta, int act, void* dx,
                              void* dx2, const void* dx_acc, const void* dx2_acc, void* stream) {
  ALDM_CHECK_ARG(x && (dy || (dy_ws && dy_splits >= 1)) && dx && gamma && beta && B > 0 && HW > 0, "groupnorm_bwd: bad args");
  const int C = C1 + C2;
  ALDM_CHECK_ARG(C % groups == 0 && (C / groups) % 4 == 0 && C1 % 4 == 0 && (C2 == 0 || x2), "groupnorm_bwd: bad channels");
  ALDM_CHECK_ARG(C / groups <= GNB_MAXCG, "groupnorm_bwd: at most %d channels per group", GNB_MAXCG);
  const long long nquads = (long long)HW * (C / groups / 4);
  const AldmDiv dq = aldm_make_div((unsigned)(C / groups / 4));
  ALDM_CHECK_ARG(nquads <= 32 * GN_THREADS, "groupnorm_bwd: strip of %lld quads exceeds the register-resident limit", nquads);
#define ALDM_GNB(QPT)                                                                                                  \
  hipLaunchKernelGGL(groupnorm_bwd_kernel<QPT>, dim3(B * groups), dim3(GN_THREADS), 0, (hipStream_t)stream,            \
                     (const bf16*)x, (const bf16*)x2, (const bf16*)dy, HW, C1, C2, groups, eps, gamma, beta, act,      \
                     (bf16*)dx, (bf16*)dx2, (const bf16*)dx_acc, (const bf16*)dx2_acc, dq, dy_ws, dy_splits,           \
                     (long long)B * HW * C)
#define ALDM_GNB_WIDE(QPT)                                                                                             \
  hipLaunchKernelGGL((groupnorm_bwd_kernel<QPT, 1024>), dim3(B * groups), dim3(1024), 0, (hipStream_t)stream,          \
                     (const bf16*)x, (const bf16*)x2, (const bf16*)dy, HW, C1, C2, groups, eps, gamma, beta, act,      \
                     (bf16*)dx, (bf16*)dx2, (const bf16*)dx_acc, (const bf16*)dx2_acc, dq, dy_ws, dy_splits,           \
                     (long long)B * HW * C)
  if (nquads <= 4 * GN_THREADS) ALDM_GNB(4);
  else if (nquads <= 4 * 1024) ALDM_GNB_WIDE(4);
  else if (nquads <= 8 * 1024) ALDM_GNB_WIDE(8);
  else ALDM_GNB_WIDE(16);
#undef ALDM_GNB_WIDE
#undef ALDM_GNB
  return aldm_launch_status("groupnorm_bwd");
}

extern "C" int aldm_groupnorm_bwd(const void* x, const void* x2, const void* dy, int B, int HW, int C1, int C2, int groups,
                                  float eps, const float* gamma, const float* beta, int act, void* dx, void* dx2,
                                  const void* dx_acc, const void* dx2_acc, void* stream) {
  return groupnorm_bwd_impl(x, x2, dy, nullptr, 0, B, HW, C1, C2, groups, eps, gamma, beta, act, dx, dx2, dx_acc, dx2_acc, stream);
}

extern "C" int aldm_groupnorm_bwd_partials(const void* x, const void* x2, const float* dy_ws, int dy_splits, int B, int HW, int C1,
                                           int C2, int groups, float eps, const float* gamma, const float* beta, int act,
                                           void* dx, void* dx2, const void* dx_acc, const void* dx2_acc, void* stream) {
  ALDM_CHECK_ARG(dy_ws && dy_splits >= 1, "groupnorm_bwd_partials: null workspace / bad splits");
  return groupnorm_bwd_impl(x, x2, nullptr, dy_ws, dy_splits, B, HW, C1, C2, groups, eps, gamma, beta, act, dx, dx2, dx_acc, dx2_acc,
                            stream);
}

extern "C" int aldm_layernorm_bwd(const void* x, const void* dy, int M, int C, const float* gamma, float eps, void* dx,
                                  const void* dx_acc, void* stream) {
  ALDM_CHECK_ARG(x && dy && dx && gamma && M > 0 && C % 8 == 0 && C <= 64 * 8 * LN_MAXC, "layernorm_bwd: bad args");
  if (C <= 256)
    hipLaunchKernelGGL(layernorm_bwd_kernel<32>, dim3(cdiv(M, 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x,
                       (const bf16*)dy, M, C, gamma, eps, (bf16*)dx, (const bf16*)dx_acc);
  else
    hipLaunchKernelGGL(layernorm_bwd_kernel<64>, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x,
                       (const bf16*)dy, M, C, gamma, eps, (bf16*)dx, (const bf16*)dx_acc);
  return aldm_launch_status("layernorm_bwd");
}

extern "C" int aldm_geglu_fwd(const void* h, long long M, int I, void* out, void* stream) {
  ALDM_CHECK_ARG(h && out && M > 0 && I % 16 == 0, "geglu_fwd: bad args");
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(nblk(M * (I / 4), 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)h, M, I, (bf16*)out);
  return aldm_launch_status("geglu_fwd");
}

extern "C" int aldm_geglu_bwd(const void* h, const void* dout, long long M, int I, void* dh, void* stream) {
  ALDM_CHECK_ARG(h && dout && dh && M > 0 && I % 16 == 0, "geglu_bwd: bad args");
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(nblk(M * (I / 4), 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)h, (const bf16*)dout, M, I, (bf16*)dh);
  return aldm_launch_status("geglu_bwd");
}

extern "C" int aldm_add_bf16(const void* a, const void* b, long long n, void* c, void* stream) {
  ALDM_CHECK_ARG(a && b && c && n > 0, "add_bf16: bad args");
  hipLaunchKernelGGL(add_bf16_kernel, dim3(nblk((n + 7) / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)b, n, (bf16*)c);
  return aldm_launch_status("add_bf16");
}

extern "C" int aldm_upsample_nearest_bwd(const void* dy, int B, int IH, int IW, int OH, int OW, int C, void* dx, void* stream) {
  ALDM_CHECK_ARG(dy && dx && B > 0 && IH > 0 && IW > 0 && OH >= IH && OW >= IW && C % 8 == 0, "upsample_nearest_bwd: bad args");
  hipLaunchKernelGGL(upsample_nearest_bwd_kernel, dim3(nblk((long long)B * IH * IW * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)dy, B, IH, IW, OH, OW, C, (bf16*)dx);
  return aldm_launch_status("upsample_nearest_bwd");
}

extern "C" int aldm_tn_small(const void* P, int Rp, const void* Q, int ldq, int Qc, int M, const void* rows_dev, void* stream) {
  ALDM_CHECK_ARG(P && Q && rows_dev && M > 0 && Qc > 0 && (Rp == 32 || Rp == 64), "tn_small: bad args");
  if (ldq % 8 == 0 && Qc % 8 == 0 && (reinterpret_cast<uintptr_t>(Q) & 15) == 0 && (reinterpret_cast<uintptr_t>(P) & 15) == 0) {
    // MFMA path: 128 q-columns per workgroup, the m-range cut so that ~256 workgroups run (64-row granules)
    const int qt = cdiv(Qc, 128);
    int msplit = 256 / qt;
    if (msplit < 1) msplit = 1;
    int mpb = cdiv(cdiv(M, msplit), 32) * 32;
    if (mpb < 64) mpb = 64;
    dim3 grid(qt, cdiv(M, mpb));
    if (Rp == 32)
      hipLaunchKernelGGL(tn_mfma_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)P, (const bf16*)Q, M, ldq, Qc, (const TnRow*)rows_dev, mpb);
    else
      hipLaunchKernelGGL(tn_mfma_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)P, (const bf16*)Q, M, ldq, Qc, (const TnRow*)rows_dev, mpb);
    return aldm_launch_status("tn_small");
  }
  const int qt = cdiv(Qc, 64);
  int msplit = 512 / qt;
  if (msplit < 1) msplit = 1;
  int mpb = cdiv(cdiv(M, msplit), 32) * 32;
  if (mpb < 64) mpb = 64;
  dim3 grid(qt, cdiv(M, mpb));
  if (Rp == 32)
    hipLaunchKernelGGL(tn_small_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)P, (const bf16*)Q, M, ldq, Qc, (const TnRow*)rows_dev, mpb);
  else
    hipLaunchKernelGGL(tn_small_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)P, (const bf16*)Q, M, ldq, Qc, (const TnRow*)rows_dev, mpb);
  return aldm_launch_status("tn_small");
}

extern "C" int aldm_tn_batched(const void* jobs_dev, int njobs, int total_wgs, int Rp, void* stream) {
  ALDM_CHECK_ARG(jobs_dev && njobs > 0 && total_wgs > 0 && (Rp == 32 || Rp == 64), "tn_batched: bad args");
  if (Rp == 32)
    hipLaunchKernelGGL(tn_mfma_batched_kernel<32>, dim3(total_wgs), dim3(256), 0, (hipStream_t)stream, (const TnJob*)jobs_dev, njobs);
  else
    hipLaunchKernelGGL(tn_mfma_batched_kernel<64>, dim3(total_wgs), dim3(256), 0, (hipStream_t)stream, (const TnJob*)jobs_dev, njobs);
  return aldm_launch_status("tn_batched");
}

extern "C" int aldm_lora_pack(const void* jobs_dev, int njobs, void* stream) {
  ALDM_CHECK_ARG(jobs_dev && njobs > 0, "lora_pack: bad args");
  hipLaunchKernelGGL(lora_pack_kernel, dim3(njobs), dim3(256), 0, (hipStream_t)stream, (const PackJob*)jobs_dev);
  return aldm_launch_status("lora_pack");
}

extern "C" int aldm_transpose_tokens(const void* in, int ld_in, int B, int N, int C, int Npad, void* out, void* stream) {
  ALDM_CHECK_ARG(in && out && B > 0 && N > 0 && C > 0 && Npad >= N, "transpose_tokens: bad args");
  if (ld_in % 8 == 0 && C % 8 == 0 && Npad % 8 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    hipLaunchKernelGGL(transpose_tokens_tr_kernel, dim3(cdiv(Npad, 64), cdiv(C, 128), B), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)in, ld_in, N, C, Npad, (bf16*)out);
    return aldm_launch_status("transpose_tokens");
  }
  hipLaunchKernelGGL(transpose_tokens_kernel, dim3(cdiv(Npad, 64), cdiv(C, 64), B), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)in, ld_in, N, C, Npad, (bf16*)out);
  return aldm_launch_status("transpose_tokens");
}

extern "C" int aldm_mse_grad(const float* pred, const float* target, long long n, float grad_scale, void* dpred, float* loss,
                             void* stream) {
  ALDM_CHECK_ARG(pred && target && dpred && loss && n > 0, "mse_grad: bad args");
  hipLaunchKernelGGL(mse_grad_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, pred, target, n, grad_scale, (bf16*)dpred, loss);
  return aldm_launch_status("mse_grad");
}

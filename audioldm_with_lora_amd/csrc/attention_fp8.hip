// fp8 (OCP e4m3) operand variant of the flash-style attention core (attention.hip) -- BASELINE config 5:
// "fp8 MFMA fused QKV + LoRA-epilogue attention ... fp8 (e4m3) Q/K/V/P operands with fp32 accumulate".
//
// Same structure as the bf16 kernel: S^T = K Q^T on v_mfma_f32_32x32x16_fp8_fp8 with a lane owning one query column, the
// fp32 score tile converted in registers and re-used as the B operand of O^T += V^T P^T.  Q / K / V^T arrive as bf16 (the
// QKV GEMM's outputs) and are converted while they are staged: the LDS images hold one byte per element, so the fragment
// reads are 8 bytes (K) and 2 x 4 bytes (V^T) instead of 16 and 2 x 8.  P is scaled by 2^8 before the conversion (p <= 1
// would otherwise sit in e4m3's subnormal range for long sequences: 1/1000 < 2^-9); the running sum carries the same factor,
// so it cancels in the final normalisation.  No per-tensor scales: Q, K, V are O(1) after the LayerNorm + projection.
//
// Measured against the bf16 kernel in DESIGN.md section 5: at d = 32 the core is softmax-VALU-bound, the fp8 MFMA runs at
// the bf16 rate (non-scaled fp8 forms) and the conversions add VALU work, so this variant exists for config 5's operand
// precision, not for speed.
#include "common.h"

namespace {

constexpr int KV = 64;

template <int DP>
struct Cfg8 {
  static constexpr int DK = DP / 16;
  static constexpr int DT = (DP + 31) / 32;
  static constexpr int KS = DP + 8;                 // K row stride in bytes (fp8)
  static constexpr int VS = KV + 8;                 // V^T row stride in bytes (fp8)
  static constexpr int KBYTES = KV * KS;
  static constexpr int VBYTES = DT * 32 * VS;
  static constexpr int LDS = 2 * (KBYTES + VBYTES);
};

__device__ __forceinline__ uint2 bf16x8_to_fp8(const bf16x8 v) {
  const uint4 u = __builtin_bit_cast(uint4, v);
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
  int o[2] = {0, 0};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo = __uint_as_float(w[i] << 16), hi = __uint_as_float(w[i] & 0xffff0000u);
    if (i & 1) o[i >> 1] = __builtin_amdgcn_cvt_pk_fp8_f32(lo, hi, o[i >> 1], true);
    else o[i >> 1] = __builtin_amdgcn_cvt_pk_fp8_f32(lo, hi, o[i >> 1], false);
  }
  return make_uint2((unsigned)o[0], (unsigned)o[1]);
}

__device__ __forceinline__ long pack2(unsigned lo, unsigned hi) { return (long)(((unsigned long long)hi << 32) | lo); }

template <int DP, int NW>
__global__ __launch_bounds__(64 * NW) void attention_fp8_kernel(const bf16* __restrict__ q, int ldq, const bf16* __restrict__ k, int ldk,
                                                                const bf16* __restrict__ vt, int vt_ld, long long vt_bs, int N, int D,
                                                                float c /* scale*log2(e) */, bf16* __restrict__ out, int out_ld) {
  using C8 = Cfg8<DP>;
  constexpr int T = 64 * NW;
  constexpr int DK = C8::DK, DT = C8::DT, KS = C8::KS, VS = C8::VS;
  constexpr int KCH = KV * (DP / 8);        // 8-element chunks in a K tile
  constexpr int VCH = DP * (KV / 8);
  constexpr int KPT = (KCH + T - 1) / T, VPT = (VCH + T - 1) / T;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;                              // [2][KV][KS]
  char* Vs = smem + 2 * C8::KBYTES;             // [2][DT*32][VS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  int qblk = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  {                                             // XCD-aware (batch, head) order, as in attention.hip
    const int nq = gridDim.x, H = gridDim.y, npairs = H * gridDim.z;
    if ((npairs & 7) == 0) {
      const int L = blockIdx.x + nq * (blockIdx.y + H * blockIdx.z);
      const int xcd = L & 7, slot = L >> 3;
      const int pl = slot / nq;
      qblk = slot - pl * nq;
      const int pair = pl * 8 + xcd;
      b = pair / H;
      head = pair - b * H;
    }
  }
  const int q0 = (qblk * NW + wave) * 32;
  const bf16* qb = q + (long long)b * N * ldq + head * D;
  const bf16* kb = k + (long long)b * N * ldk + head * D;
  const bf16* vb = vt + (long long)b * vt_bs + (long long)head * D * vt_ld;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  {                                             // d-padding rows of both V^T buffers stay zero
    const int npad = DT * 32 - D;
    for (int i = tid; i < 2 * npad * (VS / 4); i += T) {
      const int buf = i / (npad * (VS / 4)), rem = i - buf * npad * (VS / 4);
      reinterpret_cast<unsigned*>(Vs + buf * C8::VBYTES + D * VS)[rem] = 0u;
    }
  }

  long qf[DK];                                  // B operand: lane (r, hh) holds Q[q0 + r][16 ks + 8 hh .. +7] as 8 fp8
#pragma unroll
  for (int ks = 0; ks < DK; ++ks) {
    const int col = 16 * ks + 8 * hh;
    const bf16x8 v = (q0 + r < N && col < D) ? *reinterpret_cast<const bf16x8*>(qb + (long long)(q0 + r) * ldq + col) : zero8;
    const uint2 f = bf16x8_to_fp8(v);
    qf[ks] = pack2(f.x, f.y);
  }

  bf16x8 kreg[KPT], vreg[VPT];
  auto prefetch = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int cidx = tid + i * T;
      const int row = cidx / (DP / 8), ch = cidx - row * (DP / 8);
      bf16x8 v = zero8;
      if (cidx < KCH && kv0 + row < N && ch * 8 < D) v = *reinterpret_cast<const bf16x8*>(kb + (long long)(kv0 + row) * ldk + ch * 8);
      kreg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int cidx = tid + i * T;
      const int row = cidx >> 3, ch = cidx & 7;
      bf16x8 v = zero8;
      const int kvb = kv0 + ch * 8;
      if (cidx < VCH && row < D && kvb < N) {
        v = *reinterpret_cast<const bf16x8*>(vb + (long long)row * vt_ld + kvb);
        if (kvb + 8 > N) {
#pragma unroll
          for (int j = 0; j < 8; ++j) if (kvb + j >= N) v[j] = (bf16)0.f;
        }
      }
      vreg[i] = v;
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int cidx = tid + i * T;
      const int row = cidx / (DP / 8), ch = cidx - row * (DP / 8);
      if (cidx < KCH) *reinterpret_cast<uint2*>(Ks + buf * C8::KBYTES + row * KS + ch * 8) = bf16x8_to_fp8(kreg[i]);
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int cidx = tid + i * T;
      const int row = cidx >> 3, ch = cidx & 7;
      if (cidx < VCH) *reinterpret_cast<uint2*>(Vs + buf * C8::VBYTES + row * VS + ch * 8) = bf16x8_to_fp8(vreg[i]);
    }
  };

  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles = (N + KV - 1) / KV;
  prefetch(0);
  stage(0);
  __syncthreads();

  for (int it = 0; it < ntiles; ++it) {
    const int buf = it & 1, kv0 = it * KV;
    if (it + 1 < ntiles) prefetch(kv0 + KV);

    f32x16 s[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[sub][i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < DK; ++ks) {
        const uint2 kf = *reinterpret_cast<const uint2*>(Ks + buf * C8::KBYTES + (sub * 32 + r) * KS + 16 * ks + 8 * hh);
        s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(pack2(kf.x, kf.y), qf[ks], s[sub], 0, 0, 0);
      }
    }
    if (kv0 + KV > N) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int kvr = kv0 + sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (kvr >= N) s[sub][i] = -INFINITY;
        }
    }
    float mx = s[0][0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[0][i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[1][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * c);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    const float m_arg = m_new - 8.f;            // p * 2^8: keeps the fp8 P operand out of e4m3's subnormal range
    float psum = 0.f;
    long pf[4];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int h8 = 0; h8 < 2; ++h8) {
        int w0 = 0, w1 = 0;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const float p0 = __builtin_amdgcn_exp2f(fmaf(s[sub][8 * h8 + j], c, -m_arg));
          const float p1 = __builtin_amdgcn_exp2f(fmaf(s[sub][8 * h8 + j + 1], c, -m_arg));
          psum += p0 + p1;
          if (j < 4) w0 = (j & 2) ? __builtin_amdgcn_cvt_pk_fp8_f32(p0, p1, w0, true) : __builtin_amdgcn_cvt_pk_fp8_f32(p0, p1, w0, false);
          else w1 = (j & 2) ? __builtin_amdgcn_cvt_pk_fp8_f32(p0, p1, w1, true) : __builtin_amdgcn_cvt_pk_fp8_f32(p0, p1, w1, false);
        }
        pf[sub * 2 + h8] = pack2((unsigned)w0, (unsigned)w1);
      }
    l_run = l_run * alpha + psum;
    if (!__all(alpha == 1.0f)) {
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] *= alpha;
    }
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {          // 4 K-steps of 16 keys; the lane's 8 keys = {16 s2 + 4 hh + 0..3, + 8..11}
        const char* vrow = Vs + buf * C8::VBYTES + (t * 32 + r) * VS + 16 * s2 + 4 * hh;
        const unsigned lo = *reinterpret_cast<const unsigned*>(vrow);
        const unsigned hi = *reinterpret_cast<const unsigned*>(vrow + 8);
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(pack2(lo, hi), pf[s2], o[t], 0, 0, 0);
      }
    if (it + 1 < ntiles) stage(buf ^ 1);
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;               // l carries the same 2^8 as O: it cancels here
  if (q0 + r < N) {
    bf16* orow = out + ((long long)b * N + q0 + r) * out_ld + head * D;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = t * 32 + 8 * g + 4 * hh;
        if (d0 < D) {
          bf16x4 v = {(bf16)(o[t][4 * g] * inv), (bf16)(o[t][4 * g + 1] * inv), (bf16)(o[t][4 * g + 2] * inv), (bf16)(o[t][4 * g + 3] * inv)};
          *reinterpret_cast<bf16x4*>(orow + d0) = v;
        }
      }
  }
}

template <int DP, int NW>
int launch8(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld, long long vt_bs, int B, int N, int H, int D,
            float scale, void* out, int out_ld, hipStream_t st) {
  using C8 = Cfg8<DP>;
  auto kern = attention_fp8_kernel<DP, NW>;
  static unsigned long long attr_done = 0;   // per-device bit mask (aldm_set_max_lds)
  if (C8::LDS > 48 * 1024)
    if (int rc = aldm_set_max_lds(reinterpret_cast<const void*>(kern), C8::LDS, &attr_done, "attention_fp8")) return rc;
  hipLaunchKernelGGL(kern, dim3(cdiv(N, 32 * NW), H, B), dim3(64 * NW), C8::LDS, st, (const bf16*)q, ldq, (const bf16*)k, ldk,
                     (const bf16*)vt, vt_ld, vt_bs, N, D, scale * 1.44269504088896340736f, (bf16*)out, out_ld);
  return aldm_launch_status("attention_fp8");
}

template <int DP>
int launch8_d(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld, long long vt_bs, int B, int N, int H, int D,
              float scale, void* out, int out_ld, hipStream_t st) {
  if (N >= 192) return launch8<DP, 4>(q, ldq, k, ldk, vt, vt_ld, vt_bs, B, N, H, D, scale, out, out_ld, st);
  if (N >= 64) return launch8<DP, 2>(q, ldq, k, ldk, vt, vt_ld, vt_bs, B, N, H, D, scale, out, out_ld, st);
  return launch8<DP, 1>(q, ldq, k, ldk, vt, vt_ld, vt_bs, B, N, H, D, scale, out, out_ld, st);
}

}  // namespace

extern "C" int aldm_attention_fp8(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld, long long vt_batch_stride,
                                  int B, int N, int H, int d, float scale, void* out, int out_ld, void* stream) {
  ALDM_CHECK_ARG(q && k && vt && out, "attention_fp8: null pointer");
  ALDM_CHECK_ARG(B > 0 && N > 0 && H > 0 && d > 0, "attention_fp8: bad dims");
  ALDM_CHECK_ARG(d % 8 == 0 && ldq % 8 == 0 && ldk % 8 == 0 && vt_ld % 8 == 0 && out_ld % 4 == 0, "attention_fp8: d/ld must be multiples of 8");
  ALDM_CHECK_ARG(vt_ld >= ((N + 7) / 8) * 8, "attention_fp8: vt_ld %d too small for N %d", vt_ld, N);
  hipStream_t st = (hipStream_t)stream;
#define ALDM_ATTN8(DPV) return launch8_d<DPV>(q, ldq, k, ldk, vt, vt_ld, vt_batch_stride, B, N, H, d, scale, out, out_ld, st)
  if (d <= 16) ALDM_ATTN8(16);
  if (d <= 32) ALDM_ATTN8(32);
  if (d <= 48) ALDM_ATTN8(48);
  if (d <= 64) ALDM_ATTN8(64);
  if (d <= 80) ALDM_ATTN8(80);
#undef ALDM_ATTN8
  aldm_set_error("attention_fp8: head dim %d > 80 unsupported", d);
  return ALDM_E_UNSUPPORTED;
}

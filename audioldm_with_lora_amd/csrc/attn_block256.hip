// ONE launch for the front half of the fused-LoRA attention module at the UNet's 252-token level (C = 384, 8 heads x 48,
// N = 63 x 4 = 252 tokens per sample for a 10 s clip, 64 x 4 = 256 in training; 10 of the 32 Attention modules of
// UNet2DConditionModel.forward [REF script/train/train_audioldm_lora.py:539-546] / the DDIM loop
// [REF script/inference/generate_audio.py:47-52]):
//     LayerNorm (folded) -> to_q | to_k | to_v with the LoRA side channel -> softmax(Q K^T) V        per (sample, head) workgroup.
// The 64-token level's kernel (attn_block64.hip) with four times the tokens: at this level the projection (13 us) and the
// attention (8 us) are two latency chains of a few GFLOP on a full grid; a (sample, head) workgroup of 8 waves that keeps its
// 256 tokens of X in REGISTERS (2 token tiles of 16 per wave, 96 VGPRs), streams the head's 144 weight rows (+ LoRA-A rows)
// once through a three-stage LDS ring and leaves Q in registers and K | V^T in LDS runs both as one chain on 64 CUs: Q | K | V
// never touch HBM (3 x 1.5 MB written and read back per module) and one kernel boundary + one ramp disappear.
//
//   * 8 wave64 (two per SIMD); wave w owns tokens 32 w .. 32 w + 31 (token tiles tt = 0, 1) through both phases.
//     v_mfma_f32_16x16x32_bf16 issued "swapped" (weight rows = A operand) so a lane owns ONE token column per tile: LayerNorm
//     mean / rstd and the softmax state are lane-local; every weight fragment read from LDS feeds two MFMAs.
//   * weight stream HBM/L2 -> registers -> LDS (plain 16-byte loads, P tiles in flight, ds_write_b128 into an XOR-swizzled image),
//     tile = 16 output columns x full K = 12 KB (+ the tile's 16 x 32 LoRA-B rows); tiles: RT LoRA-A tiles, then 3 q + 3 k + 3 v.
//   * epilogue y = rstd (acc - mean s_n) + c_n (folded LayerNorm, statistics from the producer's ln_parts) as in aldm_igemm;
//     LoRA: T'' = T - mean sA + cA / rstd rounded to bf16 in registers, one more K = 32 step per tile against the pre-scaled B rows.
//   * attention per query tile: S^T = K Q^T over 16 key tiles (Q = the projection's accumulators, already a B operand; K read
//     with the matching k-slot order), softmax over the 256 keys in registers, O^T = V^T P^T with P as the B operand.
#include "igemm_core.h"   // make_rsrc

namespace {

struct Blk256Args {
  const bf16* x;            // [B * N][C] raw hidden state
  const float* ln_parts;    // [B * N][np][2] row partials from the producer of x
  const bf16* w;            // [3C][Kpad] LayerNorm-folded q | k | v weights (q pre-scaled)
  const float* bias;        // [3C] c_n = W beta (+ bias)
  const float* ln_s;        // [3C] row sums of the folded weights
  const bf16* lora_a;       // [Rp][Kpad] folded LoRA-A rows of q | k | v (or null)
  const bf16* lora_b;       // [3C][Rp] pre-scaled LoRA-B
  const float* ln_sa;       // [Rp]
  const float* ln_ca;       // [Rp]
  bf16* out;                // [B * N][C] attention output (heads concatenated)
  int np, Kpad, Rp, N, H, B;
  float eps;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int C, int D, int RT /* LoRA rank tiles of 16: 0, 1 or 2 */>
__global__ __launch_bounds__(512) void attn_block256_kernel(const Blk256Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  aldm_touch_kernargs<sizeof(Blk256Args)>();
  constexpr int NW = 8, NTOK = 32 * NW, CPR = C / 8;          // 16-byte chunks per weight row
  constexpr int KSTEPS = C / 32;
  constexpr int DTL = D / 16;                                 // 16-column tiles per q / k / v section
  constexpr int NTILES = RT + 3 * DTL;
  constexpr int DKS = (D + 31) / 32;                          // k-steps of the Q K^T contraction
  constexpr int KSTR = D * 2 + 16;                            // K row stride (odd multiple of 16 B: conflict-free ds_read_b64)
  constexpr int VSTR = NTOK * 2 + 136;                        // V^T row stride: (VSTR / 8) % 32 == 17 -> conflict-free ds_read_b64 (as the 64-token kernel's 136)
  constexpr int WTILE = 16 * C * 2, WSTAGE = WTILE + 1024;    // weight tile + its 16 x 32 LoRA-B rows
  constexpr int WCH = 16 * CPR;                               // 16-byte chunks of a tile
  constexpr int LPT = (WCH + 511) / 512;                      // loads per thread per tile
  constexpr int P = 4;                                        // weight tiles in flight in registers, behind the (up to) two waiting in LDS
  constexpr int NS = 3;                                       // LDS stages: tile t is multiplied while t + 1 waits and t + 2 is written
  constexpr int NKT = NTOK / 16;                              // key tiles
  static_assert(CPR % 16 == 0 && D % 16 == 0 && (KSTR / 16) % 2 == 1, "C % 128 == 0; D % 16 == 0");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Wr = smem;                                      // [NS][WSTAGE]
  char* const Ks = Wr + NS * WSTAGE;                          // [NTOK][KSTR]
  char* const Vt = Ks + NTOK * KSTR;                          // [D][VSTR]
  float* const sn_l = reinterpret_cast<float*>(Vt + D * VSTR);   // [3 D] row sums s_n of this head's q | k | v columns
  float* const cn_l = sn_l + 3 * D;                               // [3 D] c_n

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  // the 8 heads of a sample on ONE XCD (workgroups b, b + 8, ... share an L2): they all read the sample's X
  int head, b;
  if ((p.B & 7) == 0) { b = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * p.H)); head = (blockIdx.x >> 3) % p.H; }
  else { head = blockIdx.x % p.H; b = blockIdx.x / p.H; }
  const int N = p.N;
  const long long row0 = (long long)b * N;
  int tok[2];
  bool live[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) { tok[tt] = 32 * wave + 16 * tt + n; live[tt] = tok[tt] < N; }

  const __amdgpu_buffer_rsrc_t rs_x = aldm_igemm_detail::make_rsrc(p.x + row0 * C, (unsigned)(N * C * 2));
  const __amdgpu_buffer_rsrc_t rs_w = aldm_igemm_detail::make_rsrc(p.w, (unsigned)(3u * C * (unsigned)p.Kpad * 2u));
  const __amdgpu_buffer_rsrc_t rs_a = aldm_igemm_detail::make_rsrc(RT ? (const void*)p.lora_a : (const void*)p.w, (unsigned)((RT ? p.Rp : 16) * p.Kpad * 2));
  const __amdgpu_buffer_rsrc_t rs_b = aldm_igemm_detail::make_rsrc(RT ? (const void*)p.lora_b : (const void*)p.w, (unsigned)(RT ? 3u * C * (unsigned)p.Rp * 2u : 64u));

  // ---- the small operands first: LayerNorm partials of this lane's two tokens, s_n | c_n of the head's columns, LoRA vectors ----
  float ln_a[2] = {0.f, 0.f}, ln_q[2] = {0.f, 0.f};
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    if (live[tt]) {
      const float* pp = p.ln_parts + (row0 + tok[tt]) * (p.np * 2);
      float2 v[16];                                          // all partials in flight at once (a running sum would wait per load)
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = j < p.np ? *reinterpret_cast<const float2*>(pp + 2 * j) : make_float2(0.f, 0.f);
#pragma unroll
      for (int j = 0; j < 16; ++j) { ln_a[tt] += v[j].x; ln_q[tt] += v[j].y; }
    }
  }
  float sn_v = 0.f, cn_v = 0.f;
  if (tid < 3 * D) {
    const int sec = tid / D, r = tid - sec * D;
    sn_v = p.ln_s[sec * C + head * D + r];
    cn_v = p.bias[sec * C + head * D + r];
  }
  f32x4 sa[RT ? RT : 1], ca[RT ? RT : 1];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    sa[t] = *reinterpret_cast<const f32x4*>(p.ln_sa + 16 * t + 4 * g);
    ca[t] = *reinterpret_cast<const f32x4*>(p.ln_ca + 16 * t + 4 * g);
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- X fragments of this wave's 2 x 16 tokens straight into registers (B operand): lane (n, g) holds X[tok][32 ks + 8 g .. + 7].
  //      They stay there across all weight tiles, so X never needs LDS; rows >= N read as zeros through the descriptor. ----
  bf16x8 xf[2][KSTEPS];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
      xf[tt][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_x, tok[tt] * (C * 2) + (32 * ks + 8 * g) * 2, 0, 0));
  __builtin_amdgcn_sched_barrier(0);

  // ---- the weight stream: HBM/L2 -> registers (plain 16-byte loads, P tiles in flight) -> LDS (NS stages) ----
  auto wrow0 = [&](int t) {
    const int j = t - RT, sec = j / DTL, jj = j - sec * DTL;
    return sec * C + head * D + 16 * jj;
  };
  constexpr unsigned OOB = 0x80000000u;
  unsigned w_src[LPT];
  int w_dst[LPT];                                             // per-lane chunk of a tile: source byte offset (row-relative), swizzled LDS offset
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    const int c = tid + 512 * i, row = c / CPR, ph = c - row * CPR;
    w_src[i] = c < WCH ? (unsigned)(row * (p.Kpad * 2) + ph * 16) : OOB;
    w_dst[i] = c < WCH ? row * (C * 2) + ((ph ^ (row & 15)) * 16) : -1;
  }
  const int b_src = (lane >> 2) * (p.Rp * 2) + (lane & 3) * 16;
  u32x4 wreg[P][LPT], breg[P];
  auto gload = [&](int t) {
    const int slot = t % P;
    const bool la = t < RT;
    const int r0 = la ? 16 * t : wrow0(t);
    const int base = r0 * (p.Kpad * 2);
#pragma unroll
    for (int i = 0; i < LPT; ++i) wreg[slot][i] = __builtin_amdgcn_raw_buffer_load_b128(la ? rs_a : rs_w, w_src[i], base, 0);
    if (RT > 0 && !la && wave == (t & 7)) breg[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, r0 * (p.Rp * 2) + b_src, 0, 0);
  };
  auto lstore = [&](int t) {
    const int slot = t % P;
    char* const dst = Wr + (t % NS) * WSTAGE;
#pragma unroll
    for (int i = 0; i < LPT; ++i)
      if (w_dst[i] >= 0) *reinterpret_cast<u32x4*>(dst + w_dst[i]) = wreg[slot][i];
    if (RT > 0 && t >= RT && wave == (t & 7)) *reinterpret_cast<u32x4*>(dst + WTILE + lane * 16) = breg[slot];
  };
  const int w_rd = n * (C * 2);                               // this lane's row of a tile image; chunk (4 ks + g) ^ n
#pragma unroll
  for (int t = 0; t < P && t < NTILES; ++t) gload(t);
  __builtin_amdgcn_sched_barrier(0);

  // ---- LayerNorm statistics of this lane's tokens from the producer's partials; s_n | c_n of the head's columns ----
  float mean[2], rstd[2], irs[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    mean[tt] = rstd[tt] = irs[tt] = 0.f;
    if (live[tt]) {
      mean[tt] = ln_a[tt] * (1.f / C);
      rstd[tt] = rsqrtf(fmaxf(ln_q[tt] * (1.f / C) - mean[tt] * mean[tt], 0.f) + p.eps);
      irs[tt] = 1.f / rstd[tt];
    }
  }
  if (tid < 3 * D) { sn_l[tid] = sn_v; cn_l[tid] = cn_v; }
  lstore(0);
  if (P < NTILES) gload(P);
  lstore(1);
  if (P + 1 < NTILES) gload(P + 1);
  // raw barriers throughout: __syncthreads()' fence would drain the weight loads in flight (vmcnt(0)); the LDS writes are covered
  // by the explicit lgkmcnt(0)
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();

  bf16x8 tf[2];                                              // T'' as the B operand of the LoRA k-step: slots = ranks {4g+j | 16+4g+j}
  bf16x8 qf[2][DKS];                                         // Q of the lane's tokens as the B operand of S^T = K Q^T: slots = dims {32 ks + 4g+j | + 16}
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    tf[tt] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < DKS; ++ks) qf[tt][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }

#pragma unroll
  for (int t = 0; t < NTILES; ++t) {
    // stage t % NS holds tile t (all waves' parts: barrier passed); stage (t + 2) % NS was last read in iteration t - 1
    if (t + 2 < NTILES) {
      lstore(t + 2);
      if (t + 2 + P < NTILES) gload(t + 2 + P);
    }
    const char* Wb = Wr + (t % NS) * WSTAGE;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {                      // D[tile row 4 g + j][token n]; one fragment read, two MFMAs
      const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Wb + w_rd + (((4 * ks + g) ^ n) * 16));
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[0][ks], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[1][ks], acc[1], 0, 0, 0);
    }
    if (t < RT) {
      // LoRA-A tile: T'' = T - mean sA + cA / rstd  (the epilogue's rstd (acc - mean s) + c then also fixes the LoRA term)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int j = 0; j < 4; ++j) tf[tt][4 * t + j] = (bf16)(acc[tt][j] - mean[tt] * sa[t < RT ? t : 0][j] + ca[t < RT ? t : 0][j] * irs[tt]);
    } else {
      if (RT > 0) {
        // LoRA: one more K = 32 step, A = the pre-scaled B rows of this tile with the same k-slot order
        const char* brow = Wb + WTILE + n * 64 + g * 8;
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(brow);
        const bf16x4 hi = (RT > 1) ? *reinterpret_cast<const bf16x4*>(brow + 32) : bf16x4{0, 0, 0, 0};
        const bf16x8 bfr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr, tf[0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr, tf[1], acc[1], 0, 0, 0);
      }
      const int j0 = t - RT, sec = j0 / DTL, jj = j0 - sec * DTL;
      const f32x4 sn = *reinterpret_cast<const f32x4*>(sn_l + sec * D + 16 * jj + 4 * g), cn = *reinterpret_cast<const f32x4*>(cn_l + sec * D + 16 * jj + 4 * g);
      const int dcol = 16 * jj + 4 * g;                       // head-dim index of y[0]
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        bf16x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = live[tt] ? (bf16)(rstd[tt] * (acc[tt][j] - mean[tt] * sn[j]) + cn[j]) : (bf16)0.f;   // tokens >= N: K | V^T rows exactly zero
        if (sec == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) qf[tt][jj >> 1][(jj & 1) * 4 + j] = y[j];
        } else if (sec == 1) {
          *reinterpret_cast<bf16x4*>(Ks + tok[tt] * KSTR + dcol * 2) = y;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) *reinterpret_cast<bf16*>(Vt + (dcol + j) * VSTR + tok[tt] * 2) = y[j];
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
  }

  // ---- attention over the head's tokens; this wave's queries = its tokens, one 16-query tile at a time ----
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    if (32 * wave + 16 * tt >= N) break;                       // (wave-uniform) a whole query tile of padding
    f32x4 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DKS; ++ks) {
        const char* krow = Ks + (16 * kt + n) * KSTR + (32 * ks + 4 * g) * 2;       // K[key 16 kt + n][dims 32 ks + 4 g .. | + 16 ..]
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(krow);
        const bf16x4 hi = (32 * ks + 16 < D) ? *reinterpret_cast<const bf16x4*>(krow + 32) : bf16x4{0, 0, 0, 0};
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7), tt == 0 ? qf[0][ks] : qf[1][ks], s[kt], 0, 0, 0);   // S^T[key 16 kt + 4 g + j][query n]
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (16 * kt + 4 * g + j >= N) s[kt][j] = -INFINITY;
        mx = fmaxf(mx, s[kt][j]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f;
    bf16x8 pf[NKT / 2];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16 pb = (bf16)__builtin_amdgcn_exp2f(s[kt][j] - mx);
        l += (float)pb;                                       // the normaliser sums the SAME rounded values that multiply V
        pf[kt >> 1][(kt & 1) * 4 + j] = pb;
      }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    // (the MFMAs run with every lane active -- a V^T row of the A operand lives in lane n whether or not query n is live; only
    //  the store is guarded)
    const int tk = 32 * wave + 16 * tt + n;
    bf16* orow = p.out + (row0 + tk) * C + head * D;
#pragma unroll
    for (int t = 0; t < DTL; ++t) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < NKT / 2; ++kk) {
        const char* vrow = Vt + (16 * t + n) * VSTR + (32 * kk + 4 * g) * 2;      // V^T[16 t + n][keys 32 kk + 4 g .. | + 16 ..]
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vrow);
        const bf16x4 hi = *reinterpret_cast<const bf16x4*>(vrow + 32);
        o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7), pf[kk], o, 0, 0, 0);
      }
      const bf16x4 ov = {(bf16)(o[0] * inv), (bf16)(o[1] * inv), (bf16)(o[2] * inv), (bf16)(o[3] * inv)};
      if (tk < N) *reinterpret_cast<bf16x4*>(orow + 16 * t + 4 * g) = ov;   // O^T[d 16 t + 4 g + j][query n]
    }
  }
#endif
}

template <int C, int D, int RT>
int launch_blk256(const Blk256Args& a, hipStream_t st) {
  constexpr int LDS = 3 * (16 * C * 2 + 1024) + 256 * (D * 2 + 16) + D * (256 * 2 + 136) + 2 * 3 * D * 4;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  auto kern = attn_block256_kernel<C, D, RT>;
  static unsigned long long attr_done = 0;
  if (int rc = aldm_set_max_lds(reinterpret_cast<const void*>(kern), LDS, &attr_done, "attn_block256")) return rc;
  hipLaunchKernelGGL(kern, dim3(a.H * a.B), dim3(512), LDS, st, a);
  return aldm_launch_status("attn_block256");
}

}  // namespace

extern "C" int aldm_attn_block256(const void* x, const float* ln_parts, int ln_nparts, const void* w, int Kpad, const float* bias,
                                  const float* ln_s, const void* lora_a, const void* lora_b, int Rp, int ranks_used,
                                  const float* ln_sa, const float* ln_ca, float ln_eps, int B, int N, int H, int d, void* out,
                                  void* stream) {
  ALDM_CHECK_ARG(x && ln_parts && w && bias && ln_s && out, "attn_block256: null pointer");
  ALDM_CHECK_ARG(B > 0 && N > 0 && N <= 256 && ln_nparts > 0 && ln_nparts <= 16, "attn_block256: 1 <= N <= 256 tokens per sample, 1 <= ln_nparts <= 16");
  ALDM_CHECK_ARG(Rp == 0 || (Rp % 16 == 0 && ranks_used > 0 && ranks_used <= 32 && ranks_used <= Rp && lora_a && lora_b && ln_sa && ln_ca),
                 "attn_block256: LoRA needs Rp %% 16 == 0, 1 <= ranks_used <= min(32, Rp) and lora_a / lora_b / ln_sa / ln_ca");
  ALDM_CHECK_ARG(H == 8 && d == 48 && Kpad == 384, "attn_block256: built for C = 384 = 8 heads x 48 (the UNet's 252-token level); got H %d d %d Kpad %d", H, d, Kpad);
  Blk256Args a{(const bf16*)x, ln_parts, (const bf16*)w, bias, ln_s, (const bf16*)lora_a, (const bf16*)lora_b, ln_sa, ln_ca, (bf16*)out,
               ln_nparts, Kpad, Rp, N, H, B, ln_eps};
  hipStream_t st = (hipStream_t)stream;
  if (Rp == 0) return launch_blk256<384, 48, 0>(a, st);
  if (ranks_used <= 16) return launch_blk256<384, 48, 1>(a, st);   // rank-4 q | k | v = 12 rows: one LoRA-A tile, the zero rows skipped
  return launch_blk256<384, 48, 2>(a, st);
}

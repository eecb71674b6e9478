// Flash-style self-attention core for gfx950: S^T = K Q^T on v_mfma_f32_32x32x16_bf16 so that a lane owns ONE
// query column -- running max / sum / rescale are lane-local (one cross-half shuffle per 64 keys), the fp32
// score tile is converted in registers and re-used directly as the B operand of O^T += V^T P^T (no LDS round
// trip for P; the key order inside each 16-wide K-step is the accumulator's row permutation, which the
// V^T fragment reads reproduce with two 8-byte LDS reads).  K / V^T tiles of 64 keys are shared by the
// workgroup's waves through padded (bank-conflict-free) LDS images, register-prefetched one tile ahead.
//
// What bounds the loop (round 4; the round-3 note here -- "a SIMD does not run plain VALU work under its own MFMAs" -- was wrong:
// tools/micro/mfma_valu_gap.hip, instruction stream pinned by inline asm, hides five v_fma_f32 or three v_exp_f32 under every
// v_mfma_f32_32x32x16_bf16 of ONE wave, and two waves per SIMD overlap by themselves when no barrier re-aligns them): at UNet batch 8
// the grid is one 8-wave workgroup per CU and each 64-key tile is a dependent chain (fragment reads -> S^T MFMAs -> maximum -> exp ->
// convert -> P V MFMAs -> barrier) of ~1300 cycles that two waves per SIMD do not cover; the ablation is at the ATTN_ML / ATTN_FENCE
// switches below.  The per-key VALU work is nevertheless cut to its essentials:
//   * the running maximum enters as the INITIAL ACCUMULATOR of the S^T MFMAs (S' = K Q^T - m), so p = exp2(S') needs no
//     per-element subtraction; the maximum is only raised when a tile exceeds it by more than RESCALE_THR (deferred rescale,
//     wave-uniform branch) -- P is then bounded by 2^THR instead of 1, which fp32 accumulation and bf16 P tolerate
//   * the softmax scale is folded into the Q projection weights by the caller (PRESCALED) -- no multiply per score
//   * the row sum l = sum_k p rides in the P V MFMAs: V^T carries one extra row of ones (row DP of the LDS image), so l
//     accumulates in the O^T tile like any other output row, from the SAME bf16-rounded P that multiplies V
//   * running max over a tile with v_max3_f32 (16 instructions for 32 scores)
//
// Serves F.scaled_dot_product_attention inside diffusers AttnProcessor2_0 for the 32 Attention modules of
// UNet2DConditionModel.forward [REF script/train/train_audioldm_lora.py:539-546].  Head dims 32/48/80
// (8 heads at C = 256/384/640), sequence lengths 1000/252/64 (10 s) and 1024/256/64 (training).
#include "common.h"
#include <cstdlib>
#include <type_traits>

// Round-4 experiments kept as compile-time switches (both OFF in the product build; numbers: B = 8, N = 1000, 8 heads x 32, replayed
// graph, tools/bench_attn.py; profiles/r04_attention_ablation.txt):
//   ATTN_ML    row sum l on the matrix pipe (register-constant ones fragment, 4 more MFMAs per tile) instead of VALU adds
//   ATTN_FENCE hand-ordered tile: staging + prefetch behind the S^T MFMAs, exponentials interleaved with the P V MFMAs by sched_barrier
//   ML 0 / FENCE 0 20.3 us | 0 / 1 21.1 | 1 / 0 21.0 | 1 / 1 21.9 -- neither helps, and the ablations say why: without the running
//   maximum (21 VALU per tile) 20.6 us, without the 32 exponentials 18.2, without the global loads + staging 17.9, without all three
//   13.5: what remains -- fragment reads, 8 MFMAs, 16 conversions and the barrier of each of the 16 tiles -- is a DEPENDENT CHAIN of
//   ~1300 cycles per tile that two waves per SIMD (the grid is one 8-wave workgroup per CU at UNet batch 8) do not cover; at twice the
//   batch (two workgroups per CU) the same kernel does twice the work in 36.7 us.  The instruction mix is not the limit at this size.
#ifndef ATTN_ML
#define ATTN_ML 0
#endif
#ifndef ATTN_FENCE
#define ATTN_FENCE 0
#endif

namespace {

constexpr int KV = 64;           // keys per tile
constexpr int VS = 136;          // V^T LDS row stride in bytes (64 keys * 2 B + 8 B pad): conflict-free ds_read_b64
constexpr int VS8 = 72;          // the same for one-byte (e4m3) elements: 64 keys + 8 B pad


typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float RESCALE_THR = 5.0f;   // log2 units: the running max is raised only when a tile's max exceeds it by more than this

__device__ __forceinline__ float max3f(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

template <int DP, bool FP8 = false>
struct AttnCfg {
  static constexpr int DK = DP / 16;                       // QK^T k-steps
  // VL: the row sum l = sum_k p is accumulated by VALU adds instead of the row of ones.  Where DP is a multiple of 32 (d = 32, 64) the
  // ones row needs a d-tile of its own -- 4 of the 12 MFMAs per 64-key tile at d = 32 -- and, MFMA and VALU time being additive on this
  // chip (see the header), 32 adds (64-128 cycles) are cheaper than 4 MFMAs (128 cycles) plus their 4 fragment reads.
  static constexpr bool VL = DP % 32 == 0;                 // (round 4: these sizes sum l on the matrix pipe after all, see ML below)
  static constexpr int DT = VL ? DP / 32 : DP / 32 + 1;    // 32-row d-tiles of O^T, INCLUDING (unless VL) the row of ones at row DP
  static constexpr int ONES_T = VL ? 0 : DP / 32;          // tile / accumulator register that holds l = sum p (lanes hh = 0)
  static constexpr int ONES_I = 4 * ((DP % 32) >> 3);
  static_assert(DP % 16 == 0, "row DP must land on hh = 0, register 4 * (row / 8)");
  static constexpr int KS = FP8 ? DP + 8 : DP * 2 + (((DP / 8) % 2 == 0) ? 16 : 0);  // K LDS row stride (bf16: odd number of 16-B slots)
  static constexpr int VSB = FP8 ? VS8 : VS;               // V^T LDS row stride in bytes
  static constexpr int KBYTES = KV * KS;
  static constexpr int VBYTES = DT * 32 * VSB;
  static constexpr int TILE = KBYTES + VBYTES;
};

// SP > 1 splits the KEYS of a query block over SP waves (wave groups): every iteration stages SP tiles of 64 keys, group g
// works on tile g, and the groups' (max, sum, O^T) are merged through LDS at the end.  At N = 1000 / 252 with 8 x 8 heads a
// query block per wave gives only 2 / 0.5 waves per SIMD; the split doubles the independent instruction streams that hide
// the MFMA -> softmax -> MFMA dependency chain.
// FP8 (BASELINE config 5: "fp8 (e4m3) Q / K / V / P operands in K1 with fp32 accumulate"): the SAME kernel with one-byte LDS images --
// Q / K / V^T arrive as bf16 (the QKV GEMM's outputs) and are converted while they are staged, the fragment reads are 8 bytes (K) and
// 2 x 4 bytes (V^T), both MFMAs are v_mfma_f32_32x32x16_fp8_fp8 (the bf16 rate on gfx950: the non-scaled fp8 forms), P is scaled by 2^8
// before its conversion (p <= 1 would sit in e4m3's subnormal range for long sequences; l carries the same factor, so it cancels).
// Until round 4 this was a separate, older kernel (4 waves, no deferred rescale, predicated loads) and 4.6 us slower than bf16 at
// N = 1000; as a variant of this one it inherits every structural improvement.
template <int DP, int NW, int SP, bool PRESCALED, bool FP8 = false>
__global__ __launch_bounds__(64 * NW) void attention_kernel(const bf16* __restrict__ q, int ldq,
                                                            const bf16* __restrict__ k, int ldk,
                                                            const bf16* __restrict__ vt, int vt_ld, long long vt_bs,
                                                            int N, int D, float c /* scale*log2(e) */,
                                                            bf16* __restrict__ out, int out_ld, float* __restrict__ lse,
                                                            const int* __restrict__ kv_len) {
  aldm_touch_kernargs<96>();                // 88 bytes of explicit arguments: both lines in one round (common.h)
#ifndef ALDM_NO_KA_PREFETCH
  aldm_prefetch_next_kernargs<96>(threadIdx.x);
#endif
  using Cfg = AttnCfg<DP, FP8>;
  constexpr int T = 64 * NW;
  constexpr int DK = Cfg::DK, DT = Cfg::DT, KS = Cfg::KS, VS = Cfg::VSB;   // (VS shadows the bf16 stride: every use below is per-format)
#ifndef ATTN_FP8_PBIAS
#define ATTN_FP8_PBIAS 8.f
#endif
  constexpr float PBIAS = FP8 ? ATTN_FP8_PBIAS : 0.f;     // log2 of the factor P carries into its fp8 conversion
  using frag_t = std::conditional_t<FP8, long, bf16x8>;
  constexpr int KCH = KV * (DP / 8);        // 16-B chunks in a K tile
  constexpr int VCH = DP * (KV / 8);        // 16-B chunks in a V^T tile
  constexpr int KPT = (KCH + T - 1) / T, VPT = (VCH + T - 1) / T;
  constexpr int QW = NW / SP;               // query blocks (of 32) per workgroup
  static_assert(NW % SP == 0, "waves must divide into key groups");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;                              // [2][SP][KV][KS]
  char* Vs = smem + 2 * SP * Cfg::KBYTES;       // [2][SP][DT*32][VS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  // XCD-aware order: the dispatcher deals workgroups round-robin over the 8 XCDs, so with the natural order every XCD's L2
  // pulls every (batch, head)'s K / V^T (PMC: 70 MB fetched per launch for 16 MB of operands at N = 1000).  Give each XCD
  // whole (batch, head) pairs instead: all query blocks of a pair then share one L2.  Bijective when pairs % 8 == 0.
  int qblk = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  {
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
  const int grp = wave / QW;                    // key group of this wave
  const int q0 = (qblk * QW + (wave - grp * QW)) * 32;
  const bf16* qb = q + (long long)b * N * ldq + head * D;
  const bf16* kb = k + (long long)b * N * ldk + head * D;
  const bf16* vb = vt + (long long)b * vt_bs + (long long)head * D * vt_ld;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  // keys 0 .. Nk-1 of this batch item take part (right-padded text batches pass their true lengths; the key loop
  // then stops at the last valid tile instead of masking 512-token padding)
  const int Nk = kv_len ? max(1, min(kv_len[b], N)) : N;
  if (kv_len && qblk * QW * 32 >= Nk) {
    // every query of this workgroup is padding: its rows are never read as keys, so write zeros and leave
    for (int i = tid; i < QW * 32 * (D / 4); i += T) {
      const int row = qblk * QW * 32 + i / (D / 4), c4 = i % (D / 4);
      if (row < N) *reinterpret_cast<uint2*>(out + ((long long)b * N + row) * out_ld + head * D + c4 * 4) = make_uint2(0u, 0u);
    }
    return;
  }

  // rows D .. DT*32-1 of both V^T buffers are written once and stay for the whole kernel: zero (head-dim padding), except
  // row DP = ones (bf16 1.0 pairs): the P V MFMAs then also accumulate l = sum_k p in O^T row DP
  {
    const int npad = DT * 32 - D;
    for (int i = tid; i < 2 * SP * npad * (VS / 8); i += T) {
      const int buf = i / (npad * (VS / 8)), rem = i - buf * npad * (VS / 8);
      const unsigned fill = (!Cfg::VL && D + rem / (VS / 8) == DP) ? (FP8 ? 0x38383838u : 0x3F803F80u) : 0u;   // 1.0 as bf16 / e4m3 pairs
      reinterpret_cast<uint2*>(Vs + buf * Cfg::VBYTES + D * VS)[rem] = make_uint2(fill, fill);
    }
  }

  // Q fragments (B operand): lane (r, hh) holds Q[q0 + r][16 ks + 8 hh .. +7]
  frag_t qf[DK];
#pragma unroll
  for (int ks = 0; ks < DK; ++ks) {
    const int col = 16 * ks + 8 * hh;
    const bf16x8 qv = (q0 + r < N && col < D) ? *reinterpret_cast<const bf16x8*>(qb + (long long)(q0 + r) * ldq + col) : zero8;
    if constexpr (FP8) { const uint2 f = bf16x8_to_fp8(qv); qf[ks] = pack2(f.x, f.y); }
    else qf[ks] = qv;
  }

  // two register sets: the loads of tile-group it+2 are issued at the top of iteration it and written to LDS at the bottom of
  // iteration it+1 -- two iterations of flight time (one iteration, ~0.5 us, does not cover an L2 miss)
  bf16x8 kreg[2][SP][KPT], vreg[2][SP][VPT];
  // K / V^T tiles travel global -> registers -> LDS.  Buffer loads with the predicate folded into the descriptor's bound (an
  // out-of-range offset returns zeros): a predicated plain load compiles to a branch plus `s_waitcnt vmcnt(0)` in front of the
  // select, which would make every iteration wait for the tile requested one iteration earlier.
  // The per-lane byte offsets never change (loop-invariant VGPRs); the tile advance and the "keys < Nk" bound live in the
  // descriptor, rebuilt per tile with a few scalar instructions.  (An offset computed per tile ends up in the load's own
  // destination register, and writing that while the previous load into it is in flight costs an `s_waitcnt vmcnt(0)`.)
  constexpr unsigned OOB = 0x80000000u;
  unsigned k_off[KPT], v_off[VPT];
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    const int cidx = tid + i * T;
    const int row = cidx / (DP / 8), ch = cidx - row * (DP / 8);
    k_off[i] = (cidx < KCH && ch * 8 < D) ? (unsigned)row * (unsigned)(ldk * 2) + ch * 16 : OOB;
  }
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int cidx = tid + i * T;
    const int row = cidx >> 3, ch = cidx & 7;
    v_off[i] = (cidx < VCH && row < D) ? (unsigned)row * (unsigned)(vt_ld * 2) + ch * 16 : OOB;
  }
  auto prefetch_k = [&](int kvbase, auto setc) {
   constexpr int SET = decltype(setc)::value;
#pragma unroll
   for (int j = 0; j < SP; ++j) {
    const int kv0 = kvbase + j * KV;
    // rows kv0 .. Nk-1 of this head's K: the last valid byte is the end of row Nk-1's D columns
    const int bytes = max(0, ((Nk - kv0 - 1) * ldk + D) * 2);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(kb + (long long)kv0 * ldk), 0, kv0 < Nk ? bytes : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < KPT; ++i) kreg[SET][j][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, k_off[i], 0, 0));
   }
  };
  auto prefetch_v = [&](int kvbase, auto setc) {
   constexpr int SET = decltype(setc)::value;
#pragma unroll
   for (int j = 0; j < SP; ++j) {
    const int kv0 = kvbase + j * KV;
    // columns kv0 .. of this head's V^T rows; the bound is the end of the head's last row (a chunk that starts inside the row
    // padding or runs into the next row reads defined memory: stage_v clears every key >= Nk)
    const int bytes = max(0, (D * vt_ld - kv0) * 2);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(vb + kv0), 0, kv0 < Nk ? bytes : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < VPT; ++i) vreg[SET][j][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, v_off[i], 0, 0));
   }
  };
  auto prefetch = [&](int kvbase, auto setc) { prefetch_k(kvbase, setc); prefetch_v(kvbase, setc); };
  auto stage_k = [&](int buf, auto setc) {
   constexpr int SET = decltype(setc)::value;
#pragma unroll
   for (int j = 0; j < SP; ++j) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int cidx = tid + i * T;
      const int row = cidx / (DP / 8), ch = cidx - row * (DP / 8);
      if (cidx < KCH) {
        if constexpr (FP8) *reinterpret_cast<uint2*>(Ks + (buf * SP + j) * Cfg::KBYTES + row * KS + ch * 8) = bf16x8_to_fp8(kreg[SET][j][i]);
        else *reinterpret_cast<bf16x8*>(Ks + (buf * SP + j) * Cfg::KBYTES + row * KS + ch * 16) = kreg[SET][j][i];
      }
    }
   }
  };
  auto stage_v = [&](int buf, auto setc, int kvbase) {
   constexpr int SET = decltype(setc)::value;
#pragma unroll
   for (int j = 0; j < SP; ++j) {
    const int kv0 = kvbase + j * KV;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int cidx = tid + i * T;
      const int row = cidx >> 3, ch = cidx & 7;
      if (cidx < VCH) {
        bf16x8 v = vreg[SET][j][i];
        if (kv0 + KV > Nk) {                       // (wave-uniform) the tile that holds key Nk - 1: clear what lies beyond it --
#pragma unroll                                     //  p = 0 there, but those bytes are row padding / the next row, not zeros
          for (int e = 0; e < 8; ++e) if (kv0 + ch * 8 + e >= Nk) v[e] = (bf16)0.f;
        }
        if constexpr (FP8) {
          *reinterpret_cast<uint2*>(Vs + (buf * SP + j) * Cfg::VBYTES + row * VS + ch * 8) = bf16x8_to_fp8(v);
        } else {
          // rows are 136 B apart: 8-byte aligned only -> two 8-byte stores
          uint2* dst = reinterpret_cast<uint2*>(Vs + (buf * SP + j) * Cfg::VBYTES + row * VS + ch * 16);
          const uint4 u = __builtin_bit_cast(uint4, v);
          dst[0] = make_uint2(u.x, u.y);
          dst[1] = make_uint2(u.z, u.w);
        }
      }
    }
   }
  };
  auto stage = [&](int buf, auto setc, int kvbase) { stage_k(buf, setc); stage_v(buf, setc, kvbase); };

  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
  float m_run = 0.f;                            // running maximum of the SCALED scores (log2 domain); set by the first tile
  // ML (d = 32, 64): l = sum_k p on the matrix pipe without a row of ones in LDS -- one more MFMA per 16 keys whose A operand is a
  // REGISTER CONSTANT of ones, so every row of `lacc` is the query column's sum of the SAME bf16-rounded P that multiplies V.  Round 3
  // summed l with 16 v_pk_add_f32 per tile on the grounds that MFMA and VALU time add up; tools/micro/mfma_valu_gap.hip (instruction
  // stream pinned by inline asm) shows they do not: an MFMA costs its wave 8 issue cycles, its 32 pipe cycles run under the next five
  // VALU instructions (and under the SIMD's other wave), and this loop is VALU-issue-bound -- 4 x 8 issue cycles replace 16 packed adds.
  constexpr bool ML = Cfg::VL && ATTN_ML;
  float l_run = 0.f;                            // VL without ML: this lane's share of l (its 32 keys of every tile), VALU adds
  f32x2 l_pk = {0.f, 0.f};                      // ... accumulated as a PAIR (v_pk_add_f32: one instruction per two probabilities), folded into l_run per tile
  f32x16 lacc;
#pragma unroll
  for (int i = 0; i < 16; ++i) lacc[i] = 0.f;
  const bf16x8 ones8 = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};
  f32x16 minit;                                 // C operand of every tile's first S^T MFMA: -m_run (PRESCALED), else zeros
#pragma unroll
  for (int i = 0; i < 16; ++i) minit[i] = 0.f;

  const int ntiles = (Nk + KV - 1) / KV;
  const int niter = (ntiles + SP - 1) / SP;
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;

  // ---- S' = K Q^T (- m_run as the initial accumulator when the scores come out of the MFMA already scaled) ----
  auto scores = [&](int it, f32x16 (&s)[2]) {
    const int buf = (it & 1) * SP + grp;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int ks = 0; ks < DK; ++ks) {
        // the first K-step reads its C operand from `minit` (-m_run in every row; zeros for unscaled scores) and writes a DIFFERENT
        // register block: no 16 v_mov per 32 keys to seed the accumulator
        if constexpr (FP8) {
          const uint2 kf = *reinterpret_cast<const uint2*>(Ks + buf * Cfg::KBYTES + (sub * 32 + r) * KS + 16 * ks + 8 * hh);
          s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(pack2(kf.x, kf.y), qf[ks], ks == 0 ? minit : s[sub], 0, 0, 0);
        } else {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + buf * Cfg::KBYTES + (sub * 32 + r) * KS + (16 * ks + 8 * hh) * 2);
          s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? minit : s[sub], 0, 0, 0);
        }
      }
    }
  };
  // ---- online softmax of tile `it` (scores in s) and O^T += V^T P^T ----
  auto softmax_pv = [&](int it, f32x16 (&s)[2]) {
    const int buf = (it & 1) * SP + grp, kv0 = (it * SP + grp) * KV;
    const bool first = it == 0;
    if (kv0 + KV > Nk) {  // tail tile: mask keys >= Nk
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int kvr = kv0 + sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (kvr >= Nk) s[sub][i] = -INFINITY;
        }
    }
    // lane-local: this lane's query column; deferred rescale
#ifdef ATTN_DIAG_NOMAX
    float mx = s[0][0];
#else
    // (both chains START from a compiler-visible v_max_f32 over the last register of each MFMA result: max3f is inline asm, and the
    //  compiler inserts the MFMA -> VALU wait states only in front of its OWN instructions; every asm instruction depends on these two)
    float mx0 = fmaxf(s[0][0], s[1][15]), mx1 = fmaxf(s[1][0], s[0][15]);                    // two independent chains
#pragma unroll
    for (int i = 1; i < 15; i += 2) {
      mx0 = max3f(mx0, s[0][i], s[0][i + 1]);
      mx1 = max3f(mx1, s[1][i], s[1][i + 1]);
    }
    float mx = fmaxf(mx0, mx1);
    {   // the other 32 keys of this query column sit in lane ^ 32: v_permlane32_swap (VALU) instead of an LDS round trip.
        // COMPILER PITFALL (round 4, found through the fp8 form): with __builtin_amdgcn_permlane32_swap the `fmaxf(sw[0], sw[1])` over
        // the builtin's two results never reached the ISA -- only the swap's FIRST result was used (ROCm 7.2; an opaque copy of the second
        // operand did not change that), so every column tracked the maximum of HALF its keys (lanes >= 32 took the lower half's).
        // Harmless in bf16 (any reference point is a valid softmax shift and P stays in range), an overflow of e4m3's 448 in a quarter of
        // the rows with fp8 operands.  The instruction is therefore written out, with the wait states its operands need.
      float mxo = mx;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(mx), "+v"(mxo));
      mx = fmaxf(mx, mxo);
    }
#endif
    // excess of this tile's maximum over the running one, in the scaled (log2) domain
    const float ex = PRESCALED ? (first ? mx : mx - PBIAS) : fmaf(mx, c, -m_run);
    // (FP8: P x 2^8 must stay below e4m3's 448, so the maximum is tracked exactly -- no deferred rescale)
    if (first || !__all(ex <= (FP8 ? 0.f : RESCALE_THR))) {
      const float delta = first ? ex : fmaxf(ex, 0.f);                  // the maximum never decreases after the first tile
      m_run += delta;
      if (!first) {
        const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[t][i] *= alpha;                // includes l (row DP)
        if constexpr (ML) {
#pragma unroll
          for (int i = 0; i < 16; ++i) lacc[i] *= alpha;
        } else if constexpr (Cfg::VL) {
          l_pk *= alpha;
        }
      }
      if (PRESCALED) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int i = 0; i < 16; ++i) s[sub][i] -= first ? delta - PBIAS : delta;   // (the first tile's scores came out without any bias)
#pragma unroll
        for (int i = 0; i < 16; ++i) minit[i] = PBIAS - m_run;    // (FP8: the 2^8 of P rides in the initial accumulator too)
      }
    }
    // ---- p = exp2(s'), O^T += V^T P^T (row DP of V^T is ones: l accumulates alongside; ML: one more MFMA against a register of ones).
    // One 16-key step at a time, fenced: the 8 exponentials + 4 conversions of step s2 + 1 issue while the matrix pipe works on step
    // s2's MFMAs (an MFMA costs its wave 8 issue cycles; its 32 pipe cycles hide under up to ~24 cycles of the wave's next VALU
    // instructions -- tools/micro/mfma_valu_gap.hip).  The barrier at the end of every tile keeps the SIMD's two waves in phase, so
    // nothing but the wave's own instruction stream fills those cycles: left to the scheduler, all 32 exponentials came first and
    // the MFMAs ran back to back behind them, pipe-bound in both waves at once.
    frag_t pf[4];
    auto vfrag = [&](int t, int s2) -> frag_t {
      if constexpr (FP8) {                      // the lane's 8 keys = {16 s2 + 4 hh + 0..3, + 8..11}: two 4-byte reads
        const char* vrow = Vs + buf * Cfg::VBYTES + (t * 32 + r) * VS + 16 * s2 + 4 * hh;
        return pack2(*reinterpret_cast<const unsigned*>(vrow), *reinterpret_cast<const unsigned*>(vrow + 8));
      } else {
        const char* vrow = Vs + buf * Cfg::VBYTES + (t * 32 + r) * VS + (16 * s2 + 4 * hh) * 2;
        const uint2 lo = *reinterpret_cast<const uint2*>(vrow);
        const uint2 hi = *reinterpret_cast<const uint2*>(vrow + 16);
        return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
      }
    };
    frag_t vf[2][DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) vf[0][t] = vfrag(t, 0);
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {          // 4 K-steps of 16 keys
      const int sub = s2 >> 1, i0 = (s2 & 1) * 8;
      if (s2 + 1 < 4) {
#pragma unroll
        for (int t = 0; t < DT; ++t) vf[(s2 + 1) & 1][t] = vfrag(t, s2 + 1);   // next step's V^T fragments: requested before this step's exponentials
      }
      int pw8[2] = {0, 0};                       // FP8: the step's 8 probabilities as e4m3 bytes
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const float e0 = PRESCALED ? s[sub][i0 + i] : fmaf(s[sub][i0 + i], c, PBIAS - m_run);
        const float e1 = PRESCALED ? s[sub][i0 + i + 1] : fmaf(s[sub][i0 + i + 1], c, PBIAS - m_run);
#ifdef ATTN_DIAG_NOEXP
        const float p0 = e0, p1 = e1;
#else
        const float p0 = __builtin_amdgcn_exp2f(e0), p1 = __builtin_amdgcn_exp2f(e1);
#endif
#ifdef ATTN_FP8_CLAMP
        const float p0c = fminf(p0, 448.f), p1c = fminf(p1, 448.f);
#define P0 p0c
#define P1 p1c
#else
#define P0 p0
#define P1 p1
#endif
        if constexpr (Cfg::VL && !ML) l_pk += f32x2{p0, p1};
        if constexpr (FP8) {
          pw8[i >> 2] = (i & 2) ? __builtin_amdgcn_cvt_pk_fp8_f32(P0, P1, pw8[i >> 2], true) : __builtin_amdgcn_cvt_pk_fp8_f32(P0, P1, pw8[i >> 2], false);
        } else {
          pf[s2][i] = (bf16)p0;
          pf[s2][i + 1] = (bf16)p1;
        }
      }
      if constexpr (FP8) pf[s2] = pack2((unsigned)pw8[0], (unsigned)pw8[1]);
      if (ATTN_FENCE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        if constexpr (FP8) o[t] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(vf[s2 & 1][t], pf[s2], o[t], 0, 0, 0);
        else o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s2 & 1][t], pf[s2], o[t], 0, 0, 0);
      }
      if constexpr (ML && !FP8) lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones8, pf[s2], lacc, 0, 0, 0);
      if (ATTN_FENCE) __builtin_amdgcn_sched_barrier(0);
    }
  };
  // (wave-uniform) a key group past the last tile of an odd count sits that iteration out
  auto mine = [&](int it) { return SP == 1 || (it * SP + grp) * KV < Nk; };

  prefetch(0, Set0{});
  prefetch(SP * KV, Set1{});
  stage(0, Set0{}, 0);
  // Drain every global load issued so far (Q fragments, both prefetch sets) HERE, with the builtin the compiler's wait-count pass
  // models: otherwise the first use of the Q fragments inside the loop gets an `s_waitcnt vmcnt(0)` that, from the second
  // iteration on, waits for the K / V^T prefetch issued a few instructions earlier -- a full L2 round trip exposed per tile, in
  // every wave at once (the ISA showed exactly that).  vmcnt = 0, expcnt / lgkmcnt untouched.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  // two register sets: the loads of tile-group it+2 are issued at the top of iteration it and written to LDS at the bottom of
  // iteration it+1 -- two iterations of flight time (one iteration, ~0.5 us, does not cover an L2 miss)
  auto iteration = [&](int it, auto curc) {     // curc: the register set that is free (its tile-group was staged last iteration)
    constexpr int CUR = decltype(curc)::value;
    // Order inside a tile: the S^T MFMAs first, then -- in their shadow, nothing depends on them yet -- the next tile's staging
    // (registers -> LDS; its buffer was last read before the barrier that ended the previous iteration) and the prefetch of the tile
    // after it (scalar descriptor arithmetic + buffer loads), then the softmax and the P V MFMAs.  Fenced, because left alone the
    // scheduler put the staging behind the last P V MFMA, where nothing runs beside it.
    f32x16 s[2];
    const bool me = mine(it);
#if ATTN_FENCE
    if (me) scores(it, s);
    __builtin_amdgcn_sched_barrier(0);
    if (it + 1 < niter) stage((it & 1) ^ 1, std::integral_constant<int, CUR ^ 1>{}, (it + 1) * SP * KV);
    prefetch((it + 2) * SP * KV, curc);         // (unconditional: past the last tile the descriptor's bound is 0 -> zeros, no traffic;
                                                //  a fixed number of loads per iteration lets the compiler count them in s_waitcnt)
    __builtin_amdgcn_sched_barrier(0);
    if (me) softmax_pv(it, s);
#else
#ifndef ATTN_DIAG_NOLOAD
    prefetch((it + 2) * SP * KV, curc);
#endif
    if (me) {
      scores(it, s);
      softmax_pv(it, s);
    }
#ifndef ATTN_DIAG_NOLOAD
    if (it + 1 < niter) stage((it & 1) ^ 1, std::integral_constant<int, CUR ^ 1>{}, (it + 1) * SP * KV);
#endif
#endif
    __syncthreads();
  };
  for (int it = 0; it < niter; it += 2) {
    iteration(it, Set0{});
    if (it + 1 < niter) iteration(it + 1, Set1{});
  }
  if constexpr (Cfg::VL && !ML) l_run = l_pk[0] + l_pk[1];

  if constexpr (SP > 1) {
    // ---- merge the key groups: groups 1.. leave (m, l, O^T) in LDS (the staging buffers are dead after the last barrier),
    // group 0 folds them in.  Layout [value][lane] per wave: conflict-free 4-byte accesses. ----
    constexpr int WVALS = DT * 16 + 2;                           // m, O^T, l_run (VL)
    float* xch = reinterpret_cast<float*>(smem);
    const bool idle = grp * KV >= Nk;                            // (wave-uniform) this group never saw a tile: O^T = 0, l = 0
    if (grp > 0) {
      float* mine = xch + ((grp - 1) * QW + (wave - grp * QW)) * WVALS * 64;
      mine[lane] = idle ? -INFINITY : m_run;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) mine[(1 + t * 16 + i) * 64 + lane] = o[t][i];
      mine[(1 + DT * 16) * 64 + lane] = ML ? lacc[0] : l_run;
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g = 1; g < SP; ++g) {
      const float* peer = xch + ((g - 1) * QW + wave) * WVALS * 64;
      const float m1 = peer[lane];
      const float m_new = fmaxf(m_run, m1);                      // m_run is finite: group 0 always owns tile 0
      const float a0 = __builtin_amdgcn_exp2f(m_run - m_new), a1 = __builtin_amdgcn_exp2f(m1 - m_new);   // a1 = 0 for an idle group
      m_run = m_new;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = o[t][i] * a0 + peer[(1 + t * 16 + i) * 64 + lane] * a1;   // l (row DP) merges like any row
      lacc[0] = lacc[0] * a0 + peer[(1 + DT * 16) * 64 + lane] * a1;
      l_run = l_run * a0 + peer[(1 + DT * 16) * 64 + lane] * a1;
    }
  }

  // ---- normalise and store: lane owns query q0 + r, rows of O^T are head-dim indices ----
  // l sits in O^T row DP: register ONES_I of tile ONES_T on the hh = 0 lanes (the hh = 1 lanes hold row DP + 4 there: zero padding)
  // (VL: every row of lacc holds the whole sum for this lane's query column -- all 64 keys of every tile went through the MFMA's k)
  const float l_half = (Cfg::VL && !ML) ? l_run : o[Cfg::ONES_T][Cfg::ONES_I];   // (VALU form: this lane's 32 keys per tile; the other 32 sit in lane ^ 32)
  const float l_tot = ML ? lacc[0] : l_half + __shfl_xor(l_half, 32, 64);
  const float inv = 1.0f / l_tot;
  if (lse && hh == 0 && q0 + r < N)   // log2-domain log-sum-exp of the scaled scores: p = exp2(s*c - lse)
    lse[((long long)b * gridDim.y + head) * N + q0 + r] = m_run + __log2f(l_tot);
  if (q0 + r < N) {
    bf16* orow = out + ((long long)b * N + q0 + r) * out_ld + head * D;
#pragma unroll
    for (int t = 0; t < (DP + 31) / 32; ++t)
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

template <int DP, int NW, int SP, bool PS, bool FP8 = false>
int launch_attn(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld, long long vt_bs, int B, int N,
                int H, int D, float scale, void* out, int out_ld, float* lse, const int* kv_len, hipStream_t st) {
  using Cfg = AttnCfg<DP, FP8>;
  auto kern = attention_kernel<DP, NW, SP, PS, FP8>;
  constexpr int MERGE = SP == 1 ? 0 : (SP - 1) * (NW / SP) * (Cfg::DT * 16 + 2) * 256;   // the key groups' (m, O^T, l) exchange re-uses the staging area
  constexpr int LDS = 2 * SP * Cfg::TILE > MERGE ? 2 * SP * Cfg::TILE : MERGE;             // (one-byte images can be smaller than it)
  static unsigned long long attr_done = 0;   // per-device bit mask (aldm_set_max_lds)
  if (LDS > 48 * 1024)
    if (int rc = aldm_set_max_lds(reinterpret_cast<const void*>(kern), LDS, &attr_done, "attention")) return rc;
  dim3 grid(cdiv(N, 32 * (NW / SP)), H, B);
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), LDS, st, (const bf16*)q, ldq, (const bf16*)k, ldk, (const bf16*)vt,
                     vt_ld, vt_bs, N, D, scale * 1.44269504088896340736f, (bf16*)out, out_ld, lse, kv_len);
  return aldm_launch_status("attention");
}

template <int DP, bool PS, bool FP8 = false>
int launch_attn_d(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld, long long vt_bs, int B, int N,
                  int H, int D, float scale, void* out, int out_ld, float* lse, const int* kv_len, hipStream_t st) {
  // measured in a replayed graph: N = 1000 (d 32) 4 waves 24 us vs 2 waves 31; N = 252 (d 48) 4 waves 8.9 us vs 2 waves 9.4;
  // N = 64 (d 80) 2 waves 5.6 us vs 1 wave 6.9 -- sharing one K / V^T staging among the waves beats more, smaller workgroups
  // 8 waves (256 queries per workgroup) once that still leaves a workgroup per CU: 24.1 -> 22.9 us at N = 1000, 8 x 8 heads
  // keys split over two wave groups (SP = 2) where a query block per wave leaves SIMDs idle: N = 252 (d 48) 9.0 -> 7.2 us,
  // N = 1000 with 4 heads 16.6 -> 14.6 us; at N = 1000 x 64 (batch, head) pairs the kernel is VALU-throughput-bound (softmax)
  // and the split only adds the merge: 22.9 -> 23.4 us, so the full-size case keeps one wave per query block
#define ALDM_ATTN_ARGS q, ldq, k, ldk, vt, vt_ld, vt_bs, B, N, H, D, scale, out, out_ld, lse, kv_len, st
  if constexpr (FP8) {                        // (the tuning overrides below are not instantiated for the fp8 form)
    if (N >= 768) {
      if ((long long)cdiv(N, 256) * H * B >= 256) return launch_attn<DP, 8, 1, PS, true>(ALDM_ATTN_ARGS);
      return launch_attn<DP, 8, 2, PS, true>(ALDM_ATTN_ARGS);
    }
    if (N >= 192) return launch_attn<DP, 4, 2, PS, true>(ALDM_ATTN_ARGS);
    if (N >= 64) return launch_attn<DP, 4, 1, PS, true>(ALDM_ATTN_ARGS);
    return launch_attn<DP, 1, 1, PS, true>(ALDM_ATTN_ARGS);
  }
  static const int force = getenv("ALDM_ATTN_CFG") ? atoi(getenv("ALDM_ATTN_CFG")) : 0;   // tuning aid: 10 * waves + key split
  if (force == 81) return launch_attn<DP, 8, 1, PS>(ALDM_ATTN_ARGS);
  if (force == 82) return launch_attn<DP, 8, 2, PS>(ALDM_ATTN_ARGS);
  if (force == 41) return launch_attn<DP, 4, 1, PS>(ALDM_ATTN_ARGS);
  if (force == 42) return launch_attn<DP, 4, 2, PS>(ALDM_ATTN_ARGS);
  if (force == 21) return launch_attn<DP, 2, 1, PS>(ALDM_ATTN_ARGS);
  if constexpr (DP == 32) {
    if (force == 162) return launch_attn<DP, 16, 2, PS>(ALDM_ATTN_ARGS);   // 16 waves = 4 per SIMD, keys split over two wave groups
  }
  if (N >= 768) {
    if ((long long)cdiv(N, 256) * H * B >= 256) return launch_attn<DP, 8, 1, PS>(ALDM_ATTN_ARGS);
    return launch_attn<DP, 8, 2, PS>(ALDM_ATTN_ARGS);
  }
  if (N >= 192) return launch_attn<DP, 4, 2, PS>(ALDM_ATTN_ARGS);
  if (N >= 64) return launch_attn<DP, 4, 1, PS>(ALDM_ATTN_ARGS);   // N = 64 x 64 (batch, head) pairs: 4 waves 5.2 us vs 2 waves 5.7 us
  return launch_attn<DP, 1, 1, PS>(ALDM_ATTN_ARGS);
#undef ALDM_ATTN_ARGS
}

// ---- row softmax (VAE mid-block attention runs QK^T / PV through the GEMM kernel) ----
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, int rows, int cols, int ld_in,
                                                           float c, bf16* __restrict__ p, int ld_out) {
  __shared__ float red[8];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* sr = s + (long long)row * ld_in;
  float mx = -INFINITY;
  for (int i = tid; i < cols; i += 256) mx = fmaxf(mx, sr[i]);
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int i = tid; i < cols; i += 256) sum += __builtin_amdgcn_exp2f((sr[i] - mx) * c);
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  bf16* pr = p + (long long)row * ld_out;
  for (int i = tid; i < ld_out; i += 256) pr[i] = i < cols ? (bf16)(__builtin_amdgcn_exp2f((sr[i] - mx) * c) * inv) : (bf16)0.f;
}

}  // namespace

template <bool PS, bool FP8 = false>
static int attention_impl(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld,
                          long long vt_batch_stride, int B, int N, int H, int d, float scale, void* out, int out_ld,
                          float* lse, const int* kv_len, void* stream) {
  ALDM_CHECK_ARG(q && k && vt && out, "attention: null pointer");
  ALDM_CHECK_ARG(B > 0 && N > 0 && H > 0 && d > 0, "attention: bad dims");
  ALDM_CHECK_ARG(d % 8 == 0 && ldq % 8 == 0 && ldk % 8 == 0 && vt_ld % 8 == 0 && out_ld % 4 == 0, "attention: d/ld must be multiples of 8");
  ALDM_CHECK_ARG(vt_ld >= ((N + 7) / 8) * 8, "attention: vt_ld %d too small for N %d", vt_ld, N);
  hipStream_t st = (hipStream_t)stream;
#define ALDM_ATTN(DPV) return launch_attn_d<DPV, PS, FP8>(q, ldq, k, ldk, vt, vt_ld, vt_batch_stride, B, N, H, d, scale, out, out_ld, lse, kv_len, st)
  if (d <= 16) ALDM_ATTN(16);
  if (d <= 32) ALDM_ATTN(32);
  if (d <= 48) ALDM_ATTN(48);
  if (d <= 64) ALDM_ATTN(64);
  if (d <= 80) ALDM_ATTN(80);
#undef ALDM_ATTN
  aldm_set_error("attention: head dim %d > 80 is handled by the GEMM path", d);
  return ALDM_E_UNSUPPORTED;
}

extern "C" int aldm_attention(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld,
                              long long vt_batch_stride, int B, int N, int H, int d, float scale, void* out, int out_ld,
                              void* stream) {
  return attention_impl<false>(q, ldq, k, ldk, vt, vt_ld, vt_batch_stride, B, N, H, d, scale, out, out_ld, nullptr, nullptr, stream);
}

// Q already carries scale * log2(e) (the caller folded it into the Q projection weights): the scores leave the MFMA in their
// final log2 domain and the running maximum enters as the MFMA's initial accumulator -- no per-score multiply or subtract.
extern "C" int aldm_attention_prescaled(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld,
                                        long long vt_batch_stride, int B, int N, int H, int d, void* out, int out_ld,
                                        void* stream) {
  return attention_impl<true>(q, ldq, k, ldk, vt, vt_ld, vt_batch_stride, B, N, H, d, 1.0f, out, out_ld, nullptr, nullptr, stream);
}

// BASELINE config 5: the same core with fp8 (OCP e4m3) Q / K / V / P MFMA operands and fp32 accumulation (attention_kernel<..., FP8>).
// scale * log2(e) == 1 means Q is pre-scaled (the UNet folds d^-0.5 log2 e into to_q): the PRESCALED form, no per-score multiply.
extern "C" int aldm_attention_fp8(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld, long long vt_batch_stride,
                                  int B, int N, int H, int d, float scale, void* out, int out_ld, void* stream) {
  if (fabsf(scale * 1.44269504088896340736f - 1.0f) < 1e-6f)
    return attention_impl<true, true>(q, ldq, k, ldk, vt, vt_ld, vt_batch_stride, B, N, H, d, 1.0f, out, out_ld, nullptr, nullptr, stream);
  return attention_impl<false, true>(q, ldq, k, ldk, vt, vt_ld, vt_batch_stride, B, N, H, d, scale, out, out_ld, nullptr, nullptr, stream);
}

extern "C" int aldm_attention_lse(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld,
                                  long long vt_batch_stride, int B, int N, int H, int d, float scale, void* out, int out_ld,
                                  float* lse, void* stream) {
  ALDM_CHECK_ARG(lse, "attention_lse: null lse");
  return attention_impl<false>(q, ldq, k, ldk, vt, vt_ld, vt_batch_stride, B, N, H, d, scale, out, out_ld, lse, nullptr, stream);
}

extern "C" int aldm_attention_varlen(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld,
                                     long long vt_batch_stride, int B, int N, int H, int d, float scale,
                                     const int* kv_len, void* out, int out_ld, void* stream) {
  ALDM_CHECK_ARG(kv_len, "attention_varlen: null kv_len");
  return attention_impl<false>(q, ldq, k, ldk, vt, vt_ld, vt_batch_stride, B, N, H, d, scale, out, out_ld, nullptr, kv_len, stream);
}

extern "C" int aldm_softmax_rows(const float* s, int rows, int cols, int ld_in, float scale, void* p, int ld_out,
                                 void* stream) {
  ALDM_CHECK_ARG(s && p && rows > 0 && cols > 0 && ld_in >= cols && ld_out >= cols, "softmax_rows: bad args");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, s, rows, cols, ld_in,
                     scale * 1.44269504088896340736f, (bf16*)p, ld_out);
  return aldm_launch_status("softmax_rows");
}

// Projection GEMM for the transformer blocks: y[M][N] = x[M][K] W[N][K]^T with K = 256 / 384 / 640 at compile time.
//
// Serves the linear layers of BasicTransformerBlock / Transformer2DModel inside UNet2DConditionModel.forward
// [REF script/train/train_audioldm_lora.py:539-546] / the DDIM loop [REF script/inference/generate_audio.py:47-52]:
//   proj_in (1x1 conv), to_q | to_k | to_v (one GEMM, LayerNorm folded, peft LoRA side channel [REF train:378-385], V stored
//   token-major), to_out.0 (+ LoRA + bias + residual), the GEGLU projection (LayerNorm folded).
// These ran on the implicit-GEMM CONVOLUTION kernel (igemm_core.h): tap cursor, descriptor range logic, magic divisions, an
// LDS transposition in the epilogue -- 674 VALU instructions per wave against 52 MFMAs (profiles/r02d_pmc_sq.json).  With K this
// short the structure is turned around:
//   * a wave owns 16 MI rows of x for the WHOLE K: the fragments go global -> registers once (B operand of the "swapped"
//     v_mfma_f32_16x16x32_bf16, weights = A operand) and never touch LDS;
//   * the workgroup (4 waves, 64 MI rows) walks over its range of output columns in N-tiles of NT columns: a tile = NT weight
//     rows x full K, streamed global -> LDS by LDS-DMA (buffer_load ... lds) into a 3-stage ring, XOR-swizzled on the SOURCE
//     side so that every ds_read_b128 fragment read is conflict-free; one counted s_waitcnt + one s_barrier per N-tile;
//   * no K loop state at all, no divisions except the V^T (batch, token) split, and the accumulators are stored STRAIGHT from
//     registers: the rows of a weight tile are handed to the MFMA in a permuted order (lane (c, q) reads row
//     32 t + 8 (r >> 2) + 4 s + (r & 3) for sub-tile 2 t + s), so that a lane ends up with 8 CONSECUTIVE output columns of one row
//     -> one 16-byte store, no LDS transposition.  The V^T tiles use the un-swapped operand order and the same trick on the row
//     side (a lane owns 8 consecutive tokens of one channel).
//   * LoRA: T = x A^T from LoRA-A fragments loaded straight into registers (16 ranks per MFMA tile), rounded to bf16 in
//     registers and applied to every N-tile as one more K = 32 step against the pre-scaled B rows (k order = accumulator row
//     permutation, reproduced by two 8-byte reads of B);
//   * LayerNorm folded (statistics from the producer's rowstat table), LayerNorm statistics for the NEXT consumer, bias and
//     residual (residual tile by LDS-DMA, wave-private) in the epilogue.
#include "igemm_core.h"   // make_rsrc / lds_ptr_t / wait_vmcnt / FastDiv

namespace {

using aldm_igemm_detail::FastDiv;
using aldm_igemm_detail::fdiv;
using aldm_igemm_detail::lds_ptr_t;
using aldm_igemm_detail::make_fastdiv;
using aldm_igemm_detail::wait_vmcnt;

// Kernel arguments come in two parts.  The leading SCALARS (16 dwords at most) are pre-loaded into SGPRs with the wave
// (-amdgpu-kernarg-preload-count): everything the weight-tile DMA and the x-fragment loads need, so those can be issued in the
// kernel's first ~100 cycles.  The rest (PgArgs, by value) lives in the kernel-argument segment, which is cold in a replayed graph:
// its first fetch costs ~1 us -- now spent with tile 0 and the x fragments already in flight.
struct PgArgs {
  const bf16* lora_a; const bf16* lora_b;
  const float* bias; const float* ln_s; const float* ln_sa; const float* ln_ca; const float* ln_parts;
  const bf16* res; bf16* out; bf16* vt; float* rowstat; bf16* lora_t;
  int Rp, ln_np, out_ld;
  int vt_col0, vt_ld, OHW, vt_vec;        // vt_vec: tokens per V^T store (8, 4 or 1)
  int dual;                               // EPI_VT: tiles at / beyond vt_col0 are ALSO stored row-major
  long long vt_bs;
  FastDiv fd_ohw;
  float ln_eps;
  unsigned long long* diag;                // diagnostic builds (make DIAG=1, tools/diag_pgemm.py): s_memtime stamps; null otherwise
};
#ifndef PG_LDS_PAD
#define PG_LDS_PAD 0
#endif
constexpr int PG_STRUCT_OFFSET = 48;         // of PgArgs in the kernel-argument segment: 11 scalar dwords, 8-byte aligned
struct PgHost {                            // host-side launch record: the scalar arguments + the struct
  const bf16* x; const bf16* w;
  int M, N, nranges, tpr;                  // column ranges per row block; N-tiles per range
  PgArgs a;
};

#ifdef ALDM_DIAG   // stamps are compiled in only on request (even a never-taken branch perturbs the wait-count bookkeeping)
#define PG_STAMP(i) if (p.diag && lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == nwg - 1)) { unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); p.diag[(blockIdx.x == 0 ? 0 : 64) + (i)] = t_; }
#else
#define PG_STAMP(i)
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

enum { PG_LN = 1, PG_RES = 2, PG_RSTAT = 4 };
enum { EPI_STD = 0, EPI_VT = 1, EPI_GEGLU = 2 };

typedef float f32x2 __attribute__((ext_vector_type(2)));
// two fp32 -> one dword of bf16 (v_cvt_pk_bf16_f32 with both sources; converting element by element costs a cvt AND a v_perm each)
__device__ __forceinline__ bf16x2 pg_pk(float a, float b) { return __builtin_convertvector(f32x2{a, b}, bf16x2); }
__device__ __forceinline__ bf16x8 pg_cat(bf16x2 a, bf16x2 b, bf16x2 c2, bf16x2 d) {
  return __builtin_shufflevector(__builtin_shufflevector(a, b, 0, 1, 2, 3), __builtin_shufflevector(c2, d, 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ int pg_hsw(int R) { return ((R & 3) | (((R >> 3) & 1) << 2)) << 1; }   // chunk XOR of weight row R

template <int K, int NT, int MI, int RT, int EPI, int FL, int NW>
__global__ __launch_bounds__(64 * NW) void pgemm_kernel(const bf16* __restrict__ xg, const bf16* __restrict__ wg, const int M, const int N,
                                                    const int nranges, const int T, const unsigned fd_mul, const unsigned fd_shift,
                                                    const int nwg, const PgArgs p_segment) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool LN = (FL & PG_LN) != 0, RES = (FL & PG_RES) != 0, RSTAT = (FL & PG_RSTAT) != 0;
  constexpr int BM = 16 * MI * NW, NTHR = 64 * NW, KS = K / 32, CPR = K / 8, NJ = NT / 16;
  constexpr int WT = NT * K * 2;                              // bytes of a weight tile
  constexpr int PWW = WT / 1024 / NW;                         // its 1 KB DMA pieces per wave
  constexpr int RB = RES ? BM * NT * 2 : 0;                   // residual tile (wave w: rows 16 MI w ..)
  constexpr int PRW = RES ? (16 * MI * NT * 2) / 1024 : 0;
  constexpr int STG = WT + RB, P = PWW + PRW;
  constexpr int RCH = NT / 8;                                 // 16-byte chunks per residual row
  static_assert(WT % (1024 * NW) == 0, "weight tile = whole DMA pieces per wave");
  static_assert(!RES || (16 * MI * NT * 2) % 1024 == 0, "residual tile = whole DMA pieces per wave");
  static_assert(EPI != EPI_GEGLU || NT == 64, "GEGLU tiles are 64 packed columns (16 value | 16 gate blocks)");
  static_assert(NJ % 2 == 0, "column sub-tiles come in pairs");
  static_assert(P < 64, "vmcnt immediate");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  int wgid;
  {   // XCD-aware order (blocks b, b + 8, ... share an L2): the column ranges of one row block run on one XCD
    const int bid = blockIdx.x;                               // (nwg = gridDim.x as an argument: the grid size itself is an
    const int xcd = bid & 7, qq = nwg >> 3, r = nwg & 7;      //  IMPLICIT kernel argument, i.e. a load from the cold segment)
    wgid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (bid >> 3);
  }
  const int mblk = fdiv(wgid, FastDiv{fd_mul, fd_shift}), nr = wgid - mblk * nranges;
  const int BNW = T * NT, BNWP = BNW;                        // columns of the range
  const int m0 = mblk * BM, n_wg0 = nr * BNW;
  const int mw0 = wave * 16 * MI;
  const int SR = T < 3 ? T : 3;                                // ring stages in use
  float* const stat = reinterpret_cast<float*>(smem + SR * STG);   // [2][BM]: mean, rstd
  float* const cvec = stat + 2 * BM;                          // [2][BNW]: c_n (bias), s_n
  char* const LB = reinterpret_cast<char*>(cvec + 2 * BNWP);  // [BNW][Rp] bf16: the range's pre-scaled LoRA-B rows

  const __amdgpu_buffer_rsrc_t rs_x = aldm_igemm_detail::make_rsrc(xg, (unsigned)M * (unsigned)(K * 2));
  const __amdgpu_buffer_rsrc_t rs_w = aldm_igemm_detail::make_rsrc(wg, (unsigned)N * (unsigned)(K * 2));

  // ---- the weight stream.  Per-lane source offsets of this wave's pieces of a tile: physical chunk u of the linear LDS image =
  //      (row R, chunk pc); it receives logical chunk pc ^ hsw(R) (both-sides-or-neither: the fragment reads apply the same XOR).
  unsigned w_off[PWW];
#pragma unroll
  for (int pp = 0; pp < PWW; ++pp) {
    const int u = (wave * PWW + pp) * 64 + lane;
    const int R = u / CPR, pc = u - R * CPR;
    w_off[pp] = (unsigned)(R * (K * 2) + ((pc ^ pg_hsw(R)) * 16));
  }
  auto issue_w = [&](int t) {                                 // N-tile t of the range -> stage t % 3
    char* const sb = smem + (t % 3) * STG;
    const int soff = (n_wg0 + t * NT) * (K * 2);
#pragma unroll
    for (int pp = 0; pp < PWW; ++pp)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(sb + (wave * PWW + pp) * 1024), 16, w_off[pp], soff, 0, 0);
  };
  issue_w(0);                                                 // tile 0 is on its way ~100 cycles into the kernel

  // ---- x fragments of this wave's rows, straight into registers.  Lane (c, q) holds x[row rl(i, c)][32 ks + 8 q .. + 7] with
  //      rl(i, c) = 4 MI (c >> 2) + 4 i + (c & 3): as the A operand of the un-swapped (V^T) product a lane then owns 4 MI
  //      consecutive rows.  Rows >= M read as zeros through the descriptor's range check. ----
  int mrow[MI];
  bf16x8 xf[MI][KS];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    mrow[i] = m0 + mw0 + 4 * MI * (c >> 2) + 4 * i + (c & 3);
    const int off = mrow[i] * (K * 2) + q * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[i][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off + ks * 64, 0, 0));
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- everything below needs the argument struct: its (cold) fetch overlaps the loads above.  The struct is read through a
  //      laundered pointer into the kernel-argument segment: left to itself the compiler hoists these (invariant) scalar loads to
  //      the kernel's first instruction, and the first SGPR it then recycles waits for all of them -- ~1 us before tile 0's DMA. ----
  PgArgs p;
  {
    typedef const char __attribute__((address_space(4))) * ka_ptr_t;
    ka_ptr_t ka = (ka_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka) : : "memory");
    aldm_touch_kernargs<PG_STRUCT_OFFSET + sizeof(PgArgs)>();   // every 64-byte line of the segment in ONE round trip (common.h)
#ifndef ALDM_NO_KA_PREFETCH
    aldm_prefetch_next_kernargs<PG_STRUCT_OFFSET + sizeof(PgArgs)>(threadIdx.x);
#endif
    typedef const PgArgs __attribute__((address_space(4))) * pa_ptr_t;
    pa_ptr_t ps = (pa_ptr_t)(ka + PG_STRUCT_OFFSET);
    p.lora_a = ps->lora_a; p.lora_b = ps->lora_b; p.bias = ps->bias; p.ln_s = ps->ln_s; p.ln_sa = ps->ln_sa; p.ln_ca = ps->ln_ca;
    p.ln_parts = ps->ln_parts; p.res = ps->res; p.out = ps->out; p.vt = ps->vt; p.rowstat = ps->rowstat; p.lora_t = ps->lora_t;
    p.Rp = ps->Rp; p.ln_np = ps->ln_np; p.out_ld = ps->out_ld; p.vt_col0 = ps->vt_col0; p.vt_ld = ps->vt_ld; p.OHW = ps->OHW;
    p.vt_vec = ps->vt_vec; p.vt_bs = ps->vt_bs; p.fd_ohw.mul = ps->fd_ohw.mul; p.fd_ohw.shift = ps->fd_ohw.shift; p.dual = ps->dual;
    p.ln_eps = ps->ln_eps; p.diag = ps->diag;
  }
  PG_STAMP(0)
  // ---- LoRA-A fragments (A operand: lane (c, q) holds A[16 rt + c][32 ks + 8 q ..]) ----
  bf16x8 af[RT > 0 ? RT : 1][RT > 0 ? KS : 1];
  f32x4 sa[RT > 0 ? RT : 1], ca[RT > 0 ? RT : 1];
  if constexpr (RT > 0) {
    const __amdgpu_buffer_rsrc_t rs_a = aldm_igemm_detail::make_rsrc(p.lora_a, (unsigned)p.Rp * (unsigned)(K * 2));
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        af[rt][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (16 * rt + c) * (K * 2) + q * 16 + ks * 64, 0, 0));
      if constexpr (LN) {
        sa[rt] = *reinterpret_cast<const f32x4*>(p.ln_sa + 16 * rt + 4 * q);
        ca[rt] = *reinterpret_cast<const f32x4*>(p.ln_ca + 16 * rt + 4 * q);
      }
    }
  }
  // ---- LayerNorm statistics of this lane's rows from the producer's partial sums: the four lanes (q = 0 .. 3) that share row
  //      mrow[i] split its partial pairs, so every load is in flight at once (a running sum over a run-time count waits for each
  //      load in turn: one L2 round trip per partial on the path to the first MFMA) ----
  float2 pv[MI][4];
  if constexpr (LN) {
    // (buffer loads, lanes / partials past the end read zeros through the descriptor's range check: a predicated plain load
    //  compiles to a branch and a full s_waitcnt around EACH of them -- eight serial round trips in this prologue)
    const __amdgpu_buffer_rsrc_t rs_lp = aldm_igemm_detail::make_rsrc(p.ln_parts, (unsigned)M * (unsigned)(p.ln_np * 8));
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = q + 4 * u;
        const unsigned off = j < p.ln_np ? (unsigned)((mrow[i] * p.ln_np + j) * 8) : 0x80000000u;
        pv[i][u] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs_lp, off, 0, 0));
      }
    }
  }
  // ---- column vectors of the range: c_n (bias; zeros when there is none) and s_n, two columns per thread (BNW <= 512), by buffer
  //      loads (columns past the range / a missing bias read zeros through the descriptor: no branch, no wait here); parked in
  //      LDS further down.  (An LDS-DMA form of this was tried: one cold launch in ~100 read a stale c_n / s_n.) ----
  constexpr int CVU = 512 / NTHR;                             // columns per thread
  float cv0[CVU], cv1[CVU];
  {
    const __amdgpu_buffer_rsrc_t rs_c = aldm_igemm_detail::make_rsrc(p.bias ? (const void*)p.bias : (const void*)wg, p.bias ? (unsigned)N * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rs_s = aldm_igemm_detail::make_rsrc(LN ? (const void*)p.ln_s : (const void*)wg, LN ? (unsigned)N * 4u : 0u);
#pragma unroll
    for (int u = 0; u < CVU; ++u) {
      const int nn = tid + NTHR * u;
      const unsigned off = nn < BNW ? (unsigned)((n_wg0 + nn) * 4) : 0x80000000u;
      cv0[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_c, off, 0, 0));
      cv1[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_s, off, 0, 0));
    }
  }
  // ---- the remaining DMA streams: LoRA-B rows of the range, the residual tiles, weight tile 1 ----
  unsigned r_off[PRW > 0 ? PRW : 1];
  __amdgpu_buffer_rsrc_t rs_r = rs_x;
  if constexpr (RES) {
    rs_r = aldm_igemm_detail::make_rsrc(p.res, (unsigned)M * (unsigned)(p.out_ld * 2));
#pragma unroll
    for (int pp = 0; pp < PRW; ++pp) {
      const int u = pp * 64 + lane;                           // chunk of the wave's [16 MI][RCH] image
      const int rl = u / RCH, pc = u - rl * RCH;
      const int f = MI == 2 ? (((rl >> 1) & 1) | (((rl >> 3) & 3) << 1)) : ((rl >> 1) & 7);
      r_off[pp] = (unsigned)((m0 + mw0 + rl) * (p.out_ld * 2) + ((pc ^ (f & (RCH - 1))) * 16));
    }
  }
  auto issue_r = [&](int t) {
    if constexpr (RES) {
      char* const sb = smem + (t % 3) * STG;
      const int roff = (n_wg0 + t * NT) * 2;
#pragma unroll
      for (int pp = 0; pp < PRW; ++pp)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_r, (lds_ptr_t)(sb + WT + (wave * PRW + pp) * 1024), 16, r_off[pp], roff, 0, 0);
    }
  };
  auto issue = [&](int t) { issue_w(t); issue_r(t); };
  if constexpr (RT > 0) {
    const __amdgpu_buffer_rsrc_t rs_lb = aldm_igemm_detail::make_rsrc(p.lora_b, (unsigned)N * (unsigned)(p.Rp * 2));
    const int npc = (BNW * p.Rp * 2) >> 10;
    for (int pc = wave; pc < npc; pc += NW)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_lb, (lds_ptr_t)(LB + pc * 1024), 16, (unsigned)(n_wg0 * (p.Rp * 2) + pc * 1024 + lane * 16), 0, 0, 0);
  }
  issue_r(0);
  if (T > 1) issue(1);                                        // (the loop's first wait leaves exactly this tile in flight)
  PG_STAMP(1)
  __builtin_amdgcn_sched_barrier(0);

  // ---- consumed INSIDE the tile loop, after tile 0's main MFMAs (their operands -- x fragments, weight tile 0 -- were requested
  //      ~1 us before the loads below could even be issued): the LayerNorm statistics of this lane's rows, and T = x A^T corrected
  //      for the folded LayerNorm and rounded to bf16 = the B operand of the LoRA k-step.
  //      k slots {8 q + jj}: ranks {4 q + jj | 16 + 4 q + (jj - 4)} (the accumulator rows this lane holds) ----
  float mean[MI], rstd[MI];
  bf16x8 tf[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) { mean[i] = 0.f; rstd[i] = 1.f; tf[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
  auto stats_and_t = [&]() {
    if constexpr (LN) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        float s1 = (pv[i][0].x + pv[i][1].x) + (pv[i][2].x + pv[i][3].x);
        float s2 = (pv[i][0].y + pv[i][1].y) + (pv[i][2].y + pv[i][3].y);
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        mean[i] = s1 * (1.f / K);
        rstd[i] = rsqrtf(fmaxf(s2 * (1.f / K) - mean[i] * mean[i], 0.f) + p.ln_eps);
      }
    }
    if constexpr (RT > 0) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const float irs = 1.f / rstd[i];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          f32x4 ta = {0.f, 0.f, 0.f, 0.f}, tb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ks += 2) {
            ta = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rt][ks], xf[i][ks], ta, 0, 0, 0);
            tb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rt][ks + 1], xf[i][ks + 1], tb, 0, 0, 0);
          }
          ta += tb;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = ta[e];
            if constexpr (LN) v = v - mean[i] * sa[rt][e] + ca[rt][e] * irs;
            tf[i][4 * rt + e] = (bf16)v;
          }
          // training: T = x A^T is an operand of the LoRA gradient products (dB = s dY^T T): the first column range of every row
          // block stores its copy, [M][Rp] bf16, ranks 16 rt + 4 q .. + 3 of row mrow[i]
          if (p.lora_t && nr == 0 && mrow[i] < M)
            *reinterpret_cast<bf16x4*>(p.lora_t + (long long)mrow[i] * p.Rp + 16 * rt + 4 * q) =
                bf16x4{tf[i][4 * rt], tf[i][4 * rt + 1], tf[i][4 * rt + 2], tf[i][4 * rt + 3]};
        }
      }
    }
    PG_STAMP(2)
    // park the column vectors (and, for the V^T epilogue, the statistics of every row: it needs those of OTHER lanes' rows) in
    // LDS; one extra barrier per workgroup, behind tile 0's MFMAs, where the loads have long landed
#pragma unroll
    for (int u = 0; u < CVU; ++u) {
      const int nn = tid + NTHR * u;
      if (nn < BNW) { cvec[nn] = cv0[u]; cvec[BNWP + nn] = cv1[u]; }
    }
    if constexpr (LN && EPI == EPI_VT) {
      if (q == 0) {
#pragma unroll
        for (int i = 0; i < MI; ++i) { stat[mrow[i] - m0] = mean[i]; stat[BM + mrow[i] - m0] = rstd[i]; }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // per-lane fragment read addresses: sub-tile j reads weight row R_j of the tile image
  int rb_sw[NJ], vq_sw[NJ], lb_sw[NJ];                        // swapped order: 8 consecutive output columns per lane
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    int R;
    if constexpr (EPI == EPI_GEGLU) {   // j = 0, 1: value columns o = 8 (c >> 2) + 4 j + (c & 3); j = 2, 3: their gates
      const int o = 8 * (c >> 2) + 4 * (j & 1) + (c & 3);
      R = (o >> 4) * 32 + (o & 15) + 16 * (j >> 1);
    } else {
      R = 32 * (j >> 1) + 8 * (c >> 2) + 4 * (j & 1) + (c & 3);
    }
    rb_sw[j] = R * (K * 2);
    vq_sw[j] = (q ^ pg_hsw(R)) * 16;
    lb_sw[j] = R * (p.Rp * 2) + q * 8;
  }
  const int rb_vt = c * (K * 2), vq_vt = (q ^ pg_hsw(c)) * 16;   // natural order (V^T tiles): sub-tile j reads row 16 j + c

  float rs1[MI], rs2[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) rs1[i] = rs2[i] = 0.f;
  f32x4 vmean[MI], vrstd[MI];                                 // V^T orientation: statistics of the 4 MI rows this lane owns
  bool vstat_ready = false;

  for (int t = 0; t < T; ++t) {
    if (t == 2) { PG_STAMP(40) }
    if (t + 1 < T) wait_vmcnt<P>(); else wait_vmcnt<0>();     // tile t landed (this wave's pieces); at most the next tile in flight
    if (t == 2) { PG_STAMP(41) }
    __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0): LDS writes / reads of the previous tile
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");                             // (the barrier intrinsic carries no memory semantics: nothing may move
    __builtin_amdgcn_sched_barrier(0);                        //  an LDS read of this tile above it)
    if (t < 12) { PG_STAMP(3 + 2 * t) }
    if (t + 2 < T) issue(t + 2);
    if (t == 2) { PG_STAMP(43) }
    const char* const Ws = smem + (t % 3) * STG;
    const int n_t = n_wg0 + t * NT;
    const bool vt_tile = EPI == EPI_VT && n_t >= p.vt_col0;

    // (dual: the trainer's q | k | v and out-projection dX launches store every tile BOTH ways -- row-major for the next GEMM,
    //  token-major for the flash kernels -- by running the tile's MFMAs in both operand orders; the matrix pipe has the room)
    const bool do_std = !vt_tile || p.dual;
    f32x4 acc[MI][NJ];
    auto acc_zero = [&]() {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto mfma_vt = [&]() {
      // V^T tile: un-swapped operand order, lane (c, q) owns rows 4 MI q .. 4 MI q + 4 MI - 1 of column 16 j + c
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 wf[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          wf[j] = *reinterpret_cast<const bf16x8*>(Ws + j * (16 * K * 2) + rb_vt + (((ks & 3) * 64) ^ vq_vt) + (ks >> 2) * 256);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i][ks], wf[j], acc[i][j], 0, 0, 0);
      }
    };
    acc_zero();
    if (do_std) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 wf[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          wf[j] = *reinterpret_cast<const bf16x8*>(Ws + rb_sw[j] + (((ks & 3) * 64) ^ vq_sw[j]) + (ks >> 2) * 256);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i][ks], acc[i][j], 0, 0, 0);
      }
    } else if constexpr (EPI == EPI_VT) {
      mfma_vt();
    }
#ifdef PG_VERIFY   // race hunt (tools/dbg_qkv.py): after the first barrier, is every DMA'd operand really in LDS?  counts mismatches
    if (t == 0 && p.diag) {
      for (int j = tid; j < BNW; j += NTHR) {
        if (p.bias && cvec[j] != p.bias[n_wg0 + j]) atomicAdd(&p.diag[48], 1ull);
        if (LN && cvec[BNWP + j] != p.ln_s[n_wg0 + j]) atomicAdd(&p.diag[49], 1ull);
      }
      if constexpr (RT > 0) {
        const unsigned short* lb_l = reinterpret_cast<const unsigned short*>(LB);
        const unsigned short* lb_g = reinterpret_cast<const unsigned short*>(p.lora_b) + (long long)n_wg0 * p.Rp;
        for (int j = tid; j < BNW * p.Rp; j += NTHR)
          if (lb_l[j] != lb_g[j]) atomicAdd(&p.diag[50], 1ull);
      }
      const unsigned short* w_l = reinterpret_cast<const unsigned short*>(Ws);
      for (int u = tid; u < NT * CPR; u += NTHR) {          // physical chunk u = (row R, chunk pc) holds logical chunk pc ^ hsw(R)
        const int R = u / CPR, pc = u - R * CPR;
        const unsigned short* g8 = reinterpret_cast<const unsigned short*>(wg) + (long long)(n_t + R) * K + ((pc ^ pg_hsw(R)) * 8);
        bool bad = false;
        for (int e = 0; e < 8; ++e) bad = bad || (w_l[u * 8 + e] != g8[e]);
        if (bad) atomicAdd(&p.diag[51], 1ull);
      }
    }
#endif
    if (t == 2) { asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[MI - 1][NJ - 1][3])); PG_STAMP(44) }
    if (t == 0) stats_and_t();
    if (do_std) {
      if constexpr (RT > 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const char* brow = LB + t * NT * (p.Rp * 2) + lb_sw[j];
          const bf16x4 lo = *reinterpret_cast<const bf16x4*>(brow);
          const bf16x4 hi = RT > 1 ? *reinterpret_cast<const bf16x4*>(brow + 32) : bf16x4{0, 0, 0, 0};
          const bf16x8 bfr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr, tf[i], acc[i][j], 0, 0, 0);
        }
      }
      // ---- epilogue, straight from the accumulators: lane (c, q) holds columns 8 q .. 8 q + 7 of each 32-column group ----
      if constexpr (EPI == EPI_GEGLU) {
        const int pv = (q >> 1) * 32 + (q & 1) * 8;           // packed index of the first value column; its gate: + 16
        const float* cb = cvec + t * NT + pv;
        const f32x4 cv_a = *reinterpret_cast<const f32x4*>(cb), cv_b = *reinterpret_cast<const f32x4*>(cb + 4);
        const f32x4 cg_a = *reinterpret_cast<const f32x4*>(cb + 16), cg_b = *reinterpret_cast<const f32x4*>(cb + 20);
        f32x4 sv_a, sv_b, sg_a, sg_b;
        if constexpr (LN) {
          const float* sb2 = cb + BNWP;
          sv_a = *reinterpret_cast<const f32x4*>(sb2); sv_b = *reinterpret_cast<const f32x4*>(sb2 + 4);
          sg_a = *reinterpret_cast<const f32x4*>(sb2 + 16); sg_b = *reinterpret_cast<const f32x4*>(sb2 + 20);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          float o8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float val = acc[i][e >> 2][e & 3], gate = acc[i][2 + (e >> 2)][e & 3];
            const float cvv = e < 4 ? cv_a[e & 3] : cv_b[e & 3], cgg = e < 4 ? cg_a[e & 3] : cg_b[e & 3];
            if constexpr (LN) {
              const float svv = e < 4 ? sv_a[e & 3] : sv_b[e & 3], sgg = e < 4 ? sg_a[e & 3] : sg_b[e & 3];
              val = rstd[i] * (val - mean[i] * svv) + cvv;
              gate = rstd[i] * (gate - mean[i] * sgg) + cgg;
            } else {
              val += cvv;
              gate += cgg;
            }
            o8[e] = val * gelu_erf_f(gate);
          }
          const bf16x8 o = pg_cat(pg_pk(o8[0], o8[1]), pg_pk(o8[2], o8[3]), pg_pk(o8[4], o8[5]), pg_pk(o8[6], o8[7]));
          if (mrow[i] < M) *reinterpret_cast<bf16x8*>(p.out + (long long)mrow[i] * p.out_ld + (n_t >> 1) + 8 * q) = o;
        }
      } else {
#pragma unroll
        for (int tp = 0; tp < NJ / 2; ++tp) {
          const int nl = t * NT + 32 * tp + 8 * q;             // range-relative first column
          const f32x4 c_a = *reinterpret_cast<const f32x4*>(cvec + nl), c_b = *reinterpret_cast<const f32x4*>(cvec + nl + 4);
          f32x4 s_a, s_b;
          if constexpr (LN) { s_a = *reinterpret_cast<const f32x4*>(cvec + BNWP + nl); s_b = *reinterpret_cast<const f32x4*>(cvec + BNWP + nl + 4); }
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            bf16x8 r8 = {0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (RES) {
              const int rl = 4 * MI * (c >> 2) + 4 * i + (c & 3);
              const int f = MI == 2 ? (((rl >> 1) & 1) | (((rl >> 3) & 3) << 1)) : ((rl >> 1) & 7);
              r8 = *reinterpret_cast<const bf16x8*>(Ws + WT + (mw0 + rl) * (NT * 2) + (((4 * tp + q) ^ (f & (RCH - 1))) * 16));
            }
            float v8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float v = acc[i][2 * tp + (e >> 2)][e & 3];
              const float cc = e < 4 ? c_a[e & 3] : c_b[e & 3];
              if constexpr (LN) v = rstd[i] * (v - mean[i] * (e < 4 ? s_a[e & 3] : s_b[e & 3])) + cc;
              else v += cc;
              if constexpr (RES) v += (float)r8[e];
              v8[e] = v;
            }
            const bf16x8 o = pg_cat(pg_pk(v8[0], v8[1]), pg_pk(v8[2], v8[3]), pg_pk(v8[4], v8[5]), pg_pk(v8[6], v8[7]));
            if constexpr (RSTAT) {
#pragma unroll
              for (int e = 0; e < 8; ++e) { const float f2 = (float)o[e]; rs1[i] += f2; rs2[i] = fmaf(f2, f2, rs2[i]); }
            }
            if (mrow[i] < M) *reinterpret_cast<bf16x8*>(p.out + (long long)mrow[i] * p.out_ld + n_wg0 + nl) = o;
          }
        }
      }
    }
    if constexpr (EPI == EPI_VT) if (vt_tile) {
      if (do_std) {                                           // dual: the same tile again, un-swapped
        acc_zero();
        mfma_vt();
      }
      if constexpr (RT > 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const char* brow = LB + (t * NT + 16 * j + c) * (p.Rp * 2) + q * 8;
          const bf16x4 lo = *reinterpret_cast<const bf16x4*>(brow);
          const bf16x4 hi = RT > 1 ? *reinterpret_cast<const bf16x4*>(brow + 32) : bf16x4{0, 0, 0, 0};
          const bf16x8 bfr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf[i], bfr, acc[i][j], 0, 0, 0);
        }
      }
      if (LN && !vstat_ready) {                               // (uniform) first V^T tile: the statistics of the rows this lane owns
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          vmean[i] = *reinterpret_cast<const f32x4*>(stat + mw0 + 4 * MI * q + 4 * i);
          vrstd[i] = *reinterpret_cast<const f32x4*>(stat + BM + mw0 + 4 * MI * q + 4 * i);
        }
        vstat_ready = true;
      }
      const int mq = m0 + mw0 + 4 * MI * q;                   // first of this lane's 4 MI rows
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int nl = t * NT + 16 * j + c;
        const float cc = cvec[nl], ss = LN ? cvec[BNWP + nl] : 0.f;
        bf16 o[4 * MI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[i][j][e];
            if constexpr (LN) v = vrstd[i][e] * (v - vmean[i][e] * ss) + cc;
            else v += cc;
            o[4 * i + e] = (bf16)v;
          }
        bf16* const vrow = p.vt + (long long)(n_t + 16 * j + c - p.vt_col0) * p.vt_ld;
        if (p.vt_vec >= 4 * MI) {                             // the lane's rows are one aligned group inside one sample
          if (mq < M) {
            const int b = fdiv(mq, p.fd_ohw), pix = mq - b * p.OHW;
            bf16* dst = vrow + (long long)b * p.vt_bs + pix;
            if constexpr (MI == 2) *reinterpret_cast<bf16x8*>(dst) = bf16x8{o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]};
            else *reinterpret_cast<bf16x4*>(dst) = bf16x4{o[0], o[1], o[2], o[3]};
          }
        } else if (p.vt_vec == 4) {                           // MI == 2: two groups of four
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int mm = mq + 4 * i;
            if (mm < M) {
              const int b = fdiv(mm, p.fd_ohw), pix = mm - b * p.OHW;
              *reinterpret_cast<bf16x4*>(vrow + (long long)b * p.vt_bs + pix) = bf16x4{o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]};
            }
          }
        } else {
#pragma unroll
          for (int u = 0; u < 4 * MI; ++u) {
            const int mm = mq + u;
            if (mm < M) {
              const int b = fdiv(mm, p.fd_ohw), pix = mm - b * p.OHW;
              vrow[(long long)b * p.vt_bs + pix] = o[u];
            }
          }
        }
      }
    }
    if (t < 12) { PG_STAMP(4 + 2 * t) }
  }
  PG_STAMP(30)
  if constexpr (RSTAT) {
    // LayerNorm hand-over: this range's (sum, sum of squares) of every row, of the values AS STORED (bf16)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float a = rs1[i], b = rs2[i];
      a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
      a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
      if (q == 0 && mrow[i] < M) *reinterpret_cast<float2*>(p.rowstat + ((long long)mrow[i] * nranges + nr) * 2) = make_float2(a, b);
    }
  }
#endif
}

template <int K, int NT, int MI, int RT, int EPI, int FL, int NW>
int pg_launch(const PgHost& h, hipStream_t st) {
  constexpr int BM = 16 * MI * NW;
  constexpr int STG = NT * K * 2 + ((FL & PG_RES) ? BM * NT * 2 : 0);
  const int BNW = h.tpr * NT;
  const int lds = (h.tpr < 3 ? h.tpr : 3) * STG + 2 * BM * 4 + 2 * BNW * 4 + BNW * h.a.Rp * 2 + PG_LDS_PAD;
  if (lds > 160 * 1024) {
    aldm_set_error("pgemm: %d B of LDS (K %d, tile %d x %d, %d tiles per range)", lds, K, BM, NT, h.tpr);
    return ALDM_E_UNSUPPORTED;
  }
  auto kern = pgemm_kernel<K, NT, MI, RT, EPI, FL, NW>;
  static unsigned long long attr_done = 0;
  if (int rc = aldm_set_max_lds(reinterpret_cast<const void*>(kern), 160 * 1024, &attr_done, "pgemm")) return rc;
  const int grid = cdiv(h.M, BM) * h.nranges;
  const FastDiv fd = make_fastdiv((unsigned)h.nranges);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, st, h.x, h.w, h.M, h.N, h.nranges, h.tpr, fd.mul, fd.shift, grid, h.a);
  return aldm_launch_status("pgemm");
}

// the instantiations the transformer blocks use; anything else is refused (the caller keeps aldm_igemm)
template <int K, int NT, int MI, int NW>
int pg_dispatch(const PgHost& a, int rt, int epi, int fl, hipStream_t st) {
#define PG_CASE(RT_, EPI_, FL_) if constexpr ((RT_) == 0 || K <= 384 || MI == 1) { if (rt == (RT_) && epi == (EPI_) && fl == (FL_)) return pg_launch<K, NT, MI, RT_, EPI_, FL_, NW>(a, st); }
  PG_CASE(0, EPI_STD, 0)                                     // plain linear (+ bias)
  PG_CASE(0, EPI_STD, PG_RSTAT)                              // proj_in: + statistics for norm1
  PG_CASE(0, EPI_STD, PG_RES)
  PG_CASE(0, EPI_STD, PG_RES | PG_RSTAT)                     // to_out.0 without an adapter
  PG_CASE(1, EPI_STD, 0)                                     // a LoRA-wrapped linear on its own
  PG_CASE(2, EPI_STD, 0)
  PG_CASE(1, EPI_STD, PG_RES)
  PG_CASE(1, EPI_STD, PG_RES | PG_RSTAT)                     // to_out.0 + LoRA + residual + statistics for the next norm
  PG_CASE(2, EPI_STD, PG_RES)
  PG_CASE(2, EPI_STD, PG_RES | PG_RSTAT)                     //   (adapter rank 17 .. 32)
  PG_CASE(0, EPI_VT, PG_LN)                                  // to_q | to_k | to_v, LayerNorm folded
  PG_CASE(1, EPI_VT, PG_LN)                                  //   + LoRA (combined rank <= 16)
  PG_CASE(2, EPI_VT, PG_LN)                                  //   + LoRA (combined rank <= 32)
  PG_CASE(0, EPI_VT, 0)
  PG_CASE(1, EPI_VT, 0)
  PG_CASE(2, EPI_VT, 0)
  if constexpr (NT == 64) {
    PG_CASE(0, EPI_GEGLU, PG_LN)                             // GEGLU projection, LayerNorm folded
    PG_CASE(0, EPI_GEGLU, 0)
  }
#undef PG_CASE
  aldm_set_error("pgemm: no instantiation for LoRA tiles %d, epilogue %d, flags %d (%d waves x 16*%d rows, %d columns)", rt, epi, fl, NW, MI, NT);
  return ALDM_E_UNSUPPORTED;
}

template <int K>
int pg_dispatch_k(const PgHost& a, int nw, int mi, int nt, int rt, int epi, int fl, hipStream_t st) {
  if (nw == 4) {
    if (nt == 64 && mi == 1) return pg_dispatch<K, 64, 1, 4>(a, rt, epi, fl, st);
    if (nt == 32 && mi == 1) return pg_dispatch<K, 32, 1, 4>(a, rt, epi, fl, st);
    if (mi == 2 && (K <= 384 || rt == 0)) {                   // MI = 2: 64 / 96 / 160 registers of x per lane; at K = 640 there is
      if (nt == 64) return pg_dispatch<K, 64, 2, 4>(a, rt, epi, fl, st);   // no room for the LoRA-A fragments beside them
      if (nt == 32) return pg_dispatch<K, 32, 2, 4>(a, rt, epi, fl, st);
    }
  } else if (nw == 8) {                                       // two waves per SIMD: one wave's MFMAs run under the other's epilogue
    if (nt == 64 && mi == 1) return pg_dispatch<K, 64, 1, 8>(a, rt, epi, fl, st);
    if (nt == 32 && mi == 1) return pg_dispatch<K, 32, 1, 8>(a, rt, epi, fl, st);
  }
  aldm_set_error("pgemm: no tile %d waves x 16*%d rows x %d columns for K = %d", nw, mi, nt, K);
  return ALDM_E_UNSUPPORTED;
}

}  // namespace

// debugging hook (tools/diag_pgemm.py), not part of the drop-in boundary: 128 x u64 device buffer for in-kernel time stamps
static unsigned long long* g_pg_diag = nullptr;
extern "C" void aldm_pgemm_set_diag(void* buf) { g_pg_diag = (unsigned long long*)buf; }

extern "C" int aldm_pgemm_supported(int K) { return K == 256 || K == 384 || K == 640; }

// Launch shape.  These GEMMs are bound by what one CU can pull through its vector-memory path (~64 B/clk), not by the MFMA rate:
// a workgroup reads 64 mi rows of x (K 2 B each) once and tiles_per_range * nt (+ LoRA-A) weight rows.  Pick the shape with the
// fewest bytes per CU over ceil(workgroups / 256) rounds; a fixed cost per workgroup stands for its prologue / epilogue latency.
extern "C" int aldm_pgemm_plan(aldm_pgemm_t* g) {
  ALDM_CHECK_ARG(g && aldm_pgemm_supported(g->K) && g->M > 0 && g->N > 0 && g->N % 64 == 0, "pgemm_plan: bad shape");
  if (g->mi && g->nt && g->tiles_per_range && g->waves) return ALDM_OK;
  const int rt = g->Rp ? (g->ranks_used <= 16 ? 1 : 2) : 0;
  double best = 1e30;
  int bmi = 0, bnt = 0, btpr = 0, bnw = 0;
  for (int nw = 4; nw <= 8; nw += 4) {
    if (g->waves && g->waves != nw) continue;
    for (int mi = 1; mi <= (((g->K <= 384 || rt == 0) && nw == 4) ? 2 : 1); ++mi) {
      if (g->mi && g->mi != mi) continue;
      for (int nt = 32; nt <= 64; nt += 32) {
        if ((g->nt && g->nt != nt) || (g->geglu && nt != 64) || (g->vt && g->vt_col0 % nt)) continue;
        const int ntiles = g->N / nt;
        for (int tpr = 1; tpr <= ntiles && tpr * nt <= 512; ++tpr) {
          if (ntiles % tpr || (g->tiles_per_range && g->tiles_per_range != tpr)) continue;
          if (g->max_ranges > 0 && ntiles / tpr > g->max_ranges) continue;
          const int bm = 16 * mi * nw;
          const long long stg = (long long)nt * g->K * 2 + (g->res ? bm * nt * 2 : 0);
          const long long lds = (tpr < 3 ? tpr : 3) * stg + 2 * bm * 4 + 2 * tpr * nt * 4 + (long long)tpr * nt * g->Rp * 2;
          if (lds > 160 * 1024) continue;
          const long long wgs = (long long)cdiv(g->M, bm) * (ntiles / tpr);
          const double bytes = (double)bm * g->K * 2 + (double)(tpr * nt + 16 * rt) * g->K * 2 + (g->res ? (double)bm * tpr * nt * 2 : 0.0) +
                               (double)bm * tpr * nt * (g->geglu ? 1 : 2);
          const double rounds = (double)((wgs + 255) / 256);
          const double cost = rounds * (bytes + 48.0 * 1024) * (nw == 8 ? 1.05 : 1.0);   // (8 waves won only on the GEGLU launches: table)
          if (cost < best) { best = cost; bmi = mi; bnt = nt; btpr = tpr; bnw = nw; }
        }
      }
    }
  }
  if (!bmi) {
    aldm_set_error("pgemm_plan: no launch shape for M %d N %d K %d (waves %d mi %d nt %d tiles_per_range %d)", g->M, g->N, g->K, g->waves, g->mi, g->nt, g->tiles_per_range);
    return ALDM_E_UNSUPPORTED;
  }
  g->mi = bmi; g->nt = bnt; g->tiles_per_range = btpr; g->waves = bnw;
  return ALDM_OK;
}

extern "C" int aldm_pgemm(const aldm_pgemm_t* g0, void* stream) {
  ALDM_CHECK_ARG(g0 && g0->x && g0->w && g0->out, "pgemm: null x / w / out");
  aldm_pgemm_t gg = *g0;
  if (!(gg.mi && gg.nt && gg.tiles_per_range && gg.waves)) {
    if (int rc = aldm_pgemm_plan(&gg)) return rc;
  }
  const aldm_pgemm_t* g = &gg;
  ALDM_CHECK_ARG(aldm_pgemm_supported(g->K), "pgemm: K must be 256, 384 or 640 (got %d)", g->K);
  ALDM_CHECK_ARG(g->M > 0 && g->N > 0 && g->N % 64 == 0, "pgemm: N must be a multiple of 64 (got %d)", g->N);
  ALDM_CHECK_ARG((unsigned long long)g->M * g->K * 2 < 0x80000000ull && (unsigned long long)g->N * g->K * 2 < 0x80000000ull,
                 "pgemm: operands beyond 32-bit buffer offsets");
  const int mi = g->mi, nt = g->nt;
  const int nw = g->waves;
  ALDM_CHECK_ARG((nw == 4 || nw == 8) && (mi == 1 || mi == 2) && (nt == 32 || nt == 64), "pgemm: waves must be 4 / 8, mi 1 / 2, nt 32 / 64 (got %d, %d, %d)", nw, mi, nt);
  ALDM_CHECK_ARG(g->tiles_per_range > 0 && (g->N / nt) % g->tiles_per_range == 0 && g->tiles_per_range * nt <= 512,
                 "pgemm: tiles_per_range %d must divide N / nt = %d and span at most 512 columns", g->tiles_per_range, g->N / nt);
  int rt = 0;
  if (g->Rp) {
    ALDM_CHECK_ARG(g->lora_a && g->lora_b && (g->Rp == 32 || g->Rp == 64) && g->ranks_used > 0 && g->ranks_used <= 32 && g->ranks_used <= g->Rp,
                   "pgemm: LoRA needs lora_a / lora_b, Rp 32 / 64 and 1 <= ranks_used <= 32 (got Rp %d, %d ranks)", g->Rp, g->ranks_used);
    rt = g->ranks_used <= 16 ? 1 : 2;
    ALDM_CHECK_ARG(!g->ln_s || (g->ln_sa && g->ln_ca), "pgemm: folded LayerNorm with LoRA needs ln_sa / ln_ca");
  }
  ALDM_CHECK_ARG(!g->lora_t_out || g->Rp, "pgemm: lora_t_out without an adapter");
  ALDM_CHECK_ARG(!g->ln_s || (g->ln_parts && g->ln_nparts > 0 && g->ln_nparts <= 16), "pgemm: the folded LayerNorm takes its statistics from ln_parts (1 .. 16 pairs per row)");
  ALDM_CHECK_ARG(!(g->geglu && (g->vt || g->res || g->rowstat_out || g->Rp)), "pgemm: GEGLU launches take no V^T / residual / statistics / LoRA");
  ALDM_CHECK_ARG(!g->geglu || nt == 64, "pgemm: GEGLU needs nt = 64");
  ALDM_CHECK_ARG(!(g->vt && (g->res || g->rowstat_out)), "pgemm: V^T launches take no residual / statistics");
  ALDM_CHECK_ARG(!g->vt || ((g->vt_col0 > 0 || g->vt_dual) && g->vt_col0 >= 0 && g->vt_col0 % nt == 0 && g->OHW > 0 && g->M % g->OHW == 0 && g->vt_ld >= g->OHW),
                 "pgemm: vt_col0 %d must be a positive multiple of nt %d (0 with vt_dual), M a multiple of OHW", g->vt_col0, nt);
  ALDM_CHECK_ARG(!g->vt_dual || g->vt, "pgemm: vt_dual without vt");
  ALDM_CHECK_ARG(g->out_ld % 8 == 0 && ((uintptr_t)g->out & 15) == 0, "pgemm: out rows must be 16-byte aligned");
  ALDM_CHECK_ARG(!g->res || ((unsigned long long)g->M * g->out_ld * 2 < 0x80000000ull && ((uintptr_t)g->res & 15) == 0), "pgemm: residual too large / misaligned");

  PgHost h;
  h.x = (const bf16*)g->x; h.w = (const bf16*)g->w; h.M = g->M; h.N = g->N;
  h.tpr = g->tiles_per_range; h.nranges = g->N / nt / g->tiles_per_range;
  PgArgs& a = h.a;
  a.lora_a = (const bf16*)g->lora_a; a.lora_b = (const bf16*)g->lora_b;
  a.bias = g->bias; a.ln_s = g->ln_s; a.ln_sa = g->ln_sa; a.ln_ca = g->ln_ca; a.ln_parts = g->ln_parts;
  a.res = (const bf16*)g->res; a.out = (bf16*)g->out; a.vt = (bf16*)g->vt; a.rowstat = g->rowstat_out; a.lora_t = (bf16*)g->lora_t_out;
  a.Rp = g->Rp; a.ln_np = g->ln_nparts; a.out_ld = g->out_ld;
  a.vt_col0 = g->vt ? g->vt_col0 : 0x7fffffff; a.vt_ld = g->vt_ld; a.OHW = g->OHW > 0 ? g->OHW : g->M; a.vt_bs = g->vt_batch_stride;
  a.vt_vec = 1;
  a.dual = g->vt_dual ? 1 : 0;
  if (g->vt) {
    const bool al8 = a.OHW % 8 == 0 && g->vt_ld % 8 == 0 && g->vt_batch_stride % 8 == 0 && ((uintptr_t)g->vt & 15) == 0;
    const bool al4 = a.OHW % 4 == 0 && g->vt_ld % 4 == 0 && g->vt_batch_stride % 4 == 0 && ((uintptr_t)g->vt & 7) == 0;
    a.vt_vec = al8 ? 8 : (al4 ? 4 : 1);
  }
  a.fd_ohw = make_fastdiv((unsigned)a.OHW);
  a.ln_eps = g->ln_eps;
  a.diag = g_pg_diag;
  const int epi = g->geglu ? EPI_GEGLU : (g->vt ? EPI_VT : EPI_STD);
  const int fl = (g->ln_s ? PG_LN : 0) | (g->res ? PG_RES : 0) | (g->rowstat_out ? PG_RSTAT : 0);
  hipStream_t st = (hipStream_t)stream;
  switch (g->K) {
    case 256: return pg_dispatch_k<256>(h, nw, mi, nt, rt, epi, fl, st);
    case 384: return pg_dispatch_k<384>(h, nw, mi, nt, rt, epi, fl, st);
    default: return pg_dispatch_k<640>(h, nw, mi, nt, rt, epi, fl, st);
  }
}

// Flash-style attention BACKWARD for gfx950 (LoRA fine-tune step: dX through the frozen SDPA of
// UNet2DConditionModel [REF script/train/train_audioldm_lora.py:539-557]).  P is recomputed from Q, K and the
// forward's log2-domain LSE; the N x N matrices never touch HBM.  Two kernels, both built like the forward:
//   attn_bwd_dq  : a wave owns 32 QUERY columns.  S^T = K Q^T and dP^T = V dO^T (keys on rows), so
//                  lse / delta are lane-local; dS^T is converted in registers and re-used as the B operand of
//                  dQ^T += K^T dS^T (K^T token-major from LDS) -- same accumulator-as-operand trick as the forward.
//                  Also emits delta = rowsum(dO * O).
//   attn_bwd_dkv : a wave owns 32 KEY columns.  S = Q K^T and dP = dO V^T (queries on rows); P and dS feed
//                  dV^T += dO^T P and dK^T += Q^T dS with Q^T / dO^T token-major from LDS.  No atomics anywhere:
//                  every output element has exactly one owner.
// Token-major copies (Q^T, K^T, dO^T as [B][C][Npad]) come from aldm_transpose_tokens.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int TS = 64;   // tokens per staged tile
constexpr int ATTN_BWD_NW8_MIN_N = 768;   // 8-wave workgroups from this many tokens on (N = 1024: 93.5 -> 90.7 us per backward; N = 256: 33.6 -> 38.7, stays at 4)
constexpr int VS = 136;  // token-major LDS row stride (64 tokens * 2 B + 8 B pad)

template <int DP>
struct BwdCfg {
  static constexpr int DK = DP / 16;
  static constexpr int DT = (DP + 31) / 32;
  static constexpr int KS = DP * 2 + (((DP / 8) % 2 == 0) ? 16 : 0);   // row-major LDS row stride
  static constexpr int RBYTES = TS * KS;          // one row-major tile
  static constexpr int TBYTES = DT * 32 * VS;     // one token-major tile
};

// cooperative tile loaders, split in two so that a tile's global loads fly under the PREVIOUS tile's MFMAs: fetch_* issues
// every load of the next tile into registers before the compute phase, store_* writes them to LDS after it (behind the barrier
// that ends the compute phase).  Without the split every key / query tile exposed a full L2 / HBM round trip, in all waves at
// once: 16 such round trips per workgroup at N = 1024.
template <int DP, int T>
struct RowRegs { static constexpr int CH = TS * (DP / 8), NR = (CH + T - 1) / T; bf16x8 v[NR]; };
template <int DP, int T>
struct TokRegs { static constexpr int CH = BwdCfg<DP>::DT * 32 * (TS / 8), NR = (CH + T - 1) / T; bf16x8 v[NR]; };

// Buffer loads with every predicate folded into the descriptor's bound or the offset (out of range -> zeros): a predicated plain
// load compiles to a branch with an `s_waitcnt vmcnt(0)` at the join, which serialises the tile's loads with each other.
template <int DP, int T>
__device__ __forceinline__ void fetch_rows(RowRegs<DP, T>& rg, const bf16* base, int ld, int t0, int N, int D, int tid) {
  using R = RowRegs<DP, T>;
  // rows t0 .. N-1, D columns each: the last valid byte is the end of row N-1's D columns
  const int bytes = t0 < N ? ((N - t0 - 1) * ld + D) * 2 : 0;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(base + (long long)t0 * ld), 0, bytes, 0x00020000);
#pragma unroll
  for (int i = 0; i < R::NR; ++i) {
    const int c = tid + i * T;
    const int row = c / (DP / 8), ch = c - row * (DP / 8);
    const unsigned off = (c < R::CH && ch * 8 < D) ? (unsigned)row * (unsigned)(ld * 2) + ch * 16 : 0x80000000u;
    rg.v[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
  }
}
template <int DP, int T>
__device__ __forceinline__ void store_rows(char* dst, const RowRegs<DP, T>& rg, int tid) {
  using R = RowRegs<DP, T>;
  constexpr int KS = BwdCfg<DP>::KS;
#pragma unroll
  for (int i = 0; i < R::NR; ++i) {
    const int c = tid + i * T;
    const int row = c / (DP / 8), ch = c - row * (DP / 8);
    if (c < R::CH) *reinterpret_cast<bf16x8*>(dst + row * KS + ch * 16) = rg.v[i];
  }
}
template <int DP, int T>
__device__ __forceinline__ void fetch_tokmajor(TokRegs<DP, T>& rg, const bf16* base, int ldt, int t0, int N, int D, int tid) {
  using R = TokRegs<DP, T>;
  // columns t0 .. of the D token-major rows; bound = end of the last row (a chunk inside the row padding or running into the next row
  // reads defined memory: store_tokmajor clears every token >= N)
  const int bytes = t0 < N ? (D * ldt - t0) * 2 : 0;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(base + t0), 0, bytes, 0x00020000);
#pragma unroll
  for (int i = 0; i < R::NR; ++i) {
    const int c = tid + i * T;
    const int row = c >> 3, ch = c & 7;
    const unsigned off = (c < R::CH && row < D) ? (unsigned)row * (unsigned)(ldt * 2) + ch * 16 : 0x80000000u;
    rg.v[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
  }
}
template <int DP, int T>
__device__ __forceinline__ void store_tokmajor(char* dst, TokRegs<DP, T>& rg, int t0, int N, int tid) {
  using R = TokRegs<DP, T>;
#pragma unroll
  for (int i = 0; i < R::NR; ++i) {
    const int c = tid + i * T;
    const int row = c >> 3, ch = c & 7;
    if (c >= R::CH) continue;
    bf16x8 v = rg.v[i];
    const int tb = t0 + ch * 8;
    if (tb + 8 > N) {                                   // tail tile: tokens past N must read as zeros
#pragma unroll
      for (int j = 0; j < 8; ++j) if (tb + j >= N) v[j] = (bf16)0.f;
    }
    uint2* d2 = reinterpret_cast<uint2*>(dst + row * VS + ch * 16);
    const uint4 u = __builtin_bit_cast(uint4, v);
    d2[0] = make_uint2(u.x, u.y);
    d2[1] = make_uint2(u.z, u.w);
  }
}

__device__ __forceinline__ bf16x8 read_tok_frag(const char* tile, int row, int s2, int hh) {
  const char* p = tile + row * VS + (16 * s2 + 4 * hh) * 2;
  const uint2 lo = *reinterpret_cast<const uint2*>(p);
  const uint2 hi = *reinterpret_cast<const uint2*>(p + 16);
  return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
}

// ---- dQ ------------------------------------------------------------------------------------------
// XCD-aware (block, head, batch) order, as in attention.hip: every XCD gets whole (batch, head) pairs so the pair's K / V /
// Q / dO tiles are fetched into ONE L2 instead of all eight.  Bijective when (heads * batch) % 8 == 0, identity otherwise.
__device__ __forceinline__ void xcd_pair_order(int& blk, int& head, int& b) {
  const int nq = gridDim.x, H = gridDim.y, npairs = H * gridDim.z;
  blk = blockIdx.x; head = blockIdx.y; b = blockIdx.z;
  if ((npairs & 7) == 0) {
    const int L = blockIdx.x + nq * (blockIdx.y + H * blockIdx.z);
    const int xcd = L & 7, slot = L >> 3;
    const int pl = slot / nq;
    blk = slot - pl * nq;
    const int pair = pl * 8 + xcd;
    b = pair / H;
    head = pair - b * H;
  }
}

template <int DP, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dq_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                              const bf16* __restrict__ v, int ld,
                                                              const bf16* __restrict__ kT, int ldt, long long t_bs,
                                                              const bf16* __restrict__ dO, const bf16* __restrict__ O, int ldo,
                                                              const float* __restrict__ lse, float* __restrict__ delta,
                                                              int N, int D, float scale, bf16* __restrict__ dq, int lddq) {
  using Cfg = BwdCfg<DP>;
  constexpr int T = 64 * NW, DK = Cfg::DK, DT = Cfg::DT, KS = Cfg::KS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;                       // K rows   [64][KS]
  char* Vs = Ks + Cfg::RBYTES;           // V rows   [64][KS]
  char* KTs = Vs + Cfg::RBYTES;          // K^T      [DT*32][VS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  int blk, head, b;
  xcd_pair_order(blk, head, b);
  const int H = gridDim.y;
  const int q0 = (blk * NW + wave) * 32;
  const float c = scale * 1.44269504088896340736f;
  const long long rowbase = (long long)b * N;
  const bf16* qb = q + rowbase * ld + head * D;
  const bf16* kb = k + rowbase * ld + head * D;
  const bf16* vb = v + rowbase * ld + head * D;
  const bf16* dob = dO + rowbase * ldo + head * D;
  const bf16* ob = O + rowbase * ldo + head * D;
  const bf16* ktb = kT + (long long)b * t_bs + (long long)head * D * ldt;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  bf16x8 qf[DK], dof[DK];
  float dl = 0.f;
#pragma unroll
  for (int ks = 0; ks < DK; ++ks) {
    const int col = 16 * ks + 8 * hh;
    const bool ok = q0 + r < N && col < D;
    qf[ks] = ok ? *reinterpret_cast<const bf16x8*>(qb + (long long)(q0 + r) * ld + col) : zero8;
    dof[ks] = ok ? *reinterpret_cast<const bf16x8*>(dob + (long long)(q0 + r) * ldo + col) : zero8;
    const bf16x8 of = ok ? *reinterpret_cast<const bf16x8*>(ob + (long long)(q0 + r) * ldo + col) : zero8;
#pragma unroll
    for (int j = 0; j < 8; ++j) dl += (float)dof[ks][j] * (float)of[j];
  }
  dl += __shfl_xor(dl, 32, 64);            // delta_q = sum_d dO[q][d] * O[q][d]
  float L = 0.f;
  if (q0 + r < N) {
    L = lse[((long long)b * H + head) * N + q0 + r];
    if (hh == 0) delta[((long long)b * H + head) * N + q0 + r] = dl;
  }

  f32x16 acc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const int ntiles = (N + TS - 1) / TS;
  RowRegs<DP, T> rk, rv;
  TokRegs<DP, T> rkt;
  fetch_rows<DP, T>(rk, kb, ld, 0, N, D, tid);
  fetch_rows<DP, T>(rv, vb, ld, 0, N, D, tid);
  fetch_tokmajor<DP, T>(rkt, ktb, ldt, 0, N, D, tid);
  __builtin_amdgcn_s_waitcnt(0x0F70);                   // (as in attn_bwd_dkv_kernel: Q / dO fragments, lse, delta and tile 0 are waited for here)
  for (int it = 0; it < ntiles; ++it) {
    const int kv0 = it * TS;
    __syncthreads();                                    // every wave is done with the previous tile's LDS images
    store_rows<DP, T>(Ks, rk, tid);
    store_rows<DP, T>(Vs, rv, tid);
    store_tokmajor<DP, T>(KTs, rkt, kv0, N, tid);
    __syncthreads();
    // next tile: loads in flight under this tile's MFMAs (past the last tile the descriptors' bounds are 0: zeros, no traffic)
    fetch_rows<DP, T>(rk, kb, ld, kv0 + TS, N, D, tid);
    fetch_rows<DP, T>(rv, vb, ld, kv0 + TS, N, D, tid);
    fetch_tokmajor<DP, T>(rkt, ktb, ldt, kv0 + TS, N, D, tid);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < DK; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (sub * 32 + r) * KS + (16 * ks + 8 * hh) * 2);
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vs + (sub * 32 + r) * KS + (16 * ks + 8 * hh) * 2);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dp, 0, 0, 0);
      }
      bf16x8 dsf[2];
      if (kv0 + TS > N) {                               // (wave-uniform) only the tile that holds key N - 1 masks: keys >= N are zero
#pragma unroll                                          //  rows of K, whose p = exp2(-L) is not zero
        for (int i = 0; i < 16; ++i) {
          const int kvr = kv0 + sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          const float p = kvr < N ? __builtin_amdgcn_exp2f(fmaf(s[i], c, -L)) : 0.f;
          dsf[i >> 3][i & 7] = (bf16)(p * (dp[i] - dl));
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[i], c, -L));
          dsf[i >> 3][i & 7] = (bf16)(p * (dp[i] - dl));
        }
      }
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_tok_frag(KTs, t * 32 + r, sub * 2 + s2, hh), dsf[s2], acc[t], 0, 0, 0);
    }
  }
  if (q0 + r < N) {
    bf16* orow = dq + (rowbase + q0 + r) * lddq + head * D;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = t * 32 + 8 * g + 4 * hh;
        if (d0 < D) {
          bf16x4 o = {(bf16)(acc[t][4 * g] * scale), (bf16)(acc[t][4 * g + 1] * scale), (bf16)(acc[t][4 * g + 2] * scale), (bf16)(acc[t][4 * g + 3] * scale)};
          *reinterpret_cast<bf16x4*>(orow + d0) = o;
        }
      }
  }
}

// ---- dK, dV --------------------------------------------------------------------------------------
template <int DP, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dkv_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                               const bf16* __restrict__ v, int ld,
                                                               const bf16* __restrict__ qT, const bf16* __restrict__ dOT,
                                                               int ldt, long long qt_bs, long long dot_bs,
                                                               const bf16* __restrict__ dO, int ldo,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               int N, int D, float scale, bf16* __restrict__ dk,
                                                               bf16* __restrict__ dv, int lddk) {
  using Cfg = BwdCfg<DP>;
  constexpr int T = 64 * NW, DK = Cfg::DK, DT = Cfg::DT, KS = Cfg::KS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;                          // Q rows    [64][KS]
  char* dOs = Qs + Cfg::RBYTES;             // dO rows   [64][KS]
  char* QTs = dOs + Cfg::RBYTES;            // Q^T       [DT*32][VS]
  char* dOTs = QTs + Cfg::TBYTES;           // dO^T      [DT*32][VS]
  float* Ls = reinterpret_cast<float*>(dOTs + Cfg::TBYTES);   // [64] lse, [64] delta
  float* Ds = Ls + TS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  int blk, head, b;
  xcd_pair_order(blk, head, b);
  const int H = gridDim.y;
  const int kv0 = (blk * NW + wave) * 32;
  const float c = scale * 1.44269504088896340736f;
  const long long rowbase = (long long)b * N;
  const bf16* qb = q + rowbase * ld + head * D;
  const bf16* kb = k + rowbase * ld + head * D;
  const bf16* vb = v + rowbase * ld + head * D;
  const bf16* dob = dO + rowbase * ldo + head * D;
  const bf16* qtb = qT + (long long)b * qt_bs + (long long)head * D * ldt;
  const bf16* dotb = dOT + (long long)b * dot_bs + (long long)head * D * ldt;
  const float* lb = lse + ((long long)b * H + head) * N;
  const float* db = delta + ((long long)b * H + head) * N;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  bf16x8 kf[DK], vf[DK];   // B operands: lane (key column r, half hh) holds K/V[kv0 + r][16 ks + 8 hh ..]
#pragma unroll
  for (int ks = 0; ks < DK; ++ks) {
    const int col = 16 * ks + 8 * hh;
    const bool ok = kv0 + r < N && col < D;
    kf[ks] = ok ? *reinterpret_cast<const bf16x8*>(kb + (long long)(kv0 + r) * ld + col) : zero8;
    vf[ks] = ok ? *reinterpret_cast<const bf16x8*>(vb + (long long)(kv0 + r) * ld + col) : zero8;
  }
  f32x16 accv[DT], acck[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) { accv[t][i] = 0.f; acck[t][i] = 0.f; }

  const int ntiles = (N + TS - 1) / TS;
  RowRegs<DP, T> rq, rdo;
  TokRegs<DP, T> rqt, rdot;
  float rl = 0.f, rd = 0.f;                             // lse / delta of query tid of the tile (threads < TS)
  const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lb), 0, N * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(db), 0, N * 4, 0x00020000);
  auto fetch_tile = [&](int qt0) {
    fetch_rows<DP, T>(rq, qb, ld, qt0, N, D, tid);
    fetch_rows<DP, T>(rdo, dob, ldo, qt0, N, D, tid);
    fetch_tokmajor<DP, T>(rqt, qtb, ldt, qt0, N, D, tid);
    fetch_tokmajor<DP, T>(rdot, dotb, ldt, qt0, N, D, tid);
    {
      const unsigned off = tid < TS ? (unsigned)(qt0 + tid) * 4u : 0x80000000u;           // (beyond N: out of range -> 0)
      const float l = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_l, off, 0, 0));
      rd = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_d, off, 0, 0));
      rl = qt0 + tid < N ? l : 3.0e38f;                 // padding queries: exp2(s c - 3e38) = 0, so P and dS need no mask
    }
  };
  fetch_tile(0);
  // Drain the loads issued so far (the K / V fragments above, tile 0) HERE: otherwise the compiler places the wait for the K / V
  // fragments in front of the loop's first MFMA as `s_waitcnt vmcnt(0)`, where from the second iteration on it waits for the NEXT
  // tile's loads issued a few instructions earlier -- the prefetch then hides nothing (the ISA showed exactly that).
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (int it = 0; it < ntiles; ++it) {
    const int qt0 = it * TS;
    __syncthreads();                                    // every wave is done with the previous tile's LDS images
    store_rows<DP, T>(Qs, rq, tid);
    store_rows<DP, T>(dOs, rdo, tid);
    store_tokmajor<DP, T>(QTs, rqt, qt0, N, tid);
    store_tokmajor<DP, T>(dOTs, rdot, qt0, N, tid);
    if (tid < TS) { Ls[tid] = rl; Ds[tid] = rd; }
    __syncthreads();
    fetch_tile(qt0 + TS);                               // next tile: loads in flight under this tile's MFMAs (past the last tile the
                                                        // descriptors' bounds are 0: zeros, no traffic -- and a fixed load count per iteration)
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < DK; ++ks) {
        const bf16x8 qfr = *reinterpret_cast<const bf16x8*>(Qs + (sub * 32 + r) * KS + (16 * ks + 8 * hh) * 2);
        const bf16x8 dofr = *reinterpret_cast<const bf16x8*>(dOs + (sub * 32 + r) * KS + (16 * ks + 8 * hh) * 2);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[ks], s, 0, 0, 0);      // S[q rows][key cols]
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dofr, vf[ks], dp, 0, 0, 0);   // dP[q rows][key cols]
      }
      bf16x8 pf[2], dsf[2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // accumulator registers 4j .. 4j+3 are query rows sub*32 + 8j + 4hh + (0..3) of the staged tile: one 16-byte LDS read each
        // for lse and delta (every lane of a half-wave reads the same address: a broadcast)
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + sub * 32 + 8 * j + 4 * hh);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(Ds + sub * 32 + 8 * j + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * j + e;
          const float p = __builtin_amdgcn_exp2f(fmaf(s[i], c, -l4[e]));
          pf[i >> 3][i & 7] = (bf16)p;
          dsf[i >> 3][i & 7] = (bf16)(p * (dp[i] - d4[e]));
        }
      }
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          accv[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_tok_frag(dOTs, t * 32 + r, sub * 2 + s2, hh), pf[s2], accv[t], 0, 0, 0);
          acck[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_tok_frag(QTs, t * 32 + r, sub * 2 + s2, hh), dsf[s2], acck[t], 0, 0, 0);
        }
    }
  }
  if (kv0 + r < N) {
    bf16* krow = dk + (rowbase + kv0 + r) * lddk + head * D;
    bf16* vrow = dv + (rowbase + kv0 + r) * lddk + head * D;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = t * 32 + 8 * g + 4 * hh;
        if (d0 < D) {
          bf16x4 ok_ = {(bf16)(acck[t][4 * g] * scale), (bf16)(acck[t][4 * g + 1] * scale), (bf16)(acck[t][4 * g + 2] * scale), (bf16)(acck[t][4 * g + 3] * scale)};
          bf16x4 ov_ = {(bf16)accv[t][4 * g], (bf16)accv[t][4 * g + 1], (bf16)accv[t][4 * g + 2], (bf16)accv[t][4 * g + 3]};
          *reinterpret_cast<bf16x4*>(krow + d0) = ok_;
          *reinterpret_cast<bf16x4*>(vrow + d0) = ov_;
        }
      }
  }
}

template <int DP, int NW>
int launch_bwd(const void* q, const void* k, const void* v, int ld, const void* qT, const void* kT, const void* dOT, int ldt,
               long long qt_bs, long long kt_bs, long long dot_bs, const void* dO, const void* O, int ldo, const float* lse,
               float* delta, int B, int N, int H, int D, float scale, void* dq, void* dk, void* dv, int ldg, hipStream_t st) {
  using Cfg = BwdCfg<DP>;
  constexpr int lds_a = 2 * Cfg::RBYTES + Cfg::TBYTES;
  constexpr int lds_b = 2 * Cfg::RBYTES + 2 * Cfg::TBYTES + 2 * TS * 4;
  auto ka = attn_bwd_dq_kernel<DP, NW>;
  auto kb = attn_bwd_dkv_kernel<DP, NW>;
  static unsigned long long done_a = 0, done_b = 0;   // per-device bit masks (aldm_set_max_lds)
  if (int rc = aldm_set_max_lds(reinterpret_cast<const void*>(ka), lds_a, &done_a, "attention_bwd dq")) return rc;
  if (int rc = aldm_set_max_lds(reinterpret_cast<const void*>(kb), lds_b, &done_b, "attention_bwd dkv")) return rc;
  dim3 grid(cdiv(N, 32 * NW), H, B);
  hipLaunchKernelGGL(ka, grid, dim3(64 * NW), lds_a, st, (const bf16*)q, (const bf16*)k, (const bf16*)v, ld, (const bf16*)kT,
                     ldt, kt_bs, (const bf16*)dO, (const bf16*)O, ldo, lse, delta, N, D, scale, (bf16*)dq, ldg);
  int rc = aldm_launch_status("attn_bwd_dq");
  if (rc) return rc;
  hipLaunchKernelGGL(kb, grid, dim3(64 * NW), lds_b, st, (const bf16*)q, (const bf16*)k, (const bf16*)v, ld, (const bf16*)qT,
                     (const bf16*)dOT, ldt, qt_bs, dot_bs, (const bf16*)dO, ldo, lse, (const float*)delta, N, D, scale,
                     (bf16*)dk, (bf16*)dv, ldg);
  return aldm_launch_status("attn_bwd_dkv");
}

template <int DP>
int launch_bwd_d(const void* q, const void* k, const void* v, int ld, const void* qT, const void* kT, const void* dOT, int ldt,
                 long long qt_bs, long long kt_bs, long long dot_bs, const void* dO, const void* O, int ldo, const float* lse,
                 float* delta, int B, int N, int H, int D, float scale, void* dq, void* dk, void* dv, int ldg, hipStream_t st) {
  static const int nw_override = []() { const char* e = getenv("ALDM_ATTN_BWD_NW"); return e ? atoi(e) : 0; }();   // tuning aid (tools/bench_attn_bwd.py)
  if (nw_override == 8 || (nw_override == 0 && N >= ATTN_BWD_NW8_MIN_N)) return launch_bwd<DP, 8>(q, k, v, ld, qT, kT, dOT, ldt, qt_bs, kt_bs, dot_bs, dO, O, ldo, lse, delta, B, N, H, D, scale, dq, dk, dv, ldg, st);
  if (nw_override == 2) return launch_bwd<DP, 2>(q, k, v, ld, qT, kT, dOT, ldt, qt_bs, kt_bs, dot_bs, dO, O, ldo, lse, delta, B, N, H, D, scale, dq, dk, dv, ldg, st);
  if (N >= 192) return launch_bwd<DP, 4>(q, k, v, ld, qT, kT, dOT, ldt, qt_bs, kt_bs, dot_bs, dO, O, ldo, lse, delta, B, N, H, D, scale, dq, dk, dv, ldg, st);
  if (N >= 64) return launch_bwd<DP, 2>(q, k, v, ld, qT, kT, dOT, ldt, qt_bs, kt_bs, dot_bs, dO, O, ldo, lse, delta, B, N, H, D, scale, dq, dk, dv, ldg, st);
  return launch_bwd<DP, 1>(q, k, v, ld, qT, kT, dOT, ldt, qt_bs, kt_bs, dot_bs, dO, O, ldo, lse, delta, B, N, H, D, scale, dq, dk, dv, ldg, st);
}

}  // namespace

extern "C" int aldm_attention_bwd(const void* q, const void* k, const void* v, int ld, const void* qT, const void* kT,
                                  const void* dOT, int ldt, long long qT_batch_stride, long long kT_batch_stride,
                                  long long dOT_batch_stride, const void* dO, const void* O, int ldo, const float* lse,
                                  float* delta, int B, int N, int H, int d, float scale, void* dq, void* dk, void* dv,
                                  int ldg, void* stream) {
  ALDM_CHECK_ARG(q && k && v && qT && kT && dOT && dO && O && lse && delta && dq && dk && dv, "attention_bwd: null pointer");
  ALDM_CHECK_ARG(B > 0 && N > 0 && H > 0 && d > 0 && d % 8 == 0 && ld % 8 == 0 && ldo % 8 == 0 && ldt % 8 == 0 && ldg % 4 == 0,
                 "attention_bwd: bad dims");
  ALDM_CHECK_ARG(ldt >= ((N + 7) / 8) * 8, "attention_bwd: ldt %d too small for N %d", ldt, N);
  hipStream_t st = (hipStream_t)stream;
#define ALDM_BWD(DPV) return launch_bwd_d<DPV>(q, k, v, ld, qT, kT, dOT, ldt, qT_batch_stride, kT_batch_stride, dOT_batch_stride, dO, O, ldo, lse, delta, B, N, H, d, scale, dq, dk, dv, ldg, st)
  if (d <= 16) ALDM_BWD(16);
  if (d <= 32) ALDM_BWD(32);
  if (d <= 48) ALDM_BWD(48);
  if (d <= 64) ALDM_BWD(64);
  if (d <= 80) ALDM_BWD(80);
#undef ALDM_BWD
  aldm_set_error("attention_bwd: head dim %d > 80 unsupported", d);
  return ALDM_E_UNSUPPORTED;
}

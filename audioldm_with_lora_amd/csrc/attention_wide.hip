// Flash-style single-head attention for WIDE heads (d = 128 / 256 / 512) on gfx950: the AutoencoderKL mid-block attention
// (1 head, d = 512, N = 2000 .. 4096 tokens, [REF script/train/train_audioldm_lora.py:370,495-496] encode /
// [REF script/inference/generate_audio.py:47-52] decode through AudioLDMPipeline.__call__).  SURVEY.md 2.3 row K7.
//
// Why a second kernel: with d = 512 the 32x32x16 layout of attention.hip needs 256 accumulator registers for O^T alone.
// Here a wave owns 16 queries and the WHOLE head dim on v_mfma_f32_16x16x32_bf16:
//   S^T[16 keys x 16 q] = K_tile[16 x 32d] . Q^T          16 k-steps per 512-wide head, Q fragments resident (64 VGPRs)
//   O^T[16 d x 16 q]   += V^T[16 d x 32 keys] . P^T       32 (+1) d-tiles, O^T resident (132 VGPRs)
// so nothing is exchanged between waves and the fp32 score matrix (64 MB per 10 s clip) never exists.  The lane that holds
// S^T rows (keys) {4g..4g+3} of two stacked 16-key tiles is exactly the lane that must supply 8 k-values of P^T to the next
// MFMA; the k order this implies ({4g..4g+3, 16+4g..16+4g+3}) is reproduced on the V^T side by two 8-byte LDS reads.
//   * K / V^T tiles of 32 keys go global -> LDS by LDS-DMA (`buffer_load ... lds`, no staging registers), double-buffered,
//     one `s_waitcnt vmcnt(0)` + barrier per tile; both images are XOR-swizzled on the SOURCE chunk (the DMA writes lane-linear)
//     so that the ds_read_b128 (K) / ds_read_b64 (V^T) fragment reads are bank-conflict free.
//   * softmax as in attention.hip: Q arrives pre-scaled by d^-0.5 log2(e) (folded into to_q by the host), the running maximum
//     is the MFMA's initial accumulator and is only raised when a tile exceeds it by RESCALE_THR, the row sum rides in an
//     extra d-tile whose first V^T row is ones.
#include <cstdlib>
#include "igemm_core.h"   // make_rsrc / lds_ptr_t / wait_vmcnt

namespace {

using aldm_igemm_detail::lds_ptr_t;
using aldm_igemm_detail::wait_vmcnt;

constexpr float WIDE_RESCALE_THR = 5.0f;
constexpr int KT = 32;                     // keys per tile

template <int DP>
struct WideCfg {
  static constexpr int KROW = DP * 2;                       // bytes per key row of the K image (64 .. 256 x 16-B chunks)
  static constexpr int KBYTES = KT * KROW;
  static constexpr int VROWS = DP + 16;                     // + the tile whose first row is ones
  static constexpr int VBYTES = VROWS * 64;                 // 32 keys x 2 B per d-row
  static constexpr int LDS = 2 * (KBYTES + VBYTES);
  static constexpr int DKS = DP / 32;                       // k-steps of S^T
  static constexpr int DTL = DP / 16 + 1;                   // d-tiles of O^T incl. the ones tile
};

// QT query tiles of 16 per wave (round 3: QT = 2).  With one tile every wave reads the whole K tile (32 KB) and V^T tile (33 KB) from
// LDS for 65 MFMAs: four waves need 2080 LDS cycles per key tile against 1040 MFMA cycles each -- LDS-read-bound by 2x.  Two query
// tiles share every K / V^T fragment (the A operands): the same 65 KB now feed 130 MFMAs and the two pipes are balanced.  O^T and the
// Q fragments double (264 + 128 VGPRs at d = 512): one wave per SIMD, which the 130 KB of LDS imposed anyway.
template <int DP, int NW, int QT>
__global__ __launch_bounds__(64 * NW) void attention_wide_kernel(const bf16* __restrict__ q, int ldq, const bf16* __restrict__ k, int ldk,
                                                                 const bf16* __restrict__ vt, int vt_ld, long long vt_bs, int N,
                                                                 bf16* __restrict__ out, int out_ld) {
#if defined(__HIP_DEVICE_COMPILE__)
  using Cfg = WideCfg<DP>;
  constexpr int DKS = Cfg::DKS, DTL = Cfg::DTL, CPR = DP / 8;        // CPR: 16-B chunks per K row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ks = smem;                                     // [2][KT][KROW]
  char* const Vs = smem + 2 * Cfg::KBYTES;                   // [2][VROWS][64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int b = blockIdx.y;
  const int q0 = (blockIdx.x * NW + wave) * 16 * QT;
  const bf16* qb = q + (long long)b * N * ldq;
  const bf16* kb = k + (long long)b * N * ldk;
  const bf16* vb = vt + (long long)b * vt_bs;

  // K rows >= N read as zeros through the descriptor's range check (their scores are masked below)
  const __amdgpu_buffer_rsrc_t rs_k = aldm_igemm_detail::make_rsrc(kb, (unsigned)(((long long)(N - 1) * ldk + DP) * 2));
  const __amdgpu_buffer_rsrc_t rs_v = aldm_igemm_detail::make_rsrc(vb, (unsigned)((long long)DP * vt_ld * 2));

  // the ones tile of both V^T buffers: row DP = 1.0, rows DP+1 .. DP+15 = 0 (written once; the DMA never touches them)
  for (int i = tid; i < 2 * 16 * 8; i += 64 * NW) {
    const int buf = i >> 7, r8 = i & 127;                   // 16 rows x 8 x 8-byte slots
    const unsigned fill = (r8 < 8) ? 0x3F803F80u : 0u;
    reinterpret_cast<uint2*>(Vs + buf * Cfg::VBYTES + DP * 64)[r8] = make_uint2(fill, fill);
  }

  auto issue = [&](int tile, int buf) {
    const int kv0 = tile * KT;
    // K: one 16-B-chunk row of CPR chunks per key; a wave instruction moves 64 chunks
    constexpr int KINST = KT * CPR / 64;
#pragma unroll
    for (int i = 0; i < (KINST + NW - 1) / NW; ++i) {
      const int inst = wave + i * NW;                        // (wave-uniform)
      if (inst < KINST) {
        const int c = inst * 64 + lane, row = c / CPR, ph = c - row * CPR;
        const int lg = ph ^ (row & 15);                      // source chunk that lands in physical slot ph
        const unsigned off = (unsigned)(kv0 + row) * (unsigned)(ldk * 2) + lg * 16;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (lds_ptr_t)(Ks + buf * Cfg::KBYTES + inst * 1024), 16, off, 0, 0, 0);
      }
    }
    // V^T: 64 B (32 keys) per d-row, 16 rows per wave instruction
    constexpr int VINST = DP / 16;
#pragma unroll
    for (int i = 0; i < (VINST + NW - 1) / NW; ++i) {
      const int inst = wave + i * NW;
      if (inst < VINST) {
        const int row = inst * 16 + (lane >> 2), ph = lane & 3;
        const int lg = ph ^ ((row >> 2) & 3);
        const unsigned off = (unsigned)row * (unsigned)(vt_ld * 2) + (unsigned)(kv0 + lg * 8) * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lds_ptr_t)(Vs + buf * Cfg::VBYTES + inst * 1024), 16, off, 0, 0, 0);
      }
    }
  };

  // Q fragments (B operand of S^T): lane (n, g) holds Q[q0 + n][32 ks + 8 g .. + 7]
  bf16x8 qf[QT][DKS];
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < QT; ++u)
#pragma unroll
    for (int ks = 0; ks < DKS; ++ks)
      qf[u][ks] = (q0 + 16 * u + n < N) ? *reinterpret_cast<const bf16x8*>(qb + (long long)(q0 + 16 * u + n) * ldq + 32 * ks + 8 * g) : zero8;

  f32x4 o[QT][DTL];
#pragma unroll
  for (int u = 0; u < QT; ++u)
#pragma unroll
    for (int t = 0; t < DTL; ++t) o[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[QT];
#pragma unroll
  for (int u = 0; u < QT; ++u) m_run[u] = 0.f;

  const int ntiles = (N + KT - 1) / KT;
  issue(0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0): Q fragments + tile 0 (the compiler's pass models this form)
  wait_vmcnt<0>();
  __syncthreads();

  for (int tile = 0; tile < ntiles; ++tile) {
    const int buf = tile & 1;
    if (tile + 1 < ntiles) issue(tile + 1, buf ^ 1);
    const char* Kb = Ks + buf * Cfg::KBYTES;
    const char* Vb = Vs + buf * Cfg::VBYTES;
    const bool first = tile == 0;

    // ---- S'^T = K Q^T - m_run : two stacked 16-key tiles, every K fragment shared by the QT query tiles ----
    f32x4 s[QT][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int u = 0; u < QT; ++u) { const float a0 = -m_run[u]; s[u][sub] = f32x4{a0, a0, a0, a0}; }
      const int row = sub * 16 + n;                          // A operand: lane (m = n, g) holds K[key row][32 ks + 8 g ..]
#pragma unroll
      for (int ks = 0; ks < DKS; ++ks) {
        const int ph = (4 * ks + g) ^ (row & 15);
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kb + row * Cfg::KROW + ph * 16);
#pragma unroll
        for (int u = 0; u < QT; ++u) s[u][sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[u][ks], s[u][sub], 0, 0, 0);
      }
    }
    const int kv0 = tile * KT;
    if (kv0 + KT > N) {                                      // tail: keys >= N
#pragma unroll
      for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (kv0 + sub * 16 + 4 * g + j >= N) s[u][sub][j] = -INFINITY;
    }
    // ---- softmax: this lane holds 8 of the 32 keys of query n; the other 24 sit in lanes n + 16 g' ----
    bf16x8 pf[QT];                                           // B operand of O^T += V^T P^T: k slots = keys {4g+j} then {16+4g+j}
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      float mx = fmaxf(fmaxf(fmaxf(s[u][0][0], s[u][0][1]), fmaxf(s[u][0][2], s[u][0][3])),
                       fmaxf(fmaxf(s[u][1][0], s[u][1][1]), fmaxf(s[u][1][2], s[u][1][3])));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (first || !__all(mx <= WIDE_RESCALE_THR)) {
        const float delta = first ? mx : fmaxf(mx, 0.f);
        m_run[u] += delta;
        if (!first) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
          for (int t = 0; t < DTL; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) o[u][t][j] *= alpha;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int j = 0; j < 4; ++j) s[u][sub][j] -= delta;
      }
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int j = 0; j < 4; ++j) pf[u][sub * 4 + j] = (bf16)__builtin_amdgcn_exp2f(s[u][sub][j]);
    }

    // ---- O^T += V^T P^T : A operand lane (m = n, g) holds V^T[16 t + n][keys 4g..4g+3 | 16+4g..16+4g+3], shared by the query tiles ----
#pragma unroll
    for (int t = 0; t < DTL; ++t) {
      const int row = t * 16 + n;
      const int sw = (row >> 2) & 3;
      const char* vrow = Vb + row * 64 + (g & 1) * 8;
      const uint2 lo = *reinterpret_cast<const uint2*>(vrow + (((g >> 1)) ^ sw) * 16);
      const uint2 hi = *reinterpret_cast<const uint2*>(vrow + ((2 + (g >> 1)) ^ sw) * 16);
      const bf16x8 vf = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
#pragma unroll
      for (int u = 0; u < QT; ++u) o[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[u], o[u][t], 0, 0, 0);
    }
    wait_vmcnt<0>();                                         // the next tile has landed (it flew under ~65 MFMAs)
    __syncthreads();
  }

  // ---- normalise and store: lane holds O^T[16 t + 4 g + j][q0 + 16 u + n]; l = row DP of the ones tile (lanes g = 0, j = 0) ----
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    float l = o[u][DTL - 1][0];                              // zero on the g != 0 lanes (rows DP + 4g: zero padding)
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (q0 + 16 * u + n < N) {
      bf16* orow = out + ((long long)b * N + q0 + 16 * u + n) * out_ld;
#pragma unroll
      for (int t = 0; t < DTL - 1; ++t) {
        const bf16x4 v = {(bf16)(o[u][t][0] * inv), (bf16)(o[u][t][1] * inv), (bf16)(o[u][t][2] * inv), (bf16)(o[u][t][3] * inv)};
        *reinterpret_cast<bf16x4*>(orow + 16 * t + 4 * g) = v;
      }
    }
  }
#endif
}

template <int DP, int NW, int QT>
int launch_wide(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld, long long vt_bs, int B, int N, void* out,
                int out_ld, hipStream_t st) {
  using Cfg = WideCfg<DP>;
  auto kern = attention_wide_kernel<DP, NW, QT>;
  static unsigned long long attr_done = 0;
  if (int rc = aldm_set_max_lds(reinterpret_cast<const void*>(kern), Cfg::LDS, &attr_done, "attention_wide")) return rc;
  dim3 grid(cdiv(N, 16 * NW * QT), B);
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), Cfg::LDS, st, (const bf16*)q, ldq, (const bf16*)k, ldk, (const bf16*)vt, vt_ld, vt_bs, N,
                     (bf16*)out, out_ld);
  return aldm_launch_status("attention_wide");
}

}  // namespace

extern "C" int aldm_attention_wide(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld, long long vt_batch_stride,
                                   int B, int N, int d, void* out, int out_ld, void* stream) {
  ALDM_CHECK_ARG(q && k && vt && out, "attention_wide: null pointer");
  ALDM_CHECK_ARG(B > 0 && N > 0, "attention_wide: bad dims");
  ALDM_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && out_ld % 4 == 0, "attention_wide: ldq / ldk must be multiples of 8, out_ld of 4");
  ALDM_CHECK_ARG(vt_ld % 32 == 0 && vt_ld >= (N + 31) / 32 * 32, "attention_wide: vt_ld %d must be a multiple of 32 covering N %d (zero padded)", vt_ld, N);
  ALDM_CHECK_ARG((long long)N * ldk * 2 < 0x7FFFFFFFll && (long long)d * vt_ld * 2 < 0x7FFFFFFFll, "attention_wide: operands exceed 32-bit offsets");
  hipStream_t st = (hipStream_t)stream;
  // 4 waves = 64 (or 128) queries per workgroup: N = 4000 x 4 clips -> 252 workgroups, one per CU (130 KiB of LDS each at d = 512)
  static const int qt_env = getenv("ALDM_WIDE_QT") ? atoi(getenv("ALDM_WIDE_QT")) : 0;   // tuning aid: query tiles per wave (1 / 2)
  // two query tiles per wave once that still fills the chip: encode (8 x 4096 tokens: 256 workgroups) 0.79 -> 0.49 ms; decode
  // (4 x 4000: 128 workgroups on 256 CUs) would go 0.32 -> 0.38 and keeps one
  const bool two = qt_env ? qt_env == 2 : (long long)cdiv(N, 128) * B >= 192;
  if (d == 512) return two ? launch_wide<512, 4, 2>(q, ldq, k, ldk, vt, vt_ld, vt_batch_stride, B, N, out, out_ld, st)
                           : launch_wide<512, 4, 1>(q, ldq, k, ldk, vt, vt_ld, vt_batch_stride, B, N, out, out_ld, st);
  if (d == 256) return two ? launch_wide<256, 4, 2>(q, ldq, k, ldk, vt, vt_ld, vt_batch_stride, B, N, out, out_ld, st)
                           : launch_wide<256, 4, 1>(q, ldq, k, ldk, vt, vt_ld, vt_batch_stride, B, N, out, out_ld, st);
  if (d == 128) return two ? launch_wide<128, 4, 2>(q, ldq, k, ldk, vt, vt_ld, vt_batch_stride, B, N, out, out_ld, st)
                           : launch_wide<128, 4, 1>(q, ldq, k, ldk, vt, vt_ld, vt_batch_stride, B, N, out, out_ld, st);
  aldm_set_error("attention_wide: head dim %d (128 / 256 / 512 are built)", d);
  return ALDM_E_UNSUPPORTED;
}

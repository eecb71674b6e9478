// HiFi-GAN residual-block pair as ONE launch, for the vocoder's low-channel stages (C = 32 / 64 channels, 320 k - 640 k samples):
//     r' = r + conv2(lrelu(conv1(lrelu(r))))        conv1: K taps, dilation d, "same" padding;  conv2: K taps, dilation 1
// = one (conv1, conv2) pair of transformers' HifiGanResidualBlock.forward (modeling_speecht5.py:2887-2950) under SpeechT5HifiGan.forward
// inside AudioLDMPipeline.__call__ [REF script/inference/generate_audio.py:47-52]; the class is loaded at
// [REF script/train/train_audioldm_lora.py:371].
//
// Why its own kernel: on the generic implicit-GEMM kernel a pair is two launches that each stream the whole [B][T][C] tensor
// (41 MB at C = 32) in and out twice -- r, lrelu(r), t = lrelu(conv1), r' and its activated copy: ~330 MB of HBM traffic for 82 MB of
// algorithmic bytes (r in, r' out), at 57 - 360 TFLOP/s on a 64 x 64 tile that is half idle at N = 32.  Here a workgroup owns 256
// output samples: the input run (+ halo of (K - 1)(d + 1) / 2 samples each side) is activated ONCE on its way into LDS, conv1's output
// never leaves LDS, both convolutions' weights sit in REGISTERS as MFMA fragments for the whole launch (a wave owns 16 output
// channels), and the residual add, the MRF mean (alpha, res2) and the next stage's leaky-relu happen in the store epilogue.
//
//   * 8 wave64; wave w owns output-channel tile ct = w % (C / 16) and the position tiles pg, pg + PG, ... (PG = 8 / (C / 16)).
//   * v_mfma_f32_16x16x32_bf16 "swapped": A = weights [16 couts][32 (tap, cin)], B = activations [32][16 positions] read from LDS
//     at row (position + tap * dilation): one ds_read_b128 per MFMA, no im2col anywhere.
//   * D[cout 4 g + j][position n]: a lane owns 4 consecutive channels of one position: 8-byte LDS / global stores.
#include "common.h"

namespace {

struct PairArgs {
  const bf16* x;            // [B][T][C] residual stream r (raw, not activated)
  const bf16* w1; const float* b1; int ld1, dil;   // conv1 [C][ld1], K index = tap * C + cin
  const bf16* w2; const float* b2; int ld2;        // conv2
  const bf16* res2;         // optional [B][T][C]: running MRF sum
  bf16* out;                // [B][T][C]
  int B, T, tiles_per_item;
  float slope, alpha, post_slope;
  int post_act;
};

__device__ __forceinline__ bf16x8 lrelu8(bf16x8 v, float slope) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float f = (float)v[i];
    o[i] = (bf16)fmaxf(f, f * slope);          // slope < 1: max(x, slope x) = leaky_relu(x)
  }
  return o;
}

template <int C, int K>
__global__ __launch_bounds__(512) void hifigan_respair_kernel(const PairArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int CT = C / 16, PG = 8 / CT, KS = C / 32, TT = 256;
  constexpr int P2 = (K - 1) / 2;
  constexpr int TM = TT + K - 1, NMT = (TM + 15) / 16;        // conv1 outputs conv2 needs, in 16-position tiles
  constexpr int XSTR = C * 2 + 16;                            // LDS row stride (bytes) of both activation images
  constexpr int XROWS = NMT * 16 + (K - 1) * 5;               // rows the conv1 fragment reads may touch at the largest dilation (5)
  constexpr int CPR = C / 8;                                  // 16-byte chunks per position
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const xs = smem;                                      // [XROWS][XSTR]  lrelu(r) of the run, zero outside the sequence
  char* const mid = smem + XROWS * XSTR;                      // [NMT * 16][XSTR] lrelu(conv1 + b1), zero outside the sequence

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int ct = wave % CT, pg = wave / CT;
  const int d = p.dil, T = p.T;
  const int PH = P2 * (1 + d);                                // halo each side: conv2's (K-1)/2 plus conv1's d (K-1)/2
  const int TX = TT + 2 * PH;

  // ---- both convolutions' weights as A fragments, for the whole launch: lane (n, g) holds W[16 ct + n][tap * C + 32 ks + 8 g .. + 7] ----
  bf16x8 w1f[K][KS], w2f[K][KS];
#pragma unroll
  for (int tap = 0; tap < K; ++tap)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      w1f[tap][ks] = *reinterpret_cast<const bf16x8*>(p.w1 + (long long)(16 * ct + n) * p.ld1 + tap * C + 32 * ks + 8 * g);
      w2f[tap][ks] = *reinterpret_cast<const bf16x8*>(p.w2 + (long long)(16 * ct + n) * p.ld2 + tap * C + 32 * ks + 8 * g);
    }
  const f32x4 bias1 = *reinterpret_cast<const f32x4*>(p.b1 + 16 * ct + 4 * g);
  const f32x4 bias2 = *reinterpret_cast<const f32x4*>(p.b2 + 16 * ct + 4 * g);

  // A workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...: the weight fragments above are fetched once per workgroup, not once
  // per 256 samples (at C = 64, K = 11 that was 180 KB per tile through the CU's vector-memory path -- more than the tile's own 64 KB).
  for (int tile = blockIdx.x; tile < p.B * p.tiles_per_item; tile += gridDim.x) {
  const int b = tile / p.tiles_per_item, t0 = (tile - b * p.tiles_per_item) * TT;
  const bf16* xb = p.x + (long long)b * T * C;
  // ---- the input run: global -> registers -> leaky-relu -> LDS (rows outside the sequence: zeros = the convolution's padding) ----
  for (int c = tid; c < TX * CPR; c += 512) {
    const int row = c / CPR, ch = c - row * CPR;
    const int t = t0 - PH + row;
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t >= 0 && t < T) v = lrelu8(*reinterpret_cast<const bf16x8*>(xb + (long long)t * C + ch * 8), p.slope);
    *reinterpret_cast<bf16x8*>(xs + row * XSTR + ch * 16) = v;
  }
  __syncthreads();

  // ---- conv1 over the TM positions conv2 needs: mid[m] <-> t = t0 - P2 + m reads xs rows m + tap d ----
  for (int mt = pg; mt < NMT; mt += PG) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < K; ++tap)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xs + (16 * mt + n + tap * d) * XSTR + (32 * ks + 8 * g) * 2);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[tap][ks], xf, acc, 0, 0, 0);
      }
    const int t = t0 - P2 + 16 * mt + n;
    const bool in = t >= 0 && t < T;                           // conv2 zero-pads ITS input: nothing of conv1 exists outside the sequence
    bf16x4 y;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = acc[j] + bias1[j];
      y[j] = in ? (bf16)fmaxf(v, v * p.slope) : (bf16)0.f;
    }
    *reinterpret_cast<bf16x4*>(mid + (16 * mt + n) * XSTR + (16 * ct + 4 * g) * 2) = y;
  }
  __syncthreads();

  // ---- conv2 + bias + residual (+ MRF mean / next stage's activation): out position o reads mid rows o + tap ----
  for (int ot = pg; ot < TT / 16; ot += PG) {
    const int t = t0 + 16 * ot + n;
    const bool in = t < T;
    const long long off = ((long long)b * T + t) * C + 16 * ct + 4 * g;
    bf16x4 r = {0, 0, 0, 0}, r2 = {0, 0, 0, 0};
    if (in) {
      r = *reinterpret_cast<const bf16x4*>(p.x + off);        // (the run was just read: an L2 hit)
      if (p.res2) r2 = *reinterpret_cast<const bf16x4*>(p.res2 + off);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < K; ++tap)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 mf = *reinterpret_cast<const bf16x8*>(mid + (16 * ot + n + tap) * XSTR + (32 * ks + 8 * g) * 2);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[tap][ks], mf, acc, 0, 0, 0);
      }
    bf16x4 y;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = p.alpha * (acc[j] + bias2[j] + (float)r[j]) + (float)r2[j];
      if (p.post_act) v = fmaxf(v, v * p.post_slope);
      y[j] = (bf16)v;
    }
    if (in) *reinterpret_cast<bf16x4*>(p.out + off) = y;
  }
  __syncthreads();                                             // the next tile's run overwrites xs / mid
  }
#endif
}

template <int C, int K>
int launch_pair(const PairArgs& a, hipStream_t st) {
  constexpr int NMT = (256 + K - 1 + 15) / 16, XSTR = C * 2 + 16;
  constexpr int LDS = (NMT * 16 + (K - 1) * 5) * XSTR + NMT * 16 * XSTR;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  auto kern = hifigan_respair_kernel<C, K>;
  static unsigned long long attr_done = 0;
  if (int rc = aldm_set_max_lds(reinterpret_cast<const void*>(kern), LDS, &attr_done, "hifigan_respair")) return rc;
  // one 8-wave workgroup per CU at C = 64 (210 VGPRs), two at C = 32 (122 VGPRs, 47 KB of LDS): the grid is one wave of resident workgroups
  const int tiles = a.B * a.tiles_per_item, resident = 256 * (C == 32 ? 2 : 1);
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < resident ? tiles : resident)), dim3(512), LDS, st, a);
  return aldm_launch_status("hifigan_respair");
}

// ---- conv_post: Conv1d(C -> 1, K taps, "same" padding) + tanh, fp32 out -- SpeechT5HifiGan.forward's last two lines (modeling_speecht5.py:3059-3061).
// One output channel: as a GEMM it fills 1 of 64 tile columns (0.10 ms on the generic kernel: 24 TFLOP/s, 400 GB/s).  Here it is the
// HBM-bound stencil it is: a workgroup stages 256 + K - 1 positions of the (already activated) input in LDS, each thread owns one output
// and runs its K * C multiply-adds on the VALU with the weights broadcast from LDS; the input is read once, the output written once.
struct PostArgs {
  const bf16* x; const bf16* w; const float* bias; float* out;
  int B, T, C, K, tiles_per_item, act;
};

__global__ __launch_bounds__(256) void conv1d_to1_kernel(const PostArgs p) {
  constexpr int TT = 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int C = p.C, K = p.K, P = (K - 1) / 2;
  const int XSTR = C * 2 + 16;                                // padded rows: consecutive lanes' 16-byte reads spread over the banks
  char* const xs = smem;                                      // [TT + K - 1][XSTR]
  bf16* const ws = reinterpret_cast<bf16*>(smem + (TT + K - 1) * XSTR);   // [K][C]
  const int tid = threadIdx.x;
  const int b = blockIdx.x / p.tiles_per_item, t0 = (blockIdx.x - b * p.tiles_per_item) * TT;
  const int cpr = C / 8;
  const bf16* xb = p.x + (long long)b * p.T * C;
  for (int c = tid; c < (TT + K - 1) * cpr; c += 256) {
    const int row = c / cpr, ch = c - row * cpr, t = t0 - P + row;
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t >= 0 && t < p.T) v = *reinterpret_cast<const bf16x8*>(xb + (long long)t * C + ch * 8);
    *reinterpret_cast<bf16x8*>(xs + row * XSTR + ch * 16) = v;
  }
  for (int c = tid; c < K * cpr; c += 256) *reinterpret_cast<bf16x8*>(ws + c * 8) = *reinterpret_cast<const bf16x8*>(p.w + c * 8);
  __syncthreads();
  const int t = t0 + tid;
  float a0 = 0.f, a1 = 0.f;
  for (int tap = 0; tap < K; ++tap) {
    const char* xr = xs + (tid + tap) * XSTR;
    const bf16* wr = ws + tap * C;
    for (int ch = 0; ch < cpr; ++ch) {
      const bf16x8 xv = *reinterpret_cast<const bf16x8*>(xr + ch * 16);
      const bf16x8 wv = *reinterpret_cast<const bf16x8*>(wr + ch * 8);     // same address in every lane: an LDS broadcast
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        a0 = fmaf((float)xv[i], (float)wv[i], a0);
        a1 = fmaf((float)xv[i + 1], (float)wv[i + 1], a1);
      }
    }
  }
  if (t < p.T) {
    float v = a0 + a1 + (p.bias ? p.bias[0] : 0.f);
    if (p.act == ALDM_ACT_TANH) v = tanhf(v);
    p.out[(long long)b * p.T + t] = v;
  }
}

}  // namespace

extern "C" int aldm_hifigan_respair_supported(int C, int K, int dil) {
  return (C == 32 || C == 64) && (K == 3 || K == 7 || K == 11) && dil >= 1 && dil <= 5;
}

extern "C" int aldm_hifigan_respair(const void* x, int B, int T, int C, const void* w1, int ld1, const float* b1, int dil,
                                    const void* w2, int ld2, const float* b2, int K, float slope, float alpha, const void* res2,
                                    int post_act, float post_slope, void* out, void* stream) {
  ALDM_CHECK_ARG(x && w1 && b1 && w2 && b2 && out, "hifigan_respair: null pointer");
  ALDM_CHECK_ARG(B > 0 && T > 0, "hifigan_respair: bad dims");
  ALDM_CHECK_ARG(aldm_hifigan_respair_supported(C, K, dil), "hifigan_respair: built for C in {32, 64}, K in {3, 7, 11}, dilation 1..5 (got C %d K %d d %d)", C, K, dil);
  ALDM_CHECK_ARG(ld1 >= K * C && ld2 >= K * C && ld1 % 8 == 0 && ld2 % 8 == 0, "hifigan_respair: weight rows must hold K * C columns, 16-byte aligned");
  ALDM_CHECK_ARG(slope > 0.f && slope < 1.f && (post_act == ALDM_ACT_NONE || (post_act == ALDM_ACT_LRELU && post_slope > 0.f && post_slope < 1.f)),
                 "hifigan_respair: leaky-relu slopes in (0, 1); post_act NONE or LRELU");
  ALDM_CHECK_ARG((long long)B * T * C < (1ll << 31), "hifigan_respair: tensor too large for 32-bit element offsets");
  PairArgs a{(const bf16*)x, (const bf16*)w1, b1, ld1, dil, (const bf16*)w2, b2, ld2, (const bf16*)res2, (bf16*)out, B, T, cdiv(T, 256),
             slope, alpha, post_slope, post_act == ALDM_ACT_LRELU ? 1 : 0};
  hipStream_t st = (hipStream_t)stream;
#define ALDM_PAIR(CV, KV) if (C == CV && K == KV) return launch_pair<CV, KV>(a, st)
  ALDM_PAIR(32, 3); ALDM_PAIR(32, 7); ALDM_PAIR(32, 11);
  ALDM_PAIR(64, 3); ALDM_PAIR(64, 7); ALDM_PAIR(64, 11);
#undef ALDM_PAIR
  return ALDM_E_UNSUPPORTED;
}

extern "C" int aldm_conv1d_to1(const void* x, int B, int T, int C, const void* w, const float* bias, int K, int act, float* out,
                               void* stream) {
  ALDM_CHECK_ARG(x && w && out && B > 0 && T > 0, "conv1d_to1: null pointer / bad dims");
  ALDM_CHECK_ARG(C % 8 == 0 && C >= 8 && C <= 128 && K >= 1 && K <= 15 && (K & 1), "conv1d_to1: C %% 8 == 0, C <= 128, odd K <= 15 (got C %d K %d)", C, K);
  ALDM_CHECK_ARG(act == ALDM_ACT_NONE || act == ALDM_ACT_TANH, "conv1d_to1: act NONE or TANH");
  PostArgs a{(const bf16*)x, (const bf16*)w, bias, out, B, T, C, K, cdiv(T, 256), act};
  const int lds = (256 + K - 1) * (C * 2 + 16) + K * C * 2;
  static unsigned long long attr_done = 0;
  if (lds > 48 * 1024)
    if (int rc = aldm_set_max_lds(reinterpret_cast<const void*>(conv1d_to1_kernel), lds, &attr_done, "conv1d_to1")) return rc;
  hipLaunchKernelGGL(conv1d_to1_kernel, dim3((unsigned)(B * a.tiles_per_item)), dim3(256), lds, (hipStream_t)stream, a);
  return aldm_launch_status("conv1d_to1");
}

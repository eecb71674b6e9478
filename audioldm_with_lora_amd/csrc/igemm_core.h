// Implicit-GEMM convolution / linear on gfx950 MFMA (bf16 x bf16 -> fp32), with fused LoRA
// side channel, GEGLU, bias / time-embedding / residual epilogues, transposed V^T store and split-K.
//
// Serves every F.conv2d / F.linear / conv1d / conv_transpose1d phase and peft lora.Linear under
// UNet2DConditionModel.forward [REF script/train/train_audioldm_lora.py:539-546], AutoencoderKL.decode
// and SpeechT5HifiGan.forward [REF script/inference/generate_audio.py:47-52].
//
// Structure (MI355X-first, not a CUDA tiling):
//   * workgroup = 4 wave64s in a WM x WN grid; each wave owns a (BM/WM) x (BN/WN) output block built from
//     v_mfma_f32_16x16x32_bf16 tiles.  The MFMA is issued "swapped" (weights as the A operand) so each
//     lane ends up with 4 CONSECUTIVE output channels of one pixel -> 8-byte epilogue stores.
//   * K-tile = 64 channels of one filter tap.  The activation tile is gathered (NHWC, zero padding,
//     optional nearest-upsample index map, optional second source = virtual channel concat) straight
//     into registers as 16-byte chunks and written to an XOR-swizzled LDS image so that every
//     ds_read_b128 fragment read is bank-conflict free; double-buffered, one barrier per K-tile.
//   * LoRA: the rank-r projection T = X A^T rides along the main K loop as extra MFMA columns
//     (the LoRA-A rows are appended to the weight tile); T is then converted to bf16 in LDS and
//     consumed as one more K-step against (alpha/r) B -- no extra launch, no extra pass over X.
#pragma once
#include <type_traits>

#include "common.h"

namespace aldm_igemm_detail {

constexpr int BK = 64;
constexpr int THREADS = 256;

// n / d for a runtime-constant d: q = (umulhi(n, mul) + n) >> shift  (round-up magic; exact for 0 <= n < 2^31).
// A hardware integer division is ~50 instructions; the kernels divide by OH*OW and OW in their prologue AND epilogue.
struct FastDiv { unsigned mul, shift; };
__device__ __forceinline__ int fdiv(int n, FastDiv d) {
  return (int)(((unsigned long long)__umulhi((unsigned)n, d.mul) + (unsigned)n) >> d.shift);
}
static inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  f.shift = s;
  f.mul = (unsigned)((((1ull << s) - d) << 32) / d + 1);
  return f;
}

struct IgemmDev {
  const bf16* x; const bf16* x2; const bf16* w; const bf16* lora_a; const bf16* lora_b;
  bf16* lora_t_out;
  const float* bias; const float* rowbias; const bf16* res; const bf16* res2;
  void* out; bf16* vt; float* ws;
  int B, IH, IW, Cin, Cin2, Ctot, UH, UW, dilate;
  int KH, KW, sh, sw, ph, pw, dh, dw;
  int OH, OW, OHW, N, M, Kpad;
  int in_act; float in_slope;
  int rowbias_ld, geglu, out_act; float out_slope; float alpha;
  int post_act; float post_slope; bf16* out2;
  int out_f32, out_ld; long long out_bs; int out_ps, out_po;
  int vt_col0, vt_ld; long long vt_bs; int vt_dual;
  int splits, kt_per_split, nkt;
  int tiles_n;
  int tiles_m, nwg, xmap;     // 1-D grid of tiles_m * tiles_n * splits workgroups; xmap: work item -> XCD order (see igemm_work_item)
  FastDiv fd_tiles_m, fd_splits;
  unsigned x_bytes, x2_bytes, w_bytes, la_bytes, lb_bytes;
  unsigned long long* diag;   // diagnostic builds only
  FastDiv fd_ohw, fd_ow, fd_halo;   // fd_halo: / (OW + 2), halo kernel only
  FastDiv fd_tiles_n, fd_ctot, fd_kw;   // the pipe kernel's prologue: tile index / tiles_n, K cursor / Ctot, tap / KW (scalar divisions
                                        // by a run-time value are ~30 VALU instructions each, on the path to the first DMA)
  const float* ln_s; const float* ln_sa; const float* ln_ca; float ln_eps;   // LayerNorm folded into the GEMM (see below)
  const bf16* x3; const bf16* x4; int Cin3, Cin4, C3tot; unsigned x3_bytes, x4_bytes;   // fused 1x1 second-source segment (shortcut)
  float* qstat; int qtile;    // GroupNorm hand-over: per (M-tile, image slot, 4-channel quad) partial (sum, sum of squares); qtile >= 0 overrides m0 / BM
  float* rowstat;             // producer side of the LayerNorm hand-over: [M][tiles_n][2] (sum, sum of squares) per output row and N-tile
  const float* ln_parts; int ln_np;   // consumer side: the producer's table, ln_np partial pairs per row
  // GroupNorm (+ SiLU) of the input folded into the halo gather (igemm_halo.hip, GNIN instantiations); gi_gamma == nullptr: off
  const float* gi_gamma; const float* gi_beta; const float* gi_q1; const float* gi_q2;
  int gi_bm1, gi_tpi1, gi_bm2, gi_tpi2, gi_groups, gi_act; float gi_eps;
  int ws_rows;                // rows of one split-K slab of the workspace (= the launch's M; the halo tiles cut p.M per image for the bound checks).
                              // LAST on purpose: a new member anywhere above shifts the scalar-load groups of every kernel's prologue
};

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// ---- work item of a workgroup -----------------------------------------------------------------
// The grid is ONE-dimensional (tiles_m * tiles_n * splits workgroups): the hardware deals workgroups round-robin over the 8 XCDs by
// their LINEAR id (blocks b, b + 8, ... share an L2), so a 3-D grid's `blockIdx.x & 7` is the XCD only when gridDim.x % 8 == 0.
// Each XCD gets a CONTIGUOUS range of the launch's work items in one of two orders, chosen per launch on the host:
//   xmap 0 -- activation-stationary: (tile_m, tile_n) slowest, split fastest.  An XCD owns a band of output rows for every N-tile and
//             every K-split: its slice of the activation image (plus halo) stays in its 4 MiB L2, every L2 streams the whole weight
//             matrix.  Right where activations >> weights (the 4000- / 1000-pixel levels, VAE, vocoder).
//   xmap 1 -- weight-stationary: (split, tile_n) slowest, tile_m fastest.  An XCD owns a (K-slice, N-tile) range of the WEIGHTS for all
//             rows: each weight byte is fetched by one L2 instead of eight.  Right where weights >> activations (the 252- / 64-token
//             levels: 7.4 MB of weights against 0.65 MB of activations at M = 512 -- round 3 measured 45 MB fetched per launch).
// Bijective for any grid size; affects speed only.
__device__ __forceinline__ void igemm_work_item(const IgemmDev& p, int& tile_m, int& tile_n, int& split) {
  const int nwg = p.nwg, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  if (p.xmap) {
    const int sn = fdiv(w, p.fd_tiles_m);
    tile_m = w - sn * p.tiles_m;
    split = fdiv(sn, p.fd_tiles_n);
    tile_n = sn - split * p.tiles_n;
  } else {
    const int t = fdiv(w, p.fd_splits);
    split = w - t * p.splits;
    tile_m = fdiv(t, p.fd_tiles_n);
    tile_n = t - tile_m * p.tiles_n;
  }
}

// ---- epilogue helpers -------------------------------------------------------------------------
// v[0..3] are 4 consecutive output columns n..n+3 of output row m, bias etc. already added.
__device__ __forceinline__ void finish_store4(const IgemmDev& p, int m, int n, int ncols, float* v) {
  const int b = m / p.OHW;
  const int pix = m - b * p.OHW;
  const long long row = (long long)b * p.out_bs + (long long)(pix * p.out_ps + p.out_po) * p.out_ld;
  const bool vec = (n + 3 < ncols) && ((p.out_ld & 3) == 0) && ((p.out_bs & 3) == 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], p.out_act, p.out_slope);
  if (p.res) {
    if (vec) {
      bf16x4 r = *reinterpret_cast<const bf16x4*>(p.res + row + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
    } else {
      for (int j = 0; j < 4; ++j) if (n + j < ncols) v[j] += (float)p.res[row + n + j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] *= p.alpha;
  if (p.res2) {
    if (vec) {
      bf16x4 r = *reinterpret_cast<const bf16x4*>(p.res2 + row + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
    } else {
      for (int j = 0; j < 4; ++j) if (n + j < ncols) v[j] += (float)p.res2[row + n + j];
    }
  }
  if (p.out2) {
    bf16* o2 = p.out2 + row + n;
    if (vec) {
      *reinterpret_cast<bf16x4*>(o2) = bf16x4{(bf16)apply_act(v[0], p.post_act, p.post_slope), (bf16)apply_act(v[1], p.post_act, p.post_slope),
                                              (bf16)apply_act(v[2], p.post_act, p.post_slope), (bf16)apply_act(v[3], p.post_act, p.post_slope)};
    } else {
      for (int j = 0; j < 4; ++j) if (n + j < ncols) o2[j] = (bf16)apply_act(v[j], p.post_act, p.post_slope);
    }
  } else if (p.post_act) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], p.post_act, p.post_slope);
  }
  if (p.out_f32) {
    float* o = reinterpret_cast<float*>(p.out) + row + n;
    if (vec) {
      *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
      for (int j = 0; j < 4; ++j) if (n + j < ncols) o[j] = v[j];
    }
  } else {
    bf16* o = reinterpret_cast<bf16*>(p.out) + row + n;
    if (vec) {
      *reinterpret_cast<bf16x4*>(o) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    } else {
      for (int j = 0; j < 4; ++j) if (n + j < ncols) o[j] = (bf16)v[j];
    }
  }
}

__device__ __forceinline__ void add_bias4(const IgemmDev& p, int m, int n, float* v) {
  // n % 4 == 0 and N % 4 == 0 on this path (split-K reduce): whole 16-byte vectors, never per-element predicated loads
  if (p.bias) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += b[j];
  }
  if (p.rowbias) {
    const int bi = fdiv(m, p.fd_ohw);
    const float* rb = p.rowbias + (long long)bi * p.rowbias_ld + n;
    if ((p.rowbias_ld & 3) == 0) {
      const f32x4 r = *reinterpret_cast<const f32x4*>(rb);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += r[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += rb[j];
    }
  }
}

// ---- shared epilogue ---------------------------------------------------------------------------------------------
// The accumulators go through LDS (the K-loop buffers are free by then): every wave drops its 16x16 tiles into one
// fp32 [BM][BN+4] image (or its transpose for V^T tiles), then the whole workgroup walks that image in 8-column groups.
// One compact copy of the bias / GEGLU / activation / residual / store code serves all tiles (the fully unrolled
// per-tile form was ~20k instructions of cold straight-line code -- an instruction-cache disaster that cost more than
// the K loop itself), and global stores become full 16-byte row segments.
__device__ __forceinline__ void epi8(const IgemmDev& p, int m, int n, int ncols, float* v) {
  // v[0..7]: output columns n..n+7 of output row m, bias already added
  const int b = fdiv(m, p.fd_ohw);
  const int pix = m - b * p.OHW;
  const long long row = (long long)b * p.out_bs + (long long)(pix * p.out_ps + p.out_po) * p.out_ld;
  const bool vec = (n + 7 < ncols) && ((p.out_ld & 7) == 0) && ((p.out_bs & 7) == 0);
  if (p.out_act) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = apply_act(v[j], p.out_act, p.out_slope);
  }
  if (p.res) {
    if (vec) {
      const bf16x8 r = *reinterpret_cast<const bf16x8*>(p.res + row + n);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += (float)r[j];
    } else {
      for (int j = 0; j < 8; ++j) if (n + j < ncols) v[j] += (float)p.res[row + n + j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] *= p.alpha;
  if (p.res2) {
    if (vec) {
      const bf16x8 r = *reinterpret_cast<const bf16x8*>(p.res2 + row + n);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += (float)r[j];
    } else {
      for (int j = 0; j < 8; ++j) if (n + j < ncols) v[j] += (float)p.res2[row + n + j];
    }
  }
  if (p.out2) {
    bf16* o2 = p.out2 + row + n;
    if (vec) {
      bf16x8 t;
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = (bf16)apply_act(v[j], p.post_act, p.post_slope);
      *reinterpret_cast<bf16x8*>(o2) = t;
    } else {
      for (int j = 0; j < 8; ++j) if (n + j < ncols) o2[j] = (bf16)apply_act(v[j], p.post_act, p.post_slope);
    }
  } else if (p.post_act) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = apply_act(v[j], p.post_act, p.post_slope);
  }
  if (p.out_f32) {
    float* o = reinterpret_cast<float*>(p.out) + row + n;
    if (vec) {
      *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
      for (int j = 0; j < 8; ++j) if (n + j < ncols) o[j] = v[j];
    }
  } else {
    bf16* o = reinterpret_cast<bf16*>(p.out) + row + n;
    if (vec) {
      bf16x8 t;
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = (bf16)v[j];
      *reinterpret_cast<bf16x8*>(o) = t;
    } else {
      for (int j = 0; j < 8; ++j) if (n + j < ncols) o[j] = (bf16)v[j];
    }
  }
}

template <int BM, int BN>
struct EpiCfg {
  static constexpr int LD = BN + 4;                         // fp32 row stride of the normal image
  static constexpr int LDT = BM + 4;                        // row stride of the transposed (V^T) image
  static constexpr int BYTES = (BM * LD > BN * LDT ? BM * LD : BN * LDT) * 4;
};

// LEAN = true: the launch is known (host-checked) to take the standard path without an activation -- no V^T tile, no split-K,
// no GEGLU -- and every other path is compiled out: the kernel's instruction footprint shrinks from ~16k to a few thousand
// instructions, which is what its instruction-cache behaviour needs.
// EPI = 2 (GEGLU launches without residual / activation / second output): only the GEGLU path with a plain bf16 store.
template <int BM, int BN, int MI, int NI, bool VT, int NT = 256, int EPI = 0>
__device__ __forceinline__ void igemm_epilogue(const IgemmDev& p, f32x4 (&acc)[MI][NI], char* smem, bool vt_wg, int m0,
                                               int n0, int wm_off, int wn_off, int lrow, int lq, int split, int tid,
                                               const float* lnst = nullptr) {
  using E = EpiCfg<BM, BN>;
  // EPI 4 = LEAN + the GroupNorm-statistics hand-over (qstat): its own instantiation, so that the launches without it keep the
  // compact code (with the statistics code compiled into every LEAN epilogue the short-K GEMMs ran 0.3-0.7 us slower each)
  constexpr bool LEAN = EPI == 1 || EPI == 4, GLEAN = EPI == 2, SLEAN = EPI == 3;   // 3: split-K launches -- only the partial-tile store
  constexpr bool QS = EPI == 0 || EPI == 4;
#ifndef ALDM_NO_KA_PREFETCH
  aldm_prefetch_next_kernargs<sizeof(IgemmDev)>(tid);        // under the epilogue: the next launch's argument segment into this XCD's L2
#endif
  if constexpr (SLEAN) {
    // Split-K partial tile straight from the accumulators: with the swapped MFMA a lane holds 4 consecutive channels of one pixel row, so
    // the fp32 partials leave as 16-byte stores (four lanes = 64 contiguous bytes of a row) without the LDS transposition and its two
    // barriers -- the consumer (igemm_reduce_kernel / the deferred GroupNorm) reads the slab row-major either way.
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm_off + i * 16 + lrow;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn_off + j * 16 + lq * 4;
        if (m < p.M && n < p.N) *reinterpret_cast<f32x4*>(p.ws + ((long long)split * p.ws_rows + m) * p.N + n) = acc[i][j];   // (N % 4 == 0, host-checked)
      }
    }
    return;
  }
  float* Cs = reinterpret_cast<float*>(smem);
  if constexpr (LEAN && !VT) lnst = nullptr;               // LEAN keeps the V^T tile and the LayerNorm fold only in VT kernels

  // Standard path (no V^T tile, no split-K, no GEGLU).  A thread keeps ONE 8-column group (NT % (BN/8) == 0) and walks down
  // the rows, so the bias / LayerNorm column vectors are loaded once.  Rows are handled CH at a time in two phases: phase A
  // issues every global read the rows need (time-embedding row bias, residuals), phase B does the arithmetic and the stores,
  // so a thread pays one memory round trip per CH rows instead of one per row -- and phase A of the first CH rows is issued
  // HERE, before the accumulators make their LDS round trip, which hides that latency behind the transposition.
  // Measured in a replayed graph (MI355X): epilogue of the 8-wave 128x128 tile 10.2 -> 5.8 us with the two-phase walk.
  constexpr int GPR = BN / 8, RSTEP = NT / GPR, ITERS = BM / RSTEP, CH = ITERS < 4 ? ITERS : 4;
  static_assert(NT % GPR == 0 && BM % RSTEP == 0 && ITERS % CH == 0, "epilogue row walk");
  const int r0 = tid / GPR, c = (tid % GPR) * 8, n = n0 + c;
  const bool dual = VT && vt_wg && p.vt_dual;               // V^T tile that is ALSO stored row-major (trainer's q | k | v)
  const bool std_path = !GLEAN && !SLEAN && (!(VT && vt_wg) || dual) && (LEAN || (p.splits <= 1 && !p.geglu));
  const bool vec = (n + 7 < p.N) && ((p.out_ld & 7) == 0) && ((p.out_bs & 7) == 0);
  const bool rbvec = vec && ((p.rowbias_ld & 3) == 0);
  float colb[8], lns[8];
  long long rowo[CH];
  bool ok[CH];
  f32x4 rb0[CH], rb1[CH];
  bf16x8 r1[CH], r2[CH];
  // GroupNorm hand-over (p.qstat): this thread's partial (sum, sum of squares) of its two 4-channel quads, per image slot -- an
  // M-tile of the generic kernels may run across the boundary between two images (slot 1 = rows of the second one)
  int slot[CH];
  float qacc[2][2][2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};
  const int qb0 = (QS && p.qstat) ? fdiv(min(m0, p.M - 1), p.fd_ohw) : 0;
  auto phaseA = [&](int it0) {                               // addresses + every global read of rows it0 .. it0+CH-1
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int m = m0 + r0 + (it0 + u) * RSTEP;
      ok[u] = m < p.M;
      rb0[u] = rb1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      r1[u] = r2[u] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      rowo[u] = 0;
      slot[u] = 0;
      if (!ok[u]) continue;
      const int b = fdiv(m, p.fd_ohw);
      const int pix = m - b * p.OHW;
      slot[u] = b != qb0;
      rowo[u] = (long long)b * p.out_bs + (long long)(pix * p.out_ps + p.out_po) * p.out_ld;
      if (p.rowbias) {
        const float* rb = p.rowbias + (long long)b * p.rowbias_ld + n;
        if (rbvec) {
          rb0[u] = *reinterpret_cast<const f32x4*>(rb);
          rb1[u] = *reinterpret_cast<const f32x4*>(rb + 4);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (n + q < p.N) rb0[u][q] = rb[q];
            if (n + 4 + q < p.N) rb1[u][q] = rb[4 + q];
          }
        }
      }
      if (vec) {
        if (p.res) r1[u] = *reinterpret_cast<const bf16x8*>(p.res + rowo[u] + n);
        if (p.res2) r2[u] = *reinterpret_cast<const bf16x8*>(p.res2 + rowo[u] + n);
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (p.res && n + q < p.N) r1[u][q] = p.res[rowo[u] + n + q];
          if (p.res2 && n + q < p.N) r2[u][q] = p.res2[rowo[u] + n + q];
        }
      }
    }
  };
  // ACT = std::true_type only when the launch carries an activation: the SiLU / tanh / erf-GELU code (thousands of
  // instructions once inlined per value) then stays out of the instruction stream of every other launch -- before this
  // split the epilogue was 14.5k instructions and its instruction-cache misses cost more than the K loop of the small GEMMs.
  auto phaseB = [&](auto ACT, int it0) {                     // arithmetic + stores
    constexpr bool kAct = decltype(ACT)::value;
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      if (!ok[u]) continue;
      const int r = r0 + (it0 + u) * RSTEP;
      float v[8];
      if (VT && vt_wg) {                                     // dual store: the LDS image is transposed ([channel][pixel])
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = Cs[(c + q) * E::LDT + r];
      } else {
        const float* src = Cs + r * E::LD + c;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = src[q];
      }
      if (lnst) {
        const float mu = lnst[r], rs = lnst[BM + r];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = rs * (v[q] - mu * lns[q]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { v[q] += colb[q] + rb0[u][q]; v[4 + q] += colb[4 + q] + rb1[u][q]; }
      if constexpr (kAct) {
        if (p.out_act) {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = apply_act(v[q], p.out_act, p.out_slope);
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = (v[q] + (float)r1[u][q]) * p.alpha + (float)r2[u][q];
      if (p.out2) {
        bf16* o2 = p.out2 + rowo[u] + n;
        bf16x8 t;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if constexpr (kAct) t[q] = (bf16)apply_act(v[q], p.post_act, p.post_slope);
          else t[q] = (bf16)v[q];
        }
        if (vec) *reinterpret_cast<bf16x8*>(o2) = t;
        else
          for (int q = 0; q < 8; ++q) if (n + q < p.N) o2[q] = t[q];
      } else if (kAct && p.post_act) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = apply_act(v[q], p.post_act, p.post_slope);
      }
      if (p.out_f32) {
        float* o = reinterpret_cast<float*>(p.out) + rowo[u] + n;
        if (vec) {
          *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
          for (int q = 0; q < 8; ++q) if (n + q < p.N) o[q] = v[q];
        }
      } else {
        bf16* o = reinterpret_cast<bf16*>(p.out) + rowo[u] + n;
        bf16x8 t;
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = (bf16)v[q];
        if (vec) *reinterpret_cast<bf16x8*>(o) = t;       // (non-temporal stores: -7 % on isolated GEMMs, +1.2 % on the whole step)
        else
          for (int q = 0; q < 8; ++q) if (n + q < p.N) o[q] = t[q];
        if (QS && p.qstat) {
          float rs[2][2] = {{0.f, 0.f}, {0.f, 0.f}};         // this row: [quad][sum, sum of squares] of the values as stored
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float f = (float)t[q];
            rs[q >> 2][0] += f;
            rs[q >> 2][1] = fmaf(f, f, rs[q >> 2][1]);
          }
          if (slot[u]) {                                     // (static register indices on both sides: no scratch)
            qacc[1][0][0] += rs[0][0]; qacc[1][0][1] += rs[0][1]; qacc[1][1][0] += rs[1][0]; qacc[1][1][1] += rs[1][1];
          } else {
            qacc[0][0][0] += rs[0][0]; qacc[0][0][1] += rs[0][1]; qacc[0][1][0] += rs[1][0]; qacc[0][1][1] += rs[1][1];
          }
        }
        if (p.rowstat) {
          // LayerNorm hand-over: this row's sum / sum of squares over the tile's BN columns, of the values AS STORED (bf16).
          // The GPR threads of a row are consecutive lanes of one wave and are all active together (host-checked N % BN == 0).
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int q = 0; q < 8; ++q) { const float f = (float)t[q]; s1 += f; s2 = fmaf(f, f, s2); }
#pragma unroll
          for (int o2 = 1; o2 < GPR; o2 <<= 1) { s1 += __shfl_xor(s1, o2, 64); s2 += __shfl_xor(s2, o2, 64); }
          if ((tid % GPR) == 0) {
            float* dst = p.rowstat + ((long long)(m0 + r) * p.tiles_n + n0 / BN) * 2;
            dst[0] = s1;
            dst[1] = s2;
          }
        }
      }
    }
  };
  if (std_path && n < p.N) {
    // column vectors: two 16-byte loads each when the 8 columns are in range (per-element predicated loads compile to eight
    // exec-masked blocks whose loads complete one after the other: measured 3.6 us of the epilogue on the 64x64 tile)
#pragma unroll
    for (int q = 0; q < 8; ++q) colb[q] = lns[q] = 0.f;
    if (n + 7 < p.N) {
      if (p.bias) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { colb[q] = b0[q]; colb[4 + q] = b1[q]; }
      }
      if (lnst) {
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.ln_s + n), s1 = *reinterpret_cast<const f32x4*>(p.ln_s + n + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { lns[q] = s0[q]; lns[4 + q] = s1[q]; }
      }
    } else {
      for (int q = 0; q < 8; ++q) {
        if (p.bias && n + q < p.N) colb[q] = p.bias[n + q];
        if (lnst && n + q < p.N) lns[q] = p.ln_s[n + q];
      }
    }
    phaseA(0);
  }

  __syncthreads();                                           // every wave is done with the K-loop images
  if (VT && vt_wg) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)   // un-swapped MFMA: lane holds 4 consecutive pixels of channel column lrow
        *reinterpret_cast<f32x4*>(Cs + (wn_off + j * 16 + lrow) * E::LDT + wm_off + i * 16 + lq * 4) = acc[i][j];
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)   // swapped MFMA: lane holds 4 consecutive channels of pixel row lrow
        *reinterpret_cast<f32x4*>(Cs + (wm_off + i * 16 + lrow) * E::LD + wn_off + j * 16 + lq * 4) = acc[i][j];
  }
  __syncthreads();

  if (VT && vt_wg) {
    // rows of the image are channels n, columns are pixels: vt[b][n - col0][pix .. pix+7]
    for (int g = tid; g < BN * (BM / 8); g += NT) {
      const int rn = g / (BM / 8), cm = (g - rn * (BM / 8)) * 8;
      const int n = n0 + rn, m = m0 + cm;
      if (n >= p.N || m >= p.M) continue;
      const float bb = p.bias ? p.bias[n] : 0.f;
      const float lsn = lnst ? p.ln_s[n] : 0.f;
      const float* src = Cs + rn * E::LDT + cm;
      const int b = fdiv(m, p.fd_ohw), pix = m - b * p.OHW;
      bf16* o = p.vt + (long long)b * p.vt_bs + (long long)(n - p.vt_col0) * p.vt_ld + pix;
      if (m + 7 < p.M && pix + 7 < p.OHW && ((p.vt_ld & 7) == 0) && ((pix & 7) == 0) && ((p.vt_bs & 7) == 0)) {
        bf16x8 t;
        float mu8[8], rs8[8];
        if (lnst) {                                              // cm is a multiple of 8: four 16-byte LDS reads
          const f32x4 m0v = *reinterpret_cast<const f32x4*>(lnst + cm), m1v = *reinterpret_cast<const f32x4*>(lnst + cm + 4);
          const f32x4 r0v = *reinterpret_cast<const f32x4*>(lnst + BM + cm), r1v = *reinterpret_cast<const f32x4*>(lnst + BM + cm + 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) { mu8[q] = m0v[q]; mu8[4 + q] = m1v[q]; rs8[q] = r0v[q]; rs8[4 + q] = r1v[q]; }
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) { mu8[q] = 0.f; rs8[q] = 1.f; }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = (bf16)(rs8[q] * (src[q] - mu8[q] * lsn) + bb);
        *reinterpret_cast<bf16x8*>(o) = t;
      } else {
        for (int q = 0; q < 8; ++q) {
          const int mq = m + q;
          if (mq >= p.M) break;
          const int bq = fdiv(mq, p.fd_ohw), pq = mq - bq * p.OHW;
          float val = src[q];
          if (lnst) val = lnst[BM + cm + q] * (val - lnst[cm + q] * lsn);
          p.vt[(long long)bq * p.vt_bs + (long long)(n - p.vt_col0) * p.vt_ld + pq] = (bf16)(val + bb);
        }
      }
    }
    if (!dual) return;
  }
  if constexpr (!LEAN) {
  if (!GLEAN && (SLEAN || p.splits > 1)) {
    for (int g = tid; g < BM * (BN / 8); g += NT) {
      const int r = g / (BN / 8), c = (g - r * (BN / 8)) * 8;
      const int m = m0 + r, n = n0 + c;
      if (m >= p.M || n >= p.N) continue;
      float* o = p.ws + ((long long)split * p.ws_rows + m) * p.N + n;
      const float* src = Cs + r * E::LD + c;
      if (n + 7 < p.N) {
        *reinterpret_cast<f32x4*>(o) = *reinterpret_cast<const f32x4*>(src);
        *reinterpret_cast<f32x4*>(o + 4) = *reinterpret_cast<const f32x4*>(src + 4);
      } else {
        for (int q = 0; q < 8; ++q) if (n + q < p.N) o[q] = src[q];
      }
    }
    return;
  }
  if (!SLEAN && (GLEAN || p.geglu)) {
    // image columns come in blocks of (16 value | 16 gate); 8 output columns = 8 values and their 8 gates.  A thread keeps ONE
    // column group and walks down the rows, so the bias / LayerNorm column vectors are four 16-byte loads per thread, once
    // (per-element scalar loads of them were 32-64 global loads per thread: 3-13 us of the feed-forward GEMMs' epilogue).
    constexpr int CG = BN / 16, RS = NT / CG;
    static_assert(NT % CG == 0 && BM % RS == 0, "GEGLU row walk");
    const int jo = (tid % CG) * 8;                                       // output column inside the tile
    const int cv = ((jo >> 4) << 5) + (jo & 15);                         // value column inside the tile
    const int nv = n0 + cv;
    if (nv >= p.N) return;                                               // N % 32 == 0: the whole (value | gate) group is in or out
    float bv[8], bg[8], sv[8], sg[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) bv[q] = bg[q] = sv[q] = sg[q] = 0.f;
    if (p.bias) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(p.bias + nv), a1 = *reinterpret_cast<const f32x4*>(p.bias + nv + 4);
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.bias + nv + 16), g1 = *reinterpret_cast<const f32x4*>(p.bias + nv + 20);
#pragma unroll
      for (int q = 0; q < 4; ++q) { bv[q] = a0[q]; bv[4 + q] = a1[q]; bg[q] = g0[q]; bg[4 + q] = g1[q]; }
    }
    if (lnst) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(p.ln_s + nv), a1 = *reinterpret_cast<const f32x4*>(p.ln_s + nv + 4);
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.ln_s + nv + 16), g1 = *reinterpret_cast<const f32x4*>(p.ln_s + nv + 20);
#pragma unroll
      for (int q = 0; q < 4; ++q) { sv[q] = a0[q]; sv[4 + q] = a1[q]; sg[q] = g0[q]; sg[4 + q] = g1[q]; }
    }
    for (int r = tid / CG; r < BM; r += RS) {
      const int m = m0 + r;
      if (m >= p.M) break;
      const float* src = Cs + r * E::LD + cv;
      const float mu = lnst ? lnst[r] : 0.f, rs = lnst ? lnst[BM + r] : 1.f;
      float val[8], gate[8], v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        val[q] = rs * (src[q] - mu * sv[q]) + bv[q];
        gate[q] = rs * (src[16 + q] - mu * sg[q]) + bg[q];
        v[q] = val[q] * gelu_erf_f(gate[q]);
      }
      if constexpr (GLEAN) {
        if (p.out2) {   // training: also keep the pre-activation projection h [M][N] (packed value | gate columns) for the backward
          bf16* hrow = p.out2 + (long long)m * p.N + nv;
          bf16x8 hv, hg;
#pragma unroll
          for (int q = 0; q < 8; ++q) { hv[q] = (bf16)val[q]; hg[q] = (bf16)gate[q]; }
          *reinterpret_cast<bf16x8*>(hrow) = hv;
          *reinterpret_cast<bf16x8*>(hrow + 16) = hg;
        }
        // host-checked: bf16 output, no residual / activation, alpha = 1 -> one 16-byte store
        const int b = fdiv(m, p.fd_ohw), pix = m - b * p.OHW, no = (n0 >> 1) + jo, ncols = p.N >> 1;
        bf16* o = reinterpret_cast<bf16*>(p.out) + (long long)b * p.out_bs + (long long)(pix * p.out_ps + p.out_po) * p.out_ld + no;
        bf16x8 t;
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = (bf16)v[q];
        if (no + 7 < ncols && ((p.out_ld & 7) == 0) && ((p.out_bs & 7) == 0)) *reinterpret_cast<bf16x8*>(o) = t;
        else
          for (int q = 0; q < 8; ++q) if (no + q < ncols) o[q] = t[q];
      } else {
        epi8(p, m, (n0 >> 1) + jo, p.N >> 1, v);
      }
    }
    return;
  }
  }
  // Standard path, second half: the first CH rows were prefetched before the LDS transposition (see the top of this function)
  if (n >= p.N) return;
  if (!LEAN && (p.out_act | p.post_act)) {
    phaseB(std::true_type{}, 0);
#pragma unroll
    for (int it0 = CH; it0 < ITERS; it0 += CH) {
      phaseA(it0);
      phaseB(std::true_type{}, it0);
    }
  } else {
    phaseB(std::false_type{}, 0);
#pragma unroll
    for (int it0 = CH; it0 < ITERS; it0 += CH) {
      phaseA(it0);
      phaseB(std::false_type{}, it0);
    }
  }
  if (QS && p.qstat) {
    // fold the threads that share a column group (lanes l, l + GPR, ...; then the waves through LDS -- the fp32 image is dead:
    // host-checked N % BN == 0, so every thread of the workgroup is here) and write the tile's table rows:
    //   qstat[(tile * 2 + slot) * (N / 4) + quad][2]
    __syncthreads();
    float* red = Cs;                                         // [NT / 64 waves][GPR column groups][8]
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int qd = 0; qd < 2; ++qd)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float v = qacc[s2][qd][e];
#pragma unroll
          for (int o2 = GPR; o2 < 64; o2 <<= 1) v += __shfl_xor(v, o2, 64);
          qacc[s2][qd][e] = v;
        }
    const int lane = tid & 63, wv = tid >> 6;
    if (lane < GPR) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int qd = 0; qd < 2; ++qd)
#pragma unroll
          for (int e = 0; e < 2; ++e) red[(wv * GPR + lane) * 8 + s2 * 4 + qd * 2 + e] = qacc[s2][qd][e];
    }
    __syncthreads();
    if (tid < GPR * 8) {
      const int cg = tid >> 3, j = tid & 7;                  // column group, (slot, quad, which)
      float v = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NT / 64; ++w2) v += red[(w2 * GPR + cg) * 8 + j];
      const int tile = p.qtile >= 0 ? p.qtile : m0 / BM;
      const int quad = (n0 + cg * 8) / 4 + ((j >> 1) & 1);
      p.qstat[(((long long)tile * 2 + (j >> 2)) * (p.N >> 2) + quad) * 2 + (j & 1)] = v;
    }
  }
}

// ---- main kernel ------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int RP, bool VT>
__global__ __launch_bounds__(THREADS) void igemm_kernel(const IgemmDev p) {
  aldm_touch_kernargs<sizeof(IgemmDev)>();
  static_assert(WM * WN == 4, "4 waves");
  constexpr int MI = BM / WM / 16, NI = BN / WN / 16;
  constexpr int BROWS = BN + RP;
  constexpr int A_PASSES = BM / 32, B_PASSES = BROWS / 32;
  constexpr int RT_W = RP / 16 / WN;  // LoRA r-tiles (16 wide) computed by each wave
  static_assert(RP == 0 || RT_W >= 1, "RP tiles must split over WN");
  static_assert(NI % 2 == 0 || BN / WN == 16, "GEGLU pairs");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                        // [2][BM][128 B]
  char* Bs = smem + 2 * BM * 128;         // [2][BROWS][128 B]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int tile_m, tile_n, split;
  igemm_work_item(p, tile_m, tile_n, split);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kt0 = split * p.kt_per_split;
  const int kt1 = min(p.nkt, kt0 + p.kt_per_split);

  // ---- per-thread gather state ----
  const int kchunk = tid & 7, rbase = tid >> 3;
  int a_b[A_PASSES], a_ih0[A_PASSES], a_iw0[A_PASSES];
  bool a_ok[A_PASSES];
#pragma unroll
  for (int ps = 0; ps < A_PASSES; ++ps) {
    const int m = m0 + rbase + 32 * ps;
    a_ok[ps] = m < p.M;
    const int mm = a_ok[ps] ? m : 0;
    const int b = fdiv(mm, p.fd_ohw);
    const int pix = mm - b * p.OHW;
    const int oh = fdiv(pix, p.fd_ow), ow = pix - oh * p.OW;
    a_b[ps] = b;
    a_ih0[ps] = oh * p.sh - p.ph;
    a_iw0[ps] = ow * p.sw - p.pw;
  }
  int kc, kkh, kkw;  // channel / tap of this thread's chunk in the current K-tile
  {
    const int k = kt0 * BK + kchunk * 8;
    const int tap = k / p.Ctot;
    kc = k - tap * p.Ctot;
    kkh = tap / p.KW;
    kkw = tap - kkh * p.KW;
  }
  bf16x8 a_reg[A_PASSES], b_reg[B_PASSES];
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  auto gather = [&](int kt) {
    // activations
    const bool tap_ok = kkh < p.KH;
    const bf16* src; int cs, Cs;
    if (kc < p.Cin) { src = p.x; cs = kc; Cs = p.Cin; } else { src = p.x2; cs = kc - p.Cin; Cs = p.Cin2; }
#pragma unroll
    for (int ps = 0; ps < A_PASSES; ++ps) {
      int ih = a_ih0[ps] + kkh * p.dh, iw = a_iw0[ps] + kkw * p.dw;
      bool ok = a_ok[ps] && tap_ok;
      if (p.dilate == 2) {
        ok = ok && ih >= 0 && iw >= 0 && !((ih | iw) & 1) && (ih >> 1) < p.IH && (iw >> 1) < p.IW;
        ih >>= 1;
        iw >>= 1;
      } else if (p.UH > 0) {
        ok = ok && (unsigned)ih < (unsigned)p.UH && (unsigned)iw < (unsigned)p.UW;
        ih = (ih * p.IH) / p.UH;
        iw = (iw * p.IW) / p.UW;
      } else {
        ok = ok && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
      }
      bf16x8 v = zero8;
      if (ok) {
        const long long off = ((long long)(a_b[ps] * p.IH + ih) * p.IW + iw) * Cs + cs;
        v = *reinterpret_cast<const bf16x8*>(src + off);
        if (p.in_act) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (bf16)apply_act((float)v[j], p.in_act, p.in_slope);
        }
      }
      a_reg[ps] = v;
    }
    // weights (+ LoRA-A rows appended)
    const long long kofs = (long long)kt * BK + kchunk * 8;
#pragma unroll
    for (int ps = 0; ps < B_PASSES; ++ps) {
      const int r = rbase + 32 * ps;
      bf16x8 v = zero8;
      if (r < BN) {
        const int n = n0 + r;
        if (n < p.N) v = *reinterpret_cast<const bf16x8*>(p.w + (long long)n * p.Kpad + kofs);
      } else if (RP > 0) {
        v = *reinterpret_cast<const bf16x8*>(p.lora_a + (long long)(r - BN) * p.Kpad + kofs);
      }
      b_reg[ps] = v;
    }
    // advance the tap/channel cursor by one K-tile
    kc += BK;
    while (kc >= p.Ctot) {
      kc -= p.Ctot;
      if (++kkw == p.KW) { kkw = 0; ++kkh; }
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int ps = 0; ps < A_PASSES; ++ps) {
      const int r = rbase + 32 * ps;
      *reinterpret_cast<bf16x8*>(As + (buf * BM + r) * 128 + swz(r, kchunk) * 16) = a_reg[ps];
    }
#pragma unroll
    for (int ps = 0; ps < B_PASSES; ++ps) {
      const int r = rbase + 32 * ps;
      *reinterpret_cast<bf16x8*>(Bs + (buf * BROWS + r) * 128 + swz(r, kchunk) * 16) = b_reg[ps];
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 tacc[MI][RT_W > 0 ? RT_W : 1];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < (RT_W > 0 ? RT_W : 1); ++j) tacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  bool vt_tile[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) vt_tile[j] = VT && (n0 + wn * (BN / WN) + j * 16 >= p.vt_col0);

  const int lrow = lane & 15, lq = lane >> 4;

  auto mma_step = [&](int buf, int ks, bool with_t) {
    const int ch = ks * 4 + lq;
    bf16x8 af[MI], wf[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int r = wm * (BM / WM) + i * 16 + lrow;
      af[i] = *reinterpret_cast<const bf16x8*>(As + (buf * BM + r) * 128 + swz(r, ch) * 16);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int r = wn * (BN / WN) + j * 16 + lrow;
      wf[j] = *reinterpret_cast<const bf16x8*>(Bs + (buf * BROWS + r) * 128 + swz(r, ch) * 16);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        if (VT && vt_tile[j])
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], wf[j], acc[i][j], 0, 0, 0);
        else
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
      }
    if (RP > 0 && with_t) {
#pragma unroll
      for (int t = 0; t < RT_W; ++t) {
        const int r = BN + (wn * RT_W + t) * 16 + lrow;
        const bf16x8 lf = *reinterpret_cast<const bf16x8*>(Bs + (buf * BROWS + r) * 128 + swz(r, ch) * 16);
#pragma unroll
        for (int i = 0; i < MI; ++i)
          tacc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lf, af[i], tacc[i][t], 0, 0, 0);
      }
    }
  };

  // ---- main loop: one barrier per K-tile, register-staged double buffer ----
  if (kt0 < kt1) {
    gather(kt0);
    stage(0);
  }
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    const bool more = kt + 1 < kt1;
    if (more) gather(kt + 1);
    mma_step(buf, 0, true);
    mma_step(buf, 1, true);
    if (more) stage(buf ^ 1);
    __syncthreads();
  }

  // ---- LoRA: T (bf16) -> LDS, then one more K-step against the pre-scaled B ----
  if (RP > 0) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int t = 0; t < RT_W; ++t) {
        const int r = wm * (BM / WM) + i * 16 + lrow;  // T row (pixel)
        const int col = (wn * RT_W + t) * 16 + lq * 4; // T column (rank index)
        bf16x4 tv = {(bf16)tacc[i][t][0], (bf16)tacc[i][t][1], (bf16)tacc[i][t][2], (bf16)tacc[i][t][3]};
        *reinterpret_cast<bf16x4*>(As + r * 128 + swz(r, col >> 3) * 16 + (col & 7) * 2) = tv;
        if (p.lora_t_out && tile_n == 0 && m0 + r < p.M)
          *reinterpret_cast<bf16x4*>(p.lora_t_out + ((long long)split * p.M + m0 + r) * RP + col) = tv;
      }
    constexpr int RCH = RP > 0 ? RP / 8 : 1;
    constexpr int LB_CHUNKS = BN * RCH;
    for (int c = tid; c < LB_CHUNKS; c += THREADS) {
      const int r = c / RCH, ch = c - r * RCH;
      const int n = n0 + r;
      bf16x8 v = zero8;
      if (n < p.N) v = *reinterpret_cast<const bf16x8*>(p.lora_b + (long long)n * RP + ch * 8);
      *reinterpret_cast<bf16x8*>(Bs + r * 128 + swz(r, ch) * 16) = v;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < RP / 32; ++ks) mma_step(0, ks, false);
  }

  {
    bool any_vt = false;
#pragma unroll
    for (int j = 0; j < NI; ++j) any_vt = any_vt || vt_tile[j];
    igemm_epilogue<BM, BN, MI, NI, VT>(p, acc, smem, any_vt, m0, n0, wm * (BM / WM), wn * (BN / WN), lrow, lq, split, tid);
  }
}

// ---- pipelined kernel: LDS-DMA (buffer_load ... lds) ring, counted vmcnt, one raw barrier per K-tile -------
// The register-staged kernel above exposes a full HBM/L2 round trip per K-tile and spends hundreds of VALU
// instructions per K-tile on gather addresses.  Here every 16-byte chunk of the activation gather and of the
// weight tile goes global -> LDS directly (no VGPR staging) into an S-deep ring, S-1 K-tiles ahead of the
// MFMAs; the only waits in the loop are a COUNTED s_waitcnt vmcnt((S-2)*L) and one s_barrier.
//   * Requires Cin % 64 == 0 (and Cin2 % 64 == 0): a K-tile then lies inside ONE filter tap of ONE source, so
//     the tap / source / channel cursor is SCALAR.  Per-lane byte offsets (pixel * C + chunk) are recomputed
//     only when the tap or source changes; the channel advance inside a tap rides in the instruction's
//     scalar offset, so steady-state K-tiles cost one buffer_load per 16-byte chunk and nothing else.
//   * Zero padding uses the buffer descriptor's range check: padded taps get voffset 0x80000000 and the DMA
//     writes zeros.  Rows past M / N are clamped (their results are never stored).
//   * The LDS image is the same XOR-swizzled [row][64] layout: the DMA writes lane-linear, so the swizzle is
//     applied to the per-lane SOURCE chunk instead (both-sides-or-neither).
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;

#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource builtins exist in the device pass only
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
#endif

template <int BM, int BN, int WM, int WN, int RP, bool VT, int S, int EPI = 0>
__global__ __launch_bounds__(64 * WM * WN) void igemm_pipe_kernel(const IgemmDev p) {
  constexpr bool LEAN = EPI == 1 || EPI == 4;
#if defined(__HIP_DEVICE_COMPILE__)
  aldm_touch_kernargs<sizeof(IgemmDev)>();
#ifdef ALDM_DIAG
  unsigned long long dg_t_entry; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_t_entry) :: "memory");
#endif
  static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
  constexpr int NT = 64 * WM * WN;               // threads per workgroup
  constexpr int RPP = NT / 8;                    // tile rows covered by one DMA pass of the whole workgroup
  constexpr int MI = BM / WM / 16, NI = BN / WN / 16;
  constexpr int BROWS = BN + RP;
  static_assert(BM % RPP == 0 && BN % RPP == 0 && BROWS % RPP == 0, "tile rows must be whole DMA passes");
  constexpr int A_PASSES = BM / RPP, W_PASSES = BN / RPP, B_PASSES = BROWS / RPP;
  constexpr int L = A_PASSES + B_PASSES;   // LDS-DMA instructions per thread per K-tile
  constexpr int D = S - 1;                 // K-tiles in flight ahead of the MFMAs
#ifdef ALDM_NO_STAGGER
  constexpr bool STAGGER = false;
#else
  constexpr bool STAGGER = WM * WN == 8 && S >= 3;   // see the main loop
#endif
  constexpr int STAGE = (BM + BROWS) * 128;
  constexpr int RT_W = RP / 16 / WN;
  constexpr unsigned OOB = 0x80000000u;
  static_assert((D - 1) * L < 64, "vmcnt immediate");
  constexpr int LDS_LOOP = S * STAGE + (RP > 0 ? BN * 128 : 0);
  constexpr int LDS_MAIN = LDS_LOOP > EpiCfg<BM, BN>::BYTES ? LDS_LOOP : EpiCfg<BM, BN>::BYTES;   // stats live past it

  extern __shared__ __attribute__((aligned(16))) char smem[];   // [S][ A: BM x 128 B | B: BROWS x 128 B ]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  int tile_m, tile_n, split;                    // XCD-aware work-item order (igemm_work_item)
  igemm_work_item(p, tile_m, tile_n, split);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kt0 = split * p.kt_per_split;
  const int kt1 = min(p.nkt, kt0 + p.kt_per_split);

  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t rs_x2 = make_rsrc(p.x2 ? (const void*)p.x2 : (const void*)p.x, p.x2 ? p.x2_bytes : p.x_bytes);
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);
  const __amdgpu_buffer_rsrc_t rs_la = make_rsrc(RP > 0 ? (const void*)p.lora_a : (const void*)p.w, RP > 0 ? p.la_bytes : p.w_bytes);

  // this lane DMA-writes physical chunk (tid & 7) of row (tid >> 3) + 32*pass; the logical (source) chunk is
  // the swizzle's inverse image (the XOR is an involution and depends on (row >> 1) & 7 only, not on pass)
  const int rbase = tid >> 3;
  const int kchunk = (tid & 7) ^ ((rbase >> 1) & 7);
  int a_pix0[A_PASSES], a_ih0[A_PASSES], a_iw0[A_PASSES], a_m[A_PASSES];
#pragma unroll
  for (int ps = 0; ps < A_PASSES; ++ps) {
    const int m = min(m0 + rbase + RPP * ps, p.M - 1);
    const int b = fdiv(m, p.fd_ohw);
    const int pix = m - b * p.OHW;
    const int oh = fdiv(pix, p.fd_ow), ow = pix - oh * p.OW;
    a_pix0[ps] = b * p.IH * p.IW;
    a_ih0[ps] = oh * p.sh - p.ph;
    a_iw0[ps] = ow * p.sw - p.pw;
    a_m[ps] = m;                                  // the fused 1x1 segment reads output pixel m of x3 | x4
  }
  unsigned b_off[B_PASSES];
#pragma unroll
  for (int ps = 0; ps < B_PASSES; ++ps) {
    const int r = rbase + RPP * ps;
    const int row = ps < W_PASSES ? min(n0 + r, p.N - 1) : r - BN;
    b_off[ps] = (unsigned)row * (unsigned)p.Kpad * 2u + kchunk * 16;
  }
  // scalar K cursor.  Past the last filter tap (s_kh == KH) lies the optional fused 1x1 segment over x3 | x4 (ResnetBlock2D's
  // shortcut riding in conv2's launch): same cursor, C3tot channels, the output pixel itself as the gather position.
  int s_kh, s_kw, s_c0;
  {
    const int k = kt0 * BK, kmain = p.KH * p.KW * p.Ctot;
    if (k >= kmain && p.C3tot > 0) {
      s_kh = p.KH; s_kw = 0; s_c0 = k - kmain;
    } else {
      const int tap = fdiv(k, p.fd_ctot);
      s_c0 = k - tap * p.Ctot;
      s_kh = fdiv(tap, p.fd_kw);
      s_kw = tap - s_kh * p.KW;
    }
  }
  // (scalar copies: a select between two FIELDS of the by-value kernel argument becomes a select between their addresses, which
  //  forces the whole argument struct into scratch memory -- measured: 864 B of scratch per lane and every GEMM 4-6x slower)
  const int k_cin = p.Cin, k_cin2 = p.Cin2, k_ctot = p.Ctot, k_cin3 = p.Cin3, k_cin4 = p.Cin4, k_c3tot = p.C3tot;
  bool s_fresh = true;     // tap or source changed: per-lane offsets must be recomputed
  unsigned cur_off[A_PASSES];
  int a_soff = 0, b_soff = kt0 * BK * 2;
  const int IHv = p.UH > 0 ? p.UH : p.IH, IWv = p.UW > 0 ? p.UW : p.IW;

  auto issue = [&](int kt, int stage) {
    char* sbase = smem + stage * STAGE + wave * 1024;   // + pass * 4096: this wave's 8 rows of the pass
    const bool live = kt < kt1;
    const bool ext = s_kh >= p.KH;                      // (scalar) inside the fused 1x1 segment
    const int cA = ext ? k_cin3 : k_cin;
    const bool src2 = s_c0 >= cA;
    if (live) {
      if (s_fresh) {
        if (ext) {
          const int Cs = src2 ? k_cin4 : k_cin3;
#pragma unroll
          for (int ps = 0; ps < A_PASSES; ++ps) cur_off[ps] = (unsigned)a_m[ps] * (unsigned)(Cs * 2) + kchunk * 16;
        } else {
          const int Cs = src2 ? k_cin2 : k_cin;
          const int dh = s_kh * p.dh, dw = s_kw * p.dw;
#pragma unroll
          for (int ps = 0; ps < A_PASSES; ++ps) {
            int ih = a_ih0[ps] + dh, iw = a_iw0[ps] + dw;
            bool ok = (unsigned)ih < (unsigned)IHv && (unsigned)iw < (unsigned)IWv;
            if (p.dilate == 2) {
              ok = ih >= 0 && iw >= 0 && !((ih | iw) & 1) && (ih >> 1) < p.IH && (iw >> 1) < p.IW;
              ih >>= 1;
              iw >>= 1;
            } else if (p.UH > 0) {
              if (p.UH == 2 * p.IH) ih >>= 1; else ih = (ih * p.IH) / p.UH;
              if (p.UW == 2 * p.IW) iw >>= 1; else iw = (iw * p.IW) / p.UW;
            }
            const unsigned off = (unsigned)(a_pix0[ps] + ih * p.IW + iw) * (unsigned)(Cs * 2) + kchunk * 16;
            cur_off[ps] = ok ? off : OOB;
          }
        }
        s_fresh = false;
      }
      a_soff = (s_c0 - (src2 ? cA : 0)) * 2;
      b_soff = kt * BK * 2;
    }
    if (ext) {   // descriptors built on the spot (4 SGPRs, transient): the kernel already sits at the SGPR budget
      if (src2) {
        const __amdgpu_buffer_rsrc_t rs_x4 = make_rsrc(p.x4, p.x4_bytes);
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x4, (lds_ptr_t)(sbase + ps * (RPP * 128)), 16, cur_off[ps], a_soff, 0, 0);
      } else {
        const __amdgpu_buffer_rsrc_t rs_x3 = make_rsrc(p.x3, p.x3_bytes);
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x3, (lds_ptr_t)(sbase + ps * (RPP * 128)), 16, cur_off[ps], a_soff, 0, 0);
      }
    } else if (src2) {
#pragma unroll
      for (int ps = 0; ps < A_PASSES; ++ps)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x2, (lds_ptr_t)(sbase + ps * (RPP * 128)), 16, cur_off[ps], a_soff, 0, 0);
    } else {
#pragma unroll
      for (int ps = 0; ps < A_PASSES; ++ps)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(sbase + ps * (RPP * 128)), 16, cur_off[ps], a_soff, 0, 0);
    }
#pragma unroll
    for (int ps = 0; ps < B_PASSES; ++ps) {
      if (ps < W_PASSES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(sbase + BM * 128 + ps * (RPP * 128)), 16, b_off[ps], b_soff, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_la, (lds_ptr_t)(sbase + BM * 128 + ps * (RPP * 128)), 16, b_off[ps], b_soff, 0, 0);
    }
    if (live) {   // advance the scalar cursor by one K-tile
      s_c0 += BK;
      if (s_c0 == cA && (ext ? k_cin4 : k_cin2) > 0) s_fresh = true;
      if (s_c0 >= (ext ? k_c3tot : k_ctot)) {
        s_c0 = 0;
        s_fresh = true;
        if (++s_kw == p.KW) { s_kw = 0; ++s_kh; }
      }
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 tacc[MI][RT_W > 0 ? RT_W : 1];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < (RT_W > 0 ? RT_W : 1); ++j) tacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the whole workgroup tile is either stored transposed (V^T columns) or not: vt_col0 % BN == 0 (host-checked),
  // so the MFMA operand order is a per-workgroup constant and the main loop is instantiated once per order
  const bool vt_wg = VT && (n0 >= p.vt_col0);
  bool vt_tile[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) vt_tile[j] = vt_wg;
  const int lrow = lane & 15, lq = lane >> 4;

  auto mma_step = [&](auto vtf, const char* As, const char* Bs, int ks, bool with_t) {
    const int ch = ks * 4 + lq;
    bf16x8 af[MI], wf[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int r = wm * (BM / WM) + i * 16 + lrow;
      af[i] = *reinterpret_cast<const bf16x8*>(As + r * 128 + swz(r, ch) * 16);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int r = wn * (BN / WN) + j * 16 + lrow;
      wf[j] = *reinterpret_cast<const bf16x8*>(Bs + r * 128 + swz(r, ch) * 16);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        if constexpr (decltype(vtf)::value)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], wf[j], acc[i][j], 0, 0, 0);
        else
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
      }
    if (RP > 0 && with_t) {
#pragma unroll
      for (int t = 0; t < RT_W; ++t) {
        const int r = BN + (wn * RT_W + t) * 16 + lrow;
        const bf16x8 lf = *reinterpret_cast<const bf16x8*>(Bs + r * 128 + swz(r, ch) * 16);
#pragma unroll
        for (int i = 0; i < MI; ++i)
          tacc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lf, af[i], tacc[i][t], 0, 0, 0);
      }
    }
  };

  // ---- ring: prologue fills D stages; steady state = wait(tile kt landed) -> barrier -> refill the stage
  //      freed by tile kt-1 -> MFMAs on tile kt.  Tiles past the end are dummy zero-page loads so that the
  //      vmcnt immediate stays constant.
  char* const LBs = smem + S * STAGE;     // [BN][128 B] image of the pre-scaled LoRA-B tile (RP > 0 only)
#ifdef ALDM_DIAG
  unsigned long long dg_t_pro0; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_t_pro0) : "v"(a_pix0[0] + a_ih0[0] + a_iw0[0]), "v"(b_off[0]), "s"(s_kh + s_kw + s_c0) : "memory");
#endif
  if (RP > 0) {
    // issued FIRST: vmcnt retires in order, so the first counted wait of the ring also covers these
    const __amdgpu_buffer_rsrc_t rs_lb = make_rsrc(p.lora_b, p.lb_bytes);
#pragma unroll
    for (int ps = 0; ps < W_PASSES; ++ps) {
      const int row = min(n0 + rbase + RPP * ps, p.N - 1);
      const unsigned off = (kchunk < RP / 8) ? (unsigned)row * (unsigned)(RP * 2) + kchunk * 16 : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_lb, (lds_ptr_t)(LBs + wave * 1024 + ps * (RPP * 128)), 16, off, 0, 0, 0);
    }
  }
#pragma unroll
  for (int s = 0; s < D; ++s) issue(kt0 + s, s);
#ifdef ALDM_DIAG   // diagnostic build only (tools/diag_igemm.py): per-wave cycle split of the main loop
  unsigned long long dg_t_pro1; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_t_pro1) :: "memory");
  unsigned long long dg_wait = 0, dg_bar = 0, dg_issue = 0, dg_mma = 0;
#define ALDM_STAMP(var) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); var = t_; }
#endif
  // LayerNorm folded into the GEMM (p.ln_s != nullptr): y = rstd_m (x_m . W'_n - mean_m s_n) + c_n with W' = W diag(gamma),
  // s_n = sum_k W'_nk, c_n = beta . W_n (+ bias).  The row statistics come from the SAME activation tiles the MFMAs consume:
  // TPR threads per row sum their share of every landed K-tile out of LDS -- no separate LayerNorm launch, no normalised
  // tensor in HBM.
  constexpr int TPR = NT / BM;                       // threads per tile row
  constexpr int CPT = 8 / TPR;                            // 16-byte chunks of a 64-wide K-tile per thread
  const bool lnf = (!LEAN || VT) && p.ln_s != nullptr;   // LEAN non-V^T launches never carry a folded LayerNorm (host-checked)
  const bool ln_hand = lnf && p.ln_parts != nullptr;     // statistics handed over by the producer of the activations (rowstat_out)
  const int ln_row = tid / TPR, ln_c0 = (tid % TPR) * CPT;
  float ln_sum = 0.f, ln_sq = 0.f;
  if (ln_hand && tid < BM) {                             // issued here, consumed after the K loop: the loads fly under the ring
    const float* pp = p.ln_parts + (long long)min(m0 + tid, p.M - 1) * (p.ln_np * 2);
    for (int j = 0; j < p.ln_np; ++j) {
      const float2 v2 = *reinterpret_cast<const float2*>(pp + 2 * j);
      ln_sum += v2.x;
      ln_sq += v2.y;
    }
  }
  auto ln_accum = [&](const char* As) {
#pragma unroll
    for (int cc = 0; cc < CPT; ++cc) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(As + ln_row * 128 + swz(ln_row, ln_c0 + cc) * 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = (float)v[j]; ln_sum += f; ln_sq += f * f; }
    }
  };
  auto main_loop = [&](auto vtf) {
    int st = 0, st_fill = D;
    for (int kt = kt0; kt < kt1; ++kt) {
#ifdef ALDM_DIAG
      unsigned long long t0, t1, t2, t3, t4;
      ALDM_STAMP(t0)
#endif
      wait_vmcnt<(D - 1) * L>();
#ifdef ALDM_DIAG
      ALDM_STAMP(t1)
#endif
      __builtin_amdgcn_s_barrier();
#ifdef ALDM_DIAG
      ALDM_STAMP(t2)
#endif
      // STAGGER (8-wave tiles, ring >= 3): every SIMD holds two waves of this workgroup, and the barrier above puts them in phase -- both
      // stall in the LDS-DMA issue (the CU's vector-memory path accepts ~29 B/clk) and then both want the matrix pipe.  Waves 4-7
      // therefore issue the next K-tile's DMA AFTER their MFMAs: while one wave of a SIMD issues, the other multiplies (guide,
      // "two waves that run the SAME program with one barrier per block: try a stagger", split by wave number >= 4).  The stage
      // being refilled was last read before the barrier, so the later issue is as safe as the earlier one.  Measured in the replayed
      // step: 128x128w8 at M = 32000 27.7 -> 27.2 and 24.9 -> 24.1 us, the small 8-wave tiles +-0.1 us.
      if (!(STAGGER && wave >= 4)) issue(kt + D, st_fill);
#ifdef ALDM_DIAG
      ALDM_STAMP(t3)
#endif
      if (lnf && !ln_hand) ln_accum(smem + st * STAGE);
      mma_step(vtf, smem + st * STAGE, smem + st * STAGE + BM * 128, 0, true);
      mma_step(vtf, smem + st * STAGE, smem + st * STAGE + BM * 128, 1, true);
      if (STAGGER && wave >= 4) issue(kt + D, st_fill);
#ifdef ALDM_DIAG
      asm volatile("s_nop 0" :: "v"(acc[0][0][0]));
      ALDM_STAMP(t4)
      dg_wait += t1 - t0; dg_bar += t2 - t1; dg_issue += t3 - t2; dg_mma += t4 - t3;
#endif
      st = (st + 1 == S) ? 0 : st + 1;
      st_fill = (st_fill + 1 == S) ? 0 : st_fill + 1;
    }
  };
  if (VT && vt_wg) main_loop(std::true_type{}); else main_loop(std::false_type{});
#ifdef ALDM_DIAG
  unsigned long long dg_t_loop_end; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_t_loop_end) :: "memory");
#endif
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  float* const lnst = lnf ? reinterpret_cast<float*>(smem + LDS_MAIN) : nullptr;   // [2][BM]: mean, rstd (past every other image)
  if (lnf) {
    const float invk = 1.f / (float)(p.Ctot * p.KH * p.KW);
    if (ln_hand) {
      if (tid < BM) {
        const float mu = ln_sum * invk;
        lnst[tid] = mu;
        lnst[BM + tid] = rsqrtf(fmaxf(ln_sq * invk - mu * mu, 0.f) + p.ln_eps);
      }
    } else {
#pragma unroll
      for (int o = 1; o < TPR; o <<= 1) { ln_sum += __shfl_xor(ln_sum, o, 64); ln_sq += __shfl_xor(ln_sq, o, 64); }
      if ((tid % TPR) == 0) {
        const float mu = ln_sum * invk;
        lnst[ln_row] = mu;
        lnst[BM + ln_row] = rsqrtf(fmaxf(ln_sq * invk - mu * mu, 0.f) + p.ln_eps);
      }
    }
    __syncthreads();
  }

  // ---- LoRA: T (bf16) -> LDS stage 0, then one more K-step against the pre-scaled B ----
  if (RP > 0) {
    char* As = smem;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int t = 0; t < RT_W; ++t) {
        const int r = wm * (BM / WM) + i * 16 + lrow;
        const int col = (wn * RT_W + t) * 16 + lq * 4;
        if (lnf) {   // T'' = t - mean sA + cA / rstd : the epilogue's rstd (acc - mean s) + c then also fixes the LoRA term
          const float mu = lnst[r], irs = 1.f / lnst[BM + r];
          const f32x4 sa4 = *reinterpret_cast<const f32x4*>(p.ln_sa + col), ca4 = *reinterpret_cast<const f32x4*>(p.ln_ca + col);
#pragma unroll
          for (int e = 0; e < 4; ++e) tacc[i][t][e] = tacc[i][t][e] - mu * sa4[e] + ca4[e] * irs;
        }
        bf16x4 tv = {(bf16)tacc[i][t][0], (bf16)tacc[i][t][1], (bf16)tacc[i][t][2], (bf16)tacc[i][t][3]};
        *reinterpret_cast<bf16x4*>(As + r * 128 + swz(r, col >> 3) * 16 + (col & 7) * 2) = tv;
        if (p.lora_t_out && tile_n == 0 && m0 + r < p.M)
          *reinterpret_cast<bf16x4*>(p.lora_t_out + ((long long)split * p.M + m0 + r) * RP + col) = tv;
      }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < RP / 32; ++ks) {
      if (VT && vt_wg) mma_step(std::true_type{}, smem, LBs, ks, false);
      else mma_step(std::false_type{}, smem, LBs, ks, false);
    }
  }
  igemm_epilogue<BM, BN, MI, NI, VT, NT, EPI>(p, acc, smem, vt_wg, m0, n0, wm * (BM / WM), wn * (BN / WN), lrow, lq, split, tid, lnst);
#ifdef ALDM_DIAG
  if (p.diag && lane == 0) {
    unsigned long long dg_t_end; asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_t_end) :: "memory");
    unsigned long long* o = p.diag + ((long long)blockIdx.x * 4 + wave) * 8;
    o[0] = dg_wait; o[1] = dg_bar; o[2] = dg_issue; o[3] = dg_mma;
    o[4] = dg_t_loop_end - dg_t_entry - (dg_wait + dg_bar + dg_issue + dg_mma);   // prologue (everything before / around the loop)
    o[5] = dg_t_end - dg_t_loop_end;                                               // LoRA tail + epilogue incl. store drain
    o[6] = dg_t_pro0 - dg_t_entry;   // kernel entry -> ready to fill the ring (kernel arguments, descriptors, index arithmetic)
    o[7] = dg_t_pro1 - dg_t_pro0;    // issue of the first D ring stages (+ the LoRA-B tile)
  }
#endif
#endif
}

// ---- split-K reduce + epilogue ----------------------------------------------------------------
static __global__ __launch_bounds__(256) void igemm_reduce_kernel(const IgemmDev p) {
  aldm_touch_kernargs<sizeof(IgemmDev)>();
  const int ncols = p.geglu ? (p.N >> 1) : p.N;
  const int nq = ncols >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)p.M * nq) return;
  const int m = (int)(idx / nq);
  const int n = (int)(idx - (long long)m * nq) * 4;
  if (p.geglu) {
    const int nv = ((n >> 4) << 5) + (n & 15), ng = nv + 16;
    float val[4] = {0, 0, 0, 0}, gate[4] = {0, 0, 0, 0};
    for (int s = 0; s < p.splits; ++s) {
      const float* wsr = p.ws + ((long long)s * p.M + m) * p.N;
      const f32x4 a = *reinterpret_cast<const f32x4*>(wsr + nv);
      const f32x4 g = *reinterpret_cast<const f32x4*>(wsr + ng);
#pragma unroll
      for (int q = 0; q < 4; ++q) { val[q] += a[q]; gate[q] += g[q]; }
    }
    if (p.bias) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { val[q] += p.bias[nv + q]; gate[q] += p.bias[ng + q]; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) val[q] *= gelu_erf_f(gate[q]);
    finish_store4(p, m, n, ncols, val);
  } else {
    // four partials per trip: the loads of a trip are in flight together (a one-load-per-trip loop pays one memory round
    // trip per split); summation stays in split order
    float v[4] = {0, 0, 0, 0};
    const float* w0 = p.ws + (long long)m * p.N + n;
    const long long sstride = (long long)p.M * p.N;
    int s = 0;
    for (; s + 4 <= p.splits; s += 4) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(w0 + (s + 0) * sstride), a1 = *reinterpret_cast<const f32x4*>(w0 + (s + 1) * sstride);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(w0 + (s + 2) * sstride), a3 = *reinterpret_cast<const f32x4*>(w0 + (s + 3) * sstride);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = (((v[q] + a0[q]) + a1[q]) + a2[q]) + a3[q];
    }
    if (s < p.splits) {                        // 1 .. 3 partials left: requested together, added in split order
      const int rem = p.splits - s;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(w0 + s * sstride);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(w0 + (s + (rem > 1 ? 1 : 0)) * sstride);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(w0 + (s + (rem > 2 ? 2 : 0)) * sstride);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] += a0[q];
        if (rem > 1) v[q] += a1[q];
        if (rem > 2) v[q] += a2[q];
      }
    }
    add_bias4(p, m, n, v);
    finish_store4(p, m, n, p.N, v);
  }
}

template <int BM, int BN, int WM, int WN, int RP, bool VT, int S, int EPI = 0>
int launch_cfg(const IgemmDev& d, hipStream_t st) {
  // S == 0: register-staged double buffer (needed when the gather applies an activation); else LDS-DMA ring
  constexpr size_t lds_loop = (S == 0 ? 2 : S) * (size_t)(BM + BN + RP) * 128 + ((S != 0 && RP > 0) ? (size_t)BN * 128 : 0);
  constexpr size_t lds = (lds_loop > (size_t)EpiCfg<BM, BN>::BYTES ? lds_loop : (size_t)EpiCfg<BM, BN>::BYTES) + 2 * BM * sizeof(float);
  static unsigned long long attr_done = 0;   // per-device bit mask (aldm_set_max_lds); one-time, idempotent, races are benign
  void (*kern)(const IgemmDev);
  if constexpr (S == 0) kern = igemm_kernel<BM, BN, WM, WN, RP, VT>;
  else kern = igemm_pipe_kernel<BM, BN, WM, WN, RP, VT, S, EPI>;
  if (int rc = aldm_set_max_lds(reinterpret_cast<const void*>(kern), (int)lds, &attr_done, "igemm")) return rc;
  if (d.qstat && !(EPI == 0 || EPI == 4)) {
    aldm_set_error("igemm: qstat_out reached an epilogue instantiation without the statistics code (EPI %d)", EPI);
    return ALDM_E_UNSUPPORTED;
  }
  if (d.qstat && d.N % BN != 0) {
    aldm_set_error("igemm: qstat_out needs Cout %d to be a multiple of the tile width %d", d.N, BN);
    return ALDM_E_ARG;
  }
  if (d.rowstat && d.N % BN != 0) {
    aldm_set_error("igemm: rowstat_out needs Cout %d to be a multiple of the tile width %d", d.N, BN);
    return ALDM_E_ARG;
  }
  if (VT && d.vt_col0 % BN != 0) {
    aldm_set_error("igemm: vt_col0 %d must be a multiple of the tile width %d", d.vt_col0, BN);
    return ALDM_E_ARG;
  }
  IgemmDev dd = d;
  dd.tiles_n = cdiv(d.N, BN);
  dd.fd_tiles_n = make_fastdiv((unsigned)dd.tiles_n);
  dd.tiles_m = cdiv(d.M, BM);
  dd.fd_tiles_m = make_fastdiv((unsigned)dd.tiles_m);
  dd.fd_splits = make_fastdiv((unsigned)d.splits);
  dd.nwg = dd.tiles_m * dd.tiles_n * d.splits;
  dim3 grid(dd.nwg, 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(S == 0 ? THREADS : 64 * WM * WN), lds, st, dd);
  return aldm_launch_status("igemm");
}

template <int BM, int BN, int WM, int WN, int S>
int launch_rp(const IgemmDev& d, int Rp, bool vt, hipStream_t st) {
  if (vt) {
    if constexpr (S != 0) {
      if (d.splits <= 1 && !d.geglu && d.out_act == ALDM_ACT_NONE && d.post_act == ALDM_ACT_NONE) {   // LEAN V^T: no split / GEGLU / act code
        if (Rp == 0) return launch_cfg<BM, BN, WM, WN, 0, true, S, 1>(d, st);
        if (Rp == 32) return launch_cfg<BM, BN, WM, WN, 32, true, S, 1>(d, st);
        return launch_cfg<BM, BN, WM, WN, 64, true, S, 1>(d, st);
      }
    }
    if (Rp == 0) return launch_cfg<BM, BN, WM, WN, 0, true, S>(d, st);
    if (Rp == 32) return launch_cfg<BM, BN, WM, WN, 32, true, S>(d, st);
    return launch_cfg<BM, BN, WM, WN, 64, true, S>(d, st);
  }
  if constexpr (S != 0) {
    if (d.splits > 1) {                                                   // split-K: only the partial-tile store is compiled in
      if (Rp == 0) return launch_cfg<BM, BN, WM, WN, 0, false, S, 3>(d, st);
      if (Rp == 32) return launch_cfg<BM, BN, WM, WN, 32, false, S, 3>(d, st);
      return launch_cfg<BM, BN, WM, WN, 64, false, S, 3>(d, st);
    }
    if (Rp == 0 && d.geglu && d.splits <= 1 && !d.res && !d.res2 && !d.out_f32 && d.out_act == ALDM_ACT_NONE &&
        d.post_act == ALDM_ACT_NONE && d.alpha == 1.f)
      return launch_cfg<BM, BN, WM, WN, 0, false, S, 2>(d, st);          // GEGLU-only epilogue with a plain bf16 store
    // the common case -- standard epilogue, no activation -- runs the LEAN instantiation (nothing else compiled in)
    if (d.splits <= 1 && !d.geglu && d.out_act == ALDM_ACT_NONE && d.post_act == ALDM_ACT_NONE && !d.ln_s) {
      if (Rp == 0 && d.qstat) return launch_cfg<BM, BN, WM, WN, 0, false, S, 4>(d, st);   // + GroupNorm statistics
      if (Rp == 0) return launch_cfg<BM, BN, WM, WN, 0, false, S, 1>(d, st);
      if (Rp == 32) return launch_cfg<BM, BN, WM, WN, 32, false, S, 1>(d, st);
      return launch_cfg<BM, BN, WM, WN, 64, false, S, 1>(d, st);
    }
  }
  if (Rp == 0) return launch_cfg<BM, BN, WM, WN, 0, false, S>(d, st);
  if (Rp == 32) return launch_cfg<BM, BN, WM, WN, 32, false, S>(d, st);
  return launch_cfg<BM, BN, WM, WN, 64, false, S>(d, st);
}

// S0 / SL = default ring depth without / with LoRA (the LoRA rows + B tile cost LDS; SL is chosen so that the
// workgroups-per-CU count does not drop, which matters more than ring depth for the short-K projection GEMMs).
// `ring` (2..4) overrides it (tuning: tools/bench_igemm.py --ring).
template <int BM, int BN, int WM, int WN, int S0, int SL>
int launch_tile(const IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st) {
  // LDS-DMA ring needs a scalar K cursor (64-channel K-tiles inside one tap of one source), no gather-side
  // activation and < 2 GiB activations (32-bit buffer offsets, 0x80000000 = "padded tap").
  const bool fast = d.in_act == ALDM_ACT_NONE && d.Cin % 64 == 0 && d.Cin2 % 64 == 0 && d.x_bytes < 0x80000000u &&
                    d.x2_bytes < 0x80000000u;
  if (!fast && (d.ln_s || d.x3)) {
    aldm_set_error("igemm: the folded LayerNorm / the fused second-source segment need the LDS-DMA path (Cin %% 64 == 0, no gather activation)");
    return ALDM_E_UNSUPPORTED;
  }
  if (!fast) return launch_rp<BM, BN, WM, WN, 0>(d, Rp, vt, st);
  int S = ring ? ring : (Rp ? SL : S0);
  // never ask for more than the 160 KiB of LDS a workgroup can own (ring + LoRA-B image + row statistics)
  auto lds_need = [&](int s) { return (size_t)s * (BM + BN + Rp) * 128 + (Rp ? (size_t)BN * 128 : 0) + 2 * BM * sizeof(float); };
  while (S > 2 && lds_need(S) > 160 * 1024) --S;
  if (S == 2) return launch_rp<BM, BN, WM, WN, 2>(d, Rp, vt, st);
  if (S == 3) return launch_rp<BM, BN, WM, WN, 3>(d, Rp, vt, st);
  return launch_rp<BM, BN, WM, WN, 4>(d, Rp, vt, st);
}

}  // namespace aldm_igemm_detail

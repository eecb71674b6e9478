/*
 * aldm_hip.h -- C-ABI of the MI355X-native AudioLDM+LoRA hot path (libaldm_hip.so).
 *
 * The reference (2025-comprehensive-design/AudioLDM-with-LoRA) has no FFI of its own: its hot
 * path is reached through the Python objects of diffusers / peft / transformers
 * (SURVEY.md section 8b).  Each entry point below replaces the device arithmetic behind one of
 * those calls; the reference call site it serves is cited per function.  A maintainer binds
 * them with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless said otherwise
 *   - the caller owns every buffer; nothing is allocated, freed or retained here
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*), contain no host
 *     synchronisation and are safe under hipGraph stream capture
 *   - return 0 on success, a negative ALDM_E_* on bad arguments, a positive hipError_t otherwise
 *   - activations are channels-last bf16: images [B][H][W][C], sequences [B][T][C]
 *   - weights are bf16, pre-packed once by the host into [N][Kpad] (K contiguous, Kpad % 64 == 0)
 */
#ifndef ALDM_HIP_H
#define ALDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALDM_OK 0
#define ALDM_E_ARG (-1)      /* inconsistent sizes / null pointer */
#define ALDM_E_ALIGN (-2)    /* channel counts or pointers not 16-byte friendly */
#define ALDM_E_UNSUPPORTED (-3)

enum { ALDM_ACT_NONE = 0, ALDM_ACT_SILU = 1, ALDM_ACT_LRELU = 2, ALDM_ACT_TANH = 3, ALDM_ACT_GELU = 4 /* exact erf GELU */ };
enum { ALDM_OUT_BF16 = 0, ALDM_OUT_F32 = 1 };

const char* aldm_version(void);
const char* aldm_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear on MFMA (bf16 in, fp32 accumulate).
 *   out[row(b,oh,ow)][n] = epilogue( sum_{kh,kw,c} X[b][ih][iw][c] * W[n][(kh*KW+kw)*Ctot + c] )
 *   ih = oh*stride_h + kh*dil_h - pad_h   (after the optional nearest-upsample map, below)
 * Serves: every F.conv2d / F.linear / conv1d / conv_transpose1d phase under
 *   UNet2DConditionModel.forward   [REF script/train/train_audioldm_lora.py:539-546]
 *   AutoencoderKL.decode, SpeechT5HifiGan.forward inside AudioLDMPipeline.__call__
 *                                   [REF script/inference/generate_audio.py:47-52]
 * and, with lora_a/lora_b set, peft lora.Linear.forward  y = Wx + (alpha/r) B(Ax)
 *                                   [REF script/train/train_audioldm_lora.py:378-385]
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  /* input: source 1 (+ optional source 2 = virtual channel concat [x | x2]) */
  const void* x;  const void* x2;
  int B, IH, IW, Cin, Cin2;
  int UH, UW;               /* >0: conv runs on the nearest-upsampled image of this size */
  int in_dilate;            /* 2: conv runs on the zero-dilated image x[2i] (adjoint of a stride-2 conv); else 0 */
  /* filter */
  const void* w;            /* [Cout][Kpad] bf16, K index = (kh*KW+kw)*(Cin+Cin2) + c */
  int Kpad;
  int KH, KW, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w;   /* dil may be negative */
  int OH, OW, Cout;
  int in_act;  float in_slope;      /* activation applied to X while it is gathered (vocoder leaky-relu) */
  /* LayerNorm folded into the GEMM (BasicTransformerBlock.norm1/2/3 feeding to_q/k/v and the GEGLU projection):
       y = rstd_m (x_m . W'_n - mean_m ln_s[n]) + bias[n]     W' = W diag(gamma),  ln_s[n] = sum_k W'[n][k],
       bias[n] = beta . W[n] (+ linear bias) folded by the host; row statistics are computed in-kernel from the raw x tiles.
     With LoRA: ln_sa[j] = sum_k (A diag(gamma))[j][k], ln_ca[j] = beta . A[j].  null = plain GEMM. */
  const float* ln_s; const float* ln_sa; const float* ln_ca; float ln_eps;
  /* LoRA side channel (linear layers): T = X A^T, acc += T B'^T with B' = (alpha/r) B */
  const void* lora_a;       /* [Rp][Kpad] bf16 */
  const void* lora_b;       /* [Cout][Rp] bf16, pre-scaled */
  int Rp;                   /* 0, 32 or 64 */
  void* lora_t_out;         /* optional [M][Rp] bf16 copy of T (training saves it) */
  /* epilogue: v = acc + bias[n] + rowbias[b][n];  GEGLU (optional);  v = out_act(v);
     v = alpha*(v + res) + res2;  out = v;  out2 = post_act(v) (if out2) -- or out = post_act(v) when
     post_act is set without out2 (HiFi-GAN: residual stream + its leaky-relu'd copy in one pass)   */
  const float* bias;        /* [Cout] fp32 or null */
  const float* rowbias;     /* [B][rowbias_ld] fp32 or null (time-embedding projection) */
  int rowbias_ld;
  int geglu;                /* 1: W rows interleaved (16 value | 16 gate); out has Cout/2 columns */
  int out_act;  float out_slope;
  const void* res;  const void* res2;   /* bf16, addressed exactly like out */
  float alpha;
  int post_act;  float post_slope;  void* out2;     /* out2: bf16, addressed exactly like out.  With geglu = 1: out2 receives the
                                                       PRE-activation projection [M][Cout] (packed value | gate columns, bias added)
                                                       that the GEGLU backward needs -- plain GEGLU launches only */
  void* out;  int out_dtype;  int out_ld;           /* columns per output row */
  long long out_batch_stride;                       /* elements between batches */
  int out_pix_stride, out_pix_offset;               /* row = pix*stride + offset (conv_transpose phases) */
  /* columns >= vt_col0 (multiple of 16) are stored transposed: vt[b][n - vt_col0][pix] bf16 */
  void* vt;  int vt_col0;  int vt_ld;  long long vt_batch_stride;
  /* split-K */
  int splits;  float* workspace;                    /* [splits][M][Cout] fp32 when splits > 1 */
  int tile;                 /* 0 = auto, else ALDM_TILE_* */
  int ring;                 /* 0 = auto, else LDS-DMA ring depth 2..4 (tuning) */
  int defer_reduce;         /* split-K only: leave the partial tiles in `workspace` and launch no reduce -- the consumer sums
                               them itself and adds bias / rowbias / res (aldm_groupnorm_partials: ResnetBlock2D convs
                               in front of a GroupNorm) */
  /* LayerNorm statistics handed from one GEMM to the next (BasicTransformerBlock: h -> norm -> projection).  The producer of h
     (bf16 output, standard epilogue, Cout a multiple of its tile width BN) writes, per output row and per N-tile, the sum and
     the sum of squares of the row's bf16-rounded values: rowstat_out [M][Cout / BN][2] fp32.  The consumer (ln_s != NULL)
     takes that table as ln_parts [M][ln_nparts][2] and derives mean / rstd from it instead of accumulating them from its
     activation tiles in the K loop -- the LayerNorm launch disappears and no GEMM pays for the statistics twice. */
  float* rowstat_out;
  const float* ln_parts;  int ln_nparts;
  /* GroupNorm statistics handed to aldm_groupnorm_apply: per M-tile of the launch, per image slot (0: the image of the tile's
     first row, 1: the next image -- an M-tile of the generic kernels may cross one image boundary; the halo tiles are
     image-aligned and use slot 0 only) and per 4-channel quad, the (sum, sum of squares) of the bf16 values stored:
     qstat_out [tiles_m][2][Cout / 4][2] fp32.  Standard bf16 epilogue only, Cout a multiple of the tile width. */
  float* qstat_out;
  /* A 1x1 convolution over a SECOND pair of sources, accumulated into the same output tile: K grows by Cin3 + Cin4 columns
     appended to every weight row (w [Cout][KH*KW*(Cin+Cin2) | Cin3+Cin4]).  x3 (| x4) are bf16 [B][OH][OW][Cin3 (Cin4)], i.e.
     they have the OUTPUT's spatial extent.  This is ResnetBlock2D's `conv_shortcut(input) + conv2(h)` as ONE launch
     (diffusers resnet.py under [REF script/train/train_audioldm_lora.py:539-546]); LDS-DMA path only (all channel counts
     multiples of 64, KH*KW*(Cin+Cin2) a multiple of 64), not with the halo tiles. */
  const void* x3;  const void* x4;  int Cin3, Cin4;
  /* with vt: the transposed columns are ALSO stored row-major in `out` (out_ld counts every column then).  The LoRA trainer's
     fused QKV GEMM needs q | k | v row-major (backward operands) and token-major (flash kernels' operands): one launch instead
     of a GEMM plus a transpose.  Standard epilogue only: no residual / activation / out2 on such a launch. */
  int vt_dual;
  /* GroupNorm (+ activation) of the INPUT applied inside the launch (halo tiles only): x (| x2) hold the raw output of the producing
     convolution(s), gnin_q1 (| gnin_q2) their qstat_out tables (gnin_bm / gnin_tpi as aldm_groupnorm_apply's bm / tpi); the halo tile is
     normalised + activated once on its way into LDS, so ResnetBlock2D's norm1 / norm2 need no launch and no round trip of the
     normalised tensor through HBM.  gnin_gamma / gnin_beta have Cin + Cin2 entries.  NULL gnin_gamma = off. */
  const float* gnin_gamma; const float* gnin_beta; const float* gnin_q1; const float* gnin_q2;
  int gnin_bm1, gnin_tpi1, gnin_bm2, gnin_tpi2, gnin_groups, gnin_act; float gnin_eps;
  /* Which operand each of the 8 XCDs' (non-coherent, 4 MiB) L2s keeps.  The workgroups of a launch are dealt round-robin over the XCDs;
     the kernel hands each XCD a contiguous range of (M-tile, N-tile, K-split) work items.  1 = activation-stationary (an XCD owns a band
     of output rows: right when activations >> weights), 2 = weight-stationary (an XCD owns a (K-slice, N-tile) range of the weights for
     every row: right at the UNet's 252- / 64-token levels, where every L2 would otherwise stream the whole weight matrix),
     0 = auto: weight-stationary iff the launch is not split-K and the weight matrix is larger than the activation image (measured:
     csrc/igemm.hip).  Speed only; results are identical. */
  int xcd_map;
} aldm_igemm_t;

enum { ALDM_TILE_AUTO = 0, ALDM_TILE_128x128 = 1, ALDM_TILE_64x64 = 2, ALDM_TILE_128x64 = 3, ALDM_TILE_64x128 = 4,
       ALDM_TILE_32x64 = 5, ALDM_TILE_128x128_W8 = 6 /* 8-wave workgroup */,
       /* 3x3 / stride-1 / pad-1 convs only: the workgroup keeps the input halo of BM/OW image rows in LDS and reads the
          nine taps from it (csrc/igemm_halo.hip); needs Cin % 64 == 0, BM % OW == 0, no LoRA / V^T; split-K by 64-channel chunk */
       ALDM_TILE_HALO_128x128 = 7, ALDM_TILE_HALO_64x128 = 8,
       ALDM_TILE_HALO_128x128_WS = 15, ALDM_TILE_HALO_64x128_WS = 16 /* the halo tiles with 8 compute + 4 loader waves (the loader waves also
          apply gnin_*).  Beyond 3x3: any odd KH x KW filter with dilation and "same" padding at stride 1, nearest up-sampling to any size,
          x3 | x4 (conv2 + conv_shortcut as one launch; ring 3, halo of at most three DMA passes), and conv1d (KH = 1, IH = OH = 1: read as a
          KW x 1 filter over an IW x 1 image -- SpeechT5HifiGan's dilated residual-block convolutions) */,
       ALDM_TILE_256x128_W8 = 9 /* 8-wave workgroup, 64x64 per wave: big-M plain convolutions (VAE, vocoder) */,
       ALDM_TILE_256x128_WS = 12 /* wave-specialised: 8 compute waves + 4 loader waves that only feed the LDS-DMA ring (csrc/igemm_ws.hip);
          plain big-M convolutions: LDS-DMA path, no LoRA / V^T / folded LayerNorm / fused 1x1 segment / GEGLU */,
       ALDM_TILE_64x128_WS = 13, ALDM_TILE_128x64_WS = 14 /* the same split on the small tiles: 4 compute + 4 loader waves (the split-K
          convolutions of the UNet's low-resolution levels; also with the fused 1x1 second-source segment) */,
       ALDM_TILE_64x128_W8 = 10, ALDM_TILE_128x64_W8 = 11 /* 8-wave forms of the small tiles: two waves per SIMD where the grid is ~one
          workgroup per CU (split-K convolutions of the low-resolution levels); LDS-DMA path, no LoRA / V^T */ };

int aldm_igemm(const aldm_igemm_t* p, void* stream);
size_t aldm_igemm_workspace_bytes(const aldm_igemm_t* p);
/* the split count the launch will really use (splits is clamped so that every split gets whole 64-wide K-tiles) */
int aldm_igemm_effective_splits(const aldm_igemm_t* p);

/* ------------------------------------------------------------------------------------------
 * Projection GEMM of the transformer blocks: out[M][N] = epilogue(x[M][K] w[N][K]^T), K = 256 / 384 / 640.
 * The linear layers of BasicTransformerBlock / Transformer2DModel under UNet2DConditionModel.forward
 * [REF script/train/train_audioldm_lora.py:539-546] (diffusers Attention.to_q / to_k / to_v / to_out.0, GEGLU.proj,
 * Transformer2DModel.proj_in) with peft lora.Linear fused [REF script/train/train_audioldm_lora.py:378-385].  Same operand
 * packing and the same arithmetic as aldm_igemm on a 1x1 filter; a kernel built for short K (csrc/pgemm.hip): x fragments
 * stay in registers for the whole K, weight tiles stream through an LDS-DMA ring, accumulators are stored straight from
 * registers.  Epilogues: bias; folded LayerNorm (ln_s + ln_parts, as aldm_igemm_t); LoRA (combined rank <= 32); residual
 * `res` (addressed like out); columns >= vt_col0 stored token-major (vt); GEGLU (w rows in (16 value | 16 gate) blocks);
 * rowstat_out [M][N / (nt * tiles_per_range)][2] = per-row (sum, sum of squares) of the stored values per column range.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* x;            /* [M][K] bf16 */
  const void* w;            /* [N][K] bf16 */
  int M, N, K;
  const float* bias;        /* [N] fp32 or null (with ln_s: c_n = W beta + bias) */
  const float* ln_s; const float* ln_sa; const float* ln_ca; float ln_eps;
  const float* ln_parts; int ln_nparts;     /* [M][ln_nparts][2]: required with ln_s */
  const void* lora_a;       /* [Rp][K] bf16 */
  const void* lora_b;       /* [N][Rp] bf16, pre-scaled */
  int Rp, ranks_used;       /* Rp 0 / 32 / 64; rows of lora_a that are not padding (<= 32) */
  int geglu;                /* out has N / 2 columns */
  const void* res;          /* bf16 [M][out_ld] or null */
  void* out; int out_ld;    /* bf16; out_ld % 8 == 0 */
  void* vt; int vt_col0; int vt_ld; long long vt_batch_stride; int OHW;   /* vt[b][n - vt_col0][pix], row m = b * OHW + pix */
  float* rowstat_out;
  /* launch shape: a workgroup of `waves` (4 / 8) wave64s owns 16 * mi * waves rows and tiles_per_range N-tiles of nt columns.
     0 = let aldm_pgemm_plan choose. */
  int mi, nt, tiles_per_range;
  int max_ranges;           /* plan only: at most this many column ranges per row (= partials per row of rowstat_out); 0 = any */
  int waves;
  void* lora_t_out;         /* optional [M][Rp] bf16 copy of T = x A^T (the LoRA trainer saves it for dB = s dY^T T), as aldm_igemm_t */
  int vt_dual;              /* columns >= vt_col0 (which may then be 0) are stored token-major in vt AND row-major in out (as aldm_igemm_t) */
} aldm_pgemm_t;

int aldm_pgemm_supported(int K);
/* fills mi / nt / tiles_per_range where they are 0 (the caller sizes rowstat_out from them); returns 0 or ALDM_E_* */
int aldm_pgemm_plan(aldm_pgemm_t* p);
int aldm_pgemm(const aldm_pgemm_t* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU) over channels-last x [B][HW][C1] (optionally the virtual channel
 * concat [x | x2], C = C1 + C2) -> y [B][HW][C] bf16.  One workgroup per (batch, group); stats in fp32.
 * F.group_norm under ResnetBlock2D.norm1/2, Transformer2DModel.norm, conv_norm_out, VAE group_norm
 * [REF script/train/train_audioldm_lora.py:539-546] (inside UNet2DConditionModel.forward)
 * ------------------------------------------------------------------------------------------ */
int aldm_groupnorm(const void* x, const void* x2, int B, int HW, int C1, int C2, int groups, float eps,
                   const float* gamma, const float* beta, int act, void* y, void* stream);

/* GroupNorm (+SiLU) over cat[x, x2] in ONE coalesced pass, with the statistics handed over by the aldm_igemm launches that
   produced x (and x2): qstat / qstat2 are their qstat_out tables; bm = rows per M-tile of the producing launch (generic
   tiles, bm <= HW), or tpi > 0 = tiles per image of an image-aligned (halo) launch.  Same arithmetic as aldm_groupnorm
   (F.group_norm under ResnetBlock2D.norm1/2, conv_norm_out, the VAE's norms), variance as E[x^2] - mean^2 in fp32.
   stat_ws (fp32 [B][64][2], may be NULL): scratch for the (mean, rstd) table; with it, images of >= 96 producer tiles (the VAE's
   65536-pixel mel images) get their statistics summed ONCE by a one-workgroup-per-image launch in front of the apply pass. */
int aldm_groupnorm_apply(const void* x, const float* qstat, int bm, int tpi, const void* x2, const float* qstat2, int bm2, int tpi2,
                         int B, int HW, int C1, int C2, int groups, float eps, const float* gamma, const float* beta, int act,
                         void* y, float* stat_ws, void* stream);

/* conv_norm_out -> SiLU -> conv_out of UNet2DConditionModel.forward [REF script/train/train_audioldm_lora.py:539-546] as ONE launch:
   GroupNorm (statistics from the producing aldm_igemm's qstat_out table, as aldm_groupnorm_apply) + SiLU applied to each image-row
   strip once on its way into LDS, then the 3x3 / pad 1 convolution to Cout <= 16 channels from there.  x bf16 [B][H][W][C] with
   W = 16 and C = 128 (the AudioLDM latent: 64 mel bins / 4); w bf16 [Cout][w_ld] in (kh, kw, c) order as ops.pack_conv leaves it;
   out fp32 [B][H][W][Cout]. */
int aldm_gn_silu_conv3x3_small(const void* x, const float* qstat, int bm, int tpi, int B, int H, int W, int C, int groups, float eps,
                               const float* gamma, const float* beta, const void* w, int w_ld, const float* bias, int Cout, float* out,
                               void* stream);

/* LayerNorm over the last dim of [M][C] bf16 (BasicTransformerBlock.norm1/2/3). */
int aldm_layernorm(const void* x, int M, int C, const float* gamma, const float* beta, float eps, void* y,
                   void* stream);
/* GroupNorm (+SiLU) whose input is still split-K partial tiles:
     x[m][c] = sum_s ws[s][m][c] + bias[c] + rowbias[b][c] + res[m][c]
   (ws fp32 [splits][B*HW][C], as left by aldm_igemm with defer_reduce; bias / rowbias / res may be NULL).  Fuses the split-K
   reduce of ResnetBlock2D.conv1 (bias + time-embedding projection) with norm2 + SiLU, and that of conv2 (bias + shortcut
   residual `res`, bf16 [B*HW][C]) with the norm of the Transformer2DModel that follows: one launch instead of two.
   sum_out (bf16 [B*HW][C], may be NULL) receives x itself -- the block output the residual stream carries on.
   x2 (bf16 [B*HW][C2], may be NULL with C2 = 0) appends C2 plain channels: the norm then runs over torch.cat([x, x2]) as
   norm1 of an up-block ResnetBlock2D does (gamma / beta / y have C + C2 channels; the group width must divide C). */
int aldm_groupnorm_partials(const float* ws, int splits, int B, int HW, int C, const float* bias, const float* rowbias,
                            int rowbias_ld, const void* res, void* sum_out, const void* x2, int C2, int groups, float eps,
                            const float* gamma, const float* beta, int act, void* y,
                            void* stream);
/* ClapTextEmbeddings: y[b*L+j] = LayerNorm(word[ids[b][j]] + type0 + pos[pid]) as bf16 [B*L][C]; pid counts the non-pad
   tokens up to and including j (offset by pad_idx; pad tokens use pid = pad_idx).  ids int64 on the device, fp32 tables.
   First op of `text_encoder(input_ids, attention_mask)` [REF script/train/train_audioldm_lora.py:513-518]. */
int aldm_embed_layernorm(const long long* ids, int B, int L, int C, const float* word, int vocab, const float* pos,
                         int npos, const float* type0, const float* gamma, const float* beta, float eps, int pad_idx,
                         void* y, void* stream);

/* ------------------------------------------------------------------------------------------
 * Flash-style multi-head self-attention core (F.scaled_dot_product_attention in diffusers
 * AttnProcessor2_0; 32 Attention modules per UNet forward).
 *   q,k: bf16 rows [B*N][ld*], head h at columns h*d .. h*d+d-1 ;  vt: [B][H*d][vt_ld] (token-contiguous V^T)
 *   out[b*N + n][h*d + i] bf16, ld = out_ld.  softmax scale = scale (1/sqrt(d)).
 * ------------------------------------------------------------------------------------------ */
int aldm_attention(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld,
                   long long vt_batch_stride, int B, int N, int H, int d, float scale, void* out, int out_ld,
                   void* stream);

/* Same core for a Q that already carries scale * log2(e): the host folds the softmax scale of
   F.scaled_dot_product_attention (diffusers AttnProcessor2_0 under [REF script/train/train_audioldm_lora.py:539-546]) into the
   to_q weights and LoRA-B rows when it packs them, so the scores leave the MFMA in their final log2 domain. */
int aldm_attention_prescaled(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld,
                             long long vt_batch_stride, int B, int N, int H, int d, void* out, int out_ld, void* stream);

/* Single-head attention with a WIDE head (d = 128 / 256 / 512): AutoencoderKL's mid-block attention (1 head x 512 over
   N = H*W = 2000..4096 tokens) inside vae.encode / vae.decode [REF script/train/train_audioldm_lora.py:370,495-496],
   [REF script/inference/generate_audio.py:47-52].  Flash-style (no N x N score matrix); q pre-scaled as for
   aldm_attention_prescaled; q, k bf16 rows [B*N][ld*]; vt [B][d][vt_ld] token-contiguous, vt_ld % 32 == 0 and ZERO beyond N. */
int aldm_attention_wide(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld, long long vt_batch_stride,
                        int B, int N, int d, void* out, int out_ld, void* stream);

/* The front half of the fused-LoRA attention module as ONE launch, for the UNet's 64-token level (C = 640 = 8 heads x 80,
   N <= 64 tokens per sample): LayerNorm (folded; statistics from the producer's ln_parts, see aldm_igemm_t) -> to_q | to_k | to_v
   with the LoRA side channel -> softmax(Q K^T) V per (sample, head) workgroup; Q | K | V stay in LDS.  Operands exactly as
   aldm_igemm takes them for the folded QKV GEMM: w [3C][Kpad] (q rows pre-scaled by d^-0.5 log2 e), bias = c_n, ln_s, lora_a
   [Rp][Kpad], lora_b [3C][Rp] (pre-scaled), ln_sa / ln_ca [Rp]; ranks_used = rows of lora_a that are not padding.
   out [B*N][C] bf16 = the heads' attention outputs, ready for to_out.  diffusers Attention under
   [REF script/train/train_audioldm_lora.py:539-546] / [REF script/inference/generate_audio.py:47-52]. */
int aldm_attn_block64(const void* x, const float* ln_parts, int ln_nparts, const void* w, int Kpad, const float* bias,
                      const float* ln_s, const void* lora_a, const void* lora_b, int Rp, int ranks_used, const float* ln_sa,
                      const float* ln_ca, float ln_eps, int B, int N, int H, int d, void* out, void* stream);
/* BASELINE config 5 for the same launch: Q / K / V / P enter the two attention products as OCP e4m3 (fp32 accumulation), quantised from the
   bf16-rounded projection outputs exactly as aldm_attention_fp8 quantises them; everything else as aldm_attn_block64. */
int aldm_attn_block64_fp8(const void* x, const float* ln_parts, int ln_nparts, const void* w, int Kpad, const float* bias,
                          const float* ln_s, const void* lora_a, const void* lora_b, int Rp, int ranks_used, const float* ln_sa,
                          const float* ln_ca, float ln_eps, int B, int N, int H, int d, void* out, void* stream);

/* The same launch for the UNet's 252-token level (C = 384 = 8 heads x 48, N <= 256 tokens per sample: 63 x 4 for a 10 s clip, 64 x 4 in
   training): a (sample, head) workgroup of 8 waves keeps its tokens of X in registers, streams the head's 144 weight rows once and runs
   the 256 x 256 attention from LDS -- the projection launch, the attention launch and the HBM round trip of Q | K | V^T between them
   become one launch (csrc/attn_block256.hip).  Operands exactly as aldm_attn_block64.  diffusers Attention under
   [REF script/train/train_audioldm_lora.py:539-546] / [REF script/inference/generate_audio.py:47-52]. */
int aldm_attn_block256(const void* x, const float* ln_parts, int ln_nparts, const void* w, int Kpad, const float* bias,
                       const float* ln_s, const void* lora_a, const void* lora_b, int Rp, int ranks_used, const float* ln_sa,
                       const float* ln_ca, float ln_eps, int B, int N, int H, int d, void* out, void* stream);

/* Same, additionally writing the log2-domain log-sum-exp of the scaled scores, lse [B][H][N] fp32 (training). */
/* Same core with a per-batch-item key count kv_len[B] (int32, device): keys >= kv_len[b] are excluded exactly as an
   additive -inf attention_mask excludes right-padded tokens, and the key loop stops at the last valid tile.  Query
   rows >= kv_len[b] that fall in all-padding workgroups are written as zeros.  Serves ClapTextSelfAttention under
   the padding mask of `text_encoder(input_ids, attention_mask)` [REF script/train/train_audioldm_lora.py:513-518]. */
int aldm_attention_varlen(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld,
                          long long vt_batch_stride, int B, int N, int H, int d, float scale, const int* kv_len,
                          void* out, int out_ld, void* stream);

/* BASELINE config 5: the same core with fp8 (OCP e4m3) Q / K / V / P MFMA operands and fp32 accumulation.  Inputs and output
   stay bf16 (converted while staged); P is scaled by 2^8 internally.  Operand-precision variant, not a faster one. */
int aldm_attention_fp8(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld,
                       long long vt_batch_stride, int B, int N, int H, int d, float scale, void* out, int out_ld,
                       void* stream);

int aldm_attention_lse(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_ld,
                       long long vt_batch_stride, int B, int N, int H, int d, float scale, void* out, int out_ld,
                       float* lse, void* stream);
/* Flash-style backward of the same op: dQ, dK, dV [B*N][ldg] (head h at columns h*d) from q,k,v row-major [B*N][ld],
   token-major copies qT,kT,dOT [B][C][ldt] (aldm_transpose_tokens), dO, O [B*N][ldo] and the forward's lse.
   delta [B][H][N] fp32 is scratch (rowsum(dO*O)).  No atomics: bitwise reproducible. */
int aldm_attention_bwd(const void* q, const void* k, const void* v, int ld, const void* qT, const void* kT,
                       const void* dOT, int ldt, long long qT_batch_stride, long long kT_batch_stride,
                       long long dOT_batch_stride, const void* dO, const void* O, int ldo, const float* lse,
                       float* delta, int B, int N, int H, int d, float scale, void* dq, void* dk, void* dv,
                       int ldg, void* stream);

/* ------------------------------------------------------------------------------------------
 * HiFi-GAN residual-block pair as ONE launch (csrc/hifigan.hip), for the vocoder's 32- / 64-channel stages:
 *     out = post_act( alpha * ( x + conv2(lrelu(conv1(lrelu(x)))) ) + res2 )
 * conv1: K taps, dilation dil, "same" zero padding; conv2: K taps, dilation 1; both C -> C with bias.  x / res2 / out bf16 [B][T][C]
 * (channels-last); w1 / w2 bf16 [C][ld] with K index = tap * C + cin (ops.pack_conv of the Conv1d weight viewed as a 1 x K filter).
 * One (convs1[q], convs2[q]) step of transformers' HifiGanResidualBlock.forward (modeling_speecht5.py:2887-2950); alpha = 1 / 3 and
 * res2 = the running sum fold SpeechT5HifiGan.forward's mean over the three residual blocks into the last pair of each block, post_act
 * = the leaky-relu in front of the next up-sampler (slope 0.1) or of conv_post (slope 0.01).  The vocoder the reference loads at
 * [REF script/train/train_audioldm_lora.py:371] and runs inside AudioLDMPipeline.__call__ [REF script/inference/generate_audio.py:47-52].
 * Supported: C in {32, 64}, K in {3, 7, 11}, dil 1..5 (aldm_hifigan_respair_supported); other stages run on aldm_igemm.
 * ------------------------------------------------------------------------------------------ */
int aldm_hifigan_respair_supported(int C, int K, int dil);
int aldm_hifigan_respair(const void* x, int B, int T, int C, const void* w1, int ld1, const float* b1, int dil, const void* w2, int ld2,
                         const float* b2, int K, float slope, float alpha, const void* res2, int post_act, float post_slope, void* out,
                         void* stream);

/* Conv1d(C -> 1 channel, K taps, "same" zero padding) + optional tanh, fp32 out [B][T]: SpeechT5HifiGan.forward's conv_post + tanh
   (modeling_speecht5.py:3059-3061) as the HBM-bound stencil it is (csrc/hifigan.hip).  x bf16 [B][T][C] channels-last (already
   activated); w bf16 [K * C] with index tap * C + cin (row 0 of ops.pack_conv's matrix); bias fp32 [1] or NULL. */
int aldm_conv1d_to1(const void* x, int B, int T, int C, const void* w, const float* bias, int K, int act, float* out, void* stream);

/* Row softmax of fp32 scores [rows][cols] (ld) -> bf16 probabilities (VAE mid-block attention, N=4000, d=512). */
int aldm_softmax_rows(const float* s, int rows, int cols, int ld_in, float scale, void* p, int ld_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Small fused elementwise kernels.
 * ------------------------------------------------------------------------------------------ */
/* sinusoidal timestep embedding [cos | sin] (diffusers Timesteps, flip_sin_to_cos) -> bf16 [B][dim]
   t: device fp32 [B] or [1] (t_stride 0/1). */
int aldm_timestep_embedding(const float* t, int t_stride, int B, int dim, void* out, void* stream);
/* y = silu(x) bf16 elementwise */
int aldm_silu(const void* x, long long n, void* y, void* stream);
/* NCHW fp32 <-> channels-last boundary conversions (the pipeline's public tensors are NCHW fp32) */
int aldm_nchw_f32_to_nhwc(const float* x, int B, int C, int HW, void* y, int y_is_f32, void* stream);
int aldm_nhwc_to_nchw_f32(const void* x, int x_is_f32, int B, int C, int HW, float* y, void* stream);
/* y_bf16 = (bf16)(x_f32 * mul), n elements */
int aldm_f32_to_bf16(const float* x, long long n, float mul, void* y, void* stream);
/* classifier-free guidance + DDIM step (eta = 0), the AudioLDMPipeline.__call__ loop body
   [REF script/inference/generate_audio.py:47-52], elementwise over channels-last fp32 latents:
     eps = eps_u + g (eps_t - eps_u)                       (cfg != 0; eps holds [uncond | text] halves)
     x0  = (x - sqrt(1-a_t) eps) / sqrt(a_t) ;  x' = sqrt(a_p) x0 + sqrt(1-a_p) eps
   coef: device fp32 table [n_steps][4] = {sqrt(a_t), sqrt(1-a_t), sqrt(a_p), sqrt(1-a_p)}; the row is
   selected ON DEVICE by step_idx[0], so a captured hipGraph replays every step unchanged.
   x (fp32, [B][n]) is updated in place; x_in (bf16, [2B][n] if cfg else [B][n]) receives the next UNet input. */
int aldm_cfg_ddim_step(const float* eps, float* x, int B, long long n_per_sample, int cfg, float guidance,
                       const float* coef, const int* step_idx, void* x_in_bf16, void* stream);
/* aldm_cfg_ddim_step + aldm_gather_row (of the NEXT step's row of `table` [n_steps][row_elems] into rowbias; table may be NULL)
   + aldm_advance_step as ONE launch at the end of a replayed denoise step: the workgroup that finishes last (ticket: one zeroed
   device word, left zero again) moves the counter, so every workgroup of the launch reads the same step index.
   AudioLDMPipeline.__call__'s loop body bookkeeping [REF script/inference/generate_audio.py:47-52]. */
int aldm_ddim_step_fused(const float* eps, float* x, int B, long long n_per_sample, int cfg, float guidance, const float* coef,
                         int* step_idx, void* x_in_bf16, const float* table, long long row_elems, float* rowbias,
                         const float* timesteps, int n_steps, float* t_out, unsigned* ticket, void* stream);
/* measurement aid: keeps `stream` busy for ~us microseconds so that later launches queue up behind it (bench.py) */
int aldm_sleep_us(int us, void* stream);
/* device-side loop counter for graph replay: step_idx[0] = (step_idx[0] + 1) mod n_steps ; t_out[0] = timesteps[step_idx[0]] */
int aldm_advance_step(int* step_idx, const float* timesteps, int n_steps, float* t_out, void* stream);
/* out[0..row_elems) = table[idx[0]][0..row_elems) with a device-side row index: the engine precomputes the time-embedding
   projections (`time_emb_proj` of all 22 resnets) of EVERY DDIM step once per prompt and each replayed step selects its row --
   the five-launch embedding chain of UNet2DConditionModel.forward leaves the per-step graph. */
int aldm_gather_row(const float* table, const int* idx, long long row_elems, float* out, void* stream);

/* noisy[b][i] = coef[b][0]*x[b][i] + coef[b][1]*noise[b][i]  (DDIMScheduler.add_noise,
   [REF script/train/train_audioldm_lora.py:504]); fp32 in, fp32 out; coef fp32 [B][2] = {sqrt(abar_t), sqrt(1-abar_t)} */
int aldm_add_noise(const float* x, const float* noise, const float* coef, int B, long long n_per_sample, float* out,
                   void* stream);
/* the same with the coefficients looked up on the device: alphas_cumprod fp32 [n_train], timesteps int64 [B] (device) --
   `noise_scheduler.add_noise(latents, noise, timesteps)` [REF script/train/train_audioldm_lora.py:503-504] as one launch */
int aldm_add_noise_t(const float* x, const float* noise, const float* alphas_cumprod, const long long* timesteps, int n_train,
                     int B, long long n_per_sample, float* out, void* stream);
/* DiagonalGaussianDistribution.sample(): params fp32 NCHW [B][2C][HW] = (mean | logvar), noise / out fp32 [B][C][HW], chw = C*HW:
   out = mean + exp(0.5 clamp(logvar, -30, 20)) noise  -- `vae.encode(x).latent_dist.sample()` [REF train:495] */
int aldm_gaussian_sample(const float* params, const float* noise, int B, long long chw, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backward kernels of the LoRA fine-tune step [REF script/train/train_audioldm_lora.py:499-565]: the base model
 * is frozen, so only dX flows through GroupNorm / LayerNorm / GEGLU / up-sampling, and only the LoRA matrices get
 * weight gradients (aldm_tn_small).  dX of convolutions / linears is aldm_igemm with transposed weights.
 * ------------------------------------------------------------------------------------------ */
/* dx_acc (dx2_acc): the tensor x (x2) feeds more than one consumer (residual / skip connections); the gradient it already
   received from the others is passed here and ADDED to the result -- the separate element-wise add launch per join
   disappears.  NULL = none.  May alias dx (dx2). */
int aldm_groupnorm_bwd(const void* x, const void* x2, const void* dy, int B, int HW, int C1, int C2, int groups,
                       float eps, const float* gamma, const float* beta, int act, void* dx, void* dx2, const void* dx_acc,
                       const void* dx2_acc, void* stream);
/* the same with dY still as the fp32 split-K partial tiles [dy_splits][B*HW][C1+C2] of the dX convolution that produced it
   (aldm_igemm with defer_reduce, no bias / residual): the reduce launch of that convolution disappears */
int aldm_groupnorm_bwd_partials(const void* x, const void* x2, const float* dy_ws, int dy_splits, int B, int HW, int C1, int C2,
                                int groups, float eps, const float* gamma, const float* beta, int act, void* dx, void* dx2,
                                const void* dx_acc, const void* dx2_acc, void* stream);
int aldm_layernorm_bwd(const void* x, const void* dy, int M, int C, const float* gamma, float eps, void* dx, const void* dx_acc,
                       void* stream);
/* GEGLU on the interleaved (16 value | 16 gate) projection h [M][2I]: out [M][I] = value * gelu(gate), and its backward */
int aldm_geglu_fwd(const void* h, long long M, int I, void* out, void* stream);
int aldm_geglu_bwd(const void* h, const void* dout, long long M, int I, void* dh, void* stream);
int aldm_add_bf16(const void* a, const void* b, long long n, void* c, void* stream);
int aldm_upsample_nearest_bwd(const void* dy, int B, int IH, int IW, int OH, int OW, int C, void* dx, void* stream);
/* out[p][q] += sum_m P[m][p] * Q[m][q]  (P [M][Rp], Q [M][ldq] bf16; Rp = 32 or 64).  Row p is scattered through the
   device table rows_dev[p] = {float* dst; int qlo, qhi, qstride; float scale}: dst[(q-qlo)*qstride] += scale*value for
   qlo <= q < qhi (fp32 atomics into the flat LoRA gradient buffer). */
int aldm_tn_small(const void* P, int Rp, const void* Q, int ldq, int Qc, int M, const void* rows_dev, void* stream);
/* The same products for MANY (P, Q) pairs in one launch (the trainer defers the 2 x 64 LoRA-gradient products of a step to the
   end of the backward pass).  jobs_dev: device array of njobs records
     { const void* P; const void* Q; const void* rows; int M, ldq, Qc, qt, mpb, wg0; }   (48 bytes)
   with qt = ceil(Qc / 128) column blocks, mpb = rows per workgroup (multiple of 32), wg0 = first workgroup id of the job
   (ascending; total_wgs = sum of qt * ceil(M / mpb)); every P has Rp columns; ldq, Qc multiples of 8, 16-byte aligned. */
int aldm_tn_batched(const void* jobs_dev, int njobs, int total_wgs, int Rp, void* stream);
/* jobs_dev[i] = {const float* src; bf16* dst; int rows, cols, src_ld, dst_ld, transpose; float scale}: one launch converts
   the flat fp32 LoRA parameters into every packed bf16 operand of the fused GEMMs. */
int aldm_lora_pack(const void* jobs_dev, int njobs, void* stream);
/* rows [B*N][ld_in] (C columns) -> [B][C][Npad] token-major, zero padded */
int aldm_transpose_tokens(const void* in, int ld_in, int B, int N, int C, int Npad, void* out, void* stream);
/* loss[0] += mean((pred-target)^2) ; dpred = 2 (pred-target)/n * grad_scale (bf16)  (F.mse_loss, [REF train:549]) */
int aldm_mse_grad(const float* pred, const float* target, long long n, float grad_scale, void* dpred, float* loss, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training helpers (configs 3/4): fused AdamW over one flat fp32 LoRA buffer
 * (torch.optim.AdamW, [REF script/train/train_audioldm_lora.py:396-403,563-565]).
 * ------------------------------------------------------------------------------------------ */
int aldm_adamw_flat(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Log-mel front end: the dataloader's mel_spectrogram_train + pad_spec
 * [REF script/data/datasets.py:301-354,385-398] as one kernel.  wav fp32 [B][T] in [-1, 1];
 * reflect padding (n_fft - hop)/2, n_fft-point STFT (n_fft must be 1024) with `window`[n_fft],
 * magnitude, mel_basis fp32 [n_mels][n_fft/2+1] with non-zero bin ranges mel_range int32 [n_mels][2],
 * log(max(., clamp_min)); out fp32 [B][target_frames][n_mels], frames past the clip are zeros.
 * ------------------------------------------------------------------------------------------ */
int aldm_log_mel(const float* wav, int B, int T, int n_fft, int hop, const float* window, const float* mel_basis,
                 const int* mel_range, int n_mels, int target_frames, float clamp_min, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif

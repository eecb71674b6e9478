// C-ABI entry of the implicit-GEMM family; the kernels live in igemm_core.h and are instantiated per tile shape in
// igemm_t*.hip (separate translation units so the build parallelises).
#include "igemm_core.h"

using namespace aldm_igemm_detail;

// one launcher per tile shape, defined in igemm_t*.hip
int aldm_launch_tile_128x128(const IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st);
int aldm_launch_tile_128x64(const IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st);
int aldm_launch_tile_64x128(const IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st);
int aldm_launch_tile_64x64(const IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st);
int aldm_launch_tile_128x128w8(const IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st);
int aldm_launch_tile_256x128w8(const IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st);
int aldm_launch_tile_256x128ws(const IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st);
int aldm_launch_tile_64x128ws(const IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st);
int aldm_launch_tile_128x64ws(const IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st);
int aldm_launch_tile_64x128w8(const IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st);
int aldm_launch_tile_128x64w8(const IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st);
int aldm_launch_halo(const IgemmDev& d, int tile, int ring, hipStream_t st);

static int pick_tile(int M, int N) {
  auto tiles = [&](int bm, int bn) { return (long long)cdiv(M, bm) * cdiv(N, bn); };
  if (N <= 64) return M >= 4096 ? ALDM_TILE_128x64 : ALDM_TILE_64x64;
  if (tiles(128, 128) >= 224) return ALDM_TILE_128x128;
  if (tiles(128, 64) >= 224) return ALDM_TILE_128x64;
  return ALDM_TILE_64x64;
}


extern "C" size_t aldm_igemm_workspace_bytes(const aldm_igemm_t* p) {
  if (!p || p->splits <= 1) return 0;
  const long long M = (long long)p->B * p->OH * p->OW;
  return (size_t)p->splits * M * p->Cout * sizeof(float);
}

extern "C" int aldm_igemm_effective_splits(const aldm_igemm_t* p) {
  if (!p || p->splits <= 1) return 1;
  const int nkt = cdiv(p->KH * p->KW * (p->Cin + p->Cin2), BK) + (p->x3 ? (p->Cin3 + (p->x4 ? p->Cin4 : 0)) / BK : 0);
  if (p->tile == ALDM_TILE_HALO_128x128 || p->tile == ALDM_TILE_HALO_64x128 || p->tile == ALDM_TILE_HALO_128x128_WS ||
      p->tile == ALDM_TILE_HALO_64x128_WS) {   // the halo tiles split by 64-channel chunk (9 taps each)
    const int nch = (p->Cin + p->Cin2) / BK;
    return cdiv(nch, cdiv(nch, p->splits > nch ? nch : p->splits));
  }
  int splits = p->splits > nkt ? nkt : p->splits;
  const int per = cdiv(nkt, splits);
  return cdiv(nkt, per);
}

extern "C" int aldm_igemm(const aldm_igemm_t* p, void* stream) {
  ALDM_CHECK_ARG(p && p->x && p->w && p->out, "igemm: null x/w/out");
  ALDM_CHECK_ARG(p->B > 0 && p->IH > 0 && p->IW > 0 && p->OH > 0 && p->OW > 0 && p->Cout > 0, "igemm: bad dims");
  ALDM_CHECK_ARG(p->Cin > 0 && p->Cin % 8 == 0 && p->Cin2 >= 0 && p->Cin2 % 8 == 0, "igemm: Cin/Cin2 must be multiples of 8 (got %d,%d)", p->Cin, p->Cin2);
  ALDM_CHECK_ARG(p->Cin2 == 0 || p->x2, "igemm: Cin2 without x2");
  ALDM_CHECK_ARG(p->KH > 0 && p->KW > 0, "igemm: bad filter");
  const int Ctot = p->Cin + p->Cin2;
  const int Ktot = p->KH * p->KW * Ctot;
  const int Cext = p->x3 ? p->Cin3 + (p->x4 ? p->Cin4 : 0) : 0;
  ALDM_CHECK_ARG(p->Kpad % BK == 0 && p->Kpad >= Ktot + Cext, "igemm: Kpad %d must be a multiple of 64 and >= %d", p->Kpad, Ktot + Cext);
  ALDM_CHECK_ARG(!p->x3 || (p->Cin3 > 0 && p->Cin3 % 64 == 0 && (!p->x4 || (p->Cin4 > 0 && p->Cin4 % 64 == 0)) && Ktot % 64 == 0 &&
                            p->Cin % 64 == 0 && p->Cin2 % 64 == 0 && p->in_act == ALDM_ACT_NONE && !p->ln_s && !p->vt && !p->geglu &&
                            p->in_dilate == 0 && p->Rp == 0),
                 "igemm: the fused 1x1 second-source segment (x3 | x4) needs the plain LDS-DMA path: every channel count and KH*KW*(Cin+Cin2) a multiple of 64");
  ALDM_CHECK_ARG(p->x3 || !p->x4, "igemm: x4 without x3");
  ALDM_CHECK_ARG(!p->vt_dual || (p->vt && !p->res && !p->res2 && !p->out2 && p->out_act == ALDM_ACT_NONE && p->post_act == ALDM_ACT_NONE &&
                                 p->alpha == 1.f && !p->rowbias && p->out_dtype == ALDM_OUT_BF16),
                 "igemm: vt_dual needs vt and the plain bf16 epilogue (bias only)");
  ALDM_CHECK_ARG(p->Rp == 0 || p->Rp == 32 || p->Rp == 64, "igemm: Rp must be 0/32/64");
  ALDM_CHECK_ARG(p->Rp == 0 || (p->lora_a && p->lora_b), "igemm: Rp without lora_a/lora_b");
  ALDM_CHECK_ARG(!p->geglu || (p->Cout % 32 == 0 && !p->rowbias), "igemm: GEGLU needs Cout %% 32 == 0");
  ALDM_CHECK_ARG(!p->vt || (p->vt_col0 % 16 == 0 && p->splits <= 1 && !p->geglu), "igemm: bad vt config");
  ALDM_CHECK_ARG(!p->vt || p->vt_col0 > 0 || p->out, "igemm: out required");
  ALDM_CHECK_ARG(p->splits <= 1 || (p->workspace && p->Cout % 4 == 0), "igemm: split-K needs workspace and Cout %% 4 == 0");
  ALDM_CHECK_ARG(p->out_ld > 0, "igemm: out_ld");
  ALDM_CHECK_ARG(!p->defer_reduce || (!p->geglu && !p->res2 && !p->out2 && p->out_act == ALDM_ACT_NONE && p->post_act == ALDM_ACT_NONE && p->alpha == 1.f),
                 "igemm: defer_reduce leaves bias / row bias / res to the consumer and supports nothing else in the epilogue");
  ALDM_CHECK_ARG(!(p->geglu && p->out2) || (p->Rp == 0 && p->splits <= 1 && !p->res && !p->res2 && p->out_dtype == ALDM_OUT_BF16 &&
                                            p->out_act == ALDM_ACT_NONE && p->post_act == ALDM_ACT_NONE && p->alpha == 1.f &&
                                            p->in_act == ALDM_ACT_NONE && p->Cin % 64 == 0 && p->Cin2 % 64 == 0 && p->Cout % 32 == 0),
                 "igemm: GEGLU with out2 (pre-activation copy) needs the plain LDS-DMA GEGLU launch (no residual / activation / split-K)");
  ALDM_CHECK_ARG(!p->ln_s || (p->KH == 1 && p->KW == 1 && p->Cin2 == 0 && p->splits <= 1 && (p->Rp == 0 || (p->ln_sa && p->ln_ca))),
                 "igemm: folded LayerNorm needs a 1x1 single-source GEMM without split-K (and ln_sa/ln_ca with LoRA)");
  ALDM_CHECK_ARG(!p->rowstat_out || (!p->vt && p->splits <= 1 && !p->geglu && p->out_dtype == ALDM_OUT_BF16 && p->Cout % 64 == 0 && !p->out2),
                 "igemm: rowstat_out needs the standard bf16 epilogue (no V^T / split-K / GEGLU / out2) and Cout %% 64 == 0");
  ALDM_CHECK_ARG(!p->qstat_out || (!p->vt && p->splits <= 1 && !p->geglu && p->out_dtype == ALDM_OUT_BF16 && p->Cout % 8 == 0 &&
                                   p->out_pix_stride <= 1 && p->Rp == 0 && !p->ln_s && !p->rowstat_out),
                 "igemm: qstat_out needs the standard bf16 epilogue (no V^T / split-K / GEGLU / LoRA / LayerNorm hand-over) and Cout %% 8 == 0");
  ALDM_CHECK_ARG(!p->ln_parts || (p->ln_s && p->ln_nparts > 0 && p->ln_nparts <= 64), "igemm: ln_parts needs ln_s and 1..64 partials per row");
  ALDM_CHECK_ARG(p->ring == 0 || (p->ring >= 2 && p->ring <= 4), "igemm: ring must be 0 (auto) or 2..4");
  ALDM_CHECK_ARG(p->in_dilate == 0 || (p->in_dilate == 2 && p->UH == 0), "igemm: in_dilate must be 0 or 2 (and excludes UH/UW)");

  IgemmDev d;
  d.x = (const bf16*)p->x; d.x2 = (const bf16*)p->x2; d.w = (const bf16*)p->w;
  d.lora_a = (const bf16*)p->lora_a; d.lora_b = (const bf16*)p->lora_b; d.lora_t_out = (bf16*)p->lora_t_out;
  d.bias = p->bias; d.rowbias = p->rowbias; d.res = (const bf16*)p->res; d.res2 = (const bf16*)p->res2;
  d.out = p->out; d.vt = (bf16*)p->vt; d.ws = p->workspace;
  d.B = p->B; d.IH = p->IH; d.IW = p->IW; d.Cin = p->Cin; d.Cin2 = p->Cin2; d.Ctot = Ctot; d.UH = p->UH; d.UW = p->UW; d.dilate = p->in_dilate;
  d.KH = p->KH; d.KW = p->KW; d.sh = p->stride_h; d.sw = p->stride_w; d.ph = p->pad_h; d.pw = p->pad_w;
  d.dh = p->dil_h; d.dw = p->dil_w;
  d.OH = p->OH; d.OW = p->OW; d.OHW = p->OH * p->OW; d.N = p->Cout; d.M = p->B * p->OH * p->OW; d.Kpad = p->Kpad;
  d.in_act = p->in_act; d.in_slope = p->in_slope;
  d.rowbias_ld = p->rowbias_ld; d.geglu = p->geglu; d.out_act = p->out_act; d.out_slope = p->out_slope;
  d.alpha = p->alpha;
  d.post_act = p->post_act; d.post_slope = p->post_slope; d.out2 = (bf16*)p->out2;
  d.out_f32 = p->out_dtype == ALDM_OUT_F32; d.out_ld = p->out_ld;
  d.out_bs = p->out_batch_stride; d.out_ps = p->out_pix_stride > 0 ? p->out_pix_stride : 1; d.out_po = p->out_pix_offset;
  d.vt_col0 = p->vt_col0; d.vt_ld = p->vt_ld; d.vt_bs = p->vt_batch_stride; d.vt_dual = p->vt ? p->vt_dual : 0;
  d.x3 = (const bf16*)p->x3; d.x4 = (const bf16*)p->x4; d.Cin3 = p->x3 ? p->Cin3 : 0; d.Cin4 = (p->x3 && p->x4) ? p->Cin4 : 0;
  d.C3tot = d.Cin3 + d.Cin4;
  d.nkt = cdiv(Ktot, BK) + d.C3tot / BK;
  d.splits = p->splits > 1 ? p->splits : 1;
  if (d.splits > d.nkt) d.splits = d.nkt;
  d.kt_per_split = cdiv(d.nkt, d.splits);
  d.splits = cdiv(d.nkt, d.kt_per_split);
  d.ws_rows = d.M;
  d.tiles_n = 0; d.tiles_m = 0; d.nwg = 0;
  ALDM_CHECK_ARG(p->xcd_map >= 0 && p->xcd_map <= 2, "igemm: xcd_map must be 0 (auto), 1 (activation-stationary) or 2 (weight-stationary)");
  {
    // which operand an XCD's L2 keeps (igemm_work_item).  Measured (round 4, profiles/r04_fetch_xcd_*.json + r04_step_table_xcd_*.txt):
    // weight-stationary more than halves the fabric fetches of the 252- / 64-token levels' launches (21.6 -> 9.8 MiB per launch on the
    // dominant symbol) and the split-K launches among them get ~1 us SLOWER (eight co-scheduled M-tiles of an XCD then ask one L2
    // channel for the same weight lines at the same instant; the re-fetches of the M-major map were Infinity-Cache hits, not HBM
    // traffic); unsplit launches with weights > activations gain (GEGLU at M = 512: 15.8 -> 14.5 us).  So: auto = weight-stationary
    // only for an unsplit launch whose weight matrix is the larger operand.
    const unsigned long long act = 2ull * p->B * p->IH * p->IW * Ctot + 2ull * p->B * p->OH * p->OW * Cext;
    const unsigned long long wgt = 2ull * p->Cout * p->Kpad;
    d.xmap = p->xcd_map ? p->xcd_map - 1 : ((wgt > act && p->splits <= 1) ? 1 : 0);
  }
  {
    const unsigned long long xb = 2ull * p->B * p->IH * p->IW * p->Cin, x2b = 2ull * p->B * p->IH * p->IW * p->Cin2;
    const unsigned long long wb = 2ull * p->Cout * p->Kpad, lb = 2ull * p->Rp * p->Kpad;
    ALDM_CHECK_ARG(wb < 0xFFFFFFFFull, "igemm: weight matrix too large for 32-bit offsets");
    const unsigned long long x3b = 2ull * p->B * p->OH * p->OW * d.Cin3, x4b = 2ull * p->B * p->OH * p->OW * d.Cin4;
    ALDM_CHECK_ARG(x3b < 0x80000000ull && x4b < 0x80000000ull, "igemm: x3 / x4 too large for 32-bit offsets");
    d.x3_bytes = (unsigned)x3b; d.x4_bytes = (unsigned)x4b;
    d.x_bytes = xb < 0xFFFFFFFFull ? (unsigned)xb : 0xFFFFFFFFu;
    d.x2_bytes = x2b < 0xFFFFFFFFull ? (unsigned)x2b : 0xFFFFFFFFu;
    d.w_bytes = (unsigned)wb;
    d.la_bytes = (unsigned)lb;
    d.lb_bytes = (unsigned)(2ull * p->Cout * p->Rp);
    d.fd_ohw = make_fastdiv((unsigned)d.OHW); d.fd_ow = make_fastdiv((unsigned)d.OW); d.fd_halo = make_fastdiv((unsigned)d.OW + 2u);
    d.fd_ctot = make_fastdiv((unsigned)d.Ctot); d.fd_kw = make_fastdiv((unsigned)d.KW);
    d.ln_s = p->ln_s; d.ln_sa = p->ln_sa; d.ln_ca = p->ln_ca; d.ln_eps = p->ln_eps;
    d.rowstat = p->rowstat_out; d.ln_parts = p->ln_parts; d.ln_np = p->ln_nparts;
    d.qstat = p->qstat_out; d.qtile = -1;
    d.gi_gamma = p->gnin_gamma; d.gi_beta = p->gnin_beta; d.gi_q1 = p->gnin_q1; d.gi_q2 = p->gnin_q2;
    d.gi_bm1 = p->gnin_bm1; d.gi_tpi1 = p->gnin_tpi1; d.gi_bm2 = p->gnin_bm2; d.gi_tpi2 = p->gnin_tpi2;
    d.gi_groups = p->gnin_groups; d.gi_act = p->gnin_act; d.gi_eps = p->gnin_eps;
#ifdef ALDM_DIAG
    d.diag = (p->splits <= 1) ? (unsigned long long*)p->workspace : nullptr;   // diagnostic build: workspace doubles as the stamp buffer
#else
    d.diag = nullptr;
#endif
  }
  hipStream_t st = (hipStream_t)stream;

  int tile = p->tile ? p->tile : pick_tile(d.M, d.N);
  if (p->gnin_gamma) {
    ALDM_CHECK_ARG(tile == ALDM_TILE_HALO_128x128 || tile == ALDM_TILE_HALO_64x128 || tile == ALDM_TILE_HALO_128x128_WS || tile == ALDM_TILE_HALO_64x128_WS, "igemm: gnin_* (GroupNorm of the input inside the launch) needs a halo tile");
    const int Ct = p->Cin + p->Cin2;
    ALDM_CHECK_ARG(p->gnin_beta && p->gnin_q1 && (p->Cin2 == 0 || p->gnin_q2) && p->gnin_groups > 0 && p->gnin_groups <= 64 && Ct % p->gnin_groups == 0 &&
                   (Ct / p->gnin_groups) % 4 == 0 && Ct <= 512 && p->UH == 0 &&
                   (p->gnin_tpi1 > 0 || (p->gnin_bm1 > 0 && p->gnin_bm1 <= p->IH * p->IW)) &&
                   (p->Cin2 == 0 || p->gnin_tpi2 > 0 || (p->gnin_bm2 > 0 && p->gnin_bm2 <= p->IH * p->IW)) &&
                   (p->gnin_act == ALDM_ACT_NONE || p->gnin_act == ALDM_ACT_SILU),
                   "igemm: gnin_*: tables for every source, group width a multiple of 4, at most 512 input channels, no up-sampling, act NONE / SILU");
  }
  const bool vt = p->vt != nullptr;
  int rc;
  switch (tile) {
    case ALDM_TILE_128x128: rc = aldm_launch_tile_128x128(d, p->Rp, vt, p->ring, st); break;
    case ALDM_TILE_128x64: rc = aldm_launch_tile_128x64(d, p->Rp, vt, p->ring, st); break;
    case ALDM_TILE_64x128: rc = aldm_launch_tile_64x128(d, p->Rp, vt, p->ring, st); break;
    case ALDM_TILE_64x64: rc = aldm_launch_tile_64x64(d, p->Rp, vt, p->ring, st); break;
    case ALDM_TILE_128x128_W8: rc = aldm_launch_tile_128x128w8(d, p->Rp, vt, p->ring, st); break;
    case ALDM_TILE_256x128_W8: rc = aldm_launch_tile_256x128w8(d, p->Rp, vt, p->ring, st); break;
    case ALDM_TILE_256x128_WS: rc = aldm_launch_tile_256x128ws(d, p->Rp, vt, p->ring, st); break;
    case ALDM_TILE_64x128_WS: rc = aldm_launch_tile_64x128ws(d, p->Rp, vt, p->ring, st); break;
    case ALDM_TILE_128x64_WS: rc = aldm_launch_tile_128x64ws(d, p->Rp, vt, p->ring, st); break;
    case ALDM_TILE_64x128_W8: rc = aldm_launch_tile_64x128w8(d, p->Rp, vt, p->ring, st); break;
    case ALDM_TILE_128x64_W8: rc = aldm_launch_tile_128x64w8(d, p->Rp, vt, p->ring, st); break;
    case ALDM_TILE_HALO_128x128:
    case ALDM_TILE_HALO_64x128:
    case ALDM_TILE_HALO_128x128_WS:
    case ALDM_TILE_HALO_64x128_WS:
      // (qstat_out is fine with the halo tiles: they are image-aligned, slot 0)
      if (p->Rp || vt || p->rowstat_out || (p->x3 && tile != ALDM_TILE_HALO_128x128_WS && tile != ALDM_TILE_HALO_64x128_WS)) {
        aldm_set_error("igemm: the halo tiles take no LoRA / V^T / row statistics (and a second-source segment only in the wave-specialised forms)");
        return ALDM_E_UNSUPPORTED;
      }
      if (d.splits > 1) {   // split-K by whole 64-channel chunks: a chunk's halo serves its nine taps in one workgroup
        const int nch = (d.Cin + d.Cin2) / BK;
        const int cps = cdiv(nch, d.splits > nch ? nch : d.splits);
        d.kt_per_split = d.KH * d.KW * cps;
        d.splits = cdiv(nch, cps);
      }
      rc = aldm_launch_halo(d, tile, p->ring, st);
      break;
    default: aldm_set_error("igemm: unknown tile %d", tile); return ALDM_E_UNSUPPORTED;
  }
  if (rc != ALDM_OK) return rc;
  if (d.splits > 1 && !p->defer_reduce) {
    const int ncols = d.geglu ? d.N / 2 : d.N;
    const long long work = (long long)d.M * (ncols / 4);
    hipLaunchKernelGGL(igemm_reduce_kernel, dim3((unsigned)cdivll(work, 256)), dim3(256), 0, st, d);
    return aldm_launch_status("igemm_reduce");
  }
  return ALDM_OK;
}

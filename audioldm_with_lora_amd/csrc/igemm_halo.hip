// 3x3 / stride-1 / pad-1 convolution with an LDS-resident input HALO tile (gfx950).
//
// Why: the generic implicit-GEMM kernel (igemm_core.h) re-gathers the activation tile once per filter tap, so a 3x3 conv
// pulls every input pixel through the CU's texture path 9 x (N / BN) times.  Measured on MI355X that path -- not the MFMA
// pipe, not HBM -- bounds those kernels (tools/diag_igemm.py: ~55 % of the main loop is LDS-DMA issue; 221 MB of L2->LDS
// traffic for the 9.4 GFLOP level-0 conv).  Here a workgroup owns BM/W whole image rows of ONE image and keeps, per
// 64-channel chunk, the (BM/W + 2) x (W + 2) halo of input pixels in LDS; the nine taps read it at nine row offsets, so
// the activation is fetched ONCE per chunk and only the weight tiles stream (S-deep LDS-DMA ring, counted vmcnt, one raw
// barrier per tap) -- 2.6x fewer bytes through the texture path at level 0.
// Measured (MI355X, in a replayed graph, level-0 conv 128->128 on 8 x 250 x 16): 21.0 us per launch = 2.5 launch floor
// + 3.3 tap-loop skeleton (waits / barriers) + 9.3 DMA + MFMA + 5.8 epilogue; the generic 128x64 tile takes 24 us.  The
// in-context tuner (tools/autotune.py) picks this kernel where it wins (the 256..384-wide level-0 / training convs).
//
//   * 8 wave64s; v_mfma_f32_16x16x32_bf16, issued "swapped" like the generic kernel so that the shared epilogue
//     (bias / time-embedding row bias / residual / activation / bf16 or fp32 store) applies unchanged.
//   * halo image = [halo pixel][64 ch] rows of 128 B with the same XOR swizzle as the generic A tile; zero padding and
//     rows past the image come from the buffer descriptor's range check (voffset 0x80000000 -> the DMA writes zeros).
//   * virtual concat (x | x2) and the folded nearest 2x up-sampling are applied in the halo gather.
//
// Serves ResnetBlock2D conv1 / conv2 and the up-sampler convs of UNet2DConditionModel.forward
// [REF script/train/train_audioldm_lora.py:539-546], their dX in the LoRA trainer, and the same blocks of AutoencoderKL.
#include "igemm_core.h"
#include "gn_stats.h"

namespace aldm_igemm_detail {

// GNIN: GroupNorm (+ SiLU) of the INPUT inside the launch (ResnetBlock2D norm1 / norm2).  The producing convolutions left their
// per-tile (sum, sum of squares) tables next to the raw tensors (qstat_out); the prologue turns them into a per-channel (scale, shift)
// table in LDS, and every halo chunk is normalised + activated IN PLACE once it has landed -- by the thread that DMA'd it, so the
// wait it already does suffices -- before the barrier that releases it to the nine taps.  Padding pixels stay the zeros the DMA's
// range check wrote (the convolution pads the ACTIVATION).  One 8 us groupnorm_apply launch and one write + read of the normalised
// tensor disappear per ResnetBlock2D convolution at the 4000-pixel level.
// Split-K (EPI 3, round 4): by whole 64-channel chunks -- a split owns chunks [split * cps, ..) with all nine taps, so its halo is
// still fetched once; the fp32 partial tiles go to the workspace slabs like every other tile's (igemm_reduce_kernel or the deferred
// GroupNorm sums them).  This is what lets the halo form fill the chip at the 1000- / 252- / 64-token levels, where an unsplit grid
// is 20 - 130 workgroups: per output tile the texture path then carries 2.5x fewer bytes than the generic tiles' nine gathers.
template <int BM, int BN, int WM, int WN, int HP /* halo DMA passes: HP * 64 rows */, int S, int EPI /* 0 full, 1 lean, 3 split-K partial, 4 lean + statistics */, bool GNIN = false>
__global__ __launch_bounds__(512) void igemm_halo_kernel(const IgemmDev p) {
#if defined(__HIP_DEVICE_COMPILE__)
  aldm_touch_kernargs<sizeof(IgemmDev)>();
  static_assert(WM * WN == 8, "8 waves");
  constexpr int NT = 512, RPP = NT / 8;                 // 64 tile rows per whole-workgroup DMA pass
  constexpr int MI = BM / WM / 16, NI = BN / WN / 16;
  constexpr int W_PASSES = BN / RPP;                    // weight-tile DMA instructions per thread per tap
  constexpr int D = S - 1;                              // taps in flight ahead of the MFMAs
  // (measured in the replayed step, round 4: with waves 4-7 issuing behind their MFMAs the level-0 launches took +1.0 .. +4.3 us
  //  (24.0 -> 25.6, 54.9 -> 59.1 us) -- the nine taps of a chunk are short items and the late issue costs the ring its lead; the
  //  generic 8-wave tile gains 2-3 % from the same stagger.  Off here.)
  constexpr bool STAGGER = false;
  constexpr int HALO_BYTES = HP * RPP * 128;
  constexpr int BSTAGE = BN * 128;
  constexpr unsigned OOB = 0x80000000u;
  static_assert(BN % RPP == 0, "weight tile rows must be whole DMA passes");
  static_assert((D - 1) * W_PASSES + HP < 64, "vmcnt immediate");

  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][halo: HP*64 x 128 B] [S][B: BN x 128 B]
  char* const Hs = smem;
  char* const Bring = smem + 2 * HALO_BYTES;
  float* const gtab = reinterpret_cast<float*>(smem + 2 * HALO_BYTES + S * BSTAGE);   // GNIN: [512] scale | [512] shift | [64] mean | [64] rstd

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int W = p.OW, H = p.OH;                         // output == (virtual) input extent for stride 1 / pad 1
  const int HW2 = W + 2;
  const int rows_pt = BM / W;                           // image rows per tile (host guarantees BM % W == 0)
  const int tpi = (H + rows_pt - 1) / rows_pt;          // tiles per image
  int wid = blockIdx.x, split = 0;
  if constexpr (EPI == 3) {                             // (split slowest: the workgroups of one split share its weight columns)
    split = fdiv(wid, p.fd_tiles_m);                    // fd_tiles_m: divisor = tiles per split (tiles_m * tiles_n), see launch_halo_v
    wid -= split * p.tiles_m;
  }
  const int tile_m = fdiv(wid, p.fd_tiles_n), tile_n = wid - tile_m * p.tiles_n;
  const int img = tile_m / tpi, ty0 = (tile_m - img * tpi) * rows_pt;
  const int m0 = img * p.OHW + ty0 * W, n0 = tile_n * BN;
  const int halo_rows = (rows_pt + 2) * HW2;

  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t rs_x2 = make_rsrc(p.x2 ? (const void*)p.x2 : (const void*)p.x, p.x2 ? p.x2_bytes : p.x_bytes);
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);

  const int rbase = tid >> 3;
  const int kchunk = (tid & 7) ^ ((rbase >> 1) & 7);    // logical (source) chunk landing in this lane's physical slot
  // halo gather: pixel index (in source pixels) of this thread's row in each pass, or -1 (zero row)
  int h_pix[HP];
  {
    const FastDiv fd_w2 = {p.fd_halo.mul, p.fd_halo.shift};
    const bool up = p.UH > 0;
#pragma unroll
    for (int ps = 0; ps < HP; ++ps) {
      const int hp = rbase + RPP * ps;
      const int hy = fdiv(hp, fd_w2), hx = hp - hy * HW2;
      const int gy = ty0 - 1 + hy, gx = hx - 1;
      const bool ok = hp < halo_rows && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      // nearest up-sampling folded into the gather: exact 2x by a shift, any other size by the same floor(dst * in / out) the generic
      // kernel applies (F.interpolate(size=(125, 8)) of a 63 x 4 latent); ok == false rows never use sy / sx
      const int sy = !up ? gy : (p.UH == 2 * p.IH ? (gy >> 1) : (gy * p.IH) / p.UH);
      const int sx = !up ? gx : (p.UW == 2 * p.IW ? (gx >> 1) : (gx * p.IW) / p.UW);
      h_pix[ps] = ok ? (img * p.IH + sy) * p.IW + sx : -1;
    }
  }
  unsigned b_off[W_PASSES];
#pragma unroll
  for (int ps = 0; ps < W_PASSES; ++ps) {
    const int row = min(n0 + rbase + RPP * ps, p.N - 1);
    b_off[ps] = (unsigned)row * (unsigned)p.Kpad * 2u + kchunk * 16;
  }

  const int nchunks = p.Ctot >> 6;
  const int c_begin = EPI == 3 ? split * (p.kt_per_split / 9) : 0;
  const int c_end = EPI == 3 ? min(nchunks, c_begin + p.kt_per_split / 9) : nchunks;
  const int nitems = (c_end - c_begin) * 9;
  // item cursor of the ISSUE side (scalar): chunk, tap
  int i_c = c_begin, i_tap = 0;
  auto issue = [&](int item, int stage) {
    const bool live = item < nitems;
    char* bdst = Bring + stage * BSTAGE + wave * 1024;
    if (live && i_tap == 0) {                            // first tap of a chunk: its halo rides along
      const int c0 = i_c << 6;
      const bool src2 = c0 >= p.Cin;
      const int Cs = src2 ? p.Cin2 : p.Cin;
      const int soff = (c0 - (src2 ? p.Cin : 0)) * 2;
      char* hdst = Hs + (i_c & 1) * HALO_BYTES + wave * 1024;
#pragma unroll
      for (int ps = 0; ps < HP; ++ps) {
        const unsigned off = h_pix[ps] >= 0 ? (unsigned)h_pix[ps] * (unsigned)(Cs * 2) + kchunk * 16 : OOB;
        if (src2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x2, (lds_ptr_t)(hdst + ps * (RPP * 128)), 16, off, soff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(hdst + ps * (RPP * 128)), 16, off, soff, 0, 0);
      }
    }
    const int ksoff = live ? (i_tap * p.Ctot + (i_c << 6)) * 2 : 0;
#pragma unroll
    for (int ps = 0; ps < W_PASSES; ++ps)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(bdst + ps * (RPP * 128)), 16, live ? b_off[ps] : OOB, ksoff, 0, 0);
    if (live) {
      if (++i_tap == 9) { i_tap = 0; ++i_c; }
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lrow = lane & 15, lq = lane >> 4;
  // halo row of this lane's pixel in each 16-pixel MFMA sub-tile (tap (0,0) = halo origin, i.e. input pixel (-1,-1))
  int a_hp[MI];
  {
    const FastDiv fd_w = p.fd_ow;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int pl = wm * (BM / WM) + i * 16 + lrow;
      const int ly = fdiv(pl, fd_w), lx = pl - ly * W;
      a_hp[i] = ly * HW2 + lx;
    }
  }

  auto mma_tap = [&](const char* As, const char* Bs, int tap_off) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ks * 4 + lq;
      bf16x8 af[MI], wf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int r = a_hp[i] + tap_off;
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
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
  };

#pragma unroll
  for (int s = 0; s < D; ++s) issue(s, s);
  if constexpr (GNIN) {
    // statistics of this image's groups from the producers' tables, then (scale, shift) per input channel; under the first DMAs
    // (gamma / beta of channel `tid` are requested first: fetched behind the statistics they would be one more dependent round trip)
    const float my_gamma = tid < p.Ctot ? p.gi_gamma[tid] : 0.f, my_beta = tid < p.Ctot ? p.gi_beta[tid] : 0.f;
    const int groups = p.gi_groups, Cg = p.Ctot / groups;
    int lpg = 1;
    while (lpg * 2 * groups <= NT && lpg < 64) lpg *= 2;
    const GnSrc s1{p.x, p.gi_q1, p.Cin, p.gi_bm1, p.gi_tpi1}, s2{p.x2, p.gi_q2, p.Cin2, p.gi_bm2 > 0 ? p.gi_bm2 : 1, p.gi_tpi2};
    const int HWs = p.IH * p.IW;
    const int g = tid / lpg, j = tid - g * lpg;
    float a = 0.f, q2 = 0.f;
    if (g < groups) gn_group_sums(s1, s2, img, HWs, Cg, g, j, lpg, a, q2);
    for (int o = 1; o < lpg; o <<= 1) { a += __shfl_xor(a, o, 64); q2 += __shfl_xor(q2, o, 64); }
    if (g < groups && j == 0) {
      const float n = (float)HWs * (float)Cg;
      const float mu = a / n;
      gtab[1024 + g] = mu;
      gtab[1088 + g] = rsqrtf(fmaxf(q2 / n - mu * mu, 0.f) + p.gi_eps);
    }
    __syncthreads();
    if (tid < p.Ctot) {                                      // (Ctot <= 512 = NT, host-checked)
      const int gg = tid / Cg;
      const float sc = my_gamma * gtab[1088 + gg];
      gtab[tid] = sc;
      gtab[512 + tid] = my_beta - gtab[1024 + gg] * sc;
    }
    __syncthreads();
  }
  {
    int st = 0, st_fill = D;
    int c_c = c_begin, c_tap = 0, c_dy = 0, c_dx = 0;    // compute-side cursor
    for (int item = 0; item < nitems; ++item) {
      // item's data (and its halo, if it opens a chunk) must have landed; the D-1 younger items may still be in flight,
      // and they carry HP more DMA instructions when one of them opens a chunk
      bool halo_young = false;
#pragma unroll
      for (int d = 1; d < D; ++d) {
        const int t = c_tap + d;
        halo_young = halo_young || ((t == 9 || t == 18) && item + d < nitems);
      }
      if (halo_young) wait_vmcnt<(D - 1) * W_PASSES + HP>(); else wait_vmcnt<(D - 1) * W_PASSES>();
      if constexpr (GNIN) {
        if (c_tap == 0) {                                    // (uniform) this item opens a chunk: normalise the chunks this thread DMA'd
          const int cb = (c_c << 6) + kchunk * 8;            // first of the 8 channels in this thread's 16-byte slot
          const f32x4 sc0 = *reinterpret_cast<const f32x4*>(gtab + cb), sc1 = *reinterpret_cast<const f32x4*>(gtab + cb + 4);
          const f32x4 sh0 = *reinterpret_cast<const f32x4*>(gtab + 512 + cb), sh1 = *reinterpret_cast<const f32x4*>(gtab + 512 + cb + 4);
          char* hb = Hs + (c_c & 1) * HALO_BYTES + wave * 1024 + lane * 16;
#pragma unroll
          for (int ps = 0; ps < HP; ++ps) {
            if (h_pix[ps] >= 0) {
              const bf16x8 v = *reinterpret_cast<const bf16x8*>(hb + ps * (RPP * 128));
              bf16x8 o;
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                float t = fmaf((float)v[k], k < 4 ? sc0[k & 3] : sc1[k & 3], k < 4 ? sh0[k & 3] : sh1[k & 3]);
                if (p.gi_act == ALDM_ACT_SILU) t = silu_f(t);
                o[k] = (bf16)t;
              }
              *reinterpret_cast<bf16x8*>(hb + ps * (RPP * 128)) = o;
            }
          }
          __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0): the rewritten chunks are in LDS before the barrier releases them
        }
      }
      __builtin_amdgcn_s_barrier();
      // (STAGGER as in igemm_pipe_kernel: waves 4-7 issue the next item's DMA behind their MFMAs, so that the two waves of a SIMD are
      //  not both in the DMA issue and then both on the matrix pipe)
      if (!(STAGGER && wave >= 4)) issue(item + D, st_fill);
      mma_tap(Hs + (c_c & 1) * HALO_BYTES, Bring + st * BSTAGE, c_dy * HW2 + c_dx);
      if (STAGGER && wave >= 4) issue(item + D, st_fill);
      st = (st + 1 == S) ? 0 : st + 1;
      st_fill = (st_fill + 1 == S) ? 0 : st_fill + 1;
      ++c_tap;
      if (++c_dx == 3) { c_dx = 0; ++c_dy; }
      if (c_tap == 9) { c_tap = 0; c_dy = 0; ++c_c; }
    }
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  // rows of this tile past the bottom of the image alias the next image's first rows: cut M for the epilogue's bound checks
  IgemmDev q = p;
  q.M = min(p.M, min(m0 + BM, (img + 1) * p.OHW));
  q.qtile = tile_m;                                         // GroupNorm hand-over table: one row per (image, tile in image), slot 0 only
  igemm_epilogue<BM, BN, MI, NI, false, NT, EPI>(q, acc, smem, false, m0, n0, wm * (BM / WM), wn * (BN / WN), lrow, lq, split, tid, nullptr);
#endif
}

// Wave-specialised form (round 4): the eight compute waves of the halo tile plus FOUR loader waves that own the LDS-DMA ring -- the
// halo of the next chunk and the nine weight tiles -- and, with GNIN, the in-place normalisation of every halo chunk (the thread that
// DMA'd a piece normalises it once it has landed, before the barrier that releases the chunk).  Per tap the compute waves then go from
// the barrier straight into the fragment reads and MFMAs: no DMA issue stall and no GroupNorm arithmetic on the waves that multiply
// (igemm_ws.hip has the measurements behind the split).  Same LDS images, cursors, split-K rule and epilogue as igemm_halo_kernel.
// EXT: the fused 1x1 second-source segment (conv2 + conv_shortcut of a ResnetBlock2D as ONE GEMM, aldm_igemm x3 | x4): after its 3x3
// chunks a workgroup walks its share of the segment's 64-channel chunks, each ONE item -- the chunk's pixels arrive as a halo image
// of x3 / x4 (same gather, the border rows are the price of one code path) and are read at the centre tap.  Every such item opens a
// halo, and the loaders run two items ahead, so the halo buffers are THREE (u % 3) and the ring depth is fixed at 3.  Split-K: a split
// takes its main chunks as before plus an even share of the segment's chunks.
template <int BM, int BN, int WM, int WN, int HP, int S, int EPI, bool GNIN = false, bool EXT = false>
__global__ __launch_bounds__(768) void igemm_halo_ws_kernel(const IgemmDev p) {
#if defined(__HIP_DEVICE_COMPILE__)
  aldm_touch_kernargs<sizeof(IgemmDev)>();
  static_assert(WM * WN == 8, "8 compute waves");
  constexpr int NTC = 512, NTL = 256, RPP = NTL / 8;    // 32 rows per DMA pass of the four loader waves
  constexpr int MI = BM / WM / 16, NI = BN / WN / 16;
  constexpr int HPL = HP * 64 / RPP;                    // halo DMA passes per loader thread per chunk
  constexpr int W_PASSES = BN / RPP;
  constexpr int D = S - 1;
  constexpr int HALO_BYTES = HP * 64 * 128;
  constexpr int BSTAGE = BN * 128;
  constexpr unsigned OOB = 0x80000000u;
  static_assert((D - 1) * (W_PASSES + HPL) < 64, "vmcnt immediate");
  static_assert(!EXT || (S == 3 && !GNIN), "fused-shortcut form: ring 3, no gnin_*");
  constexpr int NB = EXT ? 3 : 2;                       // halo buffers

  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][halo: HP*64 x 128 B] [S][B: BN x 128 B] [GNIN tables]
  char* const Hs = smem;
  char* const Bring = smem + NB * HALO_BYTES;
  float* const gtab = reinterpret_cast<float*>(smem + NB * HALO_BYTES + S * BSTAGE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= NTC / 64;
  const int W = p.OW, H = p.OH;
  // General filter geometry (round 4, for the vocoder's dilated 1-D convolutions, which arrive re-interpreted as KH x 1 filters over a
  // T x 1 image -- aldm_launch_halo): KH x KW taps with dilation (dh, dw), "same" padding (ph, pw), stride 1.  The halo of a tile of
  // rows_pt image rows is (rows_pt + (KH - 1) dh) x (W + (KW - 1) dw) pixels; tap (kh, kw) reads it at row offset kh dh HW2 + kw dw.
  const int KWt = p.KW, NTAP = p.KH * p.KW;
  const int HW2 = W + (p.KW - 1) * p.dw;
  const int rows_pt = BM / W;
  const int tpi = (H + rows_pt - 1) / rows_pt;
  int wid = blockIdx.x, split = 0;
  if constexpr (EPI == 3) {
    split = fdiv(wid, p.fd_tiles_m);
    wid -= split * p.tiles_m;
  }
  const int tile_m = fdiv(wid, p.fd_tiles_n), tile_n = wid - tile_m * p.tiles_n;
  const int img = tile_m / tpi, ty0 = (tile_m - img * tpi) * rows_pt;
  const int m0 = img * p.OHW + ty0 * W, n0 = tile_n * BN;
  const int halo_rows = (rows_pt + (p.KH - 1) * p.dh) * HW2;
  const int nchunks = p.Ctot >> 6;
  const int cps = EPI == 3 ? fdiv(p.kt_per_split, p.fd_kw) : 0;       // chunks per split; fd_kw here: / (KH KW), set by launch_halo_v
  const int c_begin = EPI == 3 ? split * cps : 0;
  const int c_end = EPI == 3 ? min(nchunks, c_begin + cps) : nchunks;
  // the segment's chunks of this workgroup: [e_begin, e_end) of C3tot / 64
  const int next = EXT ? (p.C3tot >> 6) : 0;
  const int eps = EPI == 3 ? (next + p.splits - 1) / p.splits : next;
  const int e_begin = EPI == 3 ? min(next, split * eps) : 0;
  const int e_end = min(next, e_begin + eps);
  const int nmain = (c_end - c_begin) * NTAP;
  const int nitems = nmain + (e_end - e_begin);

  if constexpr (GNIN) {
    // per-channel (scale, shift) from the producers' statistics tables: the compute waves (idle until the first tile lands) do the
    // arithmetic, every wave keeps the two barriers
    float my_gamma = 0.f, my_beta = 0.f;
    if (tid < p.Ctot) { my_gamma = p.gi_gamma[tid]; my_beta = p.gi_beta[tid]; }
    const int groups = p.gi_groups, Cg = p.Ctot / groups;
    if (tid < NTC) {
      int lpg = 1;
      while (lpg * 2 * groups <= NTC && lpg < 64) lpg *= 2;
      const GnSrc s1{p.x, p.gi_q1, p.Cin, p.gi_bm1, p.gi_tpi1}, s2{p.x2, p.gi_q2, p.Cin2, p.gi_bm2 > 0 ? p.gi_bm2 : 1, p.gi_tpi2};
      const int HWs = p.IH * p.IW;
      const int g = tid / lpg, j = tid - g * lpg;
      float a = 0.f, q2 = 0.f;
      if (g < groups) gn_group_sums(s1, s2, img, HWs, Cg, g, j, lpg, a, q2);
      for (int o = 1; o < lpg; o <<= 1) { a += __shfl_xor(a, o, 64); q2 += __shfl_xor(q2, o, 64); }
      if (g < groups && j == 0) {
        const float n = (float)HWs * (float)Cg;
        const float mu = a / n;
        gtab[1024 + g] = mu;
        gtab[1088 + g] = rsqrtf(fmaxf(q2 / n - mu * mu, 0.f) + p.gi_eps);
      }
    }
    __syncthreads();
    if (tid < p.Ctot) {                                      // (Ctot <= 512, host-checked)
      const int gg = tid / Cg;
      const float sc = my_gamma * gtab[1088 + gg];
      gtab[tid] = sc;
      gtab[512 + tid] = my_beta - gtab[1024 + gg] * sc;
    }
    __syncthreads();
  }

  if (loader) {
    // ============================ loader waves: halo + weight DMA, GroupNorm of the landed halo ============================
    const int lt = tid - NTC, lwave = wave - NTC / 64;
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rs_x2 = make_rsrc(p.x2 ? (const void*)p.x2 : (const void*)p.x, p.x2 ? p.x2_bytes : p.x_bytes);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);
    const int rbase = lt >> 3;
    const int kchunk = (lt & 7) ^ ((rbase >> 1) & 7);
    int h_pix[HPL];
    {
      const FastDiv fd_w2 = {p.fd_halo.mul, p.fd_halo.shift};
      const bool up = p.UH > 0;
#pragma unroll
      for (int ps = 0; ps < HPL; ++ps) {
        const int hp = rbase + RPP * ps;
        const int hy = fdiv(hp, fd_w2), hx = hp - hy * HW2;
        const int gy = ty0 - p.ph + hy, gx = hx - p.pw;
        const bool ok = hp < halo_rows && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        // nearest up-sampling folded into the gather: exact 2x by a shift, any other size by the same floor(dst * in / out) the generic
      // kernel applies (F.interpolate(size=(125, 8)) of a 63 x 4 latent); ok == false rows never use sy / sx
      const int sy = !up ? gy : (p.UH == 2 * p.IH ? (gy >> 1) : (gy * p.IH) / p.UH);
      const int sx = !up ? gx : (p.UW == 2 * p.IW ? (gx >> 1) : (gx * p.IW) / p.UW);
        h_pix[ps] = ok ? (img * p.IH + sy) * p.IW + sx : -1;
      }
    }
    unsigned b_off[W_PASSES];
#pragma unroll
    for (int ps = 0; ps < W_PASSES; ++ps) {
      const int row = min(n0 + rbase + RPP * ps, p.N - 1);
      b_off[ps] = (unsigned)row * (unsigned)p.Kpad * 2u + kchunk * 16;
    }
    int i_c = c_begin, i_tap = 0, i_u = 0;              // chunk, tap, unit counter (a unit = one halo image: 9 taps, or 1 in the segment)
    bool i_ext = EXT && nmain == 0;
    if (i_ext) i_c = e_begin;
    auto issue = [&](int item, int stage) {
      const bool live = item < nitems;
      char* bdst = Bring + stage * BSTAGE + lwave * 1024;
      if (live && i_tap == 0) {
        const int c0 = i_c << 6;
        char* hdst = Hs + (EXT ? i_u % 3 : (i_c & 1)) * HALO_BYTES + lwave * 1024;
        if (EXT && i_ext) {                                  // a chunk of x3 | x4 (output geometry: h_pix is the same pixel index)
          const bool s4 = c0 >= p.Cin3;
          const int Cs = s4 ? p.Cin4 : p.Cin3;
          const int soff = (c0 - (s4 ? p.Cin3 : 0)) * 2;
          const __amdgpu_buffer_rsrc_t rs_e = s4 ? make_rsrc(p.x4, p.x4_bytes) : make_rsrc(p.x3, p.x3_bytes);
#pragma unroll
          for (int ps = 0; ps < HPL; ++ps) {
            const unsigned off = h_pix[ps] >= 0 ? (unsigned)h_pix[ps] * (unsigned)(Cs * 2) + kchunk * 16 : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_e, (lds_ptr_t)(hdst + ps * (RPP * 128)), 16, off, soff, 0, 0);
          }
        } else {
          const bool src2 = c0 >= p.Cin;
          const int Cs = src2 ? p.Cin2 : p.Cin;
          const int soff = (c0 - (src2 ? p.Cin : 0)) * 2;
#pragma unroll
          for (int ps = 0; ps < HPL; ++ps) {
            const unsigned off = h_pix[ps] >= 0 ? (unsigned)h_pix[ps] * (unsigned)(Cs * 2) + kchunk * 16 : OOB;
            if (src2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x2, (lds_ptr_t)(hdst + ps * (RPP * 128)), 16, off, soff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(hdst + ps * (RPP * 128)), 16, off, soff, 0, 0);
          }
        }
      }
      const int ksoff = !live ? 0 : (EXT && i_ext) ? (NTAP * p.Ctot + (i_c << 6)) * 2 : (i_tap * p.Ctot + (i_c << 6)) * 2;
#pragma unroll
      for (int ps = 0; ps < W_PASSES; ++ps)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(bdst + ps * (RPP * 128)), 16, live ? b_off[ps] : OOB, ksoff, 0, 0);
      if (live) {
        if (EXT && i_ext) { ++i_c; ++i_u; }
        else if (++i_tap == NTAP) {
          i_tap = 0; ++i_c; ++i_u;
          if (EXT && i_c == c_end) { i_ext = true; i_c = e_begin; }
        }
      }
    };
#pragma unroll
    for (int s = 0; s < D; ++s) issue(s, s);
    int st_fill = D;
    int c_c = c_begin, c_tap = 0;
    for (int item = 0; item < nitems; ++item) {
      // how many of the D - 1 younger items in flight carry a halo: an item does if it opens a unit (tap 0 of a chunk; every item of the segment)
      int young = 0;
#pragma unroll
      for (int d = 1; d < D; ++d) {
        const int t = c_tap + d, it = item + d;
        const bool opens = it < nitems && (it >= nmain || (it < nmain && (t == NTAP || t == 2 * NTAP)));
        young += opens ? 1 : 0;
      }
      if (young == 0) wait_vmcnt<(D - 1) * W_PASSES>();
      else if (young == 1 || D < 3) wait_vmcnt<(D - 1) * W_PASSES + HPL>();
      else wait_vmcnt<(D - 1) * W_PASSES + (D > 2 ? 2 : 1) * HPL>();
      if constexpr (GNIN) {
        if (c_tap == 0) {                                    // this item opens a chunk: normalise the pieces this thread DMA'd
          const int cb = (c_c << 6) + kchunk * 8;
          const f32x4 sc0 = *reinterpret_cast<const f32x4*>(gtab + cb), sc1 = *reinterpret_cast<const f32x4*>(gtab + cb + 4);
          const f32x4 sh0 = *reinterpret_cast<const f32x4*>(gtab + 512 + cb), sh1 = *reinterpret_cast<const f32x4*>(gtab + 512 + cb + 4);
          char* hb = Hs + (c_c & 1) * HALO_BYTES + lwave * 1024 + lane * 16;
#pragma unroll
          for (int ps = 0; ps < HPL; ++ps) {
            if (h_pix[ps] >= 0) {
              const bf16x8 v = *reinterpret_cast<const bf16x8*>(hb + ps * (RPP * 128));
              bf16x8 o;
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                float t = fmaf((float)v[k], k < 4 ? sc0[k & 3] : sc1[k & 3], k < 4 ? sh0[k & 3] : sh1[k & 3]);
                if (p.gi_act == ALDM_ACT_SILU) t = silu_f(t);
                o[k] = (bf16)t;
              }
              *reinterpret_cast<bf16x8*>(hb + ps * (RPP * 128)) = o;
            }
          }
          __builtin_amdgcn_s_waitcnt(0xC07F);
        }
      }
      __builtin_amdgcn_s_barrier();                          // item handed over; the stage of item - 1 is free
      issue(item + D, st_fill);
      st_fill = (st_fill + 1 == S) ? 0 : st_fill + 1;
      if (item < nmain && ++c_tap == NTAP) { c_tap = 0; ++c_c; }
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    return;                                                  // (an ended wave counts as arrived at every later barrier)
  }

  // ===================================== compute waves =====================================
  const int wm = wave / WN, wn = wave % WN;
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lrow = lane & 15, lq = lane >> 4;
  int a_hp[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int pl = wm * (BM / WM) + i * 16 + lrow;
    const int ly = fdiv(pl, p.fd_ow), lx = pl - ly * W;
    a_hp[i] = ly * HW2 + lx;
  }
  {
    int st = 0, c_c = c_begin, c_tap = 0, c_dy = 0, c_dx = 0, c_u = 0;
    for (int item = 0; item < nitems; ++item) {
      __builtin_amdgcn_s_barrier();
      const bool seg = EXT && item >= nmain;                 // (uniform) an item of the 1x1 segment: centre tap of its own halo image
      const char* As = Hs + (EXT ? c_u % 3 : (c_c & 1)) * HALO_BYTES;
      const char* Bs = Bring + st * BSTAGE;
      const int tap_off = seg ? HW2 + 1 : c_dy * p.dh * HW2 + c_dx * p.dw;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ch = ks * 4 + lq;
        bf16x8 af[MI], wf[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int r = a_hp[i] + tap_off;
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
          for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
      }
      st = (st + 1 == S) ? 0 : st + 1;
      if (seg) { ++c_u; continue; }
      ++c_tap;
      if (++c_dx == KWt) { c_dx = 0; ++c_dy; }
      if (c_tap == NTAP) { c_tap = 0; c_dy = 0; ++c_c; ++c_u; }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  IgemmDev q = p;
  q.M = min(p.M, min(m0 + BM, (img + 1) * p.OHW));
  q.qtile = tile_m;
  igemm_epilogue<BM, BN, MI, NI, false, NTC, EPI>(q, acc, smem, false, m0, n0, wm * (BM / WM), wn * (BN / WN), lrow, lq, split, tid, nullptr);
#endif
}

template <int BM, int BN, int WM, int WN, int HP, int S, int EPI, bool GNIN = false, bool WS = false, bool EXT = false>
int launch_halo_v(const IgemmDev& d, hipStream_t st) {
  static_assert(!EXT || WS, "the fused-shortcut form exists for the wave-specialised kernel only");
  constexpr size_t lds_loop = (EXT ? 3 : 2) * (size_t)HP * 64 * 128 + (size_t)S * BN * 128 + (GNIN ? (1024 + 128) * sizeof(float) : 0);
  constexpr size_t lds = lds_loop > (size_t)EpiCfg<BM, BN>::BYTES ? lds_loop : (size_t)EpiCfg<BM, BN>::BYTES;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static unsigned long long attr_done = 0;   // per-device bit mask (aldm_set_max_lds); one per instantiation (WS included)
  void (*kern)(const IgemmDev);
  if constexpr (EXT) kern = igemm_halo_ws_kernel<BM, BN, WM, WN, HP, S, EPI, false, true>;
  else if constexpr (WS) kern = igemm_halo_ws_kernel<BM, BN, WM, WN, HP, S, EPI, GNIN>;
  else kern = igemm_halo_kernel<BM, BN, WM, WN, HP, S, EPI, GNIN>;
  if (int rc = aldm_set_max_lds(reinterpret_cast<const void*>(kern), (int)lds, &attr_done, "igemm_halo")) return rc;
  const int rows_pt = BM / d.OW;
  const int need = WS ? (rows_pt + (d.KH - 1) * d.dh) * (d.OW + (d.KW - 1) * d.dw) : (rows_pt + 2) * (d.OW + 2);
  if (BM % d.OW != 0 || need > HP * 64) {
    aldm_set_error("igemm_halo: image width %d / filter %dx%d does not fit the %dx%d halo tile (%d halo rows of %d)", d.OW, d.KH, d.KW, BM, BN, need, HP * 64);
    return ALDM_E_UNSUPPORTED;
  }
  IgemmDev dd = d;
  if (WS) dd.fd_kw = make_fastdiv((unsigned)(d.KH * d.KW));    // the wave-specialised kernel: kt_per_split / taps = chunks per split
  dd.tiles_n = cdiv(d.N, BN);
  dd.fd_tiles_n = make_fastdiv((unsigned)dd.tiles_n);
  dd.fd_halo = make_fastdiv((unsigned)(WS ? d.OW + (d.KW - 1) * d.dw : d.OW + 2));
  const int tpi = cdiv(d.OH, rows_pt);
  dd.tiles_m = d.B * tpi * dd.tiles_n;                      // (split-K form) tiles per split: the divisor that peels the split off blockIdx.x
  dd.fd_tiles_m = make_fastdiv((unsigned)dd.tiles_m);
  dim3 grid(d.B * tpi * dd.tiles_n * (EPI == 3 ? d.splits : 1), 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(WS ? 768 : 512), lds, st, dd);
  return aldm_launch_status("igemm_halo");
}

template <int BM, int BN, int WM, int WN, int HP, int S>
int launch_halo_ext(const IgemmDev& d, hipStream_t st) {     // conv2 + conv_shortcut as one GEMM (x3 | x4 segment), wave-specialised form
  if (d.gi_gamma) {
    aldm_set_error("igemm_halo: gnin_* launches take no second-source segment");
    return ALDM_E_UNSUPPORTED;
  }
  if (d.splits > 1) return launch_halo_v<BM, BN, WM, WN, HP, S, 3, false, true, true>(d, st);
  if (d.out_act == ALDM_ACT_NONE && d.post_act == ALDM_ACT_NONE)
    return d.qstat ? launch_halo_v<BM, BN, WM, WN, HP, S, 4, false, true, true>(d, st) : launch_halo_v<BM, BN, WM, WN, HP, S, 1, false, true, true>(d, st);
  return launch_halo_v<BM, BN, WM, WN, HP, S, 0, false, true, true>(d, st);
}

template <int BM, int BN, int WM, int WN, int HP, int S, bool WS = false>
int launch_halo(const IgemmDev& d, hipStream_t st) {
  if (d.splits > 1) {
    if (d.gi_gamma) {
      aldm_set_error("igemm_halo: gnin_* launches are not split");
      return ALDM_E_UNSUPPORTED;
    }
    return launch_halo_v<BM, BN, WM, WN, HP, S, 3, false, WS>(d, st);
  }
  if (d.gi_gamma) {                                          // GroupNorm of the input folded in: the ResnetBlock2D convolutions (lean epilogues)
    if (d.out_act != ALDM_ACT_NONE || d.post_act != ALDM_ACT_NONE) {
      aldm_set_error("igemm_halo: gnin_* launches take no output activation");
      return ALDM_E_UNSUPPORTED;
    }
    return d.qstat ? launch_halo_v<BM, BN, WM, WN, HP, S, 4, true, WS>(d, st) : launch_halo_v<BM, BN, WM, WN, HP, S, 1, true, WS>(d, st);
  }
  if (d.out_act == ALDM_ACT_NONE && d.post_act == ALDM_ACT_NONE)
    return d.qstat ? launch_halo_v<BM, BN, WM, WN, HP, S, 4, false, WS>(d, st) : launch_halo_v<BM, BN, WM, WN, HP, S, 1, false, WS>(d, st);
  return launch_halo_v<BM, BN, WM, WN, HP, S, 0, false, WS>(d, st);
}

}  // namespace aldm_igemm_detail

// tile: ALDM_TILE_HALO_128x128 / _64x128 or their wave-specialised forms (_WS).  The caller (aldm_igemm) has validated the generic arguments.
int aldm_launch_halo(const aldm_igemm_detail::IgemmDev& d0, int tile, int ring, hipStream_t st) {
  using namespace aldm_igemm_detail;
  IgemmDev d = d0;
  const bool ws = tile == ALDM_TILE_HALO_128x128_WS || tile == ALDM_TILE_HALO_64x128_WS;
  if (ws && d.KH == 1 && d.OH == 1 && d.IH == 1 && d.UH == 0 && d.ph == 0 && d.dh == 1 && d.sh == 1 && d.C3tot == 0 && !d.gi_gamma) {
    // a 1-D convolution (conv1d = KH 1 over a 1 x T image): the same memory read as a T x 1 image under a KW x 1 filter -- the tile is then
    // BM consecutive time steps, its halo BM + (KW - 1) dw rows, the weight layout [(tap, c)] is unchanged
    d.KH = d.KW; d.KW = 1; d.dh = d.dw; d.dw = 1; d.ph = d.pw; d.pw = 0;
    d.OH = d.OW; d.OW = 1; d.IH = d.IW; d.IW = 1; d.sh = d.sw; d.sw = 1;
    d.fd_ow = make_fastdiv(1u);
    d.fd_ohw = make_fastdiv((unsigned)d.OHW);
  }
  const bool same = ws ? (d.KH % 2 == 1 && d.KW % 2 == 1 && 2 * d.ph == (d.KH - 1) * d.dh && 2 * d.pw == (d.KW - 1) * d.dw &&
                          ((d.KH == 3 && d.KW == 3 && d.dh == 1 && d.dw == 1) || (d.C3tot == 0 && !d.gi_gamma && d.UH == 0)))
                       : (d.KH == 3 && d.KW == 3 && d.ph == 1 && d.pw == 1 && d.dh == 1 && d.dw == 1);
  const bool ok = same && d.sh == 1 && d.sw == 1 &&
                  d.in_act == ALDM_ACT_NONE && d.Cin % 64 == 0 && d.Cin2 % 64 == 0 && d.dilate == 0 &&
                  !d.ln_s && !d.geglu && d.x_bytes < 0x80000000u && d.x2_bytes < 0x80000000u &&
                  ((d.UH == 0 && d.UW == 0) || (d.UH > 0 && d.UW > 0)) && d.OH == (d.UH ? d.UH : d.IH) && d.OW == (d.UW ? d.UW : d.IW);
  if (!ok) {
    aldm_set_error("igemm_halo: needs a stride-1 'same' convolution on the LDS-DMA path (Cin %% 64 == 0, no gather activation): 3x3, or -- wave-specialised tiles -- any odd filter with dilation");
    return ALDM_E_UNSUPPORTED;
  }
  if (d.C3tot > 0) {                                          // the fused 1x1 segment: wave-specialised tiles, three-pass halo at most
    const bool fits = d.UH == 0 && d.C3tot % 64 == 0 && d.Cin3 % 64 == 0 && d.x3_bytes < 0x80000000u && d.x4_bytes < 0x80000000u;
    if (fits && tile == ALDM_TILE_HALO_128x128_WS && (128 / d.OW + 2) * (d.OW + 2) <= 192) return launch_halo_ext<128, 128, 4, 2, 3, 3>(d, st);
    if (fits && tile == ALDM_TILE_HALO_64x128_WS) return launch_halo_ext<64, 128, 2, 4, 2, 3>(d, st);
    aldm_set_error("igemm_halo: the second-source segment needs a wave-specialised halo tile whose halo fits three DMA passes (OW >= 8 at 128 rows)");
    return ALDM_E_UNSUPPORTED;
  }
  if (tile == ALDM_TILE_HALO_128x128_WS) {                  // 8 compute + 4 loader waves; ring 3 (4 where asked)
    if ((128 / d.OW + (d.KH - 1) * d.dh) * (d.OW + (d.KW - 1) * d.dw) > 192) {
      if (ring == 4) return launch_halo<128, 128, 4, 2, 5, 4, true>(d, st);
      return launch_halo<128, 128, 4, 2, 5, 3, true>(d, st);
    }
    if (ring == 4) return launch_halo<128, 128, 4, 2, 3, 4, true>(d, st);
    return launch_halo<128, 128, 4, 2, 3, 3, true>(d, st);
  }
  if (tile == ALDM_TILE_HALO_64x128_WS) {
    if (ring == 4) return launch_halo<64, 128, 2, 4, 2, 4, true>(d, st);
    return launch_halo<64, 128, 2, 4, 2, 3, true>(d, st);
  }
  if (tile == ALDM_TILE_HALO_128x128) {
    // halo rows the tile needs: (128 / OW + 2) image rows of OW + 2 pixels.  192 (three DMA passes) covers OW <= 16 -- the UNet's
    // latents; the VAE's 64-wide images need 264: five passes, 80 KB for the two halo buffers
    if ((128 / d.OW + 2) * (d.OW + 2) > 192) {
      if (ring == 2) return launch_halo<128, 128, 4, 2, 5, 2>(d, st);
      if (ring == 4) return launch_halo<128, 128, 4, 2, 5, 4>(d, st);
      return launch_halo<128, 128, 4, 2, 5, 3>(d, st);
    }
    if (ring == 2) return launch_halo<128, 128, 4, 2, 3, 2>(d, st);
    if (ring == 4) return launch_halo<128, 128, 4, 2, 3, 4>(d, st);
    return launch_halo<128, 128, 4, 2, 3, 3>(d, st);
  }
  if (ring == 2) return launch_halo<64, 128, 2, 4, 2, 2>(d, st);
  if (ring == 4) return launch_halo<64, 128, 2, 4, 2, 4>(d, st);
  return launch_halo<64, 128, 2, 4, 2, 3>(d, st);
}

// Wave-specialised implicit-GEMM tile for the big-M plain convolutions (AutoencoderKL decode / encode, the HiFi-GAN stages with
// C >= 128, the UNet's widest level): 256 x 128 output tile, EIGHT compute waves (64 x 64 each, v_mfma_f32_16x16x32_bf16) plus
// FOUR loader waves that do nothing but feed the LDS-DMA ring.
//
// Why (round 4): on the ordinary tiles every wave issues its share of the K-tile's `buffer_load ... lds` instructions and then
// multiplies.  The CU's vector-memory path accepts ~29 B/clk, so the issue STALLS (50 - 65 % of a wave's cycles in the diagnostic
// build) and the stalled wave cannot issue MFMAs; the barrier of every K-tile keeps the SIMD's waves in phase, so DMA time and
// MFMA time add up (256 x 128 8-wave tile: ~3300 cycles per K-tile for 1650 cycles of DMA and 1024 of MFMA: 0.77 PFLOP/s, half of
// what the tile's 87 flop / byte allows).  Here the loader waves absorb the stalls: per K-tile they wait for their own tile to land,
// join the workgroup barrier and refill the stage the barrier just freed, while the compute waves go from the same barrier straight
// into the MFMAs.  A K-tile then costs max(DMA, MFMA) + the barrier.
//
// Same operand layout, XOR-swizzled LDS image, scalar K cursor, padding by descriptor range and epilogue (igemm_epilogue) as
// igemm_pipe_kernel; plain launches only: LDS-DMA path (Cin % 64 == 0), no LoRA side channel, no V^T store, no folded LayerNorm, no
// fused 1x1 second-source segment.  Serves F.conv2d / conv1d under AutoencoderKL.decode, SpeechT5HifiGan.forward
// [REF script/inference/generate_audio.py:47-52] and vae.encode [REF script/train/train_audioldm_lora.py:495-496].
#include "igemm_core.h"

namespace aldm_igemm_detail {

// The same split pays on the SMALL tiles of the UNet's 252- / 64-token levels (round 4, second half): there the grid is ~one workgroup per
// CU, a K-tile costs a wave ~420 cycles of stalled DMA issue + ~280 of fragment reads and MFMAs + the barrier, and neither a second
// wave per SIMD nor a stagger hid one behind the other -- because every wave still had to issue its share of the DMA.
// With 4 compute waves + NL loader waves the K-tile costs max(DMA, MFMA): instantiated as 64x128 and 128x64 (+ the fused 1x1
// second-source segment those levels' conv2 launches carry).
//
// RS (register-staged loaders, ring = 2): the loader waves fetch with plain buffer_load_dwordx4 into VGPRs and ds_write_b128 the tile
// into the SAME LDS image one barrier later, instead of `buffer_load ... lds` (two tiles in flight in 2 L x 4 VGPRs, two LDS stages).
// An experiment, kept as a selectable form: attn_block64.hip's weight stream runs twice as fast this way, here it does NOT pay --
// 15 - 25 % slower than the LDS-DMA loaders on every shape tried (M 512 N 640 K 5760: 18.5 against 15.8 us; M 8000 N 256 K 2304: 26.1
// against 20.5; M 32000 N 256 K 2304 on 256x128: 54.2 against 43.1; profiles/r04_ws_register_staged_loaders.log): the return trip
// load -> ds_write -> lgkmcnt(0) must fit between two barriers, and the compiler's waits keep ~one tile in flight at issue time.
typedef unsigned int ws_u32x4 __attribute__((ext_vector_type(4)));
template <int BM, int BN, int WM, int WN, int NL, int S, int EPI, bool RS = false>
__global__ __launch_bounds__(64 * (WM * WN + NL)) void igemm_ws_kernel(const IgemmDev p) {
#if defined(__HIP_DEVICE_COMPILE__)
  aldm_touch_kernargs<sizeof(IgemmDev)>();
  constexpr int NTC = 64 * WM * WN, NTL = 64 * NL;     // compute / loader threads
  constexpr int MI = BM / WM / 16, NI = BN / WN / 16;  // MFMA tiles per compute wave
  constexpr int RPP = NTL / 8;                         // tile rows covered by one DMA pass of the loader waves
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be whole DMA passes of the loader waves");
  constexpr int A_PASSES = BM / RPP, W_PASSES = BN / RPP;
  constexpr int L = A_PASSES + W_PASSES;               // LDS-DMA instructions per loader thread per K-tile
  constexpr int D = S - 1;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr unsigned OOB = 0x80000000u;
  static_assert((D - 1) * L < 64, "vmcnt immediate");
  static_assert(!RS || S == 2, "register staging: two LDS stages (the loader's registers are the third)");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [S][ A: BM x 128 B | B: BN x 128 B ]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= NTC / 64;
  int tile_m, tile_n, split;
  igemm_work_item(p, tile_m, tile_n, split);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kt0 = split * p.kt_per_split;
  const int kt1 = min(p.nkt, kt0 + p.kt_per_split);

  if (loader) {
    // ================================ loader waves: the LDS-DMA ring and nothing else ================================
    const int lt = tid - NTC, lwave = wave - NTC / 64;
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rs_x2 = make_rsrc(p.x2 ? (const void*)p.x2 : (const void*)p.x, p.x2 ? p.x2_bytes : p.x_bytes);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);
    const int rbase = lt >> 3;
    const int kchunk = (lt & 7) ^ ((rbase >> 1) & 7);   // the swizzle's inverse image: applied on the SOURCE chunk (see igemm_pipe_kernel)
    int a_pix0[A_PASSES], a_ih0[A_PASSES], a_iw0[A_PASSES], a_m[A_PASSES];
#pragma unroll
    for (int ps = 0; ps < A_PASSES; ++ps) {
      const int m = min(m0 + rbase + RPP * ps, p.M - 1);
      a_m[ps] = m;                                      // the fused 1x1 segment reads output pixel m of x3 | x4
      const int b = fdiv(m, p.fd_ohw);
      const int pix = m - b * p.OHW;
      const int oh = fdiv(pix, p.fd_ow), ow = pix - oh * p.OW;
      a_pix0[ps] = b * p.IH * p.IW;
      a_ih0[ps] = oh * p.sh - p.ph;
      a_iw0[ps] = ow * p.sw - p.pw;
    }
    unsigned b_off[W_PASSES];
#pragma unroll
    for (int ps = 0; ps < W_PASSES; ++ps) {
      const int row = min(n0 + rbase + RPP * ps, p.N - 1);
      b_off[ps] = (unsigned)row * (unsigned)p.Kpad * 2u + kchunk * 16;
    }
    // scalar K cursor (tap, channel); past the last filter tap (s_kh == KH) lies the optional fused 1x1 segment over x3 | x4
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
    const int k_cin = p.Cin, k_cin2 = p.Cin2, k_ctot = p.Ctot, k_cin3 = p.Cin3, k_cin4 = p.Cin4, k_c3tot = p.C3tot;
    bool s_fresh = true;
    unsigned cur_off[A_PASSES];
    int a_soff = 0, b_soff = kt0 * BK * 2;
    const int IHv = p.UH > 0 ? p.UH : p.IH, IWv = p.UW > 0 ? p.UW : p.IW;

    auto issue = [&](int kt, int stage, ws_u32x4* R) {      // R (RS only): the L registers that receive the tile instead of LDS
      char* sbase = smem + stage * STAGE + lwave * 1024;   // + pass * RPP * 128: this wave's 8 rows of the pass
      const bool live = kt < kt1;
      const bool ext = s_kh >= p.KH;                    // (scalar) inside the fused 1x1 segment
      const int cA = ext ? k_cin3 : k_cin;
      const bool src2 = s_c0 >= cA;
      if (live) {
        if (s_fresh && ext) {
          const int Cs = src2 ? k_cin4 : k_cin3;
#pragma unroll
          for (int ps = 0; ps < A_PASSES; ++ps) cur_off[ps] = (unsigned)a_m[ps] * (unsigned)(Cs * 2) + kchunk * 16;
          s_fresh = false;
        }
        if (s_fresh) {
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
          s_fresh = false;
        }
        a_soff = (s_c0 - (src2 ? cA : 0)) * 2;
        b_soff = kt * BK * 2;
      }
      auto fetch = [&](const __amdgpu_buffer_rsrc_t rs, int i, char* dst, unsigned voff, int soff) {
        if constexpr (RS) R[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, voff, soff, 0, 0);
      };
      if (ext) {   // descriptors built on the spot (transient SGPRs), as in igemm_pipe_kernel
        const __amdgpu_buffer_rsrc_t rs_e = src2 ? make_rsrc(p.x4, p.x4_bytes) : make_rsrc(p.x3, p.x3_bytes);
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) fetch(rs_e, ps, sbase + ps * (RPP * 128), live ? cur_off[ps] : OOB, a_soff);
      } else if (src2) {
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) fetch(rs_x2, ps, sbase + ps * (RPP * 128), live ? cur_off[ps] : OOB, a_soff);
      } else {
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) fetch(rs_x, ps, sbase + ps * (RPP * 128), live ? cur_off[ps] : OOB, a_soff);
      }
#pragma unroll
      for (int ps = 0; ps < W_PASSES; ++ps) fetch(rs_w, A_PASSES + ps, sbase + BM * 128 + ps * (RPP * 128), live ? b_off[ps] : OOB, b_soff);
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
    if constexpr (RS) {
      // tile t: fetched into R[t & 1] two iterations ahead, written to LDS stage t & 1 one iteration ahead (after the barrier that says
      // every compute wave is done with tile t - 2, the stage's previous tenant), visible to the compute waves at barrier t.
      ws_u32x4 R0[L], R1[L];
      auto put = [&](int stage, const ws_u32x4* R) {
        char* sbase = smem + stage * STAGE + lwave * 1024;
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) *reinterpret_cast<ws_u32x4*>(sbase + ps * (RPP * 128) + lane * 16) = R[ps];
#pragma unroll
        for (int ps = 0; ps < W_PASSES; ++ps) *reinterpret_cast<ws_u32x4*>(sbase + BM * 128 + ps * (RPP * 128) + lane * 16) = R[A_PASSES + ps];
      };
      issue(kt0, 0, R0);
      issue(kt0 + 1, 1, R1);
      put(0, R0);
      issue(kt0 + 2, 0, R0);
      __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): tile kt0 is written
      for (int kt = kt0; kt < kt1; kt += 2) {
        __builtin_amdgcn_s_barrier();                   // tile kt handed over; tile kt - 1's stage (1) is free
        put(1, R1);
        issue(kt + 3, 1, R1);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (kt + 1 >= kt1) break;
        __builtin_amdgcn_s_barrier();                   // tile kt + 1 handed over; stage 0 is free
        put(0, R0);
        issue(kt + 4, 0, R0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
      }
      __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0): the dummy fetches past the end
      __builtin_amdgcn_s_barrier();
      return;
    }
#pragma unroll
    for (int s = 0; s < D; ++s) issue(kt0 + s, s, nullptr);
    int st_fill = D;
    for (int kt = kt0; kt < kt1; ++kt) {
      wait_vmcnt<(D - 1) * L>();                        // tile kt (this wave's rows of it) has landed
      __builtin_amdgcn_s_barrier();                     // ... and every compute wave is done with tile kt - 1
      issue(kt + D, st_fill, nullptr);                  // refill that stage while the compute waves multiply tile kt
      st_fill = (st_fill + 1 == S) ? 0 : st_fill + 1;
    }
    wait_vmcnt<0>();                                    // the dummy tiles past the end too: the epilogue re-uses the ring's LDS
    __builtin_amdgcn_s_barrier();
    return;                                             // (an ended wave counts as arrived at every later barrier)
  }

  // ===================================== compute waves: fragment reads + MFMAs =====================================
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = lane & 15, lq = lane >> 4;
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int st = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    __builtin_amdgcn_s_barrier();                       // tile kt is in LDS (every loader waited for its rows before arriving)
    const char* As = smem + st * STAGE;
    const char* Bs = As + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
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
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    st = (st + 1 == S) ? 0 : st + 1;
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0): this wave's last fragment reads
  __builtin_amdgcn_s_barrier();                         // every DMA has landed and every fragment is read: the ring's LDS is free
  igemm_epilogue<BM, BN, MI, NI, false, NTC, EPI>(p, acc, smem, false, m0, n0, wm * (BM / WM), wn * (BN / WN), lrow, lq, split, tid, nullptr);
#endif
}

template <int BM, int BN, int WM, int WN, int NL, int S, int EPI, bool RS = false>
int launch_ws(const IgemmDev& d, hipStream_t st) {
  constexpr size_t ring = (size_t)S * (BM + BN) * 128;
  constexpr size_t lds = (ring > (size_t)EpiCfg<BM, BN>::BYTES ? ring : (size_t)EpiCfg<BM, BN>::BYTES) + 2 * BM * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static unsigned long long attr_done = 0;
  auto kern = igemm_ws_kernel<BM, BN, WM, WN, NL, S, EPI, RS>;
  if (int rc = aldm_set_max_lds(reinterpret_cast<const void*>(kern), (int)lds, &attr_done, "igemm_ws")) return rc;
  if ((d.qstat || d.rowstat) && d.N % BN != 0) {
    aldm_set_error("igemm: qstat_out / rowstat_out need Cout %d to be a multiple of the tile width %d", d.N, BN);
    return ALDM_E_ARG;
  }
  IgemmDev dd = d;
  dd.tiles_n = cdiv(d.N, BN);
  dd.fd_tiles_n = make_fastdiv((unsigned)dd.tiles_n);
  dd.tiles_m = cdiv(d.M, BM);
  dd.fd_tiles_m = make_fastdiv((unsigned)dd.tiles_m);
  dd.fd_splits = make_fastdiv((unsigned)d.splits);
  dd.nwg = dd.tiles_m * dd.tiles_n * d.splits;
  hipLaunchKernelGGL(kern, dim3(dd.nwg), dim3(64 * (WM * WN + NL)), lds, st, dd);
  return aldm_launch_status("igemm_ws");
}

template <int BM, int BN, int WM, int WN, int NL, int S, bool RS = false>
int launch_ws_epi(const IgemmDev& d, hipStream_t st) {
  const bool lean = d.splits <= 1 && d.out_act == ALDM_ACT_NONE && d.post_act == ALDM_ACT_NONE;
  if (d.splits > 1) return launch_ws<BM, BN, WM, WN, NL, S, 3, RS>(d, st);
  if (lean && d.qstat) return launch_ws<BM, BN, WM, WN, NL, S, 4, RS>(d, st);
  if (lean) return launch_ws<BM, BN, WM, WN, NL, S, 1, RS>(d, st);
  return launch_ws<BM, BN, WM, WN, NL, S, 0, RS>(d, st);
}

static int ws_refuse(const IgemmDev& d, int Rp, bool vt) {
  const bool fast = d.in_act == ALDM_ACT_NONE && d.Cin % 64 == 0 && d.Cin2 % 64 == 0 && d.x_bytes < 0x80000000u && d.x2_bytes < 0x80000000u;
  if (!fast || Rp != 0 || vt || d.ln_s || d.geglu) {
    aldm_set_error("igemm: the wave-specialised tiles take plain launches only (LDS-DMA path, no LoRA / V^T / folded LayerNorm / GEGLU)");
    return ALDM_E_UNSUPPORTED;
  }
  return ALDM_OK;
}

}  // namespace aldm_igemm_detail

int aldm_launch_tile_256x128ws(const aldm_igemm_detail::IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st) {
  using namespace aldm_igemm_detail;
  if (int rc = ws_refuse(d, Rp, vt)) return rc;
  if (ring == 2) return launch_ws_epi<256, 128, 4, 2, 4, 2, true>(d, st);     // ring 2 = register-staged loaders
  return launch_ws_epi<256, 128, 4, 2, 4, 3>(d, st);
}
// 4 compute waves + 4 loader waves on the small tiles; ring 3 or 4 (24 KB per stage)
int aldm_launch_tile_64x128ws(const aldm_igemm_detail::IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st) {
  using namespace aldm_igemm_detail;
  if (int rc = ws_refuse(d, Rp, vt)) return rc;
  if (ring == 2) return launch_ws_epi<64, 128, 2, 2, 4, 2, true>(d, st);
  if (ring == 4) return launch_ws_epi<64, 128, 2, 2, 4, 4>(d, st);
  return launch_ws_epi<64, 128, 2, 2, 4, 3>(d, st);
}
int aldm_launch_tile_128x64ws(const aldm_igemm_detail::IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st) {
  using namespace aldm_igemm_detail;
  if (int rc = ws_refuse(d, Rp, vt)) return rc;
  if (ring == 2) return launch_ws_epi<128, 64, 2, 2, 4, 2, true>(d, st);
  if (ring == 4) return launch_ws_epi<128, 64, 2, 2, 4, 4>(d, st);
  return launch_ws_epi<128, 64, 2, 2, 4, 3>(d, st);
}

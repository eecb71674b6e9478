// Instantiates 8-wave (512-thread) forms of the SMALL workgroup tiles of the LDS-DMA implicit-GEMM kernel (see igemm_core.h), for the
// split-K convolutions of the UNet's 252- / 64-token levels: there the grid is about one workgroup per CU, and with four waves (one per
// SIMD) a K-tile's LDS-DMA issue (~50 % of a wave's cycles: the vector-memory path delivers ~70 GB/s per CU), its fragment reads + MFMAs
// (~33 %) and its barrier ADD UP in every wave.  Eight waves halve each wave's DMA instructions per K-tile and put two waves on every
// SIMD, so one wave's MFMAs run under the other's DMA issue.  64x128: 2 x 4 waves of 32x32; 128x64: 4 x 2 waves of 32x32.
#include "igemm_core.h"
namespace aldm_igemm_detail {
template <int BM, int BN, int WM, int WN, int S>
int launch_s8(const IgemmDev& d, hipStream_t st) {
  if (d.splits > 1) return launch_cfg<BM, BN, WM, WN, 0, false, S, 3>(d, st);                  // split-K: only the partial-tile store
  if (!d.geglu && d.out_act == ALDM_ACT_NONE && d.post_act == ALDM_ACT_NONE && !d.ln_s) {       // LEAN: see igemm_core.h
    if (d.qstat) return launch_cfg<BM, BN, WM, WN, 0, false, S, 4>(d, st);
    return launch_cfg<BM, BN, WM, WN, 0, false, S, 1>(d, st);
  }
  return launch_cfg<BM, BN, WM, WN, 0, false, S>(d, st);
}
template <int BM, int BN, int WM, int WN>
int launch_s8_ring(const IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st) {
  const bool fast = d.in_act == ALDM_ACT_NONE && d.Cin % 64 == 0 && d.Cin2 % 64 == 0 && d.x_bytes < 0x80000000u && d.x2_bytes < 0x80000000u;
  if (!fast || Rp != 0 || vt) { aldm_set_error("igemm: the 8-wave small tiles need the LDS-DMA path, no LoRA side channel and no V^T store"); return ALDM_E_UNSUPPORTED; }
  if (ring == 2) return launch_s8<BM, BN, WM, WN, 2>(d, st);
  if (ring == 4) return launch_s8<BM, BN, WM, WN, 4>(d, st);
  return launch_s8<BM, BN, WM, WN, 3>(d, st);
}
}  // namespace aldm_igemm_detail
int aldm_launch_tile_64x128w8(const aldm_igemm_detail::IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st) {
  return aldm_igemm_detail::launch_s8_ring<64, 128, 2, 4>(d, Rp, vt, ring, st);
}
int aldm_launch_tile_128x64w8(const aldm_igemm_detail::IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st) {
  return aldm_igemm_detail::launch_s8_ring<128, 64, 4, 2>(d, Rp, vt, ring, st);
}

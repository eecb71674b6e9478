// Instantiates the 8-wave (512-thread) 128x128 workgroup tile of the LDS-DMA implicit-GEMM kernel (see igemm_core.h):
// each wave owns 32x64 of the tile, so a wave issues half the DMA instructions per MFMA of the 4-wave 64x64 tile.
#include "igemm_core.h"
namespace aldm_igemm_detail {
template <int S>
int launch_w8(const IgemmDev& d, int Rp, bool vt, hipStream_t st) {
  if (vt) {
    if (Rp == 0) return launch_cfg<128, 128, 4, 2, 0, true, S>(d, st);
    return launch_cfg<128, 128, 4, 2, 64, true, S>(d, st);
  }
  if (d.splits <= 1 && !d.geglu && d.out_act == ALDM_ACT_NONE && d.post_act == ALDM_ACT_NONE && !d.ln_s) {   // LEAN: see igemm_core.h
    if (Rp == 0 && d.qstat) return launch_cfg<128, 128, 4, 2, 0, false, S, 4>(d, st);   // + GroupNorm statistics (EPI 4)
    if (Rp == 0) return launch_cfg<128, 128, 4, 2, 0, false, S, 1>(d, st);
    return launch_cfg<128, 128, 4, 2, 64, false, S, 1>(d, st);
  }
  if (Rp == 0) return launch_cfg<128, 128, 4, 2, 0, false, S>(d, st);
  return launch_cfg<128, 128, 4, 2, 64, false, S>(d, st);
}
}  // namespace aldm_igemm_detail
int aldm_launch_tile_128x128w8(const aldm_igemm_detail::IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st) {
  using namespace aldm_igemm_detail;
  const bool fast = d.in_act == ALDM_ACT_NONE && d.Cin % 64 == 0 && d.Cin2 % 64 == 0 && d.x_bytes < 0x80000000u && d.x2_bytes < 0x80000000u;
  if (!fast || Rp == 32) { aldm_set_error("igemm: the 8-wave tile needs the LDS-DMA path and Rp in {0, 64}"); return ALDM_E_UNSUPPORTED; }
  if (ring == 3) return launch_w8<3>(d, Rp, vt, st);
  return launch_w8<2>(d, Rp, vt, st);
}

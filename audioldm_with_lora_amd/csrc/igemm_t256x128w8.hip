// Instantiates the 8-wave (512-thread) 256x128 workgroup tile of the LDS-DMA implicit-GEMM kernel (see igemm_core.h): each wave owns
// 64x64 of the tile = 16 MFMAs per 8 fragment reads per k-step, and a K-tile costs the workgroup 6 DMA passes for 256 x 128 x 64
// MACs (the 128x128 tile: 4 passes for half the work).  For the big-M layers of the VAE and the vocoder (M >= 64k rows, thousands
// of workgroups), where neither launch latency nor workgroup count is the limit.  No LoRA / V^T forms: plain convolutions only.
#include "igemm_core.h"
namespace aldm_igemm_detail {
template <int S>
int launch_256(const IgemmDev& d, hipStream_t st) {
  if (d.splits <= 1 && !d.geglu && d.out_act == ALDM_ACT_NONE && d.post_act == ALDM_ACT_NONE && !d.ln_s) {   // LEAN: see igemm_core.h
    if (d.qstat) return launch_cfg<256, 128, 4, 2, 0, false, S, 4>(d, st);
    return launch_cfg<256, 128, 4, 2, 0, false, S, 1>(d, st);
  }
  return launch_cfg<256, 128, 4, 2, 0, false, S>(d, st);
}
}  // namespace aldm_igemm_detail
int aldm_launch_tile_256x128w8(const aldm_igemm_detail::IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st) {
  using namespace aldm_igemm_detail;
  const bool fast = d.in_act == ALDM_ACT_NONE && d.Cin % 64 == 0 && d.Cin2 % 64 == 0 && d.x_bytes < 0x80000000u && d.x2_bytes < 0x80000000u;
  if (!fast || Rp != 0 || vt) { aldm_set_error("igemm: the 256x128 tile needs the LDS-DMA path and takes no LoRA / V^T"); return ALDM_E_UNSUPPORTED; }
  if (ring == 3) return launch_256<3>(d, st);
  return launch_256<2>(d, st);
}

// Instantiates the implicit-GEMM kernels for the 64x64 workgroup tile (see igemm_core.h).
#include "igemm_core.h"
int aldm_launch_tile_64x64(const aldm_igemm_detail::IgemmDev& d, int Rp, bool vt, int ring, hipStream_t st) {
  return aldm_igemm_detail::launch_tile<64, 64, 2, 2, 4, 3>(d, Rp, vt, ring, st);
}

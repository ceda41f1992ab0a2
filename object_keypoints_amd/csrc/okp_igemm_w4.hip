// 4-wave 256x256 tiles of the implicit-GEMM kernel: every wave owns a 128x128 sub-tile, i.e. 256 accumulator
// registers per lane.  They live in the AGPR file (one wave per SIMD may use 256 arch + 256 acc registers), which is
// why this translation unit is compiled WITHOUT -amdgpu-mfma-vgpr-form (see build.py).  A 128x128 wave tile reads
// (128+128) fragment rows per 128x128 outputs from LDS, a third less than the 8-wave kernel's 64x128 wave tiles:
// the 8-wave main loop keeps the LDS port ~100 % busy (fragment reads + LDS-DMA writes), this one ~75 %.
#include "okp_igemm_kernel.h"

int okp_launch_igemm_w4(const okp_conv* plan, const OkpIgemmParams& p, int tile, hipStream_t stream) {
  if (plan->dtype != OKP_BF16) {
    okp_set_error("okp_conv_forward: tile %d is bf16 only", tile);
    return OKP_EINVAL;
  }
  if (tile == 10) return launch_cfg<__bf16, 256, 256, 2, 2, 2, 128, 16>(plan, p, stream);
  return launch_cfg<__bf16, 256, 256, 2, 2, 2, 128, 32>(plan, p, stream);
}

// Launcher of the tap-list implicit-GEMM kernel (okp_igemm_kernel.h): tile codes -> template instances.
#include "okp_igemm_kernel.h"

namespace {

template <typename T>
int launch_tile(const okp_conv* plan, const OkpIgemmParams& p, int tile, hipStream_t stream) {
  switch (tile) {
    case 8: return launch_cfg<T, 64, 64, 2, 2, 4, 128, 16>(plan, p, stream);   // 64x64 on 16x16 MFMA tiles
    case 6: return launch_cfg<T, 256, 256, 4, 2, 2, 128, 16>(plan, p, stream); // 256x256 on 16x16 MFMA tiles
    case 4: return launch_cfg<T, 128, 256, 2, 2, 3, 64>(plan, p, stream);     // 2 workgroups per CU, half-slice ring
    case 3: return launch_cfg<T, 256, 256, 4, 2, 2, 128>(plan, p, stream);
    case 2: return launch_cfg<T, 128, 128, 2, 2, 2, 128>(plan, p, stream);
    default: return launch_cfg<T, 64, 64, 2, 2, 4, 128>(plan, p, stream);
  }
}

// OKP_F32X3: the fp32-storage tiles with the split-product loop (32x32x16 fp16 MFMAs)
int launch_tile_x3(const okp_conv* plan, const OkpIgemmParams& p, int tile, hipStream_t stream) {
  switch (tile) {
    case 4: return launch_cfg<F32S, 128, 256, 2, 2, 3, 64>(plan, p, stream);
    case 3: case 6: return launch_cfg<F32S, 256, 256, 2, 4, 2, 128>(plan, p, stream);   // 128 x 64 wave tiles: a wave splits two pixel fragments per four weight fragments
    case 2: return launch_cfg<F32S, 128, 128, 2, 2, 2, 128>(plan, p, stream);
    default: return launch_cfg<F32S, 64, 64, 2, 2, 4, 128>(plan, p, stream);
  }
}

}  // namespace

int okp_select_tile(int dtype, int cout_pad, long P) {
  // Fill the 256 CUs first; only then grow the tile (bigger tiles re-read less from L2).
  auto tiles = [&](int b) { return ((P + b - 1) / b) * ((cout_pad + b - 1) / b); };
  // 16-bit 256x256 tiles run on 16x16x32 MFMAs (same FLOPs per LDS byte as 32x32x16, measured 4-7 % faster: the
  // shorter MFMA gives the scheduler twice as many slots to hide the fragment reads in); fp32 keeps 32x32x2
  if (tiles(256) >= 256 && cout_pad >= 192) return okp_is16(dtype) ? 6 : 3;
  if (tiles(128) >= 256) return 2;
  return okp_is16(dtype) ? 8 : 1;             // 64x64: the 16x16 variant measured 2-5 % faster, 128x128: no difference
}

int okp_launch_igemm(const okp_conv* plan, const OkpIgemmParams& p, int tile, hipStream_t stream) {
  if (tile == 13) return okp_launch_igemm_patch(plan, p, stream);       // patch-resident 3x3 kernel (okp_igemm_patch.hip)
  if (tile == 14) {                                                       // ... its 32x32x16 instantiation (16-bit plans; MFMA-shape A/B)
    if (!okp_is16(plan->dtype)) { okp_set_error("okp_conv_forward: tile 14 is the 16-bit patch-resident kernel on 32x32x16 MFMAs"); return OKP_EINVAL; }
    OkpIgemmParams q = p;
    q.mfma32 = 1;
    return okp_launch_igemm_patch(plan, q, stream);
  }
  if (tile == 0) tile = okp_select_tile(plan->dtype, p.cout_pad, (long)p.N * p.Ho * p.Wo);
  if (plan->dtype == OKP_BF16) return launch_tile<__bf16>(plan, p, tile, stream);
  if (plan->dtype == OKP_F16) return launch_tile<_Float16>(plan, p, tile, stream);
  if (plan->dtype == OKP_F32X3) return launch_tile_x3(plan, p, tile, stream);
  return launch_tile<float>(plan, p, tile, stream);
}

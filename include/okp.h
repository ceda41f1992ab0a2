/*
 * okp.h — C ABI of libokp_hip.so: the MI355X (gfx950) keypoint-inference hot path.
 *
 * Drop-in boundary for the reference's `perception.models` / `perception.pipeline`
 * hot path (ethz-asl/object_keypoints).  The reference has no FFI of its own — its
 * device work is stock PyTorch ops called from Python — so each entry point below
 * names the reference call site(s) it replaces (file:line under the reference tree).
 * Plain pointers and sizes only: no torch / HIP types cross this boundary.
 *
 * Conventions
 *   - Every `*_dev` / `void* x` argument is a DEVICE pointer (HBM) owned by the caller.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Launch
 *     functions only enqueue work: no allocation, no synchronisation, graph-capturable.
 *   - Activations are NHWC ("channels-last"), element type `dtype` (OKP_F32 / OKP_BF16 / OKP_F16),
 *     addressed as  base + (pixel * pix_stride + channel) elements, so a tensor argument
 *     may be a channel slice of a wider tensor (that is how concat / split are free).
 *   - Return value: 0 on success, negative OKP_E* otherwise; okp_last_error() gives text.
 *     Shape / dtype / alignment violations are reported, never silently patched —
 *     the reference's convention is assert / ValueError (pipeline.py:67,183; camera_utils.py:16,144).
 */
#ifndef OKP_H
#define OKP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OKP_ABI_VERSION 7     /* 2: okp_camera.model, OKP_F16, okp_stem_create_dtype, okp_radtan...  3: OKP_F32X3  4: okp_stream_wait_stream
                                 5: OKP_F32X3 in okp_stem_create_dtype / okp_stem_forward_nchw (fp32 NHWC output) and okp_fire_forward; tile 13 for OKP_F32X3 plans; tile 14
                                 6: okp_conv_args.src_pairs / out_pairs (pair-format tensors between split-product 3x3 convolutions), okp_stem_forward_nchw_pairs
                                 7: okp_conv_patch_applies; okp_group_objects: `reduced_dev` (device k-means reduction of surplus votes);
                                    okp_conv_set_range_flag / okp_stem_set_range_flag (fp16-range guard of split-product plans), `range_flag_dev` of
                                    okp_capacity_overflow, okp_cast and okp_add_f16_f32; views of 2 GiB and more as the output of an OKP_F32X3
                                    stem and as sources of OKP_F32X3 plans on the patch-resident kernel */

/* OKP_F16: IEEE half activations / weights, fp32 accumulate (BASELINE configs[4]).
 * OKP_F32X3 (okp_conv plans only): fp32 activations, weights and results like OKP_F32 - every tensor argument of such a plan is an
 * fp32 tensor - but the products run on the fp16 matrix pipe as the three-term split x*w = x_hi*w_hi + x_lo*w_hi + x_hi*w_lo
 * (x = x_hi + x_lo, two fp16 halves = 22 significant bits; fp32 accumulation).  Results agree with OKP_F32 to ~1e-6 relative
 * (the reference's fp32 tolerance, perception/pipeline.py:20-27, holds) at a multiple of its speed; operands must stay below
 * 65504 in magnitude. */
enum { OKP_F32 = 0, OKP_BF16 = 1, OKP_F16 = 2, OKP_F32X3 = 3 };
enum { OKP_ACT_NONE = 0, OKP_ACT_RELU = 1, OKP_ACT_SIGMOID = 2 };
enum {
  OKP_OK = 0,
  OKP_EINVAL = -1,   /* bad argument (shape, dtype, alignment, capacity) */
  OKP_EHIP = -2,     /* HIP runtime error */
  OKP_ENOMEM = -3
};

const char* okp_last_error(void);
int okp_abi_version(void);
/* Number of visible HIP devices and the gfx arch string of device `dev` (e.g. "gfx950"). */
int okp_device_count(void);
int okp_device_arch(int dev, char* buf, int buflen);

/* ------------------------------------------------------------------------------------
 * Dense convolution as a tap-list implicit GEMM (MFMA).
 *
 * Replaces torch conv2d / conv_transpose2d + batch_norm + add + relu as used by
 *   convolution            corner_net_lite/core/models/py_utils/utils.py:143-156
 *   residual               .../py_utils/utils.py:158-185   (conv2 + projected skip + add + relu = ONE plan, 2 sources)
 *   fire_module 1x1s       .../CornerNet_Squeeze.py:13-15,22-30
 *   unpool + merge         .../CornerNet_Squeeze.py:35-36, py_utils/modules.py:64-65 (4 sub-pixel plans)
 *   inter-stack merge      .../py_utils/modules.py:89-91   (two 1x1+BN summed = ONE plan, 2 sources)
 *   prediction_module      perception/models.py:13-18
 *
 * out[n, ho*out_step+out_oy, wo*out_step+out_ox, co] =
 *     act( bias[co] + residual[...same pixel..., co]
 *          + sum_taps sum_ci  W_tap[co][ci] * src[tap.src][n, ho*conv_stride+tap.dy, wo*conv_stride+tap.dx, ci] )
 * with zero for out-of-range source pixels.  BatchNorm (eval mode, perception/models.py:67,81)
 * is folded into W/bias by the caller before okp_conv_create.
 * ---------------------------------------------------------------------------------- */
typedef struct okp_tap {
  int32_t src;      /* 0 or 1: which source tensor this tap reads */
  int32_t dy, dx;   /* source pixel = (ho*conv_stride[src] + dy, wo*conv_stride[src] + dx) */
  const float* w;   /* HOST pointer, row-major [cout][cin[src]] fp32 */
} okp_tap;

typedef struct okp_conv okp_conv;   /* opaque plan: packed weights + bias + slice table in HBM */

/* cin[s]*sizeof(dtype) must be a multiple of 16 bytes; cout a multiple of 8; n_taps <= 32. */
okp_conv* okp_conv_create(int dtype, int n_src, const int32_t* cin, const int32_t* conv_stride,
                          int32_t cout, int32_t n_taps, const okp_tap* taps,
                          const float* bias /* HOST [cout] or NULL */, int act);
/* OKP_F32X3 plans with a term count PER TAP (the sensitivity-guided mixed configuration): tap_terms[t] = 3 multiplies tap t's
 * products as the three-term split (fp32-grade), tap_terms[t] = 1 as the single fp16 product x_hi * w_hi (both operands rounded to
 * fp16 for that product only - the tensors stay fp32), at a third of the matrix-pipe time.  Meant for convolutions whose rounding
 * error reaches the output attenuated - the 3x3 convolutions inside residual blocks (py_utils/utils.py:158-185: conv1, conv2), behind
 * a BatchNorm-scaled branch - while the taps of the skip path keep three terms; tests/precision/attribute.py prices a choice on given
 * weights.  tap_terms == NULL: all taps three terms (= okp_conv_create(OKP_F32X3, ...)). */
okp_conv* okp_conv_create_x3(int n_src, const int32_t* cin, const int32_t* conv_stride, int32_t cout, int32_t n_taps,
                             const okp_tap* taps, const uint8_t* tap_terms, const float* bias, int act);
/* OKP_F32X3 range guard (ABI 7).  A split-product plan halves its fp32 operands into fp16: a value beyond +-65504 (or a non-finite one)
 * becomes (inf, -inf), its products NaN, and the next ReLU turns the NaN into 0 - silently, where the reference's fp32 arithmetic
 * (py_utils/utils.py:143-156) returns numbers.  With a flag attached, every launch of the plan (okp_conv_forward on any tile, and
 * okp_fire_forward / okp_heads_forward through their squeeze / first-layer plan) ORs 1 into *flag_dev when the magnitude of one of its
 * RESULTS - a value the next split-product layer will halve, including the squeeze tile of the one-launch fire module and the hidden layer
 * of the heads - passes 65504 (a running maximum; with every input in range no NaN can arise, so infinity is the only way out);
 * okp_stem_set_range_flag does the same for the stem and checks the frames it reads exactly (NaN included), as okp_cast / okp_add_f16_f32
 * do for the fp16 data they hand to split-product consumers.  The
 * flag is never cleared by the library: the caller zeroes it before a pass and reads it after (or hands it to okp_capacity_overflow).
 * flag_dev: DEVICE int32, NULL = off (the default).  Set once, before the plan is used from several threads or captured in a graph. */
int okp_conv_set_range_flag(okp_conv* plan, int32_t* flag_dev);
void okp_conv_destroy(okp_conv* plan);

typedef struct okp_tensor {        /* an NHWC view */
  void* data;                      /* device pointer to element (n=0,y=0,x=0,c=0) of the view */
  int32_t h, w;                    /* spatial size */
  int32_t pix_stride;              /* elements between consecutive pixels (>= channels of the view) */
  int64_t bytes;                   /* bytes addressable from `data` (bounds for hardware range checks); < 2 GiB - except the output of an OKP_F32X3
                                      stem and a source of an OKP_F32X3 plan whose launch runs on the patch-resident kernel (tile 13): those two
                                      kernels address frame by frame, one FRAME must stay below 2 GiB (ABI 7) */
} okp_tensor;

typedef struct okp_conv_args {
  int32_t n;                       /* batch */
  int32_t ho, wo;                  /* output pixel grid of the GEMM (before out_step mapping) */
  okp_tensor src[2];
  okp_tensor out;                  /* out.h/out.w are the FULL output tensor's spatial size */
  int32_t out_step, out_oy, out_ox;/* sub-pixel placement (1,0,0 for ordinary convs) */
  okp_tensor res;                  /* optional residual, same spatial mapping as out; data==NULL if none */
  int32_t tile;                    /* 0 = auto; 1 = 64x64, 2 = 128x128, 3 = 256x256 (32x32 MFMA); 4 = 128x256 on a half-slice ring;
                                      6 / 8 = 256x256 / 64x64 on 16x16 MFMAs (16-bit types); 13 = patch-resident 3x3 kernel (16-bit types and
                                      OKP_F32X3 plans, cout_pad % 256 == 0, whole 16x16-pixel blocks; OKP_EINVAL where it does not apply);
                                      14 = the same 16-bit kernel on 32x32x16 MFMAs (MFMA-shape A/B; never the heuristic's choice) */
  /* Optional fused depth-wise branch (the fire-module tail, CornerNet_Squeeze.py:15-17,25-30): the same launch
   * also computes  dw_out[..., c] = act(dw_bias[c] + dw_res[..., c] + sum_{3x3 taps} dw_w[tap][c] * src[0][..., c])
   * for c in [0, cout) with the plan's conv_stride[0] and pad 1, so that `expand 1x1 || depth-wise 3x3` of one
   * squeeze tensor is ONE kernel.  dw_w_dev: DEVICE fp32 [9][cout]; dw_bias_dev: DEVICE fp32 [cout];
   * dw_out / dw_res: views with the spatial mapping of out / res.  dw_w_dev == NULL disables the branch. */
  const float* dw_w_dev;
  const float* dw_bias_dev;
  okp_tensor dw_out;
  okp_tensor dw_res;
  /* Sub-pixel classes in ONE launch (transposed convolution 4x4/s2 = four 2x2 convolutions, one per output
   * parity): with n_classes == 4 the plan's taps are four equal consecutive groups; class k uses group k only and
   * writes to (out_oy + k/2, out_ox + k%2) with out_step 2.  0 or 1 = off. */
  int32_t n_classes;
  /* OKP_F32X3 plans only (the mixed configuration: fp32 stream, fp16 residual branches).  out16: optional fp16 view with the spatial
   * mapping of `out` that receives the result rounded to fp16 as well (what a single-term consumer running on the fp16 kernels reads);
   * with out16 given, out.data may be NULL (h, w, pix_stride still describe the output grid) and only the fp16 copy is written.
   * res_is_f16 != 0: `res` is an fp16 tensor (a residual branch computed by the fp16 kernels).  Zero / NULL = off. */
  okp_tensor out16;
  int32_t res_is_f16;
  /* out_subsample == 2 (needs out16, out_step 1): the fp32 result is kept at even rows and columns of the output grid only, `out` being a
   * tensor of ceil(ho / 2) x ceil(wo / 2) pixels; out16 (and res) keep the full grid.  For a stream tensor whose only fp32 reader is the
   * stride-2 projected skip of the next residual block (py_utils/utils.py:177-185: stem -> pre[1], pre[1] -> pre[2]) while the block's
   * conv1 reads the fp16 copy: three quarters of the fp32 bytes are never read. */
  int32_t out_subsample;
  /* PAIR FORMAT (OKP_F32X3 plans on the patch-resident kernel, tile 13, only).  A pair-format tensor has the geometry of an fp32 tensor
   * (4 bytes per element, the same h / w / pix_stride / bytes) but every aligned group of 8 channels holds
   *     [ hi(c0) .. hi(c0+7) | lo(c0) .. lo(c0+7) ]   as 2 x 8 fp16,   hi = fp16(x), lo = fp16(x - hi)
   * - exactly what the kernel's in-LDS split makes of the 8 fp32 values (okp_igemm_patch_x3.hip), so a consumer that is handed a pair
   * tensor skips that split and computes bit-identical results.  A tensor whose only readers are 3x3 convolutions of split-product plans
   * (conv1 -> conv2 inside a residual block, hourglass -> cnvs, py_utils/utils.py:158-185, py_utils/modules.py:84-92) is written this way
   * by its producer: the conversion happens once per element in the producer's epilogue instead of once per element, chunk and consumer
   * tile on landed LDS patches, between the K-steps.
   * src_pairs: bit s set = src[s] is in pair format (cin[s] % 32 == 0).  out_pairs != 0: `out` is written in pair format (cout % 8 == 0;
   * the residual, if any, stays fp32).  Any other tile, plan type, out16 / res_is_f16 / out_subsample / depth-wise branch: OKP_EINVAL. */
  int32_t src_pairs;
  int32_t out_pairs;
} okp_conv_args;

int okp_conv_forward(const okp_conv* plan, const okp_conv_args* args, void* stream);
/* The tile code (1..8, 13; never 14) the launch heuristic picks for these args when args->tile == 0. */
int okp_conv_select_tile(const okp_conv* plan, const okp_conv_args* args);
/* 1 if the patch-resident kernel (tile 13: the one kernel that reads / writes pair-format tensors) applies to this plan and these args
 * (only n, ho, wo, n_classes, the sources' pix_stride, dw_w_dev, out16 / res_is_f16 / out_subsample are read: a shape-only query needs no
 * pointers), else 0 - what a caller that forces args->tile = 13 or requests src_pairs / out_pairs asks first instead of getting OKP_EINVAL. */
int okp_conv_patch_applies(const okp_conv* plan, const okp_conv_args* args);
/* Multiply-accumulates one okp_conv_forward performs for these args (algorithmic, unpadded). */
int64_t okp_conv_macs(const okp_conv* plan, const okp_conv_args* args);

/* ------------------------------------------------------------------------------------
 * Whole fire module in ONE launch (bf16): squeeze 1x1 (+bn1) -> [expand 1x1 || depth-wise 3x3] (+bn2) -> (+x) -> ReLU,
 * the squeeze tensor staying in LDS (fire_module.forward, corner_net_lite/core/models/CornerNet_Squeeze.py:22-30).
 * squeeze: 1-tap plan cin -> mid (bias = folded bn1, no activation); expand: 1-tap plan mid -> half (bias = first half
 * of folded bn2); dw_w_dev [9][half] / dw_bias_dev [half]: depth-wise weights with the second half of bn2 folded.
 * cin and mid multiples of 64, mid == half <= 256.  out has 2*half channels: [expand | depth-wise].
 * Plans of one type: OKP_BF16 / OKP_F16 (okp_fire2.hip: the shapes okp_fire_forward reports in its error message), or OKP_F32X3
 * (okp_fire_x3.hip: fp32 x / out, three-term products; 256 -> 128 -> 256 at stride 1 with skip).  OKP_EINVAL for other shapes: run the
 * squeeze plan and the fused tail (okp_conv_forward with dw_*) instead.
 * ---------------------------------------------------------------------------------- */
typedef struct okp_fire_args {
  int32_t n;
  okp_tensor x;        /* input, cin channels */
  okp_tensor out;      /* output, 2*half channels, spatial size ceil(h/stride) x ceil(w/stride) */
  int32_t stride;      /* 1 or 2 (applies to both branches) */
  int32_t skip;        /* 1: add x (needs stride 1 and cin == 2*half) */
} okp_fire_args;
int okp_fire_forward(const okp_conv* squeeze, const okp_conv* expand, const float* dw_w_dev, const float* dw_bias_dev,
                     const okp_fire_args* args, void* stream);

/* A chain of up to OKP_FIRE_CHAIN_MAX consecutive stride-1 fire modules cin -> cin/2 -> cin with skip, in ONE launch with
 * the activations resident in LDS (one workgroup per frame): cin = 512 on maps of at most 4 x 4 pixels (the six
 * fire_module(512, 512) of the innermost hourglass level, CornerNet_Squeeze.py:10-51, modules [2,2,2,2,4]) and cin = 384
 * on maps of at most 8 x 8 pixels (the pairs of fire_module(384, 384) one level up).
 * Module m is given as for okp_fire_forward: squeeze[m], expand[m] plans, depth-wise weights / bias on the device.
 * Entry / exit form (the whole innermost hourglass level - low1, low2, low3 of hg_module n = 1, modules.py:36-66 - in one launch):
 * when module 0 is the stride-2 fire module 384 -> 256 -> 512 (make_hg_layer's first) and the last one the fire module
 * 512 -> 192 -> 384 (make_layer_revr's last), with one to six fire(512, 512) between them, x is the 384-channel map of at most
 * 8 x 8 pixels the entry module reads and out the 384-channel map of ceil(h / 2) x ceil(w / 2) pixels the exit module writes. */
#define OKP_FIRE_CHAIN_MAX 8
int okp_fire_chain_forward(int32_t n_modules, okp_conv* const* squeeze, okp_conv* const* expand,   /* plans gain a fragment-order weight copy on first use */
                           const float* const* dw_w_dev, const float* const* dw_bias_dev,
                           int32_t n, const okp_tensor* x, const okp_tensor* out, void* stream);

/* ------------------------------------------------------------------------------------
 * Depth-wise 3x3 convolution (pad 1, stride 1|2) + bias + optional residual + activation.
 * Replaces fire_module.conv_3x3 + its half of bn2 / skip-add / relu
 * (corner_net_lite/core/models/CornerNet_Squeeze.py:16-17,25-30).
 * w: DEVICE fp32 [9][c] (tap-major, BN scale folded), bias: DEVICE fp32 [c].
 * ---------------------------------------------------------------------------------- */
int okp_dwconv3x3_forward(int dtype, int32_t n, int32_t c, int32_t conv_stride,
                          const okp_tensor* src, const float* w_dev, const float* bias_dev,
                          const okp_tensor* res, const okp_tensor* out, int act, void* stream);

/* Element type conversion of a contiguous tensor of `count` elements (OKP_F32 <-> OKP_F16 / OKP_BF16, round to nearest even): the
 * boundary between the fp32 skip stream and an fp16 sub-network in the mixed configuration (KeypointNet(compute_dtype="float32mix")
 * runs the innermost hourglass levels, modules.py:25-66, in fp16). */
int okp_cast(int src_dtype, const void* src_dev, int dst_dtype, void* dst_dev, int64_t count, int32_t* range_flag_dev /* ABI 7; may be NULL */, void* stream);
/* range_flag_dev (conversions TO fp32, and okp_add_f16_f32): what these two make of fp16 tensors is read by split-product plans; with a flag
 * given they OR 1 into it when a value they write is outside the fp16 range or not finite (okp_conv_set_range_flag). */

/* out = act(a + b) on contiguous tensors of `count` elements, a fp16, b and out fp32: the closing add of a residual block whose branch
 * ran on the fp16 kernels while the skip is the fp32 stream itself (residual without projection, py_utils/utils.py:184-185). */
int okp_add_f16_f32(const void* a_f16_dev, const float* b_dev, float* out_dev, int64_t count, int act, int32_t* range_flag_dev /* ABI 7; may be NULL */, void* stream);

/* Stream `waiter` waits for everything enqueued on stream `signaller` so far (both on the current device): the fork / join of the hourglass'
 * up1 branch that runs on a side stream next to the low path (hg_module.forward, py_utils/modules.py:50-66 - the reference runs the two
 * one after the other).  Same semantics as an event record + stream wait, with an event that carries NO system-scope fence
 * (hipEventDisableSystemFence): the dependency is device-to-device, kernel boundaries already release / acquire at device scope, and the
 * cache writeback a default event adds to the signalling stream is what a fork costs the critical path.  Safe under stream capture. */
int okp_stream_wait_stream(void* waiter, void* signaller);

/* ------------------------------------------------------------------------------------
 * Frame packing for the 7x7/s2 stem: NCHW fp32 (n,3,h,w) -> NHWC4 `dtype` with a zero halo,
 * out dims (n, h+6, out_w, 4), image at (3,3).  The public input layout is the reference's
 * (perception/pipeline.py:24-28: frames N x 3 x 511 x 511 fp32).
 * ---------------------------------------------------------------------------------- */
int okp_pack_frames(int dtype, const float* frames_nchw_dev, int32_t n, int32_t h, int32_t w,
                    void* out_dev, int32_t out_w, void* stream);

/* Same packing fused with the reference's frame normalisation (perception/datasets/video.py:55-56,215):
 * uint8 RGB frames, NHWC (n,h,w,3), already resized/cropped  ->  ((u8 / 255) - mean[c]) / std[c]  in fp32 with the
 * reference's operation order (bit-exact with NumPy float32), then packed as above.  A quarter of the fp32 H2D bytes. */
int okp_pack_frames_u8(int dtype, const uint8_t* frames_nhwc_dev, int32_t n, int32_t h, int32_t w,
                       const float* mean3, const float* std3, void* out_dev, int32_t out_w, void* stream);

/* Raw camera frames to the stem input in one pass: uint8 RGB NHWC (n, src_h, src_w, 3) -> bilinear resize to
 * (resized_h, resized_w) -> crop (h, w) at (crop_y, crop_x) -> ((u8 / 255) - mean) / std -> packed as okp_pack_frames.
 * Replaces albumentations.SmallestMaxSize + CenterCrop + the normalisation of the reference's data path
 * (perception/datasets/video.py:95-96,215; 720x1280 -> 511x908 -> crop 511x511 at (0, 198)).  The resize restates
 * cv::resize(INTER_LINEAR) for 8-bit images (OpenCV 3.4 fixed-point arithmetic, 11-bit weights); cv2 is absent from the
 * build environment, so this entry point is pinned against the oracle's restatement and by properties only. */
int okp_preprocess_u8(int dtype, const uint8_t* frames_nhwc_dev, int32_t n, int32_t src_h, int32_t src_w,
                      int32_t resized_h, int32_t resized_w, int32_t crop_y, int32_t crop_x, int32_t h, int32_t w,
                      const float* mean3, const float* std3, void* out_dev, int32_t out_w, void* stream);

/* ------------------------------------------------------------------------------------
 * The 7x7 / stride-2 / pad-3 stem convolution 3 -> 128 + folded BatchNorm + ReLU in bf16 as its own kernel
 * (write-bound layer: whole 64-byte NHWC lines are stored straight from the MFMA accumulators).
 * Replaces hg.pre[0] = convolution(7, 3, 128, stride=2) (corner_net_lite/core/models/py_utils/utils.py:143-156,
 * CornerNet_Squeeze.py:84).  w: HOST fp32 [128][3][7][7] with BatchNorm folded, bias: HOST fp32 [128].
 * packed: the bf16 output of okp_pack_frames(_u8) for (n,h,w) frames: {data, h+6, out_w, pix_stride 4, bytes}.
 * out: NHWC bf16 view of size ((h-1)/2+1) x ((w-1)/2+1), >= 128 channels, 64-byte aligned pixels.
 * ---------------------------------------------------------------------------------- */
typedef struct okp_stem okp_stem;
okp_stem* okp_stem_create(const float* w_host, const float* bias_host);                       /* bf16 */
okp_stem* okp_stem_create_dtype(int dtype, const float* w_host, const float* bias_host);    /* OKP_BF16, OKP_F16, or OKP_F32X3: the split-product
                                                                                               form (fp32 NHWC output; okp_stem_forward_nchw only) */
int okp_stem_set_range_flag(okp_stem* stem, int32_t* flag_dev);   /* OKP_F32X3 stems: see okp_conv_set_range_flag */
void okp_stem_destroy(okp_stem* stem);
int okp_stem_forward(const okp_stem* stem, int32_t n, int32_t h, int32_t w, const okp_tensor* packed, const okp_tensor* out, void* stream);
/* Same layer straight from the reference's input layout, fp32 NCHW frames (n,3,h,w): the bf16 rounding and the zero
 * padding happen while the input patch is staged in LDS, so okp_pack_frames and its round trip through HBM are not needed.
 * OKP_F32X3 plans: `out` is an fp32 NHWC view; every product is the three-term fp16 split (fp32-grade, as OKP_F32X3 convolution plans). */
int okp_stem_forward_nchw(const okp_stem* stem, int32_t n, int32_t h, int32_t w, const float* frames_nchw_dev, const okp_tensor* out, void* stream);
/* OKP_F32X3 stems: the same launch with `out` written in PAIR FORMAT (okp_conv_args.src_pairs: [8 x fp16 hi | 8 x fp16 lo] per 8 channels
 * in the geometry of the fp32 tensor) - for a stem whose readers, conv1 and the projected skip of pre[1] (py_utils/utils.py:158-185), both
 * run on the patch-resident split-product kernel. */
int okp_stem_forward_nchw_pairs(const okp_stem* stem, int32_t n, int32_t h, int32_t w, const float* frames_nchw_dev, const okp_tensor* out, void* stream);

/* ------------------------------------------------------------------------------------
 * Final 1x1 convolutions of the three heads, NHWC -> NCHW fp32, optional sigmoid per output.
 * Replaces prediction_module[-1] (perception/models.py:17) for heat/depth/centre heads and
 * the deployed wrapper's sigmoid (scripts/package_model.py:28).
 * Output channel o reads 32 input channels starting at in_c_off[o]; w_dev is [n_out][32] fp32.
 * out_ptr[o] is the DEVICE address of plane (n=0) of that output; planes of successive frames
 * are out_n_stride[o] floats apart.
 * ---------------------------------------------------------------------------------- */
#define OKP_HEAD_MAX_OUT 32
typedef struct okp_head_out_args {
  int32_t n, h, w;
  okp_tensor src;
  int32_t n_out;
  int32_t in_c_off[OKP_HEAD_MAX_OUT];
  int32_t act[OKP_HEAD_MAX_OUT];
  float* out_ptr[OKP_HEAD_MAX_OUT];
  int64_t out_n_stride[OKP_HEAD_MAX_OUT];
  const float* w_dev;      /* [n_out][32] */
  const float* bias_dev;   /* [n_out] */
} okp_head_out_args;
int okp_head_out_forward(int dtype, const okp_head_out_args* args, void* stream);

/* The three prediction heads of one stack in one launch (bf16, 128 features): l1 = the fused 256 -> 384 first layers
 * (BN folded, ReLU), l2 = the block-diagonal 384 -> 96 second layers, `args` as for okp_head_out_forward (its `src` is
 * ignored: the 96-channel tensor never exists), x = the 256-channel backbone output.  Replaces prediction_module x 3
 * (perception/models.py:13-18,21-53) and the deployed wrapper's sigmoid (scripts/package_model.py:28).
 * OKP_F32X3 plans (ABI 6): the split-product form of the same launch; x is then a PAIR-FORMAT tensor (okp_conv_args.out_pairs: written
 * by the `cnvs` convolution in front of the heads), fp32 results. */
int okp_heads_forward(const okp_conv* l1, const okp_conv* l2, const okp_head_out_args* args, const okp_tensor* x, void* stream);

/* ------------------------------------------------------------------------------------
 * Per-map heat-map peak extraction.  Replaces KeypointExtractionComponent._extract_keypoints /
 * _compute_points (perception/pipeline.py:46-79) and nms (perception/models.py:55-58):
 *   box  = 5x5 ones convolution, zero padding, fp32 accumulation in row-major tap order
 *   peak = box == maxpool5x5(box) (padding ignored)  and  box > 0.5      [bit-exact contract]
 *   per peak (row-major order): confidence = sum of p over the clipped 5x5 window,
 *   (x, y) = sum(p * (x, y)) / confidence.
 * heat: [n_maps][h][w] fp32 (any height; w <= 2340: the map is walked in LDS-resident strips).  Outputs per map, capacity `cap` peaks:
 *   count[n_maps]        total peaks found (may exceed cap; only the first cap are stored)
 *   yx[n_maps][cap][2]   int32 (y, x)
 *   xyc[n_maps][cap][3]  fp32 (x, y, confidence)
 * ---------------------------------------------------------------------------------- */
int okp_peak_nms(const float* heat_dev, int32_t n_maps, int32_t h, int32_t w, int32_t cap,
                 int32_t* count_dev, int32_t* yx_dev, float* xyc_dev, void* stream);

/* nms(x, size) = x * (x == max_pool2d(x, size, stride 1, pad size/2)) on [n_maps][h][w] fp32 maps
 * (perception/models.py:55-58; padding never wins the max).  size must be odd, <= 15. */
int okp_nms_maxpool(const float* x_dev, int32_t n_maps, int32_t h, int32_t w, int32_t size,
                    float* out_dev, void* stream);

/* ------------------------------------------------------------------------------------
 * Geometry (fp64 on device).
 * okp_camera: pinhole + distortion.  model OKP_CAM_EQUIDISTANT: Kalibr "equidistant" = OpenCV fisheye, d = k1..k4
 * (FisheyeCamera, utils/camera_utils.py:64-81); model OKP_CAM_RADTAN: Kalibr "radtan" = OpenCV plumb-bob, d = k1, k2, p1, p2
 * (RadTanPinholeCamera, camera_utils.py:45-62: cv2.undistortPoints with P = K, five fixed-point iterations in OpenCV 3.4).
 * Every entry point that undistorts (okp_unproject_depth, okp_lift_peaks, okp_triangulate_dlt, okp_camera_undistort)
 * follows the camera's model.
 * ---------------------------------------------------------------------------------- */
enum { OKP_CAM_EQUIDISTANT = 0, OKP_CAM_RADTAN = 1 };
typedef struct okp_camera {
  double fx, fy, cx, cy;
  double d[4];
  int32_t model;
  int32_t reserved;
} okp_camera;

/* Replaces DetectionToPoint.__call__ (perception/pipeline.py:164-171) =
 * FisheyeCamera.undistort (utils/camera_utils.py:75-81; cv2.fisheye.undistortPoints with P=K)
 * -> round half-even -> clip to [0,max_x]x[0,max_y] -> z = depth[map][y][x]
 * -> PinholeCamera.unproject (camera_utils.py:31-34): K^-1 [xu, yu, 1]^T * z.
 * xy: [m][2] fp32 pixel coords; map_id: [m] int32 index into depth maps [n_maps][h][w] fp32;
 * out: [m][3] fp64.  Entries with map_id < 0 are skipped (out = NaN). */
int okp_unproject_depth(const okp_camera* cam, const float* xy_dev, const int32_t* map_id_dev, int32_t m,
                        const float* depth_dev, int32_t h, int32_t w, int32_t max_x, int32_t max_y,
                        double* out_dev, void* stream);

/* Device-resident form of the same lifting for a whole batch: consumes okp_peak_nms outputs in place
 * (no host round trip between NMS and 3D).  For map i, peak j < min(count[i], cap):
 * out[i][j] = (X, Y, Z, confidence) with (X,Y,Z) as in okp_unproject_depth using depth map i;
 * the remaining slots are filled with NaN.  out: [n_maps][cap][4] fp64. */
int okp_lift_peaks(const okp_camera* cam, const int32_t* count_dev, const float* xyc_dev, int32_t n_maps, int32_t cap,
                   const float* depth_dev, int32_t h, int32_t w, int32_t max_x, int32_t max_y,
                   double* out_dev, void* stream);

/* ------------------------------------------------------------------------------------
 * Object grouping on the device (batched form of ObjectExtraction.__call__, perception/pipeline.py:104-153):
 * the peaks of map 0 are object centres; every peak of map k >= 1 votes for the centre nearest to
 * (pixel centre + centre-offset map value at its rounded position), votes farther than `max_dist` (20 px in the
 * reference) are dropped; per object and keypoint type at most `type_count[k-1]` peaks are kept: for a type with
 * count 1 the most confident one; for a multi-instance type the first `max_sel` votes in peak order go to `sel`, and where an object
 * received MORE votes than the type has instances the reference's k-means reduction (pipeline.py:143-148: unseeded sklearn KMeans,
 * reproducible as a set only) is done here as a deterministic Lloyd iteration - one lane per vote (the object's first 64), fp64,
 * farthest-point initialisation seeded by each of the four most confident votes in turn, least inertia wins - into `reduced`.
 *   count [n][K] int32, xyc [n][K][cap][3] fp32 (okp_peak_nms outputs), centers [n][K-1][2][h][w] fp32
 *   n_obj   [n] int32                         objects per frame (= min(count[n][0], max_obj)); max_obj <= 64
 *   sel     [n][max_obj][K-1][max_sel] int32  selected peak indices into map k, -1 = empty slot
 *   n_votes [n][max_obj][K-1] int32           votes the object received for that type (before the cut)
 *   assign  [n][K][cap] int32                 object index each peak voted for (-1: none / dropped / map 0)
 *   pred    [n][K][cap][2] fp64               the predicted centre (x, y) each peak voted with
 *   reduced [n][max_obj][K-1][max_sel][2] fp32 (ABI 7; may be NULL = no reduction): for a type with type_count > 1 (<= max_sel) and
 *           n_votes > type_count the type_count cluster centres (x, y), NaN everywhere else
 * One wave per frame; K <= 8, max_sel <= 8.
 * ---------------------------------------------------------------------------------- */
int okp_group_objects(const int32_t* count_dev, const float* xyc_dev, const float* centers_dev, int32_t n, int32_t K,
                      int32_t cap, int32_t h, int32_t w, const int32_t* type_count /* HOST [K-1] */, float max_dist,
                      int32_t max_obj, int32_t max_sel, int32_t* n_obj_dev, int32_t* sel_dev, int32_t* n_votes_dev,
                      int32_t* assign_dev, double* pred_dev, float* reduced_dev, void* stream);

/* The fixed-capacity tensors can truncate where the reference (which keeps every peak, pipeline.py:73) cannot: bit 0 of flag[0] is set iff
 * some map has more than `cap` peaks or some centre map (map 0 of each frame of K maps) more than `max_obj`.  range_flag_dev (ABI 7, may be
 * NULL): the fp16-range flag of the split-product network that made the maps (okp_conv_set_range_flag); bit 1 of flag[0] is set iff it is
 * non-zero - one device-side word says whether a batch's results can be trusted.  No sync. */
int okp_capacity_overflow(const int32_t* count_dev, int32_t n_maps, int32_t K, int32_t cap, int32_t max_obj, const int32_t* range_flag_dev,
                          int32_t* flag_dev, void* stream);

/* Replaces StereoCamera.triangulate (utils/camera_utils.py:92-110) and the labelling tool's
 * 2-view DLT (scripts/label.py:285-305): undistort both views (P=K) -> optional Hartley-Sturm
 * correction against F (cv2.correctMatches) -> DLT null vector of the 4x4 system built from
 * P1 = K_l [I 0], P2 = K_r T_RL[:3] -> dehomogenise.  T_RL: row-major 3x4; F: row-major 3x3
 * (used only when correct_matches != 0).  left/right: [m][2] fp32;  out: [m][3] fp64 (left frame). */
int okp_triangulate_dlt(const okp_camera* left, const okp_camera* right, const double* T_RL,
                        const double* F, int correct_matches,
                        const float* left_xy_dev, const float* right_xy_dev, int32_t m,
                        double* out_dev, void* stream);

/* Undistort only (FisheyeCamera.undistort, camera_utils.py:75-81 / RadTanPinholeCamera.undistort, :57-62, by cam->model):
 * xy [m][2] fp32 -> out [m][2] fp64.  okp_fisheye_undistort is the same entry point under its round-1 name. */
int okp_camera_undistort(const okp_camera* cam, const float* xy_dev, int32_t m, double* out_dev, void* stream);
int okp_fisheye_undistort(const okp_camera* cam, const float* xy_dev, int32_t m, double* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OKP_H */

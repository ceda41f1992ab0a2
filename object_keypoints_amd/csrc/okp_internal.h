// Internal declarations shared by the HIP translation units of libokp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "okp.h"

#define OKP_MAX_TAPS 32

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// The two 16-bit activation / weight types (OKP_BF16, OKP_F16).  Both feed the same MFMA shapes at the same rate
// (16x16x32 and 32x32x16, fp32 accumulate); everything that differs between them - the builtin, the vector types and the
// conversions - is behind this trait, so that the 16-bit kernels are written once and instantiated for both.
template <typename T> struct H16;
template <> struct H16<__bf16> {
  using x2 = bf16x2; using x4 = bf16x4; using x8 = bf16x8;
  static constexpr int kDtype = OKP_BF16;
  // (the host pass of hipcc parses the kernels too: the builtins exist for the device only)
  static __device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#else
    return c;
#endif
  }
  static __device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#else
    return c;
#endif
  }
  // the two elements of a packed dword as fp32 (bf16 is the upper half of an fp32: one shift / one mask)
  static __device__ __forceinline__ float lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
  static __device__ __forceinline__ float hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
};
template <> struct H16<_Float16> {
  using x2 = f16x2; using x4 = f16x4; using x8 = f16x8;
  static constexpr int kDtype = OKP_F16;
  static __device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#else
    return c;
#endif
  }
  static __device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#else
    return c;
#endif
  }
  static __device__ __forceinline__ float lo(uint32_t w) { return (float)__builtin_bit_cast(f16x2, w)[0]; }
  static __device__ __forceinline__ float hi(uint32_t w) { return (float)__builtin_bit_cast(f16x2, w)[1]; }
};

// Split-product operands (OKP_F32X3; struct F32S in okp_igemm_kernel.h): 4 fp32 held by a lane -> their fp16 halves hi = fp16(x),
// lo = fp16(x - hi), two packed dwords each (the arithmetic of okp_split8: cvt_pk, two exact v_fma_mix differences, cvt_pk)
static __device__ __forceinline__ void okp_split4(const f32x4& x, u32x2& hi, u32x2& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float xa = x[2 * q], xb = x[2 * q + 1];
    f16x2 h2; h2[0] = (_Float16)xa; h2[1] = (_Float16)xb;
    const uint32_t hw = __builtin_bit_cast(uint32_t, h2);
    float da, db;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(da) : "v"(hw), "v"(xa));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(db) : "v"(hw), "v"(xb));
    f16x2 l2; l2[0] = (_Float16)da; l2[1] = (_Float16)db;
    hi[q] = hw; lo[q] = __builtin_bit_cast(uint32_t, l2);
  }
#else
  hi = u32x2{0u, 0u}; lo = u32x2{0u, 0u};
#endif
}

// OKP_F32X3 range guard: a value that a split-product consumer will halve (hi = fp16(x)) must fit fp16, |x| <= 65504 - beyond it hi is
// inf, lo = fp16(x - hi) is -inf and the three-term product is NaN, which the next ReLU (fmaxf) silently turns into 0.  Every kernel that
// PRODUCES such a value (an output tensor of a split-product plan, the squeeze tile of the one-launch fire module, the hidden layer of the
// heads) keeps a running MAXIMUM of the magnitudes it hands on (okp_range_max: one v_max / half a v_max3 per value) and raises the plan's
// flag (okp_conv_set_range_flag) when it passes 65504.  A maximum does not see NaN, and does not have to: with every input in range (their
// producers checked them: induction over the layers) a convolution's sums are finite - 2 304 products below 65504^2 - so infinity is the only
// way out of range.  Data that enters from OUTSIDE the split-product kernels is checked exactly, NaN included (okp_unsplittable): the frames
// the stem reads, and what okp_cast / okp_add_f16_f32 make of fp16 tensors for a split-product consumer (the mixed configuration).  (The
// patch-resident kernel keeps the exact per-value form as a lane mask in scalar registers: free there, where a maximum in a vector register
// cost 1.4 %; the fire module and the stem are the other way round - 5-7 % with the mask, nothing measurable with the maximum.)
#ifdef OKP_NO_RANGE_GUARD      // experiment build: what the guard costs (A/B against the committed kernels)
static __device__ __forceinline__ bool okp_unsplittable(float) { return false; }
static __device__ __forceinline__ float okp_range_max(float m, float) { return m; }
#else
static __device__ __forceinline__ bool okp_unsplittable(float v) { return !(__builtin_fabsf(v) <= 65504.f); }     // (true for NaN)
static __device__ __forceinline__ float okp_range_max(float m, float v) { return __builtin_fmaxf(m, __builtin_fabsf(v)); }
#endif
static __device__ __forceinline__ bool okp_range_exceeded(float m) { return !(m <= 65504.f); }
static __device__ __forceinline__ void okp_raise_range_flag(int32_t* flag, bool bad) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (flag != nullptr && bad) atomicOr(flag, 1);     // (no lane takes this branch in a healthy launch)
#endif
}

template <typename T> __device__ __forceinline__ uint32_t okp_pack2(float a, float b) {     // two fp32 -> one packed dword (RNE)
  // ONE vector conversion (v_cvt_pk_{bf16,f16}_f32 d, a, b).  Two scalar conversions packed afterwards are the same instruction only as long as
  // nothing else touches the dword: with a packed integer operation behind it (the 16-bit ReLU of okp_fire2's epilogue) hipcc converted each
  // value by itself and joined the halves with a v_perm_b32 - three instructions per pair
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_){a, b}, typename H16<T>::x2));
}
inline int okp_esz(int dtype) { return (dtype == OKP_F32 || dtype == OKP_F32X3) ? 4 : 2; }
inline bool okp_is16(int dtype) { return dtype == OKP_BF16 || dtype == OKP_F16; }

// One K-slice of the implicit GEMM = 128 bytes of K per row (64 bf16 / 32 fp32):
// chunks 0-3 (64 B) come from tap_lo at channel c0_lo, chunks 4-7 from tap_hi at c0_hi.
struct OkpSlice {
  uint8_t tap_lo, tap_hi, src, nvalid;   // nvalid: number of 16-B chunks that carry data (1..8)
  int32_t c0_lo, c0_hi;                  // first channel (elements) of each half
  int32_t pad;                           // OKP_F32X3 plans: 1 = this slice's products are single-term (x_hi * w_hi), 0 = three terms
};

struct OkpTapDev {
  int32_t dy, dx, src, pad;
};

// x / d for 0 <= x < 2^31 and a divisor fixed per launch: q = umulhi(x, mul) >> shift (mul == 0: d == 1).
struct OkpFastDiv {
  uint32_t mul, shift;
};
inline OkpFastDiv okp_fastdiv(uint32_t d) {
  OkpFastDiv f{0u, 0u};
  if (d <= 1) return f;
  uint32_t s = 0;
  while ((1ull << s) < d) ++s;                   // ceil(log2 d) >= 1
  f.mul = (uint32_t)((1ull << (31 + s)) / d + 1); // < 2^32 because d > 2^(s-1)
  f.shift = s - 1;
  return f;
}

struct OkpIgemmParams {
  const void* src[2];
  uint32_t src_bytes[2];
  int32_t srcH[2], srcW[2], src_pix_stride[2], conv_stride[2];
  const void* weights;     // [n_slices][cout_pad][128 B]
  uint32_t w_bytes;
  int32_t cout_pad;
  const float* bias;       // [cout_pad]
  const float* oscale;     // OKP_F32X3: per-output-channel factor applied to the accumulator before the bias (NULL otherwise)
  const OkpSlice* slices;
  int32_t n_slices;
  int32_t n_taps;
  int32_t N, Ho, Wo;
  OkpFastDiv div_howo, div_wo;
  void* out;
  uint32_t out_bytes, res_bytes;
  int32_t OH, OW, out_step, out_oy, out_ox, out_pix_stride;
  int32_t cout;
  const void* res;
  int32_t res_pix_stride;
  int32_t act;
  int32_t n_co_tiles;
  const float* dw_w;       // fused depth-wise branch (NULL = off)
  const float* dw_bias;
  void* dw_out;
  const void* dw_res;
  int32_t dw_out_pix_stride, dw_res_pix_stride;
  int32_t n_tiles;
  int32_t n_classes, tiles_per_class, slices_per_class;
  int32_t n_single_slices; // OKP_F32X3: leading single-term K-slices (the plan sorts them first)
  void* out16;             // OKP_F32X3: optional fp16 copy of the result (same pixel mapping as out); out may then be NULL
  int32_t out16_pix_stride, res16;   // res16: the residual tensor is fp16
  int32_t src_pairs, out_pairs;      // OKP_F32X3 on the patch-resident kernel: sources (bit s) / output in pair format (okp_conv_args.src_pairs / out_pairs)
  int32_t out_sub2, OH2, OW2;        // OKP_F32X3: the fp32 output keeps even rows / columns only (tensor of OH2 x OW2 pixels); out16 is full size
  int32_t mfma32;                    // tile 14: the patch-resident 16-bit kernel on 32x32x16 MFMAs (experiment)
  int32_t* range_flag;               // OKP_F32X3: the plan's range flag (okp_conv_set_range_flag) or NULL
  int64_t src_bytes64[2];            // the sources' byte spans in full (src_bytes is clamped to 32 bits): a source of the split-product patch kernel may pass 2 GiB
  OkpTapDev taps[OKP_MAX_TAPS];
};

// okp_igemm_patch.hip: 256 x (16x16 px) tiles whose input patch (halo included) stays in LDS for all taps of a
// 64-channel chunk; one step = one tap of one chunk = one 128-byte K-slice of the packed weights.
struct OkpPatchStep {          // 16 bytes, built at plan creation
  uint32_t tap_bytes;          // byte offset of this step's tap inside the patch: ((dy - oy) * 18 + (dx - ox)) * 128 (row pitch 18 px)
  uint32_t nx_c0b;             // channel byte offset of the NEXT group's patch (prefetched during this group's steps)
  uint8_t pbuf, nx_k0, nx_k1, nx_geom;  // patch buffer of this step (bit 0; split-product plans: OKP_PSTEP_CVT_* in bits 1-2); passes [k0, k1) of the next patch (geometry nx_geom) issued in this step
  uint8_t geom, tx, c0q, grp_last;   // geometry of this step's patch; tx = patch column offset of the tap (the fragment swizzle is keyed on
                               // the column); c0q = first channel / 64 of the step's chunk; grp_last = last step of its (chunk, geometry) group
};
// Split-product plans (okp_igemm_patch_x3.hip): a patch arrives in LDS as fp32 and is split ONCE, in place, into fp16 hi | lo pairs.
#define OKP_PSTEP_CVT_SELF 2u  // this step's patch finished landing in the previous step (or the tile's prologue): split it now, then a second barrier
#define OKP_PSTEP_CVT_NEXT 4u  // last step of a group of >= 2 steps: the next group's patch has landed (requested in steps 0 .. n-2), split it during this step
#define OKP_PATCH_MAX_GEOM 6
static_assert(OKP_PATCH_MAX_GEOM * 5 <= 32, "OkpPatchParams.geom_ph packs 5 bits per geometry into one u32");
struct OkpPatchGeom {          // one patch geometry: a source + the sub-lattice / window its taps read (a stride-2 3x3 has four: the parity classes)
  const void* data; uint32_t bytes;
  int32_t H, W, pix_stride;
  int32_t PW, npx;             // valid patch columns; rows * 18 (pixels of the LDS image, pitch 18)
  int32_t oy, ox;              // offset of patch pixel (0, 0) relative to conv_stride * (tile origin)
  int32_t step, conv_stride;   // source pixels per patch pixel (the conv stride for strided sources, else 1)
};
struct OkpPatchParams {
  OkpPatchGeom g[OKP_PATCH_MAX_GEOM];
  int32_t n_geom;
  uint32_t geom_ph, geom_src;  // per geometry g: patch rows in bits 5 g .. 5 g + 4; source index in bit g (what the in-loop requests need, in two registers)
  const void* src_data[2]; uint32_t src_bytes[2];
  const void* weights; uint32_t w_bytes; int32_t cout_pad, cout;
  const float* bias;
  const float* oscale;         // OKP_F32X3 plans: per-output-channel factor of the accumulator (1 / the power of two its weights were scaled with)
  const OkpPatchStep* steps; int32_t n_steps;
  int32_t N, H, W, tiles_y, tiles_x;            // H x W = GEMM pixel grid (Ho x Wo)
  int32_t n_classes, steps_per_class, tiles_per_class;   // sub-pixel classes of a transposed convolution (1 otherwise)
  int32_t OH, OW, out_step, out_oy, out_ox;     // GEMM pixel (y, x) of class c -> output pixel (y * out_step + out_oy + (c >> 1), x * out_step + out_ox + (c & 1))
  OkpFastDiv div_tiles_frame, div_tiles_x;
  void* out; uint32_t out_bytes; int32_t out_pix_stride;
  const void* res; uint32_t res_bytes; int32_t res_pix_stride;
  int32_t act, n_co_tiles, n_tiles;
  uint32_t pairs;              // okp_igemm_patch_x3.hip: bit s = source s arrives in pair format (its patches are not split), bit 2 = out is written in pair format
  int32_t* range_flag;         // okp_igemm_patch_x3.hip: the plan's range flag or NULL
  int64_t src_total_bytes[2];  // okp_igemm_patch_x3.hip: a source's whole byte span (may pass 2 GiB) and the bytes of one frame of it: the kernel builds
  int64_t src_frame_bytes[2];  // its buffer resources per tile from the tile's frame, offsets are frame-relative
#ifdef OKP_PATCH_STAMPS
  uint32_t* dbg;
#endif
};

struct OkpFire2Params {       // okp_fire2.hip: streaming fire module, 256 -> 128 -> 256, stride 1
  const void* x; uint32_t x_bytes; int32_t H, W, x_ps;
  void* out; uint32_t out_bytes; int32_t Ho, Wo, out_ps;    // Ho x Wo = ceil(H / stride) x ceil(W / stride)
  int32_t N, skip;
  const void* w1; int32_t w1_cout_pad; const float* b1;     // squeeze plan: packed [slice][cout_pad][128 B] bf16
  const void* wa; int32_t wa_cout_pad; const float* ba;     // expand plan
  const float* wd; const float* bd;                         // depth-wise [9][128] fp32, bias [128]
  const float* s1; const float* sa;                         // okp_fire_x3.hip (split-product plans): output scales of the squeeze / expand plan
  int32_t* range_flag;                                      // okp_fire_x3.hip: the squeeze plan's range flag or NULL
  int32_t SH, SW, IH, IW, IP, RPR, tiles_y, tiles_x, n_tiles;
  OkpFastDiv div_tiles_frame, div_tiles_x, div_sw, div_iw, div_rpr;
#ifdef OKP_FIRE_STAMPS
  uint32_t* dbg;               // debug build: shader-clock stamps at the phase boundaries of every workgroup's second tile
#endif
};

struct OkpFireChainModule {   // okp_fire_chain.hip
  const void* w1; int32_t w1_cout_pad; const float* b1;
  const void* wa; int32_t wa_cout_pad; const float* ba;
  const float* wd; const float* bd;
};
struct OkpFireChainParams {
  const void* x; void* out; int32_t x_ps, out_ps, H, W, n_modules;
  int32_t He, We;               // EE form (entry + exit modules): size of the map the stride-2 entry module reads
  OkpFireChainModule mod[OKP_FIRE_CHAIN_MAX];
#ifdef OKP_FIRE_STAMPS
  uint32_t* dbg;                // debug build: shader-clock stamps at the phase boundaries of the chain's second module
#endif
};

struct okp_conv {
  int dtype;
  int n_src;
  int32_t cin[2];
  int32_t conv_stride[2];
  int32_t cout, cout_pad;
  int32_t n_taps;
  OkpTapDev taps[OKP_MAX_TAPS];
  int32_t n_slices;
  int32_t n_single_slices;   // OKP_F32X3 plans built with tap_terms: leading K-slices whose products are single-term
  int act;
  void* weights_dev;
  uint32_t w_bytes;
  float* bias_dev;
  float* oscale_dev;       // OKP_F32X3 plans: 1 / (power-of-two scale the channel's weights were multiplied with before the fp16 split)
  OkpSlice* slices_dev;
  // patch-resident kernel (okp_igemm_patch.hip): step table + per-source patch geometry, or patch_steps_dev == NULL
  OkpPatchStep* patch_steps_dev;
  int32_t patch_n_geom;
  int32_t patch_src[OKP_PATCH_MAX_GEOM], patch_PW[OKP_PATCH_MAX_GEOM], patch_PH[OKP_PATCH_MAX_GEOM], patch_oy[OKP_PATCH_MAX_GEOM], patch_ox[OKP_PATCH_MAX_GEOM], patch_step[OKP_PATCH_MAX_GEOM];
  void* frag_dev;          // single-tap 16-bit plans: weights re-laid in MFMA-fragment order (built by okp_conv_create; NULL otherwise)
  int32_t* range_flag;     // OKP_F32X3 plans: device flag raised by this plan's launches when a result leaves the fp16 range (okp_conv_set_range_flag); NULL = off
  void* fragT_dev;         // the same with the channel-as-ROW order of okp_fire2 (row i of block b = channel 32 w + 8 (i >> 2) + 4 b + (i & 3));
                           // single-tap OKP_F32X3 plans: [cout / 16][hi | lo][cin / 32][lane][8 fp16], row i of wave w = channel 16 w + i (okp_fire_x3.hip)
};

void okp_set_error(const char* fmt, ...);
uint16_t okp_f32_to_16(int dtype, float f);          // host: fp32 -> bf16 / fp16 bits, round to nearest even
int okp_check_hip(hipError_t e, const char* what);

// launchers implemented in the .hip files
int okp_select_tile(int dtype, int cout_pad, long pixels);
int okp_launch_igemm(const okp_conv* plan, const OkpIgemmParams& p, int tile, hipStream_t stream);
int okp_ensure_frags(const okp_conv* plan, hipStream_t stream);   // okp_fire_chain.hip: fragment-order weight copy of a 1x1 plan
bool okp_fire2_supported(int cin, int mid, int half, int stride);
int okp_launch_fire2(int dtype, const OkpFire2Params& p, int cin, int mid, int stride, hipStream_t stream);
bool okp_fire_x3_supported(int cin, int mid, int half, int stride, int skip);     // okp_fire_x3.hip: one-launch fire module of split-product plans
int okp_launch_fire_x3(OkpFire2Params p, int stride, hipStream_t stream);
bool okp_patch_supported(const okp_conv* plan, const OkpIgemmParams& p);   // okp_igemm_patch.hip
int okp_launch_igemm_patch(const okp_conv* plan, const OkpIgemmParams& p, hipStream_t stream);
int okp_fill_patch_params(const okp_conv* plan, const OkpIgemmParams& q, OkpPatchParams& p);             // okp_igemm_patch.hip (shared by both patch kernels)
int okp_launch_igemm_patch_x3(const okp_conv* plan, const OkpPatchParams& p, hipStream_t stream);      // okp_igemm_patch_x3.hip

// HBM-bound kernels around the implicit GEMM: depth-wise 3x3 (fire-module tail), frame packing
// for the stem, and the final 1x1 of the heads (NHWC -> NCHW fp32 + sigmoid).
// All of them move 16 bytes per lane per access and keep channels contiguous across lanes.
#include <algorithm>

#include "okp_internal.h"

namespace {

template <typename T> struct Vec;      // 16 bytes of channels
template <> struct Vec<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const char* p, float (&v)[4]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = a[e];
  }
  static __device__ __forceinline__ void store(char* p, const float (&v)[4]) {
    f32x4 a = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = a;
  }
};
template <typename T> struct Vec16 {
  using x8 = typename H16<T>::x8;
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const char* p, float (&v)[8]) {
    const x8 a = *reinterpret_cast<const x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
  }
  static __device__ __forceinline__ void store(char* p, const float (&v)[8]) {
    x8 a;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (T)v[e];
    *reinterpret_cast<x8*>(p) = a;
  }
};
template <> struct Vec<__bf16> : Vec16<__bf16> {};
template <> struct Vec<_Float16> : Vec16<_Float16> {};
template <typename T> struct X8 { using t = typename H16<T>::x8; };
template <> struct X8<float> { using t = bf16x8; };          // never used for fp32; only has to name a type

struct DwParams {
  const void* src; uint32_t src_bytes; int32_t H, W, src_ps;
  const float* w; const float* bias;
  const void* res; int32_t res_ps;
  void* out; int32_t Ho, Wo, out_ps;
  int32_t N, C, stride, act;
};

// A thread owns one 16-byte channel group (its 9 x VN weights stay in registers) and SEG = 8 consecutive output
// pixels, processed in groups of G = 4: with stride 1 and a group inside one image row the 3 x (G+2) taps are loaded
// once and slid across the G outputs.  Zero padding is branch-free (out-of-range buffer offsets return 0), so all taps
// of a group are in flight together.  No MFMA accumulators here, so the kernel runs at 3-4 waves per SIMD; the
// host runs it on a side stream next to the expand GEMM of the same fire module.
template <typename T>
__global__ __launch_bounds__(256) void okp_dwconv3x3_kernel(const DwParams p) {
  constexpr int VN = Vec<T>::N;
  constexpr int ESZ = (int)sizeof(T);
  constexpr int SEG = 8, G = 4;
  constexpr uint32_t kInvalidOff = 0x80000000u;
  // the 9 x C weights live in LDS (36 B per channel): registers are kept for the tap window, 3+ waves per SIMD
  extern __shared__ __attribute__((aligned(16))) float sw[];
  for (int i = threadIdx.x; i < 9 * p.C; i += 256) sw[i] = p.w[i];
  __syncthreads();
  const int CG = p.C / VN;
  const int PL = 256 / CG;
  const int cq = threadIdx.x % CG, pl = threadIdx.x / CG;
  if (pl >= PL) return;
  const int ch = cq * VN;
  const long P = (long)p.N * p.Ho * p.Wo;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, (int)p.src_bytes, 0x00020000);
  float breg[VN];
#pragma unroll
  for (int e = 0; e < VN; ++e) breg[e] = p.bias[ch + e];
  const int cs = p.stride, H = p.H, W = p.W, ps = p.src_ps;
  const bool slide = cs == 1 && (p.Wo % G) == 0;
  auto tap_off = [&](int n, int hi, int wi) -> uint32_t {
    const bool ok = hi >= 0 && hi < H && wi >= 0 && wi < W;
    return ok ? (uint32_t)(((n * H + hi) * W + wi) * ps + ch) * (uint32_t)ESZ : kInvalidOff;
  };
  auto to_f = [&](const u32x4& raw, float (&x)[VN]) {
    if constexpr (ESZ == 2) {
      const typename X8<T>::t xv = __builtin_bit_cast(typename X8<T>::t, raw);
#pragma unroll
      for (int e = 0; e < VN; ++e) x[e] = (float)xv[e];
    } else {
      const f32x4 xv = __builtin_bit_cast(f32x4, raw);
#pragma unroll
      for (int e = 0; e < VN; ++e) x[e] = xv[e];
    }
  };
  auto finish = [&](float (&v)[VN], int n, int ho, int wo) {
    const size_t opix = ((size_t)n * p.Ho + ho) * p.Wo + wo;
    if (p.res) {
      float r[VN];
      Vec<T>::load(static_cast<const char*>(p.res) + (opix * p.res_ps + ch) * ESZ, r);
#pragma unroll
      for (int e = 0; e < VN; ++e) v[e] += r[e];
    }
    if (p.act == OKP_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < VN; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    Vec<T>::store(static_cast<char*>(p.out) + (opix * p.out_ps + ch) * ESZ, v);
  };
  for (long base = ((long)blockIdx.x * PL + pl) * SEG; base < P; base += (long)gridDim.x * PL * SEG) {
#pragma unroll
    for (int g0 = 0; g0 < SEG; g0 += G) {
      const long pix0 = base + g0;
      if (pix0 >= P) break;
      if (slide) {                                  // Wo % G == 0  =>  the whole group is in range and in one row
        const int wo0 = (int)(pix0 % p.Wo);
        const long t = pix0 / p.Wo;
        const int ho = (int)(t % p.Ho), n = (int)(t / p.Ho);
        u32x4 win[3][G + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int cc = 0; cc < G + 2; ++cc)
            win[r][cc] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)tap_off(n, ho + r - 1, wo0 + cc - 1), 0, 0);
#pragma unroll
        for (int k = 0; k < G; ++k) {
          float v[VN];
#pragma unroll
          for (int e = 0; e < VN; ++e) v[e] = breg[e];
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) {
              float x[VN];
              to_f(win[r][k + c3], x);
#pragma unroll
              for (int e = 0; e < VN; ++e) v[e] = fmaf(x[e], sw[(r * 3 + c3) * p.C + ch + e], v[e]);
            }
          finish(v, n, ho, wo0 + k);
        }
      } else {
        for (int k = 0; k < G; ++k) {
          const long pix = pix0 + k;
          if (pix >= P) break;
          const int wo = (int)(pix % p.Wo);
          const long t = pix / p.Wo;
          const int ho = (int)(t % p.Ho), n = (int)(t / p.Ho);
          u32x4 tapv[9];
#pragma unroll
          for (int q = 0; q < 9; ++q)
            tapv[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)tap_off(n, ho * cs + q / 3 - 1, wo * cs + q % 3 - 1), 0, 0);
          float v[VN];
#pragma unroll
          for (int e = 0; e < VN; ++e) v[e] = breg[e];
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            float x[VN];
            to_f(tapv[q], x);
#pragma unroll
            for (int e = 0; e < VN; ++e) v[e] = fmaf(x[e], sw[q * p.C + ch + e], v[e]);
          }
          finish(v, n, ho, wo);
        }
      }
    }
  }
}

// NCHW fp32 frames -> NHWC4 with a zero halo of 3 (the 7x7 stem reads 8 pixels x 4 channels =
// one contiguous K half-slice per kernel row).  One lane per output pixel: three strided plane
// reads (coalesced along x) and one 8/16-byte store.
template <typename T>
__global__ __launch_bounds__(256) void okp_pack_frames_kernel(const float* __restrict__ in, int N, int H, int W,
                                                               T* __restrict__ out, int OHt, int OWt) {
  const long total = (long)N * OHt * OWt;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % OWt);
    const long t = idx / OWt;
    const int y = (int)(t % OHt);
    const int n = (int)(t / OHt);
    const int sy = y - 3, sx = x - 3;
    float v[3] = {0.f, 0.f, 0.f};
    if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
      const float* base = in + ((size_t)n * 3 * H + sy) * W + sx;
      v[0] = base[0];
      v[1] = base[(size_t)H * W];
      v[2] = base[2 * (size_t)H * W];
    }
    T* o = out + idx * 4;
    o[0] = (T)v[0]; o[1] = (T)v[1]; o[2] = (T)v[2]; o[3] = (T)0.f;
  }
}

struct NormParams { float mean[3], stdv[3]; };

template <typename T>
__global__ __launch_bounds__(256) void okp_pack_frames_u8_kernel(const uint8_t* __restrict__ in, int N, int H, int W, NormParams np,
                                                                  T* __restrict__ out, int OHt, int OWt) {
  const long total = (long)N * OHt * OWt;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % OWt);
    const long t = idx / OWt;
    const int y = (int)(t % OHt);
    const int n = (int)(t / OHt);
    const int sy = y - 3, sx = x - 3;
    float v[3] = {0.f, 0.f, 0.f};
    if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
      const uint8_t* px = in + (((size_t)n * H + sy) * W + sx) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c)      // (u8 / 255 - mean) / std, one IEEE rounding per operation as NumPy float32 does
        v[c] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)px[c], 255.0f), np.mean[c]), np.stdv[c]);
    }
    T* o = out + idx * 4;
    o[0] = (T)v[0]; o[1] = (T)v[1]; o[2] = (T)v[2]; o[3] = (T)0.f;
  }
}


// ---- resize (bilinear, 8-bit fixed point) + centre crop + normalise + pack, raw camera frames -> stem input ----------
// Restates the arithmetic of cv::resize(INTER_LINEAR) for CV_8U as published in OpenCV 3.4 (imgproc/resize.cpp:
// resizeGeneric_ with HResizeLinear<uchar,int,short,2048> and VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>),
// which is what the reference's data path runs through albumentations.SmallestMaxSize + CenterCrop
// (perception/datasets/video.py:95-96): per destination column  fx = (float)((dx + 0.5) * scale - 0.5),
// sx = floor(fx), weights round((1 - fx) * 2048), round(fx * 2048) as int16; rows alike; then
//   dst = ( ((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2 ) >> 2   with  h = S[sx] * a0 + S[sx + 1] * a1.
// cv2 is not importable in the build container: PARITY UNPINNED against OpenCV itself; pinned against the oracle's
// restatement (oracle/preprocess.py, integer arithmetic: bit-exact) and by properties (tests/test_oracle_preprocess.py).
struct ResizeParams {
  int srcH, srcW, rsH, rsW;       // source frame, resized frame (before the crop)
  int cropY, cropX, H, W;         // crop origin inside the resized frame, cropped size (the network's input size)
  double scaleY, scaleX;          // srcH / rsH, srcW / rsW as cv::resize derives them: 1.0 / ((double)rs / src)
  NormParams np;
};

__device__ __forceinline__ void resize_coeff(int d, double scale, int ssize, int& s0, int& s1, int& w0, int& w1) {
  float f = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  s0 = s;
  s1 = s + 1 < ssize ? s + 1 : s;                    // weight 0 there (f == 0) or the in-range neighbour
  w0 = __float2int_rn(__fmul_rn(__fsub_rn(1.f, f), 2048.f));   // saturate_cast<short>(cvRound(.)): values are within [0, 2048]
  w1 = __float2int_rn(__fmul_rn(f, 2048.f));
}

template <typename T>
__global__ __launch_bounds__(256) void okp_preprocess_u8_kernel(const uint8_t* __restrict__ in, int N, ResizeParams rp,
                                                                 T* __restrict__ out, int OHt, int OWt) {
  const long total = (long)N * OHt * OWt;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % OWt);
    const long t = idx / OWt;
    const int y = (int)(t % OHt);
    const int n = (int)(t / OHt);
    const int cy = y - 3, cx = x - 3;               // position inside the cropped frame
    float v[3] = {0.f, 0.f, 0.f};
    if (cy >= 0 && cy < rp.H && cx >= 0 && cx < rp.W) {
      int sy0, sy1, b0, b1, sx0, sx1, a0, a1;
      resize_coeff(cy + rp.cropY, rp.scaleY, rp.srcH, sy0, sy1, b0, b1);
      resize_coeff(cx + rp.cropX, rp.scaleX, rp.srcW, sx0, sx1, a0, a1);
      const uint8_t* r0 = in + ((size_t)n * rp.srcH + sy0) * rp.srcW * 3;
      const uint8_t* r1 = in + ((size_t)n * rp.srcH + sy1) * rp.srcW * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int h0 = (int)r0[sx0 * 3 + c] * a0 + (int)r0[sx1 * 3 + c] * a1;
        const int h1 = (int)r1[sx0 * 3 + c] * a0 + (int)r1[sx1 * 3 + c] * a1;
        const int u = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        const int u8 = u < 0 ? 0 : (u > 255 ? 255 : u);
        v[c] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)u8, 255.0f), rp.np.mean[c]), rp.np.stdv[c]);
      }
    }
    T* o = out + idx * 4;
    o[0] = (T)v[0]; o[1] = (T)v[1]; o[2] = (T)v[2]; o[3] = (T)0.f;
  }
}

struct HeadParams {
  const void* src; int32_t src_ps;
  int32_t N, HW, n_out;
  const float* w; const float* bias;
  int32_t in_c_off[OKP_HEAD_MAX_OUT];
  int32_t act[OKP_HEAD_MAX_OUT];
  float* out_ptr[OKP_HEAD_MAX_OUT];
  int64_t out_n_stride[OKP_HEAD_MAX_OUT];
};

// One lane per pixel; outputs are written plane by plane, so stores are contiguous across lanes.
template <typename T>
__global__ __launch_bounds__(256) void okp_head_out_kernel(const HeadParams p) {
  constexpr int VN = Vec<T>::N;
  constexpr int ESZ = (int)sizeof(T);
  __shared__ float sw[OKP_HEAD_MAX_OUT * 32];
  __shared__ float sb[OKP_HEAD_MAX_OUT];
  for (int i = threadIdx.x; i < p.n_out * 32; i += blockDim.x) sw[i] = p.w[i];
  for (int i = threadIdx.x; i < p.n_out; i += blockDim.x) sb[i] = p.bias[i];
  __syncthreads();
  const long total = (long)p.N * p.HW;
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long)gridDim.x * blockDim.x) {
    const int n = (int)(pix / p.HW);
    const int hw = (int)(pix - (long)n * p.HW);
    const char* row = static_cast<const char*>(p.src) + (size_t)pix * p.src_ps * ESZ;
    int cached_off = -1;
    float x[32];
    for (int o = 0; o < p.n_out; ++o) {
      const int off = p.in_c_off[o];
      if (off != cached_off) {           // outputs of one head share their 32 input channels
#pragma unroll
        for (int k = 0; k < 32 / VN; ++k) {
          float t[VN];
          Vec<T>::load(row + (size_t)(off + k * VN) * ESZ, t);
#pragma unroll
          for (int e = 0; e < VN; ++e) x[k * VN + e] = t[e];
        }
        cached_off = off;
      }
      float acc = sb[o];
#pragma unroll
      for (int k = 0; k < 32; ++k) acc = fmaf(x[k], sw[o * 32 + k], acc);
      if (p.act[o] == OKP_ACT_SIGMOID) acc = 1.f / (1.f + expf(-acc));
      else if (p.act[o] == OKP_ACT_RELU) acc = fmaxf(acc, 0.f);
      p.out_ptr[o][(size_t)n * p.out_n_stride[o] + hw] = acc;
    }
  }
}

int grid_for(long total, int block) {
  long g = (total + block - 1) / block;
  if (g > 256L * 16) g = 256L * 16;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int okp_dwconv3x3_forward(int dtype, int32_t n, int32_t c, int32_t conv_stride, const okp_tensor* src,
                                     const float* w_dev, const float* bias_dev, const okp_tensor* res,
                                     const okp_tensor* out, int act, void* stream) {
  if (!src || !out || !src->data || !out->data || !w_dev || !bias_dev) { okp_set_error("okp_dwconv3x3_forward: null argument"); return OKP_EINVAL; }
  const int esz = okp_esz(dtype);
  const int vn = 16 / esz;
  if (dtype != OKP_F32 && !okp_is16(dtype)) { okp_set_error("okp_dwconv3x3_forward: bad dtype %d", dtype); return OKP_EINVAL; }
  if (conv_stride != 1 && conv_stride != 2) { okp_set_error("okp_dwconv3x3_forward: stride %d unsupported", conv_stride); return OKP_EINVAL; }
  if (c <= 0 || c % vn) { okp_set_error("okp_dwconv3x3_forward: channels %d not a multiple of %d", c, vn); return OKP_EINVAL; }
  if (src->pix_stride % vn || out->pix_stride % vn || (res && res->data && res->pix_stride % vn)) {
    okp_set_error("okp_dwconv3x3_forward: pixel strides must be multiples of %d elements", vn); return OKP_EINVAL;
  }
  const int ho = (src->h + 2 - 3) / conv_stride + 1, wo = (src->w + 2 - 3) / conv_stride + 1;
  if (out->h != ho || out->w != wo) { okp_set_error("okp_dwconv3x3_forward: out is %dx%d, expected %dx%d", out->h, out->w, ho, wo); return OKP_EINVAL; }
  DwParams p;
  if (src->bytes <= 0 || src->bytes >= 0x7FFF0000ll) { okp_set_error("okp_dwconv3x3_forward: src spans %lld bytes; views must be < 2 GiB", (long long)src->bytes); return OKP_EINVAL; }
  if (c / vn > 256) { okp_set_error("okp_dwconv3x3_forward: at most %d channels", 256 * vn); return OKP_EINVAL; }
  p.src = src->data; p.src_bytes = (uint32_t)src->bytes; p.H = src->h; p.W = src->w; p.src_ps = src->pix_stride;
  p.w = w_dev; p.bias = bias_dev;
  p.res = (res && res->data) ? res->data : nullptr; p.res_ps = res ? res->pix_stride : 0;
  p.out = out->data; p.Ho = ho; p.Wo = wo; p.out_ps = out->pix_stride;
  p.N = n; p.C = c; p.stride = conv_stride; p.act = act;
  const int pl = 256 / (c / vn);
  const long groups = ((long)n * ho * wo + (long)pl * 8 - 1) / ((long)pl * 8);
  const int grid = (int)(groups < 1 ? 1 : (groups > 256L * 16 ? 256L * 16 : groups));
  if (dtype == OKP_BF16) hipLaunchKernelGGL(okp_dwconv3x3_kernel<__bf16>, dim3(grid), dim3(256), 9 * c * sizeof(float), (hipStream_t)stream, p);
  else if (dtype == OKP_F16) hipLaunchKernelGGL(okp_dwconv3x3_kernel<_Float16>, dim3(grid), dim3(256), 9 * c * sizeof(float), (hipStream_t)stream, p);
  else hipLaunchKernelGGL(okp_dwconv3x3_kernel<float>, dim3(grid), dim3(256), 9 * c * sizeof(float), (hipStream_t)stream, p);
  return okp_check_hip(hipGetLastError(), "okp_dwconv3x3 launch");
}

namespace {
// 8 elements per thread and iteration: 32 + 16 bytes per lane, grid-stride
template <typename S, typename D>
__global__ __launch_bounds__(256) void okp_cast_kernel(const S* __restrict__ src, D* __restrict__ dst, long n8, long count, int32_t* range_flag) {
  bool bad = false;         // range_flag (conversions TO fp32 for a split-product consumer): a value outside the fp16 range or not finite, exactly
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const long e0 = i * 8;
    if (e0 + 8 <= count) {
      typedef S sv8 __attribute__((ext_vector_type(8)));
      typedef D dv8 __attribute__((ext_vector_type(8)));
      const sv8 v = *reinterpret_cast<const sv8*>(src + e0);
      dv8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) { o[e] = (D)(float)v[e]; bad |= okp_unsplittable((float)v[e]); }
      *reinterpret_cast<dv8*>(dst + e0) = o;
    } else {
      for (long e = e0; e < count; ++e) { dst[e] = (D)(float)src[e]; bad |= okp_unsplittable((float)src[e]); }
    }
  }
  okp_raise_range_flag(range_flag, bad);
}
template <typename S, typename D>
void launch_cast(const void* src, void* dst, long count, int32_t* range_flag, hipStream_t stream) {
  const long n8 = (count + 7) / 8;
  const int grid = (int)std::min<long>((n8 + 255) / 256, 2048);
  hipLaunchKernelGGL((okp_cast_kernel<S, D>), dim3(grid), dim3(256), 0, stream, static_cast<const S*>(src), static_cast<D*>(dst), n8, count, range_flag);
}
}  // namespace

namespace {
__global__ __launch_bounds__(256) void okp_add_f16_f32_kernel(const _Float16* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n8, long count, int relu,
                                                              int32_t* range_flag) {
  bool bad = false;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const long e0 = i * 8;
    if (e0 + 8 <= count) {
      const f16x8 av = *reinterpret_cast<const f16x8*>(a + e0);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(b + e0), b1 = *reinterpret_cast<const f32x4*>(b + e0 + 4);
      f32x4 o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o0[e] = (float)av[e] + b0[e]; o1[e] = (float)av[4 + e] + b1[e];
        bad |= okp_unsplittable(o0[e]) || okp_unsplittable(o1[e]);        // (before the ReLU: fmaxf turns a NaN into 0)
        if (relu) { o0[e] = fmaxf(o0[e], 0.f); o1[e] = fmaxf(o1[e], 0.f); }
      }
      *reinterpret_cast<f32x4*>(out + e0) = o0;
      *reinterpret_cast<f32x4*>(out + e0 + 4) = o1;
    } else {
      for (long e = e0; e < count; ++e) { const float v = (float)a[e] + b[e]; bad |= okp_unsplittable(v); out[e] = relu ? fmaxf(v, 0.f) : v; }
    }
  }
  okp_raise_range_flag(range_flag, bad);
}
}  // namespace

extern "C" int okp_add_f16_f32(const void* a, const float* b, float* out, int64_t count, int act, int32_t* range_flag, void* stream) {
  if (!a || !b || !out || count < 1) { okp_set_error("okp_add_f16_f32: null / empty argument"); return OKP_EINVAL; }
  if (((uintptr_t)a) % 16 || ((uintptr_t)b) % 16 || ((uintptr_t)out) % 16) { okp_set_error("okp_add_f16_f32: tensors must be 16-byte aligned"); return OKP_EINVAL; }
  if (act != OKP_ACT_NONE && act != OKP_ACT_RELU) { okp_set_error("okp_add_f16_f32: activation %d", act); return OKP_EINVAL; }
  const long n8 = (count + 7) / 8;
  const int grid = (int)std::min<long>((n8 + 255) / 256, 2048);
  hipLaunchKernelGGL(okp_add_f16_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, static_cast<const _Float16*>(a), b, out, n8, (long)count, act == OKP_ACT_RELU ? 1 : 0, range_flag);
  return okp_check_hip(hipGetLastError(), "okp_add_f16_f32 launch");
}

extern "C" int okp_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t count, int32_t* range_flag, void* stream) {
  if (!src || !dst || count < 1) { okp_set_error("okp_cast: null / empty argument"); return OKP_EINVAL; }
  if (((uintptr_t)src) % 16 || ((uintptr_t)dst) % 16) { okp_set_error("okp_cast: tensors must be 16-byte aligned"); return OKP_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  if (range_flag && dst_dtype != OKP_F32) { okp_set_error("okp_cast: the range flag belongs to conversions to fp32"); return OKP_EINVAL; }
  if (src_dtype == OKP_F32 && dst_dtype == OKP_F16) launch_cast<float, _Float16>(src, dst, count, nullptr, st);
  else if (src_dtype == OKP_F32 && dst_dtype == OKP_BF16) launch_cast<float, __bf16>(src, dst, count, nullptr, st);
  else if (src_dtype == OKP_F16 && dst_dtype == OKP_F32) launch_cast<_Float16, float>(src, dst, count, range_flag, st);
  else if (src_dtype == OKP_BF16 && dst_dtype == OKP_F32) launch_cast<__bf16, float>(src, dst, count, range_flag, st);
  else { okp_set_error("okp_cast: unsupported conversion %d -> %d (fp32 <-> fp16 / bf16)", src_dtype, dst_dtype); return OKP_EINVAL; }
  return okp_check_hip(hipGetLastError(), "okp_cast launch");
}

extern "C" int okp_pack_frames(int dtype, const float* frames, int32_t n, int32_t h, int32_t w, void* out, int32_t out_w, void* stream) {
  if (!frames || !out) { okp_set_error("okp_pack_frames: null argument"); return OKP_EINVAL; }
  if (dtype != OKP_F32 && !okp_is16(dtype)) { okp_set_error("okp_pack_frames: bad dtype %d", dtype); return OKP_EINVAL; }
  if (out_w < w + 6) { okp_set_error("okp_pack_frames: out_w %d < w+6", out_w); return OKP_EINVAL; }
  const long total = (long)n * (h + 6) * out_w;
  const int grid = grid_for(total, 256);
  if (dtype == OKP_BF16) hipLaunchKernelGGL(okp_pack_frames_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames, n, h, w, (__bf16*)out, h + 6, out_w);
  else if (dtype == OKP_F16) hipLaunchKernelGGL(okp_pack_frames_kernel<_Float16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames, n, h, w, (_Float16*)out, h + 6, out_w);
  else hipLaunchKernelGGL(okp_pack_frames_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames, n, h, w, (float*)out, h + 6, out_w);
  return okp_check_hip(hipGetLastError(), "okp_pack_frames launch");
}

extern "C" int okp_pack_frames_u8(int dtype, const uint8_t* frames, int32_t n, int32_t h, int32_t w, const float* mean3, const float* std3,
                                  void* out, int32_t out_w, void* stream) {
  if (!frames || !out || !mean3 || !std3) { okp_set_error("okp_pack_frames_u8: null argument"); return OKP_EINVAL; }
  if (dtype != OKP_F32 && !okp_is16(dtype)) { okp_set_error("okp_pack_frames_u8: bad dtype %d", dtype); return OKP_EINVAL; }
  if (out_w < w + 6) { okp_set_error("okp_pack_frames_u8: out_w %d < w+6", out_w); return OKP_EINVAL; }
  NormParams np;
  for (int c = 0; c < 3; ++c) { np.mean[c] = mean3[c]; np.stdv[c] = std3[c]; }
  const long total = (long)n * (h + 6) * out_w;
  const int grid = grid_for(total, 256);
  if (dtype == OKP_BF16) hipLaunchKernelGGL(okp_pack_frames_u8_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames, n, h, w, np, (__bf16*)out, h + 6, out_w);
  else if (dtype == OKP_F16) hipLaunchKernelGGL(okp_pack_frames_u8_kernel<_Float16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames, n, h, w, np, (_Float16*)out, h + 6, out_w);
  else hipLaunchKernelGGL(okp_pack_frames_u8_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames, n, h, w, np, (float*)out, h + 6, out_w);
  return okp_check_hip(hipGetLastError(), "okp_pack_frames_u8 launch");
}

extern "C" int okp_preprocess_u8(int dtype, const uint8_t* frames, int32_t n, int32_t src_h, int32_t src_w, int32_t resized_h, int32_t resized_w,
                                 int32_t crop_y, int32_t crop_x, int32_t h, int32_t w, const float* mean3, const float* std3,
                                 void* out, int32_t out_w, void* stream) {
  if (!frames || !out || !mean3 || !std3) { okp_set_error("okp_preprocess_u8: null argument"); return OKP_EINVAL; }
  if (dtype != OKP_F32 && !okp_is16(dtype)) { okp_set_error("okp_preprocess_u8: bad dtype %d", dtype); return OKP_EINVAL; }
  if (n < 1 || src_h < 1 || src_w < 1 || resized_h < 1 || resized_w < 1 || h < 1 || w < 1 || crop_y < 0 || crop_x < 0 ||
      crop_y + h > resized_h || crop_x + w > resized_w) {
    okp_set_error("okp_preprocess_u8: crop %dx%d at (%d,%d) does not fit the resized frame %dx%d", h, w, crop_y, crop_x, resized_h, resized_w);
    return OKP_EINVAL;
  }
  if (out_w < w + 6) { okp_set_error("okp_preprocess_u8: out_w %d < w+6", out_w); return OKP_EINVAL; }
  ResizeParams rp;
  rp.srcH = src_h; rp.srcW = src_w; rp.rsH = resized_h; rp.rsW = resized_w; rp.cropY = crop_y; rp.cropX = crop_x; rp.H = h; rp.W = w;
  rp.scaleY = 1.0 / ((double)resized_h / (double)src_h);
  rp.scaleX = 1.0 / ((double)resized_w / (double)src_w);
  for (int c = 0; c < 3; ++c) { rp.np.mean[c] = mean3[c]; rp.np.stdv[c] = std3[c]; }
  const long total = (long)n * (h + 6) * out_w;
  const int grid = grid_for(total, 256);
  if (dtype == OKP_BF16) hipLaunchKernelGGL(okp_preprocess_u8_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames, n, rp, (__bf16*)out, h + 6, out_w);
  else if (dtype == OKP_F16) hipLaunchKernelGGL(okp_preprocess_u8_kernel<_Float16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames, n, rp, (_Float16*)out, h + 6, out_w);
  else hipLaunchKernelGGL(okp_preprocess_u8_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames, n, rp, (float*)out, h + 6, out_w);
  return okp_check_hip(hipGetLastError(), "okp_preprocess_u8 launch");
}

extern "C" int okp_head_out_forward(int dtype, const okp_head_out_args* a, void* stream) {
  if (!a || !a->src.data || !a->w_dev || !a->bias_dev) { okp_set_error("okp_head_out_forward: null argument"); return OKP_EINVAL; }
  if (dtype != OKP_F32 && !okp_is16(dtype)) { okp_set_error("okp_head_out_forward: bad dtype %d", dtype); return OKP_EINVAL; }
  if (a->n_out < 1 || a->n_out > OKP_HEAD_MAX_OUT) { okp_set_error("okp_head_out_forward: n_out %d out of range", a->n_out); return OKP_EINVAL; }
  const int vn = okp_is16(dtype) ? 8 : 4;
  if (a->src.pix_stride % vn) { okp_set_error("okp_head_out_forward: pixel stride must be a multiple of %d", vn); return OKP_EINVAL; }
  HeadParams p;
  p.src = a->src.data; p.src_ps = a->src.pix_stride;
  p.N = a->n; p.HW = a->h * a->w; p.n_out = a->n_out;
  p.w = a->w_dev; p.bias = a->bias_dev;
  for (int o = 0; o < a->n_out; ++o) {
    if (a->in_c_off[o] % vn || !a->out_ptr[o]) { okp_set_error("okp_head_out_forward: output %d misaligned or null", o); return OKP_EINVAL; }
    p.in_c_off[o] = a->in_c_off[o]; p.act[o] = a->act[o]; p.out_ptr[o] = a->out_ptr[o]; p.out_n_stride[o] = a->out_n_stride[o];
  }
  const long total = (long)p.N * p.HW;
  const int grid = grid_for(total, 256);
  if (dtype == OKP_BF16) hipLaunchKernelGGL(okp_head_out_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else if (dtype == OKP_F16) hipLaunchKernelGGL(okp_head_out_kernel<_Float16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(okp_head_out_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return okp_check_hip(hipGetLastError(), "okp_head_out launch");
}

// ---- cross-stream dependency without a system-scope fence (okp.h: okp_stream_wait_stream) ----------------------------------------
namespace {
constexpr int kEdgeEvents = 256;        // round-robin: a wait holds its own reference to the record it was given, re-recording an event later is safe
struct EdgeEvents {
  hipEvent_t ev[kEdgeEvents] = {};
  int device = -1;
  unsigned next = 0;
};
thread_local EdgeEvents t_edges;
}  // namespace

extern "C" int okp_stream_wait_stream(void* waiter, void* signaller) {
  int dev = 0;
  if (int e = okp_check_hip(hipGetDevice(&dev), "hipGetDevice")) return e;
  EdgeEvents& E = t_edges;
  if (E.device != dev) {                // (events belong to the device they were created on; a thread that changes device gets new ones)
    for (hipEvent_t& e : E.ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    E.device = dev; E.next = 0;
  }
  hipEvent_t& ev = E.ev[E.next++ % kEdgeEvents];
  if (!ev) { if (int e = okp_check_hip(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence), "hipEventCreateWithFlags")) return e; }
  if (int e = okp_check_hip(hipEventRecord(ev, (hipStream_t)signaller), "hipEventRecord")) return e;
  return okp_check_hip(hipStreamWaitEvent((hipStream_t)waiter, ev, 0), "hipStreamWaitEvent");
}

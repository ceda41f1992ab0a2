// Fused CornerNet-Squeeze fire module for gfx950 (bf16):  ONE launch computes
//
//     s   = W1 x + b1                                   squeeze 1x1 (+bn1, no ReLU)      Cin -> mid
//     y_a = relu(Wa s + ba (+ x[:, :half]))             expand 1x1 (+bn2 half, skip)     mid -> half
//     y_b = relu(dw3x3(s) * wd + bd (+ x[:, half:]))    depth-wise 3x3 (+bn2 half, skip) mid -> half
//
// (reference: fire_module, perception/corner_net_lite/core/models/CornerNet_Squeeze.py:10-30).
// The squeeze tensor never goes to HBM: a workgroup owns a rectangle of output pixels of one or more
// frames, computes `s` for that rectangle plus a one-pixel halo with MFMAs (K-slices of x and W1 gathered
// by LDS-DMA exactly as in okp_igemm.hip), keeps it in LDS as bf16 in the K-slice layout
// [mid/64][128 rows][128 B, chunk-swizzled], and then
//   * feeds it straight back to the MFMAs as the B operand of the expand GEMM (Wa streamed by LDS-DMA),
//   * reads the 3x3 neighbourhoods for the depth-wise branch with ds_read_b128.
// Out-of-frame halo pixels are zero in `s` (the reference zero-pads the squeeze output), which the gather
// gives for free: masked LDS-DMA lanes write zeros and the bias is multiplied by the validity flag.
// HBM traffic per module: x once (+halo overlap, mostly L2 hits) and the output once.
#include <cstdlib>
#include <cstring>

#include "okp_internal.h"

namespace {

constexpr uint32_t kInvalid = 0x80000000u;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ uint32_t swz128(int row, int chunk) {
  return (uint32_t)row * 128u + (uint32_t)((chunk ^ ((row >> 1) & 7)) << 4);
}

constexpr int kStageBytes = 65536;          // GEMM stages / fp32 output staging
constexpr int kSBufMax = 4 * 16384;         // mid <= 256
constexpr int kTabOff = kStageBytes + kSBufMax;
constexpr int kLdsBytes = kTabOff + 3 * 512;

__global__ __launch_bounds__(256) void okp_fire_kernel(const OkpFireParams p) {
  __shared__ __attribute__((aligned(16))) char smem[kLdsBytes];
  char* const sbuf = smem + kStageBytes;
  uint32_t* const tab_xoff = reinterpret_cast<uint32_t*>(smem + kTabOff);         // [128] byte offset of S pixel or kInvalid
  int* const tab_srow = reinterpret_cast<int*>(smem + kTabOff + 512);             // [128] s row of interior pixel j
  int* const tab_opix = reinterpret_cast<int*>(smem + kTabOff + 1024);            // [128] output pixel index or -1

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave >> 1, wpx = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;

  // ---- tile decode + per-pixel tables ------------------------------------------------------------
  const int tiles_per_group = p.tiles_y * p.tiles_x;
  const int grp = blockIdx.x / tiles_per_group;
  const int trem = blockIdx.x - grp * tiles_per_group;
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int SHW = p.SH * p.SW, IHW = p.IH * p.IW;
  if (tid < 128) {
    {   // S pixel tid
      const int f = tid / SHW, r = tid - f * SHW;
      const int sy = r / p.SW, sx = r - sy * p.SW;
      const int n = grp * p.FR + f;
      const int y = ty * p.IH * p.stride - 1 + sy, x = tx * p.IW * p.stride - 1 + sx;
      const bool ok = f < p.FR && n < p.N && y >= 0 && y < p.H && x >= 0 && x < p.W;
      tab_xoff[tid] = ok ? (uint32_t)(((n * p.H + y) * p.W + x) * p.x_ps) * 2u : kInvalid;
    }
    {   // interior pixel tid
      const int f = tid / IHW, r = tid - f * IHW;
      const int iy = r / p.IW, ix = r - iy * p.IW;
      const int n = grp * p.FR + f;
      const int oy = ty * p.IH + iy, ox = tx * p.IW + ix;
      const bool ok = f < p.FR && n < p.N && oy < p.Ho && ox < p.Wo;
      tab_srow[tid] = ok ? f * SHW + (iy * p.stride + 1) * p.SW + ix * p.stride + 1 : 0;
      tab_opix[tid] = ok ? (n * p.Ho + oy) * p.Wo + ox : -1;
    }
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, (int)p.w1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wa = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wa), 0, (int)p.wa_bytes, 0x00020000);

  // loader geometry (as okp_igemm.hip): lane fetches the chunk the read-side swizzle expects at its position
  const int r0 = tid >> 3;                                   // 0..31, rows r0 + 32 i
  const int c = (tid & 7) ^ ((r0 >> 1) & 7);
  uint32_t xoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t b = tab_xoff[r0 + 32 * i];
    xoff[i] = (b == kInvalid) ? kInvalid : b + (uint32_t)c * 16u;
  }
  const int n1 = p.cin >> 6, n2 = p.mid >> 6;                // 64-channel K-slices

  // =============== phase 1: s = W1 x + b1 for the 128 halo'd pixels, 128 mid channels at a time ===========
  const int mid_chunks = (p.mid + 127) >> 7;
  for (int mc = 0; mc < mid_chunks; ++mc) {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    uint32_t wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = mc * 128 + r0 + 32 * i;
      wrow[i] = row < p.w1_cout_pad ? (uint32_t)row * 128u + (uint32_t)c * 16u : kInvalid;
    }
    auto issue = [&](int s, int stage) {
      char* const wt = smem + stage * 32768 + wave * 1024;
      char* const xt = wt + 16384;
      const uint32_t ws = (uint32_t)s * (uint32_t)p.w1_cout_pad * 128u;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (lds_ptr_t)(wt + i * 4096), 16, (int)(wrow[i] + ws), 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t off = xoff[i] == kInvalid ? kInvalid : xoff[i] + (uint32_t)s * 128u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(xt + i * 4096), 16, (int)off, 0, 0, 0);
      }
    };
    issue(0, 0);
    for (int s = 0; s < n1; ++s) {
      const int stage = s & 1;
      __syncthreads();                                        // drains this wave's LDS-DMA (vmcnt 0) and syncs
      if (s + 1 < n1) issue(s + 1, stage ^ 1);
      const char* wt = smem + stage * 32768;
      const char* xt = wt + 16384;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        u32x4 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const u32x4*>(wt + swz128((wco * 2 + i) * 32 + fr, 2 * kk + fh));
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const u32x4*>(xt + swz128((wpx * 2 + j) * 32 + fr, 2 * kk + fh));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
      }
    }
    // s tile -> LDS (bf16, K-slice layout).  Row = halo'd pixel, zero where the pixel is outside the frame.
    const int slice = mc * 2 + wco;
    if (slice < n2) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int px = (wpx * 2 + j) * 32 + fr;
        const float keep = tab_xoff[px] == kInvalid ? 0.f : 1.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int ch = mc * 128 + wco * 64 + i * 32 + 8 * g + 4 * fh;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.b1 + ch);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)((acc[i][j][4 * g + e] + bv[e]) * keep);
            *reinterpret_cast<bf16x4*>(sbuf + slice * 16384 + swz128(px, 4 * i + g) + 8 * fh) = o;
          }
      }
    }
    __syncthreads();                                          // stages free for the next chunk; s visible
  }

  // =============== phase 2a: y_a = relu(Wa s + ba (+x)) on the interior pixels ===========================
  int srow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) srow[j] = tab_srow[(wpx * 2 + j) * 32 + fr];
  const int co_chunks = (p.half + 127) >> 7;
  for (int cc = 0; cc < co_chunks; ++cc) {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    uint32_t wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = cc * 128 + r0 + 32 * i;
      wrow[i] = row < p.wa_cout_pad ? (uint32_t)row * 128u + (uint32_t)c * 16u : kInvalid;
    }
    auto issue = [&](int s, int stage) {
      char* const wt = smem + stage * 16384 + wave * 1024;
      const uint32_t ws = (uint32_t)s * (uint32_t)p.wa_cout_pad * 128u;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wa, (lds_ptr_t)(wt + i * 4096), 16, (int)(wrow[i] + ws), 0, 0, 0);
    };
    issue(0, 0);
    for (int s = 0; s < n2; ++s) {
      const int stage = s & 1;
      __syncthreads();
      if (s + 1 < n2) issue(s + 1, stage ^ 1);
      const char* wt = smem + stage * 16384;
      const char* st = sbuf + s * 16384;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        u32x4 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const u32x4*>(wt + swz128((wco * 2 + i) * 32 + fr, 2 * kk + fh));
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const u32x4*>(st + swz128(srow[j], 2 * kk + fh));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();                                          // every wave done with the Wa stages: reuse as staging
    // bias in registers, transpose through LDS ([px][128 co] fp32, chunk-swizzled), coalesced NHWC rows
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co_l = (wco * 2 + i) * 32 + 8 * g + 4 * fh;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.ba + cc * 128 + co_l);     // ba padded to a multiple of 256
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int prow = (wpx * 2 + j) * 32 + fr;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + bv[e];
          *reinterpret_cast<f32x4*>(smem + prow * 512 + (((co_l >> 2) ^ (prow & 7)) << 4)) = v;
        }
      }
    __syncthreads();
    for (int it = tid; it < 128 * 16; it += 256) {
      const int q = it & 15, prow = it >> 4;
      const int opix = tab_opix[prow];
      const int co = cc * 128 + q * 8;
      if (opix >= 0 && co < p.half) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(smem + prow * 512 + (((2 * q) ^ (prow & 7)) << 4));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(smem + prow * 512 + (((2 * q + 1) ^ (prow & 7)) << 4));
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (p.skip) {
          const bf16x8 r = *reinterpret_cast<const bf16x8*>(static_cast<const char*>(p.x) + ((size_t)opix * p.x_ps + co) * 2);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)fmaxf(v[e], 0.f);
        *reinterpret_cast<bf16x8*>(static_cast<char*>(p.out) + ((size_t)opix * p.out_ps + co) * 2) = o;
      }
    }
    __syncthreads();                                          // staging free before the next chunk's LDS-DMA
  }

  // =============== phase 2b: y_b = relu(dw3x3(s) + bd (+x)) from the LDS-resident s tile =================
  const int groups = p.half >> 3;                             // 8-channel groups (half == mid)
  const int items = 128 * groups;
  for (int it = tid; it < items; it += 256) {
    const int q = it % groups, j = it / groups;
    const int opix = tab_opix[j];
    if (opix < 0) continue;
    const int row0 = tab_srow[j];
    const char* sl = sbuf + (q >> 3) * 16384;
    float v[8];
    {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bd + q * 8), b1 = *reinterpret_cast<const f32x4*>(p.bd + q * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = b0[e]; v[4 + e] = b1[e]; }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int row = row0 + (k / 3 - 1) * p.SW + (k % 3 - 1);
      const bf16x8 sv = *reinterpret_cast<const bf16x8*>(sl + swz128(row, q & 7));
      const float* wk = p.wd + k * p.half + q * 8;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wk), w1 = *reinterpret_cast<const f32x4*>(wk + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = fmaf((float)sv[e], w0[e], v[e]);
        v[4 + e] = fmaf((float)sv[4 + e], w1[e], v[4 + e]);
      }
    }
    if (p.skip) {
      const bf16x8 r = *reinterpret_cast<const bf16x8*>(static_cast<const char*>(p.x) + ((size_t)opix * p.x_ps + p.half + q * 8) * 2);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)fmaxf(v[e], 0.f);
    *reinterpret_cast<bf16x8*>(static_cast<char*>(p.out) + ((size_t)opix * p.out_ps + p.half + q * 8) * 2) = o;
  }
}

}  // namespace

// Pick the output rectangle (IH x IW, FR frames) whose halo'd footprint fits 128 LDS rows with the fewest tiles.
static void choose_tile(int Ho, int Wo, int stride, int N, OkpFireParams& p) {
  long best = -1;
  for (int ih = 1; ih <= Ho && ih <= 128; ++ih)
    for (int iw = 1; iw <= Wo && iw <= 128; ++iw) {
      const int sh = stride * (ih - 1) + 3, sw = stride * (iw - 1) + 3;
      if (sh * sw > 128 || ih * iw > 128) continue;
      const int ty = (Ho + ih - 1) / ih, tx = (Wo + iw - 1) / iw;
      int fr = 1;
      if (ty == 1 && tx == 1) { fr = 128 / (sh * sw); if (fr > N) fr = N; if (fr < 1) fr = 1; }
      const long tiles = (long)((N + fr - 1) / fr) * ty * tx;
      // fewest tiles; ties: wider rows (longer contiguous stores)
      const long score = tiles * 1024 - iw;
      if (best < 0 || score < best) {
        best = score;
        p.IH = ih; p.IW = iw; p.SH = sh; p.SW = sw; p.FR = fr; p.tiles_y = ty; p.tiles_x = tx;
      }
    }
}

extern "C" int okp_fire_forward(const okp_conv* squeeze, const okp_conv* expand, const float* dw_w_dev, const float* dw_bias_dev,
                                const okp_fire_args* a, void* stream) {
  if (!squeeze || !expand || !dw_w_dev || !dw_bias_dev || !a || !a->x.data || !a->out.data) { okp_set_error("okp_fire_forward: null argument"); return OKP_EINVAL; }
  if (squeeze->dtype != OKP_BF16 || expand->dtype != OKP_BF16) { okp_set_error("okp_fire_forward: the fused fire kernel is bf16 only"); return OKP_EINVAL; }
  if (squeeze->n_taps != 1 || expand->n_taps != 1 || squeeze->n_src != 1 || expand->n_src != 1) { okp_set_error("okp_fire_forward: squeeze/expand must be single-tap 1x1 plans"); return OKP_EINVAL; }
  const int cin = squeeze->cin[0], mid = squeeze->cout, half = expand->cout;
  if (expand->cin[0] != mid || half != mid) { okp_set_error("okp_fire_forward: expects expand cin == squeeze cout == half (sr = 2)"); return OKP_EINVAL; }
  if (cin % 64 || mid % 64 || mid > 256) { okp_set_error("okp_fire_forward: cin %d / mid %d must be multiples of 64, mid <= 256", cin, mid); return OKP_EINVAL; }
  if (a->stride != 1 && a->stride != 2) { okp_set_error("okp_fire_forward: stride %d", a->stride); return OKP_EINVAL; }
  if (a->skip && (a->stride != 1 || cin != 2 * half)) { okp_set_error("okp_fire_forward: skip needs stride 1 and cin == cout"); return OKP_EINVAL; }
  const int ho = (a->x.h - 1) / a->stride + 1, wo = (a->x.w - 1) / a->stride + 1;
  if (a->out.h != ho || a->out.w != wo) { okp_set_error("okp_fire_forward: out is %dx%d, expected %dx%d", a->out.h, a->out.w, ho, wo); return OKP_EINVAL; }
  if (a->x.pix_stride % 8 || a->out.pix_stride % 8 || a->x.pix_stride < cin || a->out.pix_stride < 2 * half ||
      ((uintptr_t)a->x.data) % 16 || ((uintptr_t)a->out.data) % 16) { okp_set_error("okp_fire_forward: views must be 16-byte aligned and wide enough"); return OKP_EINVAL; }
  if (a->x.bytes <= 0 || a->x.bytes >= 0x7FFF0000ll) { okp_set_error("okp_fire_forward: x spans %lld bytes; views must be < 2 GiB", (long long)a->x.bytes); return OKP_EINVAL; }
  static const bool force_v1 = getenv("OKP_FIRE_V1") != nullptr;      // experiments: first-generation kernel for every shape
  if (okp_fire2_supported(cin, mid, half, a->stride) && !force_v1) {
    if (a->out.bytes <= 0 || a->out.bytes >= 0x7FFF0000ll) { okp_set_error("okp_fire_forward: out spans %lld bytes; views must be < 2 GiB", (long long)a->out.bytes); return OKP_EINVAL; }
    OkpFire2Params q;
    memset(&q, 0, sizeof(q));
    q.x = a->x.data; q.x_bytes = (uint32_t)a->x.bytes; q.H = a->x.h; q.W = a->x.w; q.x_ps = a->x.pix_stride;
    q.out = a->out.data; q.out_bytes = (uint32_t)a->out.bytes; q.Ho = ho; q.Wo = wo; q.out_ps = a->out.pix_stride;
    q.N = a->n; q.skip = a->skip;
    if (int e = okp_ensure_frags(squeeze, (hipStream_t)stream)) return e;
    if (int e = okp_ensure_frags(expand, (hipStream_t)stream)) return e;
    q.w1 = squeeze->frag_dev; q.w1_cout_pad = squeeze->cout_pad; q.b1 = squeeze->bias_dev;
    q.wa = expand->frag_dev; q.wa_cout_pad = expand->cout_pad; q.ba = expand->bias_dev;
    q.wd = dw_w_dev; q.bd = dw_bias_dev;
    return okp_launch_fire2(q, cin, mid, a->stride, (hipStream_t)stream);
  }
  OkpFireParams p;
  memset(&p, 0, sizeof(p));
  p.x = a->x.data; p.x_bytes = (uint32_t)a->x.bytes; p.H = a->x.h; p.W = a->x.w; p.x_ps = a->x.pix_stride;
  p.out = a->out.data; p.Ho = ho; p.Wo = wo; p.out_ps = a->out.pix_stride;
  p.N = a->n; p.stride = a->stride; p.skip = a->skip; p.cin = cin; p.mid = mid; p.half = half;
  p.w1 = squeeze->weights_dev; p.w1_bytes = squeeze->w_bytes; p.w1_cout_pad = squeeze->cout_pad; p.b1 = squeeze->bias_dev;
  p.wa = expand->weights_dev; p.wa_bytes = expand->w_bytes; p.wa_cout_pad = expand->cout_pad; p.ba = expand->bias_dev;
  p.wd = dw_w_dev; p.bd = dw_bias_dev;
  choose_tile(ho, wo, a->stride, a->n, p);
  const long grid = (long)((a->n + p.FR - 1) / p.FR) * p.tiles_y * p.tiles_x;
  hipLaunchKernelGGL(okp_fire_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p);
  return okp_check_hip(hipGetLastError(), "okp_fire launch");
}

// Heat-map peak extraction: one workgroup per (frame, keypoint-type) map, the map (or a strip of it) and its
// 5x5 box sums resident in LDS, ordered compaction with wave-level ballot/prefix reductions.
//
// Restates KeypointExtractionComponent._extract_keypoints/_compute_points
// (reference perception/pipeline.py:46-79) and nms (perception/models.py:55-58).
// The index contract is bit-exact: the box sum is accumulated in fp32 in row-major tap order
// (dy outer, dx inner) starting from 0 with zero padding — any other association changes which
// pixels tie with their 5x5 maximum (SURVEY.md §7 "Bit-exact peaks").
#include <atomic>

#include "okp_internal.h"

namespace {

constexpr int kNmsThreads = 1024;            // 4 pixels per thread on a 64x64 map: the map is latency-, not bandwidth-bound

// The map is walked in strips of R rows (R = H when it fits: one strip for the 64x64 maps of the network).  A strip keeps
// the probabilities of rows [r0-4, r0+R+4) and the box sums of rows [r0-2, r0+R+2) in LDS: the 5x5 maximum of a box-sum
// row needs box rows +-2, each of which needs probability rows +-2.  Both arrays carry a two-pixel border of ZEROS (columns
// -2, -1, W, W+1 and the rows outside the image): the box sum adds the padding exactly as the reference's zero-padded
// convolution does (x + 0 = x in fp32, same tap order), and a zero never wins the 5x5 maximum against the non-negative box
// sums - so neither inner loop tests a bound.  Strips are processed in order by the same workgroup with a running peak
// count, so the row-major peak order of the reference is kept for maps of any size.
__global__ __launch_bounds__(kNmsThreads) void okp_peak_nms_kernel(const float* __restrict__ heat, int H, int W, int R, int cap,
                                                                   int* __restrict__ count, int* __restrict__ yx,
                                                                   float* __restrict__ xyc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int WP = W + 4;                    // padded row pitch
  float* prob = lds;                       // [(R+8) rows][WP]: image rows r0-4 .. r0+R+3, columns -2 .. W+1
  float* box = lds + (size_t)(R + 8) * WP; // [(R+4) rows][WP]: image rows r0-2 .. r0+R+1
  __shared__ int wave_tot[kNmsThreads / 64];
  const int map = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const float* src = heat + (size_t)map * H * W;
  int running = 0;
  int* my_yx = yx + (size_t)map * cap * 2;
  float* my_xyc = xyc + (size_t)map * cap * 3;
  // this thread's position in a row-major walk over W-wide rows, advanced by kNmsThreads per step without a division
  const int sy = kNmsThreads / W, sx = kNmsThreads - sy * W;
  for (int r0 = 0; r0 < H; r0 += R) {
    const int s1 = min(r0 + R, H);                                 // rows [r0, s1) are classified in this strip
    // probabilities with their zero border
    for (int i = tid; i < (R + 8) * WP; i += kNmsThreads) {
      const int ly = i / WP, lx = i - ly * WP;
      const int y = r0 - 4 + ly, x = lx - 2;
      prob[i] = (y >= 0 && y < H && x >= 0 && x < W) ? src[(size_t)y * W + x] : 0.f;
    }
    for (int i = tid; i < (R + 4) * WP; i += kNmsThreads) box[i] = 0.f;
    __syncthreads();
    // box sums of the image rows max(r0-2, 0) .. min(r0+R+2, H) - 1
    {
      const int b0 = max(r0 - 2, 0), b1 = min(r0 + R + 2, H);
      int y = b0 + tid / W, x = tid % W;
      for (; y < b1; ) {
        const float* pp = prob + (size_t)(y - (r0 - 4) - 2) * WP + x;       // tap (dy, dx) = pp[(dy + 2) * WP + dx + 2]
        float s = 0.f;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
#pragma unroll
          for (int dx = 0; dx < 5; ++dx) s = s + pp[dy * WP + dx];           // sequential fp32, row-major taps: the reference's accumulation order
        box[(size_t)(y - (r0 - 2)) * WP + x + 2] = s;
        x += sx; y += sy;
        if (x >= W) { x -= W; ++y; }
      }
    }
    __syncthreads();
    const int n_px = (s1 - r0) * W;
    int y = r0 + tid / W, x = tid % W;
    for (int base = 0; base < n_px; base += kNmsThreads) {
      const int i = base + tid;
      bool peak = false;
      if (i < n_px) {
        const float* bp = box + (size_t)(y - (r0 - 2) - 2) * WP + x;         // tap (dy, dx) = bp[(dy + 2) * WP + dx + 2]
        const float b = bp[2 * WP + 2];
        float m = b;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
#pragma unroll
          for (int dx = 0; dx < 5; ++dx) m = fmaxf(m, bp[dy * WP + dx]);
        peak = (b == m) && (b > 0.5f);
      }
      const unsigned long long ball = __ballot(peak);
      const int before = __popcll(ball & ((1ull << lane) - 1ull));
      if (lane == 0) wave_tot[wave] = __popcll(ball);
      __syncthreads();
      int off = running + before;
      int tot = 0;
#pragma unroll
      for (int w = 0; w < kNmsThreads / 64; ++w) {
        const int t = wave_tot[w];
        if (w < wave) off += t;
        tot += t;
      }
      if (peak && off < cap) {
        const int y0 = max(y - 2, 0), y1 = min(y + 3, H), x0 = max(x - 2, 0), x1 = min(x + 3, W);
        float sp = 0.f, sy2 = 0.f, sx2 = 0.f;
        for (int yy = y0; yy < y1; ++yy)
          for (int xx = x0; xx < x1; ++xx) {
            const float pv = prob[(size_t)(yy - (r0 - 4)) * WP + xx + 2];
            sp += pv;
            sy2 += pv * (float)yy;
            sx2 += pv * (float)xx;
          }
        my_yx[off * 2 + 0] = y;
        my_yx[off * 2 + 1] = x;
        my_xyc[off * 3 + 0] = sx2 / sp;
        my_xyc[off * 3 + 1] = sy2 / sp;
        my_xyc[off * 3 + 2] = sp;
      }
      running += tot;
      x += sx; y += sy;
      if (x >= W) { x -= W; ++y; }
      __syncthreads();
    }
  }
  if (tid == 0) count[map] = running;
  // unused slots are defined output (zeros): callers hand over uninitialised buffers, and bit-reproducibility checks
  // compare whole tensors
  for (int j = min(running, cap) + tid; j < cap; j += kNmsThreads) {
    my_yx[j * 2 + 0] = 0; my_yx[j * 2 + 1] = 0;
    my_xyc[j * 3 + 0] = 0.f; my_xyc[j * 3 + 1] = 0.f; my_xyc[j * 3 + 2] = 0.f;
  }
}

__global__ __launch_bounds__(256) void okp_nms_maxpool_kernel(const float* __restrict__ x, int H, int W, int size,
                                                              long total, float* __restrict__ out) {
  const int r = size / 2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(idx % W);
    const long t = idx / W;
    const int yy = (int)(t % H);
    const float* m = x + (t / H) * (long)H * W;
    const float v = m[yy * W + xx];
    float mx = v;
    for (int dy = -r; dy <= r; ++dy) {
      const int y2 = yy + dy;
      if (y2 < 0 || y2 >= H) continue;
      for (int dx = -r; dx <= r; ++dx) {
        const int x2 = xx + dx;
        if (x2 < 0 || x2 >= W) continue;
        mx = fmaxf(mx, m[y2 * W + x2]);
      }
    }
    out[idx] = v * ((v == mx) ? 1.f : 0.f);
  }
}

}  // namespace

extern "C" int okp_nms_maxpool(const float* x, int32_t n_maps, int32_t h, int32_t w, int32_t size, float* out, void* stream) {
  if (!x || !out) { okp_set_error("okp_nms_maxpool: null argument"); return OKP_EINVAL; }
  if (size < 1 || size > 15 || !(size & 1)) { okp_set_error("okp_nms_maxpool: size %d must be odd and <= 15", size); return OKP_EINVAL; }
  const long total = (long)n_maps * h * w;
  if (total <= 0) return OKP_OK;
  long grid = (total + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(okp_nms_maxpool_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, h, w, size, total, out);
  return okp_check_hip(hipGetLastError(), "okp_nms_maxpool launch");
}

extern "C" int okp_peak_nms(const float* heat, int32_t n_maps, int32_t h, int32_t w, int32_t cap, int32_t* count,
                            int32_t* yx, float* xyc, void* stream) {
  if (!heat || !count || !yx || !xyc) { okp_set_error("okp_peak_nms: null argument"); return OKP_EINVAL; }
  if (n_maps < 0 || h < 1 || w < 1 || cap < 1) { okp_set_error("okp_peak_nms: bad sizes n_maps=%d h=%d w=%d cap=%d", n_maps, h, w, cap); return OKP_EINVAL; }
  // strip height: (R + 8) probability rows + (R + 4) box-sum rows of w floats in at most 128 KiB of LDS
  constexpr long kLdsFloats = 32768;
  long rmax = (kLdsFloats / (w + 4) - 12) / 2;           // rows of w + 4 floats (two-pixel zero border)
  if (rmax < 1) { okp_set_error("okp_peak_nms: maps wider than %ld pixels are not supported (width %d)", kLdsFloats / 14 - 4, w); return OKP_EINVAL; }
  const int R = (int)(rmax < h ? rmax : h);
  if (n_maps == 0) return OKP_OK;
  const size_t lds = (size_t)(2 * R + 12) * (w + 4) * sizeof(float);
  // the attribute is per device: one bit per device id (idempotent, so a race only repeats the call)
  static std::atomic<unsigned long long> attr_set{0ull};
  int dev = 0;
  if (int e = okp_check_hip(hipGetDevice(&dev), "okp_peak_nms: hipGetDevice")) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_acquire) & bit)) {
    if (int e = okp_check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(okp_peak_nms_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsFloats * sizeof(float))),
                              "okp_peak_nms: hipFuncSetAttribute"))
      return e;
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(okp_peak_nms_kernel, dim3(n_maps), dim3(kNmsThreads), lds, (hipStream_t)stream, heat, h, w, R, cap, count, yx, xyc);
  return okp_check_hip(hipGetLastError(), "okp_peak_nms launch");
}

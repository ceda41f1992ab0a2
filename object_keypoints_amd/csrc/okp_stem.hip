// 7x7 / stride-2 stem convolution (3 -> 128 channels) + folded BatchNorm + ReLU, bf16, for gfx950.
//
// Replaces `convolution(7, 3, 128, stride=2)` = hg.pre[0] (corner_net_lite/core/models/py_utils/utils.py:143-156,
// built at corner_net_lite/core/models/CornerNet_Squeeze.py:84).  The generic tap-list kernel spends more time in
// its per-tile prologue and epilogue than in the 7 K-slices of this layer; this kernel is shaped for it instead:
//
//  * The layer is WRITE bound: 2 B x 128 channels per output pixel (1.07 GB at 64 frames) against 224 padded MACs
//    per output value.  Everything is organised so that the stores leave as whole 64-byte lines straight from the
//    accumulator registers, without a transposition through LDS.
//  * Input: the packed frame of okp_pack_frames (NHWC4 bf16 with a zero halo, image at (3,3)).  A workgroup owns an
//    8 x 32 tile of output pixels and keeps the 21 x 70 pixel input patch of that tile in LDS (11.8 KB, fetched by
//    LDS-DMA one tile ahead, double-buffered).  No im2col: with 4 channels per pixel and stride 2, the 8 K-values
//    lane (pixel j, half h) feeds to a 32x32x16 MFMA are the two packed pixels at patch column 2j + kx0 + 2h, one
//    aligned ds_read_b128.  K order: k-step s = (ky = s/2, kx0 = 4*(s%2)), k = 8h + e <-> (kx = kx0 + 2h + e/4, c = e%4);
//    channel 3 and kx = 7 carry zero weights.
//  * Pixels are the MFMA ROWS (A operand), output channels the COLUMNS (B operand): D then has lane j of a half-wave
//    on channel column j and accumulator register r on pixel 8 (r/4) + 4 h + r%4.  Weights: 14 k-steps x 2 column
//    blocks of B fragments stay in registers for the whole kernel (112 VGPRs); wave (wc, wp) owns output channels
//    [64 wc, 64 wc + 64) and tile rows [4 wp, 4 wp + 4).
//  * Column j of block b is channel 64 wc + 2 j + b (fixed at pack time), so a lane's two blocks are two ADJACENT
//    channels of the same pixel: bias is the accumulator's initial value, ReLU and the bf16 rounding happen in
//    registers, the pair is one dword and a half-wave's store is 128 contiguous bytes of one NHWC pixel (whole
//    64-byte lines, two pixels per instruction) - no transposition through LDS.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "okp_internal.h"

namespace {

constexpr int kCout = 128;
constexpr int TH = 8, TW = 32;                 // output pixels per tile
constexpr int PR = 2 * (TH - 1) + 7;           // 21 patch rows
constexpr int PC = 2 * (TW - 1) + 8;           // 70 patch columns (the 8th kernel column is zero padding of K)
constexpr int PITCH = PC * 8;                  // 560 bytes per patch row (bf16 x 4 channels per pixel)
constexpr int CHUNKS_ROW = PITCH / 16;         // 35 16-byte chunks per row
constexpr int CHUNKS = PR * CHUNKS_ROW;        // 735 per patch
constexpr int PATCH_BYTES = ((CHUNKS + 63) / 64) * 1024;   // whole LDS-DMA instructions: 12 KiB
constexpr int KSTEPS = 14;
constexpr uint32_t kInvalidOff = 0x80000000u;

struct StemParams {
  const void* src;          // packed frames [N][Hp][Wp][4] bf16, or fp32 frames [N][3][H][W]
  uint32_t src_bytes;
  int32_t Hp, Wp, H, W;
  const void* wfrag;        // [wc 2][block 2][k-step 14][lane 64][16 B]  B-operand fragments
  const float* bias;        // [128], natural channel order (split-product plan: already multiplied with the channel's weight scale)
  const float* oscale;      // split-product plan: [128] 1 / (power of two the channel's weights were multiplied with)
  void* out;
  uint32_t out_bytes;
  int32_t N, Ho, Wo, out_pix_stride;
  int32_t tiles_x, tiles_y, n_tiles;
  OkpFastDiv div_tiles_frame, div_tiles_x;
  int32_t* range_flag;      // split-product plan: raised when a frame value or a result leaves the fp16 range (okp_stem_set_range_flag), or NULL
  int64_t out_total_bytes;  // split-product plan: bytes of the whole output view (it may pass 2 GiB: the kernel addresses frame by frame)
  int64_t out_frame_bytes;  // ... and of one frame of it
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ int fastdiv(int x, const OkpFastDiv& f) {
  return f.mul ? (int)(__umulhi((uint32_t)x, f.mul) >> f.shift) : x;
}

// NCHW = true: the patch is read straight from the caller's fp32 NCHW frames (zero padding by range checks), rounded to
// bf16 and written to LDS by the threads themselves - the packing pass (okp_pack_frames) and its 138 MB round trip go away.
template <typename T, bool NCHW>
__global__ __launch_bounds__(256, 2) void okp_stem_kernel(const StemParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * PATCH_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave >> 1, wp = wave & 1;
  const int j = lane & 31, h = lane >> 5;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, (int)p.src_bytes, 0x00020000);

  // weight (B operand) fragments: resident for the whole kernel
  u32x4 wa[2][KSTEPS];
  {
    const u32x4* wf = reinterpret_cast<const u32x4*>(p.wfrag) + (size_t)wc * 2 * KSTEPS * 64 + lane;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) wa[b][s] = wf[(b * KSTEPS + s) * 64];
  }
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
  const uint32_t ps2 = (uint32_t)p.out_pix_stride * 2u;                       // bytes per output pixel
  const uint32_t st_lane = (uint32_t)(4 * h) * ps2 + (uint32_t)(64 * wc + 2 * j) * 2u;
  // this lane's two channels (64 wc + 2 j, + 1): their biases are the initial accumulator values
  const float bias0 = p.bias[64 * wc + 2 * j], bias1 = p.bias[64 * wc + 2 * j + 1];

  // patch loader: LDS-DMA instruction q of this wave covers chunks [64 (3 wave + q), +64) of the linear patch image
  constexpr int DMA_PER_WAVE = PATCH_BYTES / 1024 / 4;      // 3
  uint32_t ld_off[DMA_PER_WAVE];                            // offset inside the patch's first row/column origin
#pragma unroll
  for (int q = 0; q < DMA_PER_WAVE; ++q) {
    const int ci = 64 * (DMA_PER_WAVE * wave + q) + lane;
    const int row = ci / CHUNKS_ROW, col = ci - row * CHUNKS_ROW;
    ld_off[q] = ci < CHUNKS ? (uint32_t)row * (uint32_t)p.Wp * 8u + (uint32_t)col * 16u : kInvalidOff;
  }
  auto tile_coords = [&](int tile, int& n, int& oy0, int& ox0) {
    n = fastdiv(tile, p.div_tiles_frame);
    const int r = tile - n * p.tiles_x * p.tiles_y;
    const int ty = fastdiv(r, p.div_tiles_x);
    oy0 = ty * TH;
    ox0 = (r - ty * p.tiles_x) * TW;
  };
  auto issue_patch = [&](int tile, int buf) {
    int n, oy0, ox0;
    tile_coords(tile, n, oy0, ox0);
    // patch origin = packed pixel (2 oy0, 2 ox0); rows beyond the packed frame end past the buffer -> zeros
    const uint32_t base = (uint32_t)(((long)n * p.Hp + 2 * oy0) * p.Wp + 2 * ox0) * 8u;
    char* dst = smem + buf * PATCH_BYTES + (DMA_PER_WAVE * wave) * 1024;
#pragma unroll
    for (int q = 0; q < DMA_PER_WAVE; ++q) {
      const uint32_t off = ld_off[q] == kInvalidOff ? kInvalidOff : base + ld_off[q];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(dst + q * 1024), 16, (int)off, 0, 0, 0);
    }
  };

  // fp32 NCHW source: thread t owns patch pixels t, t + 256, ... (row-major over the 21 x 70 patch), three planes each
  constexpr int PPT = (PR * PC + 255) / 256;                // 6 pixels per thread
  float pv[NCHW ? PPT : 1][3];
  auto load_patch = [&](int tile) {
    if constexpr (NCHW) {
      int n, oy0, ox0;
      tile_coords(tile, n, oy0, ox0);
      const uint32_t plane = (uint32_t)p.H * (uint32_t)p.W * 4u;
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx / PC, col = idx - row * PC;
        const int y = 2 * oy0 - 3 + row, x = 2 * ox0 - 3 + col;
        const bool ok = idx < PR * PC && y >= 0 && y < p.H && x >= 0 && x < p.W;
        const uint32_t off = ok ? ((uint32_t)(n * 3) * (uint32_t)p.H + (uint32_t)y) * (uint32_t)p.W * 4u + (uint32_t)x * 4u : kInvalidOff;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          pv[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (int)(ok ? off + (uint32_t)c * plane : kInvalidOff), 0, 0));
      }
    }
  };
  auto store_patch = [&](int buf) {
    if constexpr (NCHW) {
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const int idx = tid + 256 * i;
        if (idx < PATCH_BYTES / 8) {
          typename H16<T>::x4 o;
          o[0] = (T)pv[i][0]; o[1] = (T)pv[i][1]; o[2] = (T)pv[i][2]; o[3] = (T)0.f;
          *reinterpret_cast<typename H16<T>::x4*>(smem + buf * PATCH_BYTES + idx * 8) = o;
        }
      }
    }
  };

  int tile = blockIdx.x;
  if (tile >= p.n_tiles) return;
  if constexpr (NCHW) { load_patch(tile); store_patch(0); }
  else issue_patch(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int buf = 0;
  for (; tile < p.n_tiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if (next < p.n_tiles) { if constexpr (NCHW) load_patch(next); else issue_patch(next, buf ^ 1); }
    int n, oy0, ox0;
    tile_coords(tile, n, oy0, ox0);
    const char* patch = smem + buf * PATCH_BYTES;
    // byte address of this lane's B chunk for (row ry of the tile, ky, kx0): ((2 ry + ky) * PC + 2 j + kx0 + 2 h) * 8
    const int lane_off = (2 * j + 2 * h) * 8;
#pragma unroll 1
    for (int g = 0; g < 2; ++g) {                 // two groups of two output rows
      f32x16 acc[2][2];
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[0][0][e] = bias0; acc[1][0][e] = bias0; acc[0][1][e] = bias1; acc[1][1][e] = bias1; }
      const int ry0 = 4 * wp + 2 * g;
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        const int ky = s >> 1, kx0 = 4 * (s & 1);
        u32x4 pf[2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
          pf[r] = *reinterpret_cast<const u32x4*>(patch + (2 * (ry0 + r) + ky) * PITCH + kx0 * 8 + lane_off);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[r][b] = H16<T>::mfma32(pf[r], wa[b][s], acc[r][b]);
      }
      if (g == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next patch has landed, older stores have drained
      // ReLU, round to bf16, channel pair -> one dword; a half-wave writes 128 contiguous bytes of one pixel.
      // Buffer stores: the lane part of the address is one VGPR, the register-dependent part a scalar offset.
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int oy = oy0 + ry0 + r;
        if (oy < p.Ho) {
          const uint32_t row_off = (uint32_t)(((long)n * p.Ho + oy) * p.Wo + ox0) * ps2;       // scalar
          const bool full = ox0 + TW <= p.Wo;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int i0 = 8 * (e >> 2) + (e & 3);                  // pixel (MFMA row) of register e is i0 + 4 h
            const uint32_t v = okp_pack2<T>(fmaxf(acc[r][0][e], 0.f), fmaxf(acc[r][1][e], 0.f));
            if (full || ox0 + i0 + 4 * h < p.Wo)
              __builtin_amdgcn_raw_buffer_store_b32(v, rs_o, (int)st_lane, (int)(row_off + (uint32_t)i0 * ps2), 0);
          }
        }
      }
    }
    if constexpr (NCHW) { if (next < p.n_tiles) store_patch(buf ^ 1); }     // (its loads were waited for with the stores above)
    __syncthreads();            // every wave's part of the next patch is in LDS; this patch is free
    buf ^= 1;
  }
}

// ---- split-product form (OKP_F32X3): fp32 NHWC output, every product as x_hi w_hi + x_lo w_hi + x_hi w_lo on the fp16 matrix pipe ----
// The generic tap-list kernel ran this layer of the float32x3 configuration at 2.1 TB/s of its output bytes (510 us per 32 frames:
// seven K-slices between a prologue and an LDS-transposed epilogue).  Same structure as the 16-bit kernel above, with three changes:
//  * the patch is kept in LDS as TWO planes in the 16-bit kernel's layout - fp16(x) and fp16(x - fp16(x)) - written by the threads that
//    fetched the fp32 NCHW pixels (the split happens once per patch pixel, not per tap);
//  * a wave owns 32 output channels (one column block) and all eight rows of the tile: its weight fragments, hi and lo, fill the 112
//    registers the 16-bit kernel spends on 64 channels x hi only;
//  * the accumulators start at bias x scale and are multiplied by 1 / scale at the end (the per-channel power of two that keeps the
//    low halves of small weights normal numbers: okp_conv_create does the same for split-product convolution plans); a half-wave's
//    store is 128 contiguous bytes (32 channels) of one fp32 NHWC pixel.
//  * PAIRS: the output in pair format (include/okp.h, okp_conv_args: [8 x fp16 hi | 8 x fp16 lo] per 8 channels - what the split-product
//    patch kernel of pre[1] would make of the fp32 values in LDS).  A lane holds one channel of a pixel: it splits its value, exchanges
//    the packed (hi, lo) with its neighbour lane and stores one dword as before - even lanes the hi halves of channels j, j + 1, odd
//    lanes the lo halves of j - 1, j - so a half-wave's store is still the 128 contiguous bytes of 32 channels of one pixel.
template <bool PAIRS>
__global__ __launch_bounds__(256, 2) void okp_stem_x3_kernel(const StemParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * PATCH_BYTES];      // [buffer][hi plane | lo plane]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  bool range_bad = false;                    // a frame value outside the fp16 range or not finite (okp_unsplittable: exact, the frames come from outside)

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, (int)p.src_bytes, 0x00020000);
  u32x4 wh[KSTEPS], wl[KSTEPS];              // B-operand fragments of this wave's 32 channels: [wave][hi | lo][k-step][lane]
  {
    const u32x4* wf = reinterpret_cast<const u32x4*>(p.wfrag) + (size_t)wave * 2 * KSTEPS * 64 + lane;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) { wh[s] = wf[s * 64]; wl[s] = wf[(KSTEPS + s) * 64]; }
  }
  // (the output resource is built per tile from the tile's FRAME - 64-bit base, frame-relative 32-bit offsets - so that an fp32 output of
  //  a whole batch, 2.1 GB at 64 frames, needs no frame chunks: okp_stem_forward_nchw lifts the 2 GiB view limit for this kernel)
  const uint32_t ps4 = (uint32_t)p.out_pix_stride * 4u;                       // bytes per output pixel
  const uint32_t st_lane = (uint32_t)(4 * h) * ps4 + (PAIRS ? (uint32_t)(32 * wave) * 4u + (uint32_t)(j >> 3) * 32u + (uint32_t)((j & 7) >> 1) * 4u + (uint32_t)(j & 1) * 16u
                                                                : (uint32_t)(32 * wave + j) * 4u);
  // v_perm_b32 selector of the stored dword from {neighbour's (hi, lo), own (hi, lo)}: even lanes [own hi, neighbour's hi], odd lanes
  // [neighbour's lo, own lo] (bytes 0-3 = second operand, 4-7 = first)
  const uint32_t pair_sel = (j & 1) ? 0x03020706u : 0x05040100u;
  const float bias0 = p.bias[32 * wave + j], osc = p.oscale[32 * wave + j];

  auto tile_coords = [&](int tile, int& n, int& oy0, int& ox0) {
    n = fastdiv(tile, p.div_tiles_frame);
    const int r = tile - n * p.tiles_x * p.tiles_y;
    const int ty = fastdiv(r, p.div_tiles_x);
    oy0 = ty * TH;
    ox0 = (r - ty * p.tiles_x) * TW;
  };
  constexpr int PPT = (PR * PC + 255) / 256;                // 6 patch pixels per thread
  float pv[PPT][3];
  auto load_patch = [&](int tile) {
    int n, oy0, ox0;
    tile_coords(tile, n, oy0, ox0);
    const uint32_t plane = (uint32_t)p.H * (uint32_t)p.W * 4u;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / PC, col = idx - row * PC;
      const int y = 2 * oy0 - 3 + row, x = 2 * ox0 - 3 + col;
      const bool ok = idx < PR * PC && y >= 0 && y < p.H && x >= 0 && x < p.W;
      const uint32_t off = ((uint32_t)(n * 3) * (uint32_t)p.H + (uint32_t)y) * (uint32_t)p.W * 4u + (uint32_t)x * 4u;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        pv[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (int)(ok ? off + (uint32_t)c * plane : kInvalidOff), 0, 0));
    }
  };
  auto store_patch = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int idx = tid + 256 * i;
      if (idx < PATCH_BYTES / 8) {
        f16x4 hi, lo;
#pragma unroll
        for (int c = 0; c < 3; ++c) { hi[c] = (_Float16)pv[i][c]; lo[c] = (_Float16)(pv[i][c] - (float)hi[c]); range_bad |= okp_unsplittable(pv[i][c]); }
        hi[3] = (_Float16)0.f; lo[3] = (_Float16)0.f;
        *reinterpret_cast<f16x4*>(smem + (2 * buf) * PATCH_BYTES + idx * 8) = hi;
        *reinterpret_cast<f16x4*>(smem + (2 * buf + 1) * PATCH_BYTES + idx * 8) = lo;
      }
    }
  };

  int tile = blockIdx.x;
  if (tile >= p.n_tiles) return;
  load_patch(tile);
  store_patch(0);
  __syncthreads();

  int buf = 0;
  for (; tile < p.n_tiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if (next < p.n_tiles) load_patch(next);
    int n, oy0, ox0;
    tile_coords(tile, n, oy0, ox0);
    const int64_t frame_off = (int64_t)n * p.out_frame_bytes, left = p.out_total_bytes - frame_off;
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(static_cast<char*>(p.out) + frame_off, 0, (int)(left < 0x7FFF0000ll ? left : 0x7FFF0000ll), 0x00020000);
    const char* const patch = smem + (2 * buf) * PATCH_BYTES;
    const int lane_off = (2 * j + 2 * h) * 8;
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {                 // four groups of two output rows
      f32x16 acc[2];
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[0][e] = bias0; acc[1][e] = bias0; }
      const int ry0 = 2 * g;
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        const int ky = s >> 1, kx0 = 4 * (s & 1);
        u32x4 ph[2], pl[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const char* a = patch + (2 * (ry0 + r) + ky) * PITCH + kx0 * 8 + lane_off;
          ph[r] = *reinterpret_cast<const u32x4*>(a);
          pl[r] = *reinterpret_cast<const u32x4*>(a + PATCH_BYTES);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) acc[r] = H16<_Float16>::mfma32(pl[r], wh[s], acc[r]);
#pragma unroll
        for (int r = 0; r < 2; ++r) acc[r] = H16<_Float16>::mfma32(ph[r], wl[s], acc[r]);
#pragma unroll
        for (int r = 0; r < 2; ++r) acc[r] = H16<_Float16>::mfma32(ph[r], wh[s], acc[r]);
      }
      if (g == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next patch's pixels have arrived, older stores have drained
      float range_m = 0.f;                          // largest result of this row pair (okp_range_max)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int oy = oy0 + ry0 + r;
        if (oy < p.Ho) {
          const uint32_t row_off = (uint32_t)(oy * p.Wo + ox0) * ps4;       // scalar, relative to the frame
          const bool full = ox0 + TW <= p.Wo;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int i0 = 8 * (e >> 2) + (e & 3);                  // pixel (MFMA row) of register e is i0 + 4 h
            const float v = fmaxf(acc[r][e] * osc, 0.f);
            range_m = okp_range_max(range_m, v);
            uint32_t word = __builtin_bit_cast(uint32_t, v);
            if constexpr (PAIRS) {
              f16x2 hl;
              hl[0] = (_Float16)v;
              hl[1] = (_Float16)(v - (float)hl[0]);                   // exact difference, as okp_split8 forms it
              const uint32_t own = __builtin_bit_cast(uint32_t, hl);
              const uint32_t nb = (uint32_t)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]: lane ^ 1
              word = __builtin_amdgcn_perm(nb, own, pair_sel);
            }
            if (full || ox0 + i0 + 4 * h < p.Wo)
              __builtin_amdgcn_raw_buffer_store_b32(word, rs_o, (int)st_lane, (int)(row_off + (uint32_t)i0 * ps4), 0);
          }
        }
      }
      range_bad |= okp_range_exceeded(range_m);
    }
    if (next < p.n_tiles) store_patch(buf ^ 1);
    __syncthreads();            // every thread's part of the next patch is in LDS; this patch is free
    buf ^= 1;
  }
  okp_raise_range_flag(p.range_flag, range_bad);
}

}  // namespace

static okp_stem* stem_create_x3(const float* w, const float* bias);

struct okp_stem {
  void* wfrag_dev;
  float* bias_dev;
  int dtype;
  float* oscale_dev;        // OKP_F32X3 only
  int32_t* range_flag;      // OKP_F32X3 only: okp_stem_set_range_flag
};


// Split-product plan: fragments [wave 4][hi | lo][k-step 14][lane 64][8 fp16], column jc of wave w = channel 32 w + jc; a channel's
// weights are multiplied by the power of two that puts the largest of them in [256, 512) before the split (its low halves are then
// normal fp16 numbers down to 2^-12 of it), the bias by the same factor, and the kernel multiplies the accumulator by its inverse.
static okp_stem* stem_create_x3(const float* w, const float* bias) {
  std::vector<uint16_t> frag((size_t)4 * 2 * KSTEPS * 64 * 8, 0);
  std::vector<float> bias_s(kCout), osc(kCout, 1.f);
  for (int co = 0; co < kCout; ++co) {
    float mx = 0.f;
    for (int i = 0; i < 147; ++i) mx = std::max(mx, std::fabs(w[co * 147 + i]));
    float sc = 1.f;
    if (mx > 0.f && std::isfinite(mx)) { int e; std::frexp(mx, &e); sc = std::ldexp(1.f, 9 - e); }
    osc[co] = 1.f / sc;
    bias_s[co] = bias[co] * sc;
    const int wv = co / 32, jc = co % 32;
    for (int s = 0; s < KSTEPS; ++s)
      for (int h = 0; h < 2; ++h) {
        const int ky = s >> 1, kx0 = 4 * (s & 1), lane = jc + 32 * h;
        uint16_t* dh = &frag[((((size_t)wv * 2 + 0) * KSTEPS + s) * 64 + lane) * 8];
        uint16_t* dl = &frag[((((size_t)wv * 2 + 1) * KSTEPS + s) * 64 + lane) * 8];
        for (int e = 0; e < 8; ++e) {
          const int kx = kx0 + 2 * h + e / 4, c = e % 4;
          if (kx < 7 && c < 3) {
            const float v = w[((co * 3 + c) * 7 + ky) * 7 + kx] * sc;
            const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
            std::memcpy(&dh[e], &hi, 2); std::memcpy(&dl[e], &lo, 2);
          }
        }
      }
  }
  okp_stem* st = new okp_stem{nullptr, nullptr, OKP_F32X3, nullptr, nullptr};
  if (okp_check_hip(hipMalloc(&st->wfrag_dev, frag.size() * 2), "okp_stem_create: hipMalloc") ||
      okp_check_hip(hipMalloc((void**)&st->bias_dev, kCout * 4), "okp_stem_create: hipMalloc") ||
      okp_check_hip(hipMalloc((void**)&st->oscale_dev, kCout * 4), "okp_stem_create: hipMalloc") ||
      okp_check_hip(hipMemcpy(st->wfrag_dev, frag.data(), frag.size() * 2, hipMemcpyHostToDevice), "okp_stem_create: copy") ||
      okp_check_hip(hipMemcpy(st->bias_dev, bias_s.data(), kCout * 4, hipMemcpyHostToDevice), "okp_stem_create: copy") ||
      okp_check_hip(hipMemcpy(st->oscale_dev, osc.data(), kCout * 4, hipMemcpyHostToDevice), "okp_stem_create: copy")) {
    okp_stem_destroy(st);
    return nullptr;
  }
  return st;
}

// w: HOST fp32 [128][3][7][7] (BatchNorm folded), bias: HOST fp32 [128]
extern "C" okp_stem* okp_stem_create(const float* w, const float* bias) { return okp_stem_create_dtype(OKP_BF16, w, bias); }

extern "C" okp_stem* okp_stem_create_dtype(int dtype, const float* w, const float* bias) {
  if (!w || !bias) { okp_set_error("okp_stem_create: null argument"); return nullptr; }
  if (dtype == OKP_F32X3) return stem_create_x3(w, bias);
  if (!okp_is16(dtype)) { okp_set_error("okp_stem_create: the stem kernel computes in bf16, fp16 or split products (dtype %d)", dtype); return nullptr; }
  std::vector<uint16_t> frag((size_t)2 * 2 * KSTEPS * 64 * 8, 0);
  for (int wc = 0; wc < 2; ++wc)
    for (int b = 0; b < 2; ++b)
      for (int s = 0; s < KSTEPS; ++s)
        for (int lane = 0; lane < 64; ++lane) {
          const int jc = lane & 31, h = lane >> 5;                // MFMA column jc, k half h
          const int co = 64 * wc + 2 * jc + b;
          const int ky = s >> 1, kx0 = 4 * (s & 1);
          uint16_t* dst = &frag[((((size_t)wc * 2 + b) * KSTEPS + s) * 64 + lane) * 8];
          for (int e = 0; e < 8; ++e) {
            const int kx = kx0 + 2 * h + e / 4, c = e % 4;
            dst[e] = (kx < 7 && c < 3) ? okp_f32_to_16(dtype, w[((co * 3 + c) * 7 + ky) * 7 + kx]) : 0;
          }
        }
  okp_stem* st = new okp_stem{nullptr, nullptr, dtype, nullptr, nullptr};
  if (okp_check_hip(hipMalloc(&st->wfrag_dev, frag.size() * 2), "okp_stem_create: hipMalloc") ||
      okp_check_hip(hipMalloc((void**)&st->bias_dev, kCout * 4), "okp_stem_create: hipMalloc") ||
      okp_check_hip(hipMemcpy(st->wfrag_dev, frag.data(), frag.size() * 2, hipMemcpyHostToDevice), "okp_stem_create: copy") ||
      okp_check_hip(hipMemcpy(st->bias_dev, bias, kCout * 4, hipMemcpyHostToDevice), "okp_stem_create: copy")) {
    if (st->wfrag_dev) (void)hipFree(st->wfrag_dev);
    if (st->bias_dev) (void)hipFree(st->bias_dev);
    delete st;
    return nullptr;
  }
  return st;
}

extern "C" int okp_stem_set_range_flag(okp_stem* st, int32_t* flag_dev) {
  if (!st || st->dtype != OKP_F32X3) { okp_set_error("okp_stem_set_range_flag: a split-product (OKP_F32X3) stem"); return OKP_EINVAL; }
  st->range_flag = flag_dev;
  return OKP_OK;
}

extern "C" void okp_stem_destroy(okp_stem* st) {
  if (!st) return;
  if (st->wfrag_dev) (void)hipFree(st->wfrag_dev);
  if (st->bias_dev) (void)hipFree(st->bias_dev);
  if (st->oscale_dev) (void)hipFree(st->oscale_dev);
  delete st;
}

extern "C" int okp_stem_forward(const okp_stem* st, int32_t n, int32_t h, int32_t w, const okp_tensor* packed, const okp_tensor* out, void* stream) {
  if (!st || !packed || !out || !packed->data || !out->data) { okp_set_error("okp_stem_forward: null argument"); return OKP_EINVAL; }
  if (n < 1 || h < 1 || w < 1) { okp_set_error("okp_stem_forward: empty problem"); return OKP_EINVAL; }
  if (st->dtype == OKP_F32X3) { okp_set_error("okp_stem_forward: the split-product stem reads fp32 NCHW frames (okp_stem_forward_nchw); packed frames go through okp_conv_forward"); return OKP_EINVAL; }
  const int ho = (h + 6 - 7) / 2 + 1, wo = (w + 6 - 7) / 2 + 1;
  if (packed->pix_stride != 4 || packed->h != h + 6 || packed->w < 2 * (wo - 1) + 8 || packed->w < w + 6 || packed->w % 2 ||
      ((uintptr_t)packed->data) % 16) {
    okp_set_error("okp_stem_forward: packed frame must be the okp_pack_frames layout (h+6 rows, even width >= w+6, 4 channels)");
    return OKP_EINVAL;
  }
  if (packed->bytes <= 0 || packed->bytes >= 0x7FFF0000ll || out->bytes <= 0 || out->bytes >= 0x7FFF0000ll) { okp_set_error("okp_stem_forward: views must be < 2 GiB"); return OKP_EINVAL; }
  if (out->h != ho || out->w != wo || out->pix_stride < kCout || (out->pix_stride * 2) % 64 || ((uintptr_t)out->data) % 64) {
    okp_set_error("okp_stem_forward: out must be %dx%d with 64-byte aligned pixels of >= 128 channels", ho, wo);
    return OKP_EINVAL;
  }
  if ((int64_t)n * ho * wo * out->pix_stride * 2 > out->bytes + (int64_t)(out->pix_stride - kCout) * 2) { okp_set_error("okp_stem_forward: out view too small"); return OKP_EINVAL; }
  StemParams p;
  std::memset(&p, 0, sizeof(p));
  p.src = packed->data; p.src_bytes = (uint32_t)packed->bytes; p.Hp = packed->h; p.Wp = packed->w;
  p.wfrag = st->wfrag_dev; p.bias = st->bias_dev;
  p.out = out->data; p.out_bytes = (uint32_t)out->bytes; p.N = n; p.Ho = ho; p.Wo = wo; p.out_pix_stride = out->pix_stride;
  p.tiles_x = (wo + TW - 1) / TW; p.tiles_y = (ho + TH - 1) / TH;
  const long tiles = (long)n * p.tiles_x * p.tiles_y;
  if (tiles >= 0x7FFFFFFFl) { okp_set_error("okp_stem_forward: too many tiles"); return OKP_EINVAL; }
  p.n_tiles = (int)tiles;
  p.div_tiles_frame = okp_fastdiv((uint32_t)(p.tiles_x * p.tiles_y));
  p.div_tiles_x = okp_fastdiv((uint32_t)p.tiles_x);
  const int resident = 256 * 2;        // two workgroups per CU: one computes while the other's stores drain
  const int grid = p.n_tiles < resident ? p.n_tiles : resident;
  if (st->dtype == OKP_BF16) hipLaunchKernelGGL((okp_stem_kernel<__bf16, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((okp_stem_kernel<_Float16, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return okp_check_hip(hipGetLastError(), "okp_stem launch");
}

static int stem_forward_nchw(const okp_stem* st, int32_t n, int32_t h, int32_t w, const float* frames_nchw_dev, const okp_tensor* out, bool pairs, void* stream) {
  if (!st || !frames_nchw_dev || !out || !out->data) { okp_set_error("okp_stem_forward_nchw: null argument"); return OKP_EINVAL; }
  if (pairs && st->dtype != OKP_F32X3) { okp_set_error("okp_stem_forward_nchw_pairs: the pair format belongs to OKP_F32X3 stems"); return OKP_EINVAL; }
  if (n < 1 || h < 1 || w < 1) { okp_set_error("okp_stem_forward_nchw: empty problem"); return OKP_EINVAL; }
  const int ho = (h + 6 - 7) / 2 + 1, wo = (w + 6 - 7) / 2 + 1;
  const int64_t src_bytes = (int64_t)n * 3 * h * w * 4;
  const int esz = st->dtype == OKP_F32X3 ? 4 : 2;          // the split-product plan writes fp32 NHWC
  // views are limited to 2 GiB (32-bit buffer offsets, bit 31 = "masked") - except the OUTPUT of the split-product kernel, which addresses
  // frame by frame: there one frame must stay below the limit, the view below 2^40
  const int64_t out_limit = st->dtype == OKP_F32X3 ? (1ll << 40) : 0x7FFF0000ll;
  if (src_bytes >= 0x7FFF0000ll || out->bytes <= 0 || out->bytes >= out_limit || (int64_t)ho * wo * out->pix_stride * esz >= 0x7FFF0000ll) {
    okp_set_error("okp_stem_forward_nchw: views must be < 2 GiB (sub-batch the frames)"); return OKP_EINVAL;
  }
  if (out->h != ho || out->w != wo || out->pix_stride < kCout || (out->pix_stride * esz) % 64 || ((uintptr_t)out->data) % 64 || ((uintptr_t)frames_nchw_dev) % 4) {
    okp_set_error("okp_stem_forward_nchw: out must be %dx%d with 64-byte aligned pixels of >= 128 channels", ho, wo);
    return OKP_EINVAL;
  }
  if ((int64_t)n * ho * wo * out->pix_stride * esz > out->bytes + (int64_t)(out->pix_stride - kCout) * esz) { okp_set_error("okp_stem_forward_nchw: out view too small"); return OKP_EINVAL; }
  StemParams p;
  std::memset(&p, 0, sizeof(p));
  p.src = frames_nchw_dev; p.src_bytes = (uint32_t)src_bytes; p.H = h; p.W = w;
  p.wfrag = st->wfrag_dev; p.bias = st->bias_dev; p.oscale = st->oscale_dev; p.range_flag = st->range_flag;
  p.out = out->data; p.out_bytes = (uint32_t)(out->bytes < 0x7FFF0000ll ? out->bytes : 0x7FFF0000ll); p.N = n; p.Ho = ho; p.Wo = wo; p.out_pix_stride = out->pix_stride;
  p.out_total_bytes = out->bytes; p.out_frame_bytes = (int64_t)ho * wo * out->pix_stride * esz;
  p.tiles_x = (wo + TW - 1) / TW; p.tiles_y = (ho + TH - 1) / TH;
  const long tiles = (long)n * p.tiles_x * p.tiles_y;
  if (tiles >= 0x7FFFFFFFl) { okp_set_error("okp_stem_forward_nchw: too many tiles"); return OKP_EINVAL; }
  p.n_tiles = (int)tiles;
  p.div_tiles_frame = okp_fastdiv((uint32_t)(p.tiles_x * p.tiles_y));
  p.div_tiles_x = okp_fastdiv((uint32_t)p.tiles_x);
  const int resident = 256 * 2;
  const int grid = p.n_tiles < resident ? p.n_tiles : resident;
  if (st->dtype == OKP_F32X3 && pairs) hipLaunchKernelGGL(okp_stem_x3_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else if (st->dtype == OKP_F32X3) hipLaunchKernelGGL(okp_stem_x3_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else if (st->dtype == OKP_BF16) hipLaunchKernelGGL((okp_stem_kernel<__bf16, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((okp_stem_kernel<_Float16, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return okp_check_hip(hipGetLastError(), "okp_stem launch");
}

extern "C" int okp_stem_forward_nchw(const okp_stem* st, int32_t n, int32_t h, int32_t w, const float* frames_nchw_dev, const okp_tensor* out, void* stream) {
  return stem_forward_nchw(st, n, h, w, frames_nchw_dev, out, false, stream);
}

extern "C" int okp_stem_forward_nchw_pairs(const okp_stem* st, int32_t n, int32_t h, int32_t w, const float* frames_nchw_dev, const okp_tensor* out, void* stream) {
  return stem_forward_nchw(st, n, h, w, frames_nchw_dev, out, true, stream);
}

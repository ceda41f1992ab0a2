// The three prediction heads of one stack in ONE launch (bf16, features = 128, gfx950):
//
//   per head h (heat, depth, centre):   h1 = relu(W1_h x + b1_h)        1x1 + BN + ReLU   256 -> 128
//                                       h2 = relu(W2_h h1 + b2_h)       1x1 + BN + ReLU   128 -> 32
//                                       out_o = act(w3_o . h2 + b3_o)   1x1 + bias        32 -> K (or 2 (K-1)), NCHW fp32
//
// Replaces prediction_module (perception/models.py:13-18) x 3 + the deployed wrapper's sigmoid
// (scripts/package_model.py:28).  As three launches (a 256 -> 384 GEMM, a block-diagonal 384 -> 96 GEMM and the
// pointwise output kernel) the 384- and 96-channel intermediates make a round trip through HBM (201 + 50 MB written
// and read per 64 frames against 134 MB of input); here they live in LDS.
//
// A workgroup (4 waves) serves ONE head for its tiles of 64 consecutive pixels; the three workgroups of a tile are
// neighbours in the grid and share the tile's x lines in L2.  The x tile (64 x 512 B) is fetched by LDS-DMA;
// GEMM 1 runs on 16x16x32 MFMAs with pixels as rows and wave w owning channels [32 w, 32 w + 32) of the head, its weight
// fragments RESIDENT in registers for the whole launch (read once from the fragment-ordered copy of the plan,
// okp_ensure_frags; streaming them per tile cost 216 KB of L2 traffic per 64 pixels), the result goes to LDS as bf16;
// GEMM 2 (K = 128, 32 output channels) gives each wave one 16-pixel block; the last layer is a 32-term dot product per
// (pixel, output) on the vector ALUs with pixels along the lanes, so the NCHW stores are contiguous.
#include <cstring>

#include "okp_internal.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr uint32_t kInvalid = 0x80000000u;

constexpr int TP = 64;                    // pixels per tile
constexpr int CIN = 256, F = 128, F2 = 32;
constexpr int KS1 = CIN / 32, KS2 = F / 32;
constexpr int OFF_X = 0;                  // [64][512 B], 16-byte chunks XOR-swizzled by row & 15
constexpr int OFF_H1 = OFF_X + TP * CIN * 2;     // [64][256 B]
constexpr int H2_PITCH = F2 + 1;          // floats per row: odd pitch, so that a column walk over pixels is conflict-free
constexpr int OFF_H2 = OFF_H1 + TP * F * 2;      // [64][33] fp32
constexpr int OFF_W3 = OFF_H2 + TP * H2_PITCH * 4;   // [n_out][32] fp32, then [n_out] bias
constexpr int LDS_BYTES = OFF_W3 + OKP_HEAD_MAX_OUT * 33 * 4;

struct HeadsParams {
  const void* x; uint32_t x_bytes; int32_t x_ps;
  long n_pix;                             // N * H * W
  int32_t HW;
  const void* w1; const float* b1;        // fragment-ordered 256 -> 384 weights: [12 waves][2][8][64][16 B]; bias [384]
  const void* w2; const float* b2;        // fragment-ordered 384 -> 96 (block diagonal): [3][2][12][64][16 B]; bias [96]
  const float* w3; const float* b3;       // [n_out][32], [n_out]
  int32_t n_out;
  int32_t head_of[OKP_HEAD_MAX_OUT];      // which head (0..2) output o reads
  int32_t act[OKP_HEAD_MAX_OUT];
  float* out_ptr[OKP_HEAD_MAX_OUT];
  int64_t out_n_stride[OKP_HEAD_MAX_OUT];
  int32_t n_tiles;
};

__device__ __forceinline__ uint32_t xoff(int row, int chunk, int rowbytes) { return (uint32_t)row * rowbytes + (uint32_t)((chunk ^ (row & 15)) << 4); }

template <typename T>
__global__ __launch_bounds__(256, 2) void okp_heads_kernel(const HeadsParams p) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, q = lane >> 4;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  float* const h2 = reinterpret_cast<float*>(smem + OFF_H2);
  float* const w3s = reinterpret_cast<float*>(smem + OFF_W3);
  for (int i = tid; i < p.n_out * 33; i += 256) w3s[i] = i < p.n_out * 32 ? p.w3[i] : p.b3[i - p.n_out * 32];
  const int ch0 = 32 * w + 2 * l16;                 // this lane's channel pair inside a head (GEMM 1)
  // LDS-DMA of the x tile: instruction i of this wave covers linear bytes [1024 (8 w + i), +1024) = two 512-byte rows
  int d_row[8];
  uint32_t d_chunk[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int lin = (8 * w + i) * 64 + lane;        // 16-byte position in the tile
    d_row[i] = lin >> 5;
    d_chunk[i] = (uint32_t)(((lin & 31) ^ (d_row[i] & 15)) << 4);      // the chunk the read-side swizzle expects there
  }

  // A workgroup serves ONE head (blockIdx.x % 3) for its tiles, so that head's GEMM-1 and GEMM-2 weight fragments stay in
  // registers for the whole launch (64 + 8 VGPRs) instead of being re-streamed from L2 for every 64-pixel tile (216 KB
  // per tile and workgroup, 885 MB per 64 frames); the three workgroups of a tile run side by side and share its x lines in L2.
  // grid = 24 m workgroups: workgroup g runs on XCD g % 8 (round-robin dispatch); the three heads of a tile are given to
  // workgroups g, g + 8, g + 16 of the same XCD, so the tile's x lines are fetched into ONE L2 instead of three.
  // (Smaller grids, for fewer tiles than resident groups, use the plain g % 3 order.)
  const bool xcd_order = gridDim.x % 24 == 0;
  const int gm = blockIdx.x >> 3;
  const int h = xcd_order ? gm % 3 : (int)(blockIdx.x % 3);
  const int tile0 = xcd_order ? (gm / 3) * 8 + (int)(blockIdx.x & 7) : (int)(blockIdx.x / 3);
  auto frag1 = [&](int ks, int b) { return static_cast<const u32x4*>(p.w1)[(size_t)(((4 * h + w) * 2 + b) * KS1 + ks) * 64 + lane]; };
  auto frag2 = [&](int kk, int b) { return static_cast<const u32x4*>(p.w2)[(size_t)((h * 2 + b) * (3 * KS2) + KS2 * h + kk) * 64 + lane]; };
  u32x4 wf[KS1][2], wg[KS2][2];
#pragma unroll
  for (int ks = 0; ks < KS1; ++ks) { wf[ks][0] = frag1(ks, 0); wf[ks][1] = frag1(ks, 1); }
#pragma unroll
  for (int kk = 0; kk < KS2; ++kk) { wg[kk][0] = frag2(kk, 0); wg[kk][1] = frag2(kk, 1); }
  const float b1a = p.b1[F * h + ch0], b1b = p.b1[F * h + ch0 + 1];
  const float b2a = p.b2[F2 * h + 2 * l16], b2b = p.b2[F2 * h + 2 * l16 + 1];

  auto issue_x = [&](int tile) {
    const long pix0 = (long)tile * TP;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long pix = pix0 + d_row[i];
      const uint32_t off = pix < p.n_pix ? (uint32_t)pix * (uint32_t)(p.x_ps * 2) + d_chunk[i] : kInvalid;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(smem + OFF_X + (8 * w + i) * 1024), 16, (int)off, 0, 0, 0);
    }
  };
  const int tstep = gridDim.x / 3;
  if (tile0 < p.n_tiles) issue_x(tile0);
  for (int tile = tile0; tile < p.n_tiles; tile += tstep) {
    const long pix0 = (long)tile * TP;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's x (issued behind GEMM 1 of the previous tile) has landed
    __syncthreads();                                // ... for every wave; the previous tile's readers of h2 are done; w3 is in LDS

    {
      // ---- GEMM 1: h1[64 px][128] = relu(W1_h x + b1_h) ------------------------------------------------------
      {
        f32x4 acc[4][2];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) { acc[pb][0] = f32x4{b1a, b1a, b1a, b1a}; acc[pb][1] = f32x4{b1b, b1b, b1b, b1b}; }
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
          u32x4 a[4];
#pragma unroll
          for (int pb = 0; pb < 4; ++pb) a[pb] = *reinterpret_cast<const u32x4*>(smem + OFF_X + xoff(16 * pb + l16, 4 * ks + q, CIN * 2));
#pragma unroll
          for (int pb = 0; pb < 4; ++pb) {
            acc[pb][0] = H16<T>::mfma16(a[pb], wf[ks][0], acc[pb][0]);
            acc[pb][1] = H16<T>::mfma16(a[pb], wf[ks][1], acc[pb][1]);
          }
        }
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t v = okp_pack2<T>(fmaxf(acc[pb][0][r], 0.f), fmaxf(acc[pb][1][r], 0.f));
            const int px = 16 * pb + 4 * q + r;
            *reinterpret_cast<uint32_t*>(smem + OFF_H1 + xoff(px, (ch0 * 2) >> 4, F * 2) + ((ch0 * 2) & 15)) = v;
          }
      }
      __syncthreads();
      // x is free (every wave has read its fragments): the next tile's x streams in under GEMM 2 and the last layer (89 -> 86 us)
      if (tile + tstep < p.n_tiles) issue_x(tile + tstep);
      // ---- GEMM 2: h2[64 px][32] = relu(W2_h h1 + b2_h); wave w takes pixel block w ---------------------------------
      {
        f32x4 acc0 = {b2a, b2a, b2a, b2a}, acc1 = {b2b, b2b, b2b, b2b};
#pragma unroll
        for (int kk = 0; kk < KS2; ++kk) {
          const u32x4 a = *reinterpret_cast<const u32x4*>(smem + OFF_H1 + xoff(16 * w + l16, 4 * kk + q, F * 2));
          acc0 = H16<T>::mfma16(a, wg[kk][0], acc0);
          acc1 = H16<T>::mfma16(a, wg[kk][1], acc1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int px = 16 * w + 4 * q + r;
          // rounded to bf16 like the stored activation of the unfused path, kept as fp32 for the dot products
          h2[px * H2_PITCH + 2 * l16] = (float)(T)fmaxf(acc0[r], 0.f);
          h2[px * H2_PITCH + 2 * l16 + 1] = (float)(T)fmaxf(acc1[r], 0.f);
        }
      }
      __syncthreads();
      // ---- last layer: out_o = act(w3_o . h2 + b3_o) for the outputs of this head; lanes along the pixels -------------
      {
        const int px = tid & 63, slot = tid >> 6;
        const long pix = pix0 + px;
        int seen = 0;
        for (int o = 0; o < p.n_out; ++o) {
          if (p.head_of[o] != h) continue;
          if ((seen++ & 3) != slot) continue;          // the head's outputs are dealt round-robin over the four waves
          float acc = w3s[p.n_out * 32 + o];
#pragma unroll
          for (int c = 0; c < F2; ++c) acc = fmaf(h2[px * H2_PITCH + c], w3s[o * 32 + c], acc);
          if (p.act[o] == OKP_ACT_SIGMOID) acc = 1.f / (1.f + expf(-acc));
          else if (p.act[o] == OKP_ACT_RELU) acc = fmaxf(acc, 0.f);
          if (pix < p.n_pix) {
            const long n = pix / p.HW;
            p.out_ptr[o][(size_t)n * p.out_n_stride[o] + (size_t)(pix - n * p.HW)] = acc;
          }
        }
      }
    }
  }
}

// ---- the same three heads for split-product plans (OKP_F32X3): fp32 tensors, every product as x_hi w_hi + x_lo w_hi + x_hi w_lo on the
// fp16 matrix pipe.  x arrives in PAIR FORMAT (include/okp.h, okp_conv_args.out_pairs: `cnvs` of the last stack writes it that way when
// its one reader is this kernel), so a tile is copied to LDS by LDS-DMA and multiplied as it lands - no conversion in this kernel.
//
// A workgroup (4 waves, two per CU) serves ONE head for its tiles of 32 consecutive pixels.  Both GEMMs run with CHANNELS as MFMA rows
// (A = weight fragments, hi and lo, resident in registers for the whole launch: 128 + 32 VGPRs) and PIXELS as columns, so a lane
// holds four adjacent channels of a pixel: h1 = relu(W1 x + b1) goes to LDS as pairs (8-byte hi / lo pieces, the pixel operand of GEMM
// 2), h2 = relu(W2 h1 + b2) as fp32 rows of 33 floats, and the last layer is the 32-term fp32 dot product of okp_head_out_kernel.
// LDS rows of 1 KiB (x) and 512 B (h1) are XOR-swizzled in 16-byte chunks by a key of the pixel row that is conflict-free for the
// ds_read_b128 lane groups of gfx950 with hi and lo chunks adjacent (searched against them: row bit 0 -> 8, bit 1 -> 4, bit 2 -> 1).
namespace x3 {

constexpr int TP = 32;
constexpr int OFF_X = 0;                                 // [32][1 KiB]: 256 channels as 32 pairs [hi | lo]
constexpr int OFF_H1 = OFF_X + TP * CIN * 4;             // [32][512 B]: 128 channels as 16 pairs
constexpr int OFF_H2 = OFF_H1 + TP * F * 4;              // [32][33] fp32
constexpr int OFF_W3 = OFF_H2 + TP * H2_PITCH * 4;       // [n_out][32] fp32, then [n_out] bias
constexpr int LDS_BYTES = OFF_W3 + OKP_HEAD_MAX_OUT * 33 * 4;
static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");

struct Params {
  const void* x; uint32_t x_bytes; int32_t x_ps;
  long n_pix;
  int32_t HW;
  const void* w1; const float* b1; const float* s1;      // fragment-ordered split weights [24 blocks of 16 channels][hi | lo][8 k-steps][64][16 B], bias, output scale
  const void* w2; const float* b2; const float* s2;      // [6 blocks][hi | lo][12 k-steps][64][16 B]
  const float* w3; const float* b3;
  int32_t n_out;
  int32_t head_of[OKP_HEAD_MAX_OUT];
  int32_t act[OKP_HEAD_MAX_OUT];
  float* out_ptr[OKP_HEAD_MAX_OUT];
  int64_t out_n_stride[OKP_HEAD_MAX_OUT];
  int32_t n_tiles;
  int32_t* range_flag;         // plan l1's range flag (okp_conv_set_range_flag) or NULL: raised when a hidden value of layer 1 leaves the fp16 range
};

__device__ __forceinline__ uint32_t row_key(int row) { return (uint32_t)(((row & 1) << 3) | ((row & 2) << 1) | ((row >> 2) & 1)); }

__global__ __launch_bounds__(256, 2) void okp_heads_x3_kernel(const Params p) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, q = lane >> 4;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  float* const h2 = reinterpret_cast<float*>(smem + OFF_H2);
  float* const w3s = reinterpret_cast<float*>(smem + OFF_W3);
  for (int i = tid; i < p.n_out * 33; i += 256) w3s[i] = i < p.n_out * 32 ? p.w3[i] : p.b3[i - p.n_out * 32];

  // head of this workgroup and its tiles: as in okp_heads_kernel (the three heads of a tile on one XCD where the grid allows)
  const bool xcd_order = gridDim.x % 24 == 0;
  const int gm = blockIdx.x >> 3;
  const int h = xcd_order ? gm % 3 : (int)(blockIdx.x % 3);
  const int tile0 = xcd_order ? (gm / 3) * 8 + (int)(blockIdx.x & 7) : (int)(blockIdx.x / 3);
  const int tstep = gridDim.x / 3;

  // resident weight fragments.  GEMM 1: wave w owns channels 32 w .. 32 w + 31 of the head (two 16-row blocks); GEMM 2: wave w owns
  // channel block w & 1 of the head's 32 and pixel block w >> 1; of the 12 k-steps of the block-diagonal plan only the head's four count
  u32x4 w1h[2][KS1], w1l[2][KS1], w2h[KS2], w2l[KS2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const u32x4* const f = static_cast<const u32x4*>(p.w1) + (size_t)(8 * h + 2 * w + cb) * 2 * KS1 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) { w1h[cb][ks] = f[ks * 64]; w1l[cb][ks] = f[(KS1 + ks) * 64]; }
  }
  {
    const u32x4* const f = static_cast<const u32x4*>(p.w2) + (size_t)(2 * h + (w & 1)) * 2 * (3 * KS2) * 64 + lane;
#pragma unroll
    for (int kk = 0; kk < KS2; ++kk) { w2h[kk] = f[(KS2 * h + kk) * 64]; w2l[kk] = f[(3 * KS2 + KS2 * h + kk) * 64]; }
  }
  const int c1 = F * h + 32 * w + 4 * q;                 // this lane's channels c1 + 16 cb .. + 3 of GEMM 1 (plan channel index)
  const int c2 = F2 * h + 16 * (w & 1) + 4 * q;          // ... of GEMM 2
  f32x4 b1v[2], s1v[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) { b1v[cb] = *reinterpret_cast<const f32x4*>(p.b1 + c1 + 16 * cb); s1v[cb] = *reinterpret_cast<const f32x4*>(p.s1 + c1 + 16 * cb); }
  const f32x4 b2v = *reinterpret_cast<const f32x4*>(p.b2 + c2), s2v = *reinterpret_cast<const f32x4*>(p.s2 + c2);

  // pixel-fragment reads: column = pixel 16 pb + l16, lane k-group q reads the hi chunk 8 ks + 2 q of its row and the lo chunk beside it
  const uint32_t fkey = ((uint32_t)(2 * q) ^ row_key(l16)) << 4;
  const uint32_t xfrag = (uint32_t)OFF_X + (uint32_t)l16 * 1024u + fkey;
  const uint32_t hfrag = (uint32_t)OFF_H1 + (uint32_t)(16 * (w >> 1) + l16) * 512u + fkey;

  auto issue_x = [&](int tile) {
    const long pix0 = (long)tile * TP;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 8 * w + i;                          // one 1 KiB row per instruction: position `lane` receives chunk lane ^ key(row)
      const long pix = pix0 + row;
      const uint32_t off = pix < p.n_pix ? (uint32_t)pix * (uint32_t)(p.x_ps * 4) + ((((uint32_t)lane) ^ row_key(row)) << 4) : kInvalid;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(smem + OFF_X + row * 1024), 16, (int)off, 0, 0, 0);
    }
  };
  if (tile0 < p.n_tiles) issue_x(tile0);
  for (int tile = tile0; tile < p.n_tiles; tile += tstep) {
    const long pix0 = (long)tile * TP;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this tile's x (issued behind GEMM 1 of the previous tile) has landed
    __syncthreads();                                       // ... for every wave; the previous tile's readers of h2 are done; w3 is in LDS
    {
      // ---- GEMM 1: h1[128][32 px] = relu(W1_h x + b1_h), three terms, fragments of k-step ks + 1 read under the MFMAs of ks ----
      f32x4 acc[2][2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) acc[cb][pb] = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x4 xh[2][2], xl[2][2];
      auto ldx = [&](int ks, int slot) {
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          const uint32_t a = (xfrag + (uint32_t)pb * 16384u) ^ (uint32_t)(ks * 128);
          xh[slot][pb] = *reinterpret_cast<const u32x4*>(smem + a);
          xl[slot][pb] = *reinterpret_cast<const u32x4*>(smem + (a ^ 16u));
        }
      };
      ldx(0, 0);
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < KS1) ldx(ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) acc[cb][pb] = H16<_Float16>::mfma16(w1l[cb][ks], xh[ks & 1][pb], acc[cb][pb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) acc[cb][pb] = H16<_Float16>::mfma16(w1h[cb][ks], xl[ks & 1][pb], acc[cb][pb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) acc[cb][pb] = H16<_Float16>::mfma16(w1h[cb][ks], xh[ks & 1][pb], acc[cb][pb]);
      }
      // h1 as pairs: channels 32 w + 16 cb + 4 q .. + 3 of the head = half (q & 1) of pair 4 w + 2 cb + (q >> 1) of pixel 16 pb + l16
      float range_m = 0.f;                                 // largest hidden value split for the second layer (okp_range_max): local to this block,
                                                           // so that nothing of the guard is live across the GEMMs
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = fmaxf(__builtin_fmaf(acc[cb][pb][e], s1v[cb][e], b1v[cb][e]), 0.f); range_m = okp_range_max(range_m, v[e]); }
          u32x2 hi, lo;
          okp_split4(v, hi, lo);
          const int px = 16 * pb + l16;
          const uint32_t pos = ((uint32_t)(2 * (4 * w + 2 * cb + (q >> 1))) ^ row_key(l16)) << 4;
          char* const row = smem + OFF_H1 + px * 512 + (q & 1) * 8;
          *reinterpret_cast<u32x2*>(row + pos) = hi;
          *reinterpret_cast<u32x2*>(row + (pos ^ 16u)) = lo;
        }
      okp_raise_range_flag(p.range_flag, okp_range_exceeded(range_m));
    }
    __syncthreads();
    // x is free (every wave has read its fragments): the next tile streams in under GEMM 2 and the last layer
    if (tile + tstep < p.n_tiles) issue_x(tile + tstep);
    {
      // ---- GEMM 2: h2[32][32 px] = relu(W2_h h1 + b2_h); wave w: channel block w & 1, pixel block w >> 1 ----
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KS2; ++kk) {
        const uint32_t a = hfrag ^ (uint32_t)(kk * 128);
        const u32x4 yh = *reinterpret_cast<const u32x4*>(smem + a), yl = *reinterpret_cast<const u32x4*>(smem + (a ^ 16u));
        acc = H16<_Float16>::mfma16(w2l[kk], yh, acc);
        acc = H16<_Float16>::mfma16(w2h[kk], yl, acc);
        acc = H16<_Float16>::mfma16(w2h[kk], yh, acc);
      }
      const int px = 16 * (w >> 1) + l16;
#pragma unroll
      for (int e = 0; e < 4; ++e) h2[px * H2_PITCH + 16 * (w & 1) + 4 * q + e] = fmaxf(__builtin_fmaf(acc[e], s2v[e], b2v[e]), 0.f);
    }
    __syncthreads();
    {
      // ---- last layer: out_o = act(w3_o . h2 + b3_o) for the outputs of this head; 32 lanes along the pixels, eight output slots ----
      const int px = tid & 31, slot = tid >> 5;
      const long pix = pix0 + px;
      int seen = 0;
      for (int o = 0; o < p.n_out; ++o) {
        if (p.head_of[o] != h) continue;
        if ((seen++ & 7) != slot) continue;
        float acc = w3s[p.n_out * 32 + o];
#pragma unroll
        for (int c = 0; c < F2; ++c) acc = fmaf(h2[px * H2_PITCH + c], w3s[o * 32 + c], acc);
        if (p.act[o] == OKP_ACT_SIGMOID) acc = 1.f / (1.f + expf(-acc));
        else if (p.act[o] == OKP_ACT_RELU) acc = fmaxf(acc, 0.f);
        if (pix < p.n_pix) {
          const long n = pix / p.HW;
          p.out_ptr[o][(size_t)n * p.out_n_stride[o] + (size_t)(pix - n * p.HW)] = acc;
        }
      }
    }
  }
}

}  // namespace x3

}  // namespace

// split-product plans: x in pair format (include/okp.h)
static int heads_forward_x3(const okp_conv* l1, const okp_conv* l2, const okp_head_out_args* a, const okp_tensor* x, void* stream) {
  if (l2->dtype != OKP_F32X3 || l1->n_taps != 1 || l2->n_taps != 1 || l1->cin[0] != CIN || l1->cout != 3 * F || l2->cin[0] != 3 * F || l2->cout != 3 * F2 ||
      l1->n_single_slices || l2->n_single_slices || !l1->fragT_dev || !l2->fragT_dev || !l1->oscale_dev || !l2->oscale_dev) {
    okp_set_error("okp_heads_forward: expects split-product 1x1 plans 256 -> 384 and 384 -> 96 (three heads of 128 features, three terms)"); return OKP_EINVAL;
  }
  if (a->n_out < 1 || a->n_out > OKP_HEAD_MAX_OUT) { okp_set_error("okp_heads_forward: n_out %d out of range", a->n_out); return OKP_EINVAL; }
  if (a->n < 1 || a->h < 1 || a->w < 1) return OKP_OK;
  if (x->pix_stride < CIN || x->pix_stride % 8 || ((uintptr_t)x->data) % 32 || x->bytes <= 0 || x->bytes >= 0x7FFF0000ll) {
    okp_set_error("okp_heads_forward: x must be a 32-byte aligned (one pair group) pair-format view of >= 256 channels in whole 8-channel groups, < 2 GiB"); return OKP_EINVAL;
  }
  x3::Params p;
  std::memset(&p, 0, sizeof(p));
  p.x = x->data; p.x_bytes = (uint32_t)x->bytes; p.x_ps = x->pix_stride;
  p.HW = a->h * a->w; p.n_pix = (long)a->n * p.HW;
  p.w1 = l1->fragT_dev; p.b1 = l1->bias_dev; p.s1 = l1->oscale_dev; p.w2 = l2->fragT_dev; p.b2 = l2->bias_dev; p.s2 = l2->oscale_dev;
  p.w3 = a->w_dev; p.b3 = a->bias_dev; p.n_out = a->n_out; p.range_flag = l1->range_flag;
  for (int o = 0; o < a->n_out; ++o) {
    if (a->in_c_off[o] % F2 || a->in_c_off[o] < 0 || a->in_c_off[o] >= 3 * F2 || !a->out_ptr[o]) { okp_set_error("okp_heads_forward: output %d: bad channel offset or null pointer", o); return OKP_EINVAL; }
    p.head_of[o] = a->in_c_off[o] / F2; p.act[o] = a->act[o]; p.out_ptr[o] = a->out_ptr[o]; p.out_n_stride[o] = a->out_n_stride[o];
  }
  const long tiles = (p.n_pix + x3::TP - 1) / x3::TP;
  if (tiles >= 0x7FFFFFFFl) { okp_set_error("okp_heads_forward: too many pixels"); return OKP_EINVAL; }
  p.n_tiles = (int)tiles;
  const long groups = tiles < 168 ? tiles : 168;            // 3 x 168 = 504 = 24 x 21 workgroups: two per CU, whole head triples per XCD
  hipLaunchKernelGGL(x3::okp_heads_x3_kernel, dim3((unsigned)(3 * groups)), dim3(256), 0, (hipStream_t)stream, p);
  return okp_check_hip(hipGetLastError(), "okp_heads_x3 launch");
}

extern "C" int okp_heads_forward(const okp_conv* l1, const okp_conv* l2, const okp_head_out_args* a, const okp_tensor* x, void* stream) {
  if (!l1 || !l2 || !a || !x || !x->data || !a->w_dev || !a->bias_dev) { okp_set_error("okp_heads_forward: null argument"); return OKP_EINVAL; }
  if (l1->dtype == OKP_F32X3) return heads_forward_x3(l1, l2, a, x, stream);
  if (!okp_is16(l1->dtype) || l2->dtype != l1->dtype || l1->n_taps != 1 || l2->n_taps != 1 || l1->cin[0] != CIN || l1->cout != 3 * F ||
      l2->cin[0] != 3 * F || l2->cout != 3 * F2) {
    okp_set_error("okp_heads_forward: expects bf16 / fp16 1x1 plans 256 -> 384 and 384 -> 96 (three heads of 128 features)"); return OKP_EINVAL;
  }
  if (a->n_out < 1 || a->n_out > OKP_HEAD_MAX_OUT) { okp_set_error("okp_heads_forward: n_out %d out of range", a->n_out); return OKP_EINVAL; }
  if (a->n < 1 || a->h < 1 || a->w < 1) return OKP_OK;
  if (x->pix_stride < CIN || x->pix_stride % 8 || ((uintptr_t)x->data) % 16 || x->bytes <= 0 || x->bytes >= 0x7FFF0000ll) {
    okp_set_error("okp_heads_forward: x must be a 16-byte aligned view of >= 256 channels, < 2 GiB"); return OKP_EINVAL;
  }
  if (int e = okp_ensure_frags(l1, (hipStream_t)stream)) return e;
  if (int e = okp_ensure_frags(l2, (hipStream_t)stream)) return e;
  HeadsParams p;
  std::memset(&p, 0, sizeof(p));
  p.x = x->data; p.x_bytes = (uint32_t)x->bytes; p.x_ps = x->pix_stride;
  p.HW = a->h * a->w; p.n_pix = (long)a->n * p.HW;
  p.w1 = l1->frag_dev; p.b1 = l1->bias_dev; p.w2 = l2->frag_dev; p.b2 = l2->bias_dev;
  p.w3 = a->w_dev; p.b3 = a->bias_dev; p.n_out = a->n_out;
  for (int o = 0; o < a->n_out; ++o) {
    if (a->in_c_off[o] % F2 || a->in_c_off[o] < 0 || a->in_c_off[o] >= 3 * F2 || !a->out_ptr[o]) { okp_set_error("okp_heads_forward: output %d: bad channel offset or null pointer", o); return OKP_EINVAL; }
    p.head_of[o] = a->in_c_off[o] / F2; p.act[o] = a->act[o]; p.out_ptr[o] = a->out_ptr[o]; p.out_n_stride[o] = a->out_n_stride[o];
  }
  const long tiles = (p.n_pix + TP - 1) / TP;
  if (tiles >= 0x7FFFFFFFl) { okp_set_error("okp_heads_forward: too many pixels"); return OKP_EINVAL; }
  p.n_tiles = (int)tiles;
  const long groups = tiles < 168 ? tiles : 168;            // 3 x 168 = 504 = 24 x 21 workgroups: two per CU, whole head triples per XCD
  if (l1->dtype == OKP_BF16) hipLaunchKernelGGL(okp_heads_kernel<__bf16>, dim3((unsigned)(3 * groups)), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(okp_heads_kernel<_Float16>, dim3((unsigned)(3 * groups)), dim3(256), 0, (hipStream_t)stream, p);
  return okp_check_hip(hipGetLastError(), "okp_heads launch");
}

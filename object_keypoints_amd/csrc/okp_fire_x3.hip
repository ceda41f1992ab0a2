// One-launch fire module of the split-product configuration (OKP_F32X3 plans: fp32 tensors, every product of the two 1x1
// convolutions as x_hi w_hi + x_lo w_hi + x_hi w_lo on the fp16 matrix pipe), 256 -> 128 -> 256 channels, stride 1, with skip -
// the fourteen modules of the two high-resolution hourglass levels (64 x 64 and 32 x 32).
//
//     s   = W1 x + b1                                   squeeze 1x1 (+bn1, no ReLU)      256 -> 128
//     y_a = relu(Wa s + ba + x[:, :128])                expand 1x1 (+bn2 half, skip)     128 -> 128
//     y_b = relu(dw3x3(s) * wd + bd + x[:, 128:])       depth-wise 3x3 (+bn2 half, skip) 128 -> 128
// (reference: fire_module, corner_net_lite/core/models/CornerNet_Squeeze.py:10-30)
//
// As two launches (squeeze; expand + depth-wise tail of okp_igemm_kernel<F32S>) the module moves 1.07 GB at 64 x 64, N = 64 - x twice
// (GEMM input + skip), the squeeze tensor written and re-read - in 283 us.  Here, as in okp_fire2.hip for the 16-bit types, x is
// read once (+ halo) and the output written once (537 MB); the squeeze tile never leaves LDS.
//
// Shape (one 8-wave workgroup per CU, persistent grid, 134 KB of LDS):
//  * a workgroup owns an IH x IW rectangle of output pixels; the squeeze tile is that rectangle plus a one-pixel halo (SH x SW <= 128
//    pixels; halo pixels outside the frame are zero in s: the reference zero-pads s);
//  * both GEMMs run on 16x16x32 fp16 MFMAs with CHANNELS as rows (A = weights, hi and lo fragments resident in registers for the
//    whole kernel for the squeeze GEMM: 64 VGPRs; the expand GEMM's are re-fetched per tile) and PIXELS as columns; wave w owns channels 16 w .. 16 w + 15 of both GEMMs, so a lane holds FOUR ADJACENT
//    channels of a pixel: 16 bytes of fp32 - one store per 16 pixels, no transposition;
//  * x streams through a 4-stage LDS ring of 128-byte K-chunks (32 fp32 channels = one MFMA k-step) filled by LDS-DMA, and every stage
//    is split ONCE, in place, into fp16 [hi | lo] pairs by all threads (one 32-byte item each) during the step before it is multiplied -
//    eight waves read the same pixel fragments, so splitting in registers would repeat the conversion eight times;
//  * the squeeze tile is written twice: as fp32 (what the depth-wise branch reads) and as [hi | lo] pairs (the expand GEMM's pixel
//    operand) over the x ring, which is free by then; the next tile's first ring stages are requested when the expand phase is done;
//  * the skip values are re-read from global memory (the tile was streamed a few microseconds earlier: L2 / Infinity Cache hits) and
//    added in fp32, exactly as the two-launch path adds them.
#include <cstdio>
#include <cstring>
#include <cstdlib>

#include "okp_igemm_kernel.h"

namespace {

constexpr int CIN = 256, MID = 128;
constexpr int SP = 128;                          // squeeze-tile rows in LDS
constexpr int PBI = 6;                           // interior pixel blocks of 16 (IP <= 96)
constexpr int NST = 4;                           // x ring stages
constexpr int MAXIH = 5;                         // interior rows per tile (5 x 16 tiles on every map of the network; a sixth row cost the depth-wise phase 9 registers)
#ifndef OKP_FX3_DWB
#define OKP_FX3_DWB 4
#endif
constexpr int KS1 = CIN / 32, KS2 = MID / 32;    // k-steps of the squeeze / expand GEMM
constexpr int NW = 8, NT = 64 * NW;
constexpr int XST = SP * 128;                    // bytes per ring stage (128 pixels x 32 fp32)
constexpr int OFF_S32 = 0;                       // [128][128 x 4 B] squeeze tile, fp32, 16-byte chunks XOR-swizzled by the row
constexpr int OFF_X = OFF_S32 + SP * MID * 4;    // x ring; after phase 1: the squeeze tile as [hi | lo] fp16 pairs, [128][16 k-groups x 32 B]
constexpr int OFF_WD = OFF_X + NST * XST;        // [9][128] fp32 depth-wise weights, then [128] bias
constexpr int OFF_TAB = OFF_WD + 10 * MID * 4;   // interior pixel ip -> byte offset relative to the tile's first pixel: [96] in x, [96] in out
constexpr int OFF_MASK = OFF_TAB + 2 * 96 * 4;   // 4 x u32 validity bits of the squeeze pixels
constexpr int OFF_BS = OFF_MASK + 16;            // [4][128] fp32: squeeze bias, squeeze output scale, expand bias, expand output scale
constexpr int LDS_TOTAL = OFF_BS + 4 * MID * 4;
static_assert(NST * XST == SP * MID * 4, "the [hi | lo] copy of the squeeze tile takes exactly the x ring");
static_assert(LDS_TOTAL <= 160 * 1024, "LDS");

// swizzle key of ring row r (128-byte rows, 16-row fragments whose lane k-group q reads chunk 2q / 2q + 1): by (r >> 1) & 7, conflict-free
// for the ds_read_b128 lane groups of gfx950 (MI355X_MICROARCH.md, LDS; searched against them)
__device__ __forceinline__ uint32_t ring_key(int row) { return (0x54541010u >> (4 * ((row >> 1) & 7))) & 7u; }

#ifdef OKP_FIRE_STAMPS
// Debug build (OKP_EXTRA_CFLAGS=-DOKP_FIRE_STAMPS, printed with OKP_FIRE_STAMPS_PRINT=1): shader-clock stamps of every wave at the phase
// boundaries of its workgroup's SECOND tile (steady state), workgroups 0..15: [wg][wave][8] u32
#define FX3_STAMP(i) do { if (second && lane == 0 && blockIdx.x < 16) { uint64_t t_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); p.dbg[(blockIdx.x * NW + w) * 8 + (i)] = (uint32_t)t_; } } while (0)
#else
#define FX3_STAMP(i) do {} while (0)
#endif

// STR = 2 (round 6, last): the stride-2 module 256 -> 128 -> 256 WITHOUT skip (64 x 64 -> 32 x 32: until now a squeeze launch plus the fused tail -
// the fp32 squeeze tensor, 134 MB per 64 frames, written and read back).  The squeeze tile is the input pixels under a tile of OH2 x 8 output
// pixels (7 x 17); the 1x1 branch samples it at (2 iy + 1, 2 ix + 1), the depth-wise branch walks every squeeze row, row sr feeding output row
// (sr - dy) / 2 where that is whole.
constexpr int OH2 = 3;                           // stride 2: output rows per tile
template <int STR>
__global__ __launch_bounds__(NT) void okp_fire_x3_kernel(const OkpFire2Params p) {
  static_assert(STR == 1 || STR == 2, "stride");
  constexpr int OH = STR == 1 ? MAXIH : OH2;     // output rows a thread of the depth-wise phase holds (loads, stores per column iteration)
  constexpr int NSR = STR * (OH - 1) + 3;        // squeeze rows it walks
  __shared__ __attribute__((aligned(16))) char smem[LDS_TOTAL];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, q = lane >> 4;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);

  // ---- once per workgroup: resident weight fragments (hi, lo), biases / output scales of this lane's four channels, depth-wise constants ----
  const int chq = 16 * w + 4 * q;                             // this lane's channels chq .. chq + 3 in both GEMMs
  u32x4 w1h[KS1], w1l[KS1];                                   // [wave][hi | lo][k-step][lane][16 B] (okp_conv_create, split-product 1x1 plans)
  {
    const u32x4* const w1 = static_cast<const u32x4*>(p.w1) + (size_t)w * 2 * KS1 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) { w1h[ks] = w1[ks * 64]; w1l[ks] = w1[(KS1 + ks) * 64]; }
  }
  const u32x4* const wa_lane = static_cast<const u32x4*>(p.wa) + (size_t)w * 2 * KS2 * 64 + lane;      // expand weights: re-fetched per tile (L2)
  // ring fragment read: row 16 pb + l16, this lane's hi chunk (logical 2q) at position 2q ^ key(row); key depends on l16 only
  const uint32_t xfrag_off = (uint32_t)l16 * 128u + (((uint32_t)(2 * q) ^ ring_key(l16)) << 4);
  const uint32_t xfrag_lo = xfrag_off ^ 16u;                  // its lo chunk: the other one of the aligned pair
  // in-place split of a ring stage: this thread's 32-byte item = pair tid & 3 of row tid >> 2
  const uint32_t split_off = (uint32_t)tid * 32u;
  const bool split_odd = ring_key(tid >> 2) & 1u;

  auto tile_origin = [&](int slot, int& n, int& y0, int& x0) {
    // XCD-aware order (as in okp_fire2_kernel): each XCD walks a contiguous range of tiles, neighbours share halo rows in its L2
    const int xq = p.n_tiles >> 3, xr = p.n_tiles & 7, xcd = slot & 7;
    const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (slot >> 3);
    n = fastdiv(tile, p.div_tiles_frame);
    const int trem = tile - n * p.tiles_y * p.tiles_x;
    const int ty = fastdiv(trem, p.div_tiles_x);
    y0 = ty * p.IH;
    x0 = (trem - ty * p.tiles_x) * p.IW;
  };
  // LDS-DMA geometry: a ring stage is 16 instructions of 8 rows; wave w issues blocks w and w + 8.  The lane FETCHES the 16-byte
  // chunk the read-side swizzle expects at its position.  (Worked out per tile from an opaque copy of the lane id: kept in registers
  // across the phases these values made the kernel spill, and a spill's reload waits for every store in flight.)
  uint32_t d_off[2];
  auto tile_setup = [&](int tile) {                          // DMA source offsets + validity bits of `tile`
    int n, y0, x0;
    tile_origin(tile, n, y0, x0);
    int lt = lane, tt = tid;
    asm volatile("" : "+v"(lt), "+v"(tt));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 8 * (w + NW * i) + (lt >> 3);
      const int sy = fastdiv(row, p.div_sw), sx = row - sy * p.SW;
      const int y = STR * y0 - 1 + sy, x = STR * x0 - 1 + sx;
      const bool ok = row < p.SH * p.SW && y >= 0 && y < p.H && x >= 0 && x < p.W;
      d_off[i] = ok ? (uint32_t)(((long)n * p.H + y) * p.W + x) * (uint32_t)(p.x_ps * 4) + (((uint32_t)lt & 7u) ^ ring_key(row)) * 16u : kInvalidOff;
    }
    if (tid < SP) {
      const int m_sy = fastdiv(tt, p.div_sw), m_sx = tt - m_sy * p.SW;
      const int y = STR * y0 - 1 + m_sy, x = STR * x0 - 1 + m_sx;
      const unsigned long long m = __ballot(tt < p.SH * p.SW && y >= 0 && y < p.H && x >= 0 && x < p.W);
      if (lane == 0) {
        uint32_t* mk = reinterpret_cast<uint32_t*>(smem + OFF_MASK);
        mk[2 * w] = (uint32_t)m;
        mk[2 * w + 1] = (uint32_t)(m >> 32);
      }
    }
  };
  auto issue_x = [&](int ks, int stage) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(smem + OFF_X + stage * XST + (w + NW * i) * 1024), 16,
                                               (int)(d_off[i] == kInvalidOff ? kInvalidOff : d_off[i] + (uint32_t)ks * 128u), 0, 0, 0);
  };
  auto split_stage = [&](int stage) {
    char* const a = smem + OFF_X + stage * XST + split_off;
    const u32x4 r0 = *reinterpret_cast<const u32x4*>(a), r1 = *reinterpret_cast<const u32x4*>(a + 16);
    u32x4 hi, lo;
    okp_split8(split_odd ? r1 : r0, split_odd ? r0 : r1, hi, lo);
    *reinterpret_cast<u32x4*>(a) = split_odd ? lo : hi;
    *reinterpret_cast<u32x4*>(a + 16) = split_odd ? hi : lo;
  };

  int tile = blockIdx.x;
  if (tile >= p.n_tiles) return;
  int stores_behind_ring = 0;                                // stores this wave issued behind its ring requests of the tile about to start
  // the first tile's requests go out BEFORE the per-workgroup constants are fetched (both are cold reads right behind a launch), and the
  // constants' loads are all in flight before the first is used (as a loop hipcc waited for each load by itself): okp_fire2.hip, round 6
  tile_setup(tile);
#pragma unroll
  for (int ks = 0; ks < NST - 1; ++ks) issue_x(ks, ks);
  {
    constexpr int NC = (10 * MID + NT - 1) / NT;
    float cv[NC], bsv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const int i = tid + k * NT;
      cv[k] = i < 10 * MID ? *(i < 9 * MID ? p.wd + i : p.bd + (i - 9 * MID)) : 0.f;
    }
    if (tid < MID) { bsv[0] = p.b1[tid]; bsv[1] = p.s1[tid]; bsv[2] = p.ba[tid]; bsv[3] = p.sa[tid]; }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const int i = tid + k * NT;
      if (i < 10 * MID) reinterpret_cast<float*>(smem + OFF_WD)[i] = cv[k];
    }
    if (tid < MID) {
      float* const bs = reinterpret_cast<float*>(smem + OFF_BS);
      bs[tid] = bsv[0]; bs[MID + tid] = bsv[1]; bs[2 * MID + tid] = bsv[2]; bs[3 * MID + tid] = bsv[3];
    }
  }
  if (tid < 16 * PBI) {
    const int iy = fastdiv(tid, p.div_iw), ix = tid - iy * p.IW;
    reinterpret_cast<uint32_t*>(smem + OFF_TAB)[tid] = tid < p.IP ? (uint32_t)(iy * p.W + ix) * (uint32_t)(p.x_ps * 4) : kInvalidOff;
    reinterpret_cast<uint32_t*>(smem + OFF_TAB)[96 + tid] = tid < p.IP ? (uint32_t)(iy * p.Wo + ix) * (uint32_t)(p.out_ps * 4) : kInvalidOff;
  }
  __syncthreads();                                           // depth-wise constants and tables are in LDS

  for (; tile < p.n_tiles; tile += gridDim.x) {
    int n, y0, x0;
    tile_origin(tile, n, y0, x0);
#ifdef OKP_FIRE_STAMPS
    const bool second = tile == (int)(blockIdx.x + gridDim.x);
#endif
    FX3_STAMP(0);
    int qt = q, tidt = tid, l16t = l16;                      // opaque copies: keeps tile-invariant address arithmetic out of registers
    asm volatile("" : "+v"(qt), "+v"(tidt), "+v"(l16t));

    // ---- phase 1: s = W1 x + b1 on the halo'd tile ------------------------------------------------------------------
    // Ring stages 0 .. 2 of this tile were requested during the previous tile's depth-wise phase (or above).  Everything older -
    // that tile's stores included: loads, stores and LDS-DMA share the counter - is waited for once here.
    f32x4 acc[SP / 16];
#pragma unroll
    for (int pb = 0; pb < SP / 16; ++pb) acc[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (the depth-wise phase of the previous tile issued MAXIH loads - consumed - and MAXIH stores per column iteration behind the ring
    //  requests: with one iteration, the usual case, those stores may stay in flight)
    if (stores_behind_ring == OH) {
      if constexpr (OH == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    static_assert(OH == 5 || OH == 3, "the counted waits above name OH");
    __builtin_amdgcn_s_barrier();                            // stages 0 .. 2 have landed (every wave's part)
    FX3_STAMP(1);
    split_stage(0);
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      // step ks multiplies stage ks (split during step ks - 1: visible behind this barrier), splits stage ks + 1 (its LDS-DMA, issued
      // two steps ago, has landed: vmcnt leaves only the two requests of stage ks + 2 in flight) and requests stage ks + 3 into the
      // slot stage ks - 1 was read from (every wave's fragment reads have returned: the explicit lgkmcnt(0) in front of the barrier)
      if (ks + 2 < KS1 && ks > 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (ks > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (ks + NST - 1 < KS1) issue_x(ks + NST - 1, (ks + NST - 1) % NST);
      // the split of stage ks + 1 sits in the middle of the step: its two reads are requested with the first fragments, its conversions
      // and writes follow the MFMAs of the first four blocks (placed in front of them the wave waited for its own split before multiplying)
      char* const sa_ = smem + OFF_X + ((ks + 1) % NST) * XST + split_off;
      u32x4 sr0 = {}, sr1 = {};
      if (ks + 1 < KS1) { sr0 = *reinterpret_cast<const u32x4*>(sa_); sr1 = *reinterpret_cast<const u32x4*>(sa_ + 16); }
      // fragment pipeline: the pixel fragments of block pb + 1 are requested before the three MFMAs of block pb, so that a wave's LDS reads
      // run under its own (and its SIMD partner's) MFMAs instead of in one burst behind the barrier that all eight waves issue together
      const char* const st = smem + OFF_X + (ks % NST) * XST;
      u32x4 xh[3], xl[3];                                     // (two blocks ahead: an LDS read takes 200-300 clocks with eight waves on the port, a block's MFMAs 100)
      xh[0] = *reinterpret_cast<const u32x4*>(st + xfrag_off);
      xl[0] = *reinterpret_cast<const u32x4*>(st + xfrag_lo);
      xh[1] = *reinterpret_cast<const u32x4*>(st + 2048 + xfrag_off);
      xl[1] = *reinterpret_cast<const u32x4*>(st + 2048 + xfrag_lo);
#define OKP_FX3_BLOCK(PB)                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      if (PB + 2 < SP / 16) {                                                                                 \
        xh[(PB + 2) % 3] = *reinterpret_cast<const u32x4*>(st + (PB + 2) * 2048 + xfrag_off);                 \
        xl[(PB + 2) % 3] = *reinterpret_cast<const u32x4*>(st + (PB + 2) * 2048 + xfrag_lo);                  \
      }                                                                                                       \
      if (PB == 3 && ks + 1 < KS1) {                                                                          \
        u32x4 hi, lo;                                                                                         \
        okp_split8(split_odd ? sr1 : sr0, split_odd ? sr0 : sr1, hi, lo);                                     \
        *reinterpret_cast<u32x4*>(sa_) = split_odd ? lo : hi;                                                 \
        *reinterpret_cast<u32x4*>(sa_ + 16) = split_odd ? hi : lo;                                            \
      }                                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      acc[PB] = H16<_Float16>::mfma16(w1l[ks], xh[PB % 3], acc[PB]);                                          \
      acc[PB] = H16<_Float16>::mfma16(w1h[ks], xl[PB % 3], acc[PB]);                                          \
      acc[PB] = H16<_Float16>::mfma16(w1h[ks], xh[PB % 3], acc[PB]);
      OKP_FX3_BLOCK(0) OKP_FX3_BLOCK(1) OKP_FX3_BLOCK(2) OKP_FX3_BLOCK(3)
      OKP_FX3_BLOCK(4) OKP_FX3_BLOCK(5) OKP_FX3_BLOCK(6) OKP_FX3_BLOCK(7)
#undef OKP_FX3_BLOCK
      __builtin_amdgcn_sched_barrier(0);
    }
    FX3_STAMP(2);
    // expand weights for this tile (dead after phase 2a): requested now, consumed behind the barrier
    u32x4 wah[KS2], wal[KS2];
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) { wah[ks] = wa_lane[ks * 64]; wal[ks] = wa_lane[(KS2 + ks) * 64]; }
    // ---- s -> LDS, as fp32 and as [hi | lo] pairs over the x ring (zero outside the frame: the reference zero-pads the squeeze output) ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // every wave's reads of the last ring stage have returned
    {
      const uint32_t* mk = reinterpret_cast<const uint32_t*>(smem + OFF_MASK);
      uint32_t mq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) mq[i] = mk[i] >> l16t;
      const int g = 2 * w + (qt >> 1);                       // this lane's k-group of 8 squeeze channels, and which half of it
      const uint32_t hl_off = (uint32_t)(qt & 1) * 8u;
      const f32x4 b1v = *reinterpret_cast<const f32x4*>(smem + OFF_BS + chq * 4), s1v = *reinterpret_cast<const f32x4*>(smem + OFF_BS + (MID + chq) * 4);
      float range_m = 0.f;                                   // largest |squeeze value| this lane splits (okp_range_max; local to the block)
#pragma unroll
      for (int pb = 0; pb < SP / 16; ++pb) {
        const bool ok = (mq[pb >> 1] >> (16 * (pb & 1))) & 1u;
        const int row = 16 * pb + l16t;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = ok ? __builtin_fmaf(acc[pb][e], s1v[e], b1v[e]) : 0.f; range_m = okp_range_max(range_m, v[e]); }
        *reinterpret_cast<f32x4*>(smem + OFF_S32 + row * 512 + ((((uint32_t)(4 * w + qt)) ^ (uint32_t)(row & 15)) << 4)) = v;
        u32x2 hi, lo;
        okp_split4(v, hi, lo);
        char* const hp = smem + OFF_X + row * 512 + hl_off;
        *reinterpret_cast<u32x2*>(hp + ((((uint32_t)(2 * g)) ^ (uint32_t)(row & 15)) << 4)) = hi;
        *reinterpret_cast<u32x2*>(hp + ((((uint32_t)(2 * g + 1)) ^ (uint32_t)(row & 15)) << 4)) = lo;
      }
      okp_raise_range_flag(p.range_flag, okp_range_exceeded(range_m));
    }
    __syncthreads();
    FX3_STAMP(3);

    // ---- phase 2a: y_a = relu(Wa s + ba + x[:, :128]) on the interior pixels, 16 bytes per lane straight to HBM --------------------
    // One block of 16 pixels per iteration (not unrolled: 24 blocks of reads + MFMAs in one basic block made hipcc hoist every LDS read
    // above the first MFMA and spill 250 registers); the skip values and offsets of block pb + 1 are requested while block pb is multiplied.
    {
      const bool full = y0 + p.IH <= p.Ho && x0 + p.IW <= p.Wo;
      const uint32_t pix0 = (uint32_t)(((long)n * p.Ho + y0) * p.Wo + x0);
      const uint32_t xb = pix0 * (uint32_t)(p.x_ps * 4) + (uint32_t)chq * 4u, ob = pix0 * (uint32_t)(p.out_ps * 4) + (uint32_t)chq * 4u;
      const int n_pb = (p.IP + 15) >> 4;
      const f32x4 bav = *reinterpret_cast<const f32x4*>(smem + OFF_BS + (2 * MID + chq) * 4), sav = *reinterpret_cast<const f32x4*>(smem + OFF_BS + (3 * MID + chq) * 4);
      // block pb: output offset, squeeze-tile row (byte offset | swizzle key) and skip values of this lane's pixel
      auto block_setup = [&](int pb, uint32_t& o_off, uint32_t& a_row, u32x4& r_raw) {
        int ip = 16 * pb + l16t;
        const bool in = ip < p.IP;
        if (!in) ip = 0;
        const int iy = fastdiv(ip, p.div_iw), ix = ip - iy * p.IW;
        const int sp = (STR * iy + 1) * p.SW + STR * ix + 1;     // the squeeze pixel the 1x1 branch samples
        a_row = (uint32_t)sp * 512u + (uint32_t)(sp & 15);
        uint32_t xr = *reinterpret_cast<const uint32_t*>(smem + OFF_TAB + ip * 4);
        if (STR == 2 || !p.skip) xr = kInvalidOff;               // no skip connection: the load below returns zeros
        uint32_t orr = *reinterpret_cast<const uint32_t*>(smem + OFF_TAB + (96 + ip) * 4);
        if (!in || (!full && (y0 + iy >= p.Ho || x0 + ix >= p.Wo))) { xr = kInvalidOff; orr = kInvalidOff; }
        o_off = orr == kInvalidOff ? kInvalidOff : ob + orr;
        r_raw = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(xr == kInvalidOff ? kInvalidOff : xb + xr), 0, 0);
      };
      uint32_t o_nx, a_nx;
      u32x4 r_nx;
      block_setup(0, o_nx, a_nx, r_nx);
      auto s_frag = [&](uint32_t a_row, int ks, int lo) {
        return *reinterpret_cast<const u32x4*>(smem + OFF_X + (a_row & ~15u) + ((((uint32_t)(2 * (4 * ks + qt) + lo)) ^ (a_row & 15u)) << 4));
      };
      // fragment pipeline two k-steps deep across the blocks: four rotating sets (set = k-step), the fragments of (pb, ks + 2) - or of
      // (pb + 1, ks - 2) - are requested before the three MFMAs of (pb, ks)
      u32x4 sh[KS2], sl[KS2];
      sh[0] = s_frag(a_nx, 0, 0); sl[0] = s_frag(a_nx, 0, 1);
      sh[1] = s_frag(a_nx, 1, 0); sl[1] = s_frag(a_nx, 1, 1);
      float range_m2 = 0.f;
#pragma unroll 1
      for (int pb = 0; pb < n_pb; ++pb) {
        const uint32_t o_off = o_nx, a_row = a_nx;
        const u32x4 r_raw = r_nx;
        if (pb + 1 < n_pb) block_setup(pb + 1, o_nx, a_nx, r_nx);
        f32x4 ac2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
          __builtin_amdgcn_sched_barrier(0);
          if (ks + 2 < KS2) { sh[ks + 2] = s_frag(a_row, ks + 2, 0); sl[ks + 2] = s_frag(a_row, ks + 2, 1); }
          else { sh[ks + 2 - KS2] = s_frag(a_nx, ks + 2 - KS2, 0); sl[ks + 2 - KS2] = s_frag(a_nx, ks + 2 - KS2, 1); }   // (behind the last block: read, not used)
          __builtin_amdgcn_sched_barrier(0);
          ac2 = H16<_Float16>::mfma16(wal[ks], sh[ks], ac2);
          ac2 = H16<_Float16>::mfma16(wah[ks], sl[ks], ac2);
          ac2 = H16<_Float16>::mfma16(wah[ks], sh[ks], ac2);
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 r = __builtin_bit_cast(f32x4, r_raw);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = fmaxf(__builtin_fmaf(ac2[e], sav[e], bav[e]) + r[e], 0.f); range_m2 = okp_range_max(range_m2, v[e]); }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_o, (int)o_off, 0, 0);
      }
      okp_raise_range_flag(p.range_flag, okp_range_exceeded(range_m2));
    }
    FX3_STAMP(4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                         // every wave's reads of the [hi | lo] tile have returned: the ring is free
    {
      const int next = tile + gridDim.x;
      if (next < p.n_tiles) {
        tile_setup(next);                                    // (the validity bits are next read behind >= 9 barriers)
#pragma unroll
        for (int ks = 0; ks < NST - 1; ++ks) issue_x(ks, ks);
      }
    }

    FX3_STAMP(5);
    // ---- phase 2b: y_b = relu(dw3x3(s) + bd + x[:, 128:]) from the fp32 squeeze tile -------------------------------------
    // thread = (4-channel group cg, column slot): it walks DOWN its column, each squeeze row feeding the three output rows that see it
    // as tap row 2, 1, 0; the nine taps' weights come from LDS per column position.
    {
      const int cg = tidt & 31;
      const float* const wl = reinterpret_cast<const float*>(smem + OFF_WD) + cg * 4;     // [tap][128] fp32, bias at tap 9
      const f32x4 breg = *reinterpret_cast<const f32x4*>(wl + 9 * MID);
      stores_behind_ring = 0;
      float range_m3 = 0.f;
      for (int ix = tidt >> 5; ix < ((p.IW + 15) & ~15); ix += 16) {
        stores_behind_ring += OH;
        const int ox = x0 + ix;
        const bool col_ok = ix < p.IW && ox < p.Wo;
        const uint32_t pix = (uint32_t)(((long)n * p.Ho + y0) * p.Wo + ox);
        uint32_t xo = pix * (uint32_t)(p.x_ps * 4) + (uint32_t)(MID + cg * 4) * 4u, oof = pix * (uint32_t)(p.out_ps * 4) + (uint32_t)(MID + cg * 4) * 4u;
        u32x4 rr[OH];
        uint32_t oo[OH];
#pragma unroll
        for (int iy = 0; iy < OH; ++iy) {
          const bool ok = col_ok && iy < p.IH && y0 + iy < p.Ho;
          oo[iy] = ok ? oof : kInvalidOff;
          rr[iy] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(ok && STR == 1 && p.skip ? xo : kInvalidOff), 0, 0);
          xo += (uint32_t)(p.W * p.x_ps * 4);
          oof += (uint32_t)(p.Wo * p.out_ps * 4);
        }
        const int ixc = ix < p.IW ? ix : 0;
        f32x4 v[OH];
#pragma unroll
        for (int iy = 0; iy < OH; ++iy) v[iy] = breg;
#pragma unroll 1
        for (int dx = 0; dx < 3; ++dx) {
          f32x4 wt[3];
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) wt[dy] = *reinterpret_cast<const f32x4*>(wl + (dy * 3 + dx) * MID);
          // ALL squeeze rows of the column position in flight before the first FMA.  With a tile-height condition per row hipcc made every read a
          // basic block of its own - ds_read -> s_waitcnt -> 12 FMAs, 21 LDS latencies one behind the other per column, most of this phase's
          // 9-10 k clocks.  Rows beyond the tile are read too (clamped into the tile): they feed only accumulator rows that are never stored
          // - and that the range maximum below leaves out.
          constexpr int BT = OKP_FX3_DWB;                      // rows per batch (registers: 4 per row)
#pragma unroll
          for (int s0 = 0; s0 < NSR; s0 += BT) {
            f32x4 sv[BT];
#pragma unroll
            for (int u = 0; u < BT; ++u) {                     // squeeze row sr is tap row dy of output row (sr - dy) / STR
              if (s0 + u < NSR) {
                const int sp = min((s0 + u) * p.SW + STR * ixc + dx, SP - 1);
                sv[u] = *reinterpret_cast<const f32x4*>(smem + OFF_S32 + sp * 512 + (((uint32_t)cg ^ (uint32_t)(sp & 15)) << 4));
              }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < BT; ++u) {
              const int sr = s0 + u;
              if (sr < NSR) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                  const int iy = (sr - dy) / STR;
                  if (sr - dy >= 0 && (sr - dy) % STR == 0 && iy < OH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[iy][e] = __builtin_fmaf(sv[u][e], wt[dy][e], v[iy][e]);
                  }
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int iy = 0; iy < OH; ++iy) {
          const f32x4 r = __builtin_bit_cast(f32x4, rr[iy]);
          const bool stored = oo[iy] != kInvalidOff;
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) { o[e] = fmaxf(v[iy][e] + r[e], 0.f); range_m3 = okp_range_max(range_m3, stored ? o[e] : 0.f); }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_o, (int)oo[iy], 0, 0);
        }
      }
      okp_raise_range_flag(p.range_flag, okp_range_exceeded(range_m3));
    }
    FX3_STAMP(6);
    // no barrier here: the next tile's phase 1 starts with one (behind its vmcnt(0)), and nothing of this phase is overwritten before it
  }
}

}  // namespace

bool okp_fire_x3_supported(int cin, int mid, int half, int stride, int skip) {
  return cin == CIN && mid == MID && half == MID && ((stride == 1 && skip) || (stride == 2 && !skip));
}

int okp_launch_fire_x3(OkpFire2Params p, int stride, hipStream_t stream) {
  // interior rectangle IH x IW: halo'd footprint <= 128 squeeze pixels, <= 96 interior pixels; minimise the squeeze pixels
  // computed per frame (halo + partial tiles), ties -> wider rows (as okp_fire2's launcher)
  long best = -1;
  for (int ih = 1; ih <= p.Ho && ih <= (stride == 1 ? MAXIH : OH2); ++ih)
    for (int iw = 1; iw <= p.Wo && iw <= 96; ++iw) {
      const int sh = stride * (ih - 1) + 3, sw = stride * (iw - 1) + 3;
      if (sh * sw > SP || ih * iw > 16 * PBI) continue;
      const long ty = (p.Ho + ih - 1) / ih, tx = (p.Wo + iw - 1) / iw;
      const long score = ty * tx * 4096 - iw;
      if (best < 0 || score < best) { best = score; p.IH = ih; p.IW = iw; p.SH = sh; p.SW = sw; p.tiles_y = (int)ty; p.tiles_x = (int)tx; }
    }
  p.IP = p.IH * p.IW;
  const long tiles = (long)p.N * p.tiles_y * p.tiles_x;
  if (tiles >= 0x7FFFFFFFl) { okp_set_error("okp_fire_forward: too many tiles"); return OKP_EINVAL; }
  p.n_tiles = (int)tiles;
  p.div_tiles_frame = okp_fastdiv((uint32_t)(p.tiles_y * p.tiles_x));
  p.div_tiles_x = okp_fastdiv((uint32_t)p.tiles_x);
  p.div_sw = okp_fastdiv((uint32_t)p.SW);
  p.div_iw = okp_fastdiv((uint32_t)p.IW);
  const dim3 grid((unsigned)(p.n_tiles < 256 ? p.n_tiles : 256)), block((unsigned)NT);
#ifdef OKP_FIRE_STAMPS
  static uint32_t* dbg = nullptr;
  if (!dbg) (void)hipMalloc((void**)&dbg, 16 * NW * 8 * 4);
  (void)hipMemsetAsync(dbg, 0, 16 * NW * 8 * 4, stream);
  p.dbg = dbg;
#endif
  if (stride == 1) hipLaunchKernelGGL(okp_fire_x3_kernel<1>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(okp_fire_x3_kernel<2>, grid, block, 0, stream, p);
#ifdef OKP_FIRE_STAMPS
  if (getenv("OKP_FIRE_STAMPS_PRINT")) {
    (void)hipStreamSynchronize(stream);
    uint32_t h[16 * NW * 8];
    (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
    printf("fire_x3 stamps (tile %d x %d interior, %d tiles): clocks per wave of the second tile: wait for the ring + older stores | squeeze GEMM (8 k-steps) | s -> LDS | expand GEMM + stores | barrier + next tile's requests | depth-wise branch\n", p.IH, p.IW, p.n_tiles);
    double sum[6] = {0, 0, 0, 0, 0, 0};
    for (int wg = 0; wg < 16; ++wg)
      for (int w = 0; w < NW; ++w) {
        const uint32_t* a = &h[(wg * NW + w) * 8];
        if (wg % 5 == 0 && (w == 0 || w == 7)) printf("wg %2d wave %d: %6u | %6u | %6u | %6u | %6u | %6u\n", wg, w, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3], a[5] - a[4], a[6] - a[5]);
        for (int i = 0; i < 6; ++i) sum[i] += a[i + 1] - a[i];
      }
    printf("mean over 16 workgroups x 8 waves: %.0f | %.0f | %.0f | %.0f | %.0f | %.0f  = %.0f clocks per tile\n", sum[0] / 128, sum[1] / 128, sum[2] / 128, sum[3] / 128, sum[4] / 128, sum[5] / 128,
           (sum[0] + sum[1] + sum[2] + sum[3] + sum[4] + sum[5]) / 128);
  }
#endif
  return okp_check_hip(hipGetLastError(), "okp_fire_x3 launch");
}

// Role-split streaming fire module, 256 -> 128 -> 256 at stride 1 (bf16 / fp16, gfx950): the instance of the two high-resolution
// hourglass levels (64x64 and 32x32), where the module is bound by HBM: x read once, out written once (268 MB per launch at 64x64,
// N=64), ~40 GFLOP of matrix work and ~0.6 GFLOP of depth-wise vector work.
//
//     s   = W1 x + b1                                   squeeze 1x1 (+bn1, no ReLU)      256 -> 128
//     y_a = relu(Wa s + ba (+ x[:, :128]))              expand 1x1 (+bn2 half, skip)     128 -> 128
//     y_b = relu(dw3x3(s) * wd + bd (+ x[:, 128:]))     depth-wise 3x3 (+bn2 half, skip) 128 -> 128
// (reference: fire_module, corner_net_lite/core/models/CornerNet_Squeeze.py:10-30)
//
// okp_fire2_kernel runs these phases one after the other in two 4-wave workgroups per CU at 256 VGPRs: x streams in only while a
// workgroup is in its squeeze K loop (a quarter of its tile time), the phases' latencies add up, 1.7 TB/s.  Here ONE 8-wave
// workgroup per CU splits the ROLES between the two waves of every SIMD, so that matrix work, vector work and the HBM stream of
// different tiles overlap by construction:
//  * waves 0-3 ("G"): squeeze GEMM of tile t+1 and its squeeze tile -> LDS.  Wave w owns 32 PIXELS of the 128-pixel halo'd
//    squeeze tile (all 128 squeeze channels), so it reads only x bytes that its own LDS-DMA fetched: the K loop has NO
//    workgroup barrier, just counted s_waitcnt vmcnt on a private 7-stage ring (2 KiB per stage: 32 pixels x 32 channels), refilled
//    a whole tile ahead: 48 KiB of x in flight per CU at any time, through both barriers of the tile.  The squeeze weights (64 KiB,
//    MFMA-fragment order) sit in LDS for the whole launch; a K-step is 2 + 8 fragment reads for 16 MFMAs (16x16x32).
//  * waves 4-7 ("D"): expand GEMM (48 MFMAs per wave) + skip + ReLU + stores, then depth-wise 3x3 + skip + ReLU + stores of tile t from
//    the LDS squeeze tile while the G waves multiply tile t+1: mostly vector work beside pure matrix work on every SIMD; the skip
//    values are re-read from L2 (the lines the tile's LDS-DMA fetched a tile earlier).
//  * two workgroup barriers per tile: B1 (D has read s(t); G has s(t+1) in registers) - G writes s(t+1) - B2 (s(t+1) published).
//  Tile geometry, LDS layouts of the squeeze tile, MFMA operand mapping (channels as rows: a lane holds eight adjacent channels of a
//  pixel, 16-byte accesses everywhere) and the arithmetic order are those of okp_fire2_kernel: results are bit-identical to it.
#include <cstdio>
#include <cstring>
#include <cstdlib>

#include "okp_internal.h"

namespace {

constexpr uint32_t kInvalid = 0x80000000u;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CIN = 256, MID = 128, HALF = 128;
constexpr int SP = 128;                          // squeeze-tile rows in LDS
constexpr int PBI = 6;                           // interior pixel blocks of 16 (IP <= 96)
constexpr int MAXIH = 6;                         // interior rows per tile
constexpr int KS1 = CIN / 32, KS2 = MID / 32;    // squeeze / expand k-steps
constexpr int NG = 4;                            // G waves (= D waves)
constexpr int NST = 7;                           // x ring stages per G wave
constexpr int SWM = 15;                          // swizzle key bits of the squeeze tile (16 chunks of 16 B per row)
constexpr int CG = MID / 8;                      // 8-channel groups of the depth-wise branch; 256 D threads / CG = 16 column slots

constexpr int OFF_S = 0;                                   // [128][128 x 2 B] squeeze tile, 16-B chunks XOR-swizzled by the row
constexpr int OFF_W1 = OFF_S + SP * MID * 2;               // [4 channel groups][2 blocks][8 k-steps][64 lanes][16 B] squeeze weights
constexpr int OFF_X = OFF_W1 + CIN * MID * 2;              // [G wave][stage][2 pixel blocks][16 rows x 64 B] x ring
constexpr int OFF_WD = OFF_X + NG * NST * 2048;            // [9][128] fp32 depth-wise weights, then [128] bias
constexpr int OFF_B1 = OFF_WD + 10 * HALF * 4;             // [128] fp32 squeeze bias
constexpr int OFF_BA = OFF_B1 + MID * 4;                   // [128] fp32 expand bias
constexpr int OFF_TAB = OFF_BA + HALF * 4;                 // interior pixel ip -> byte offset relative to the tile's first pixel: [96] in x, [96] in out;
                                                           // [96] squeeze-tile row of the pixel (the expand GEMM's B operand)
constexpr int LDS_TOTAL = OFF_TAB + 3 * 96 * 4;
static_assert(LDS_TOTAL <= 160 * 1024, "LDS");

__device__ __forceinline__ int fastdiv(int x, const OkpFastDiv& f) {
  return f.mul ? (int)(__umulhi((uint32_t)x, f.mul) >> f.shift) : x;
}

// every LDS access of this wave has completed, then the workgroup barrier; a compiler barrier for memory operations as well
// (s_barrier alone orders nothing the compiler or the LDS queue knows about: DESIGN.md 2.1).  Vector-memory operations
// (the x ring's LDS-DMA, stores) stay in flight across it.
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T>
__global__ __launch_bounds__(512) void okp_fire3_kernel(const OkpFire2Params p) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_TOTAL];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_g = w8 < NG;
  const int w = is_g ? w8 : w8 - NG;               // index within the role
  const int l16 = lane & 15, q = lane >> 4;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);

  int tile = blockIdx.x;
  if (tile >= p.n_tiles) return;

  // ---- once per workgroup: squeeze weights, depth-wise constants, squeeze bias, interior tables -> LDS ----------------------
  {
    const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, CIN * MID * 2, 0x00020000);
#pragma unroll
    for (int i = 0; i < 8; ++i)                    // 64 KiB in fragment order (okp_conv_create: fragT_dev), 1 KiB per instruction
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (lds_ptr_t)(smem + OFF_W1 + (w8 * 8 + i) * 1024), 16, (int)((uint32_t)(w8 * 8 + i) * 1024u + (uint32_t)lane * 16u), 0, 0, 0);
    for (int i = tid; i < 10 * HALF; i += 512)
      reinterpret_cast<float*>(smem + OFF_WD)[i] = i < 9 * HALF ? p.wd[i] : p.bd[i - 9 * HALF];
    if (tid < MID) reinterpret_cast<float*>(smem + OFF_B1)[tid] = p.b1[tid];
    else if (tid < MID + HALF) reinterpret_cast<float*>(smem + OFF_BA)[tid - MID] = p.ba[tid - MID];
    if (tid < 16 * PBI) {
      const int iy = fastdiv(tid, p.div_iw), ix = tid - iy * p.IW;
      reinterpret_cast<uint32_t*>(smem + OFF_TAB)[tid] = tid < p.IP ? (uint32_t)(iy * p.W + ix) * (uint32_t)(p.x_ps * 2) : kInvalid;
      reinterpret_cast<uint32_t*>(smem + OFF_TAB)[96 + tid] = tid < p.IP ? (uint32_t)(iy * p.Wo + ix) * (uint32_t)(p.out_ps * 2) : kInvalid;
      reinterpret_cast<uint32_t*>(smem + OFF_TAB)[192 + tid] = tid < p.IP ? (uint32_t)((iy + 1) * p.SW + ix + 1) : (uint32_t)(p.SW + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
  }

  auto tile_origin = [&](int slot, int& n, int& y0, int& x0) {
    // XCD-aware order (as in okp_fire2_kernel): slots with equal slot % 8 share an L2; each XCD gets a contiguous range of tiles
    const int xq = p.n_tiles >> 3, xr = p.n_tiles & 7, xcd = slot & 7;
    const int t = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (slot >> 3);
    n = fastdiv(t, p.div_tiles_frame);
    const int trem = t - n * p.tiles_y * p.tiles_x;
    const int ty = fastdiv(trem, p.div_tiles_x);
    y0 = ty * p.IH;
    x0 = (trem - ty * p.tiles_x) * p.IW;
  };

  if (is_g) {
    // =========================================== G waves ===========================================================================
    const int chq = 32 * w + 8 * q;                  // expand GEMM: this lane's eight output channels (block b, register r: chq + 4 b + r)
    u32x4 waf[2][KS2];                               // expand weights: resident
    {
      const u32x4* const wa_lane = static_cast<const u32x4*>(p.wa) + (size_t)w * 2 * KS2 * 64 + lane;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) waf[b][ks] = wa_lane[(size_t)(b * KS2 + ks) * 64];
    }
    // skip values x[:, :128] of this lane's interior pixels: six 16-byte loads - L2 hits: the tile's LDS-DMA brings the lines in.  They are requested at the end of the previous tile's
    // work, in front of the two barriers (the D waves reach B1 ahead of the G waves, which set the pace at the HBM rate): they land
    // while this wave waits there.
    auto prefetch_ex = [&](int t, u32x4 (&r_ex)[PBI]) {
      int n, y0, x0;
      tile_origin(t, n, y0, x0);
      const int live = (t < p.n_tiles) & (p.skip != 0) & !(p.RPR & 64);
      const int full = (y0 + p.IH <= p.Ho) & (x0 + p.IW <= p.Wo);
      int l16t = l16;
      asm volatile("" : "+v"(l16t));
      const uint32_t xb = (uint32_t)(((long)n * p.Ho + y0) * p.Wo + x0) * (uint32_t)(p.x_ps * 2) + (uint32_t)chq * 2u;
#pragma unroll
      for (int pb = 0; pb < PBI; ++pb) {
        const int ip = 16 * pb + l16t;
        const uint32_t xr = *reinterpret_cast<const uint32_t*>(smem + OFF_TAB + ip * 4);           // (kInvalid beyond the interior)
        const int iy = fastdiv(ip, p.div_iw), ix = ip - iy * p.IW;
        const int bad = (xr == kInvalid) | (live ^ 1) | ((full ^ 1) & ((y0 + iy >= p.Ho) | (x0 + ix >= p.Wo)));
        r_ex[pb] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(bad ? kInvalid : xb + xr), 0, 0);
      }
    };
    auto expand = [&](int t, const u32x4 (&r_raw)[PBI]) {
      int n, y0, x0;
      tile_origin(t, n, y0, x0);
      int qt = q, l16t = l16;
      asm volatile("" : "+v"(qt), "+v"(l16t));
      f32x4 ac2[PBI][2];
      uint32_t a_row[PBI], a_key[PBI];             // B operand: interior pixel block pb, column l16 -> squeeze-tile row
      {
        const f32x4 bav0 = *reinterpret_cast<const f32x4*>(smem + OFF_BA + (chq + 0) * 4), bav1 = *reinterpret_cast<const f32x4*>(smem + OFF_BA + (chq + 4) * 4);
#pragma unroll
        for (int pb = 0; pb < PBI; ++pb) {
          ac2[pb][0] = bav0;
          ac2[pb][1] = bav1;
          const uint32_t sp = *reinterpret_cast<const uint32_t*>(smem + OFF_TAB + (192 + 16 * pb + l16t) * 4);
          a_row[pb] = sp * (MID * 2);
          a_key[pb] = sp & SWM;
        }
      }
      // No tile-shape conditions in here: blocks beyond the interior multiply a valid row and store nowhere (a branch per block serialises
      // every LDS read behind the previous block's MFMAs).  The six fragments of k-step ks + 1 are requested in front of the twelve
      // MFMAs of k-step ks: this wave is alone of its kind on its SIMD, nobody else hides its LDS latency (300+ clocks under load).
      u32x4 af[KS2][PBI];
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
        for (int pb = 0; pb < PBI; ++pb) af[ks][pb] = *reinterpret_cast<const u32x4*>(smem + OFF_S + a_row[pb] + (((uint32_t)(4 * ks + qt) ^ a_key[pb]) << 4));
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
        for (int pb = 0; pb < PBI; ++pb)
#pragma unroll
          for (int b = 0; b < 2; ++b) ac2[pb][b] = H16<T>::mfma16(waf[b][ks], af[ks][pb], ac2[pb][b]);
      // order: fragments of k-steps 0, 1 | MFMAs of 0 | fragments of 2 | MFMAs of 1 | fragments of 3 | MFMAs of 2, 3
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * PBI, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * PBI, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, PBI, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * PBI, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, PBI, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * PBI, 0);
      {
        const int full = (y0 + p.IH <= p.Ho) & (x0 + p.IW <= p.Wo);
        const uint32_t ob = (uint32_t)(((long)n * p.Ho + y0) * p.Wo + x0) * (uint32_t)(p.out_ps * 2) + (uint32_t)chq * 2u;
        uint32_t o_off[PBI];
#pragma unroll
        for (int pb = 0; pb < PBI; ++pb) {
          const int ip = 16 * pb + l16t;
          const uint32_t orr = *reinterpret_cast<const uint32_t*>(smem + OFF_TAB + (96 + ip) * 4);      // (kInvalid beyond the interior)
          const int iy = fastdiv(ip, p.div_iw), ix = ip - iy * p.IW;
          const int bad = (orr == kInvalid) | ((full ^ 1) & ((y0 + iy >= p.Ho) | (x0 + ix >= p.Wo))) | ((p.RPR & 32) != 0);
          o_off[pb] = bad ? kInvalid : ob + orr;
        }
#pragma unroll
        for (int pb = 0; pb < PBI; ++pb) {
          u32x4 v;
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            v[2 * b] = okp_pack2<T>(fmaxf(ac2[pb][b][0] + H16<T>::lo(r_raw[pb][2 * b]), 0.f), fmaxf(ac2[pb][b][1] + H16<T>::hi(r_raw[pb][2 * b]), 0.f));
            v[2 * b + 1] = okp_pack2<T>(fmaxf(ac2[pb][b][2] + H16<T>::lo(r_raw[pb][2 * b + 1]), 0.f), fmaxf(ac2[pb][b][3] + H16<T>::hi(r_raw[pb][2 * b + 1]), 0.f));
          }
          __builtin_amdgcn_raw_buffer_store_b128(v, rs_o, (int)o_off[pb], 0, 0);
        }
      }
      asm volatile("" ::: "memory");             // keep the stores here (the scheduler otherwise sinks them below the depth-wise phase)
    };
    // K loop: this wave's pixels are squeeze-tile rows 32 w .. 32 w + 31 (blocks pbl = 0, 1 of 16 rows)
    // x-ring fragment read: row l16 of a block, chunk q sits at position (q + 2 (row >> 2)) & 3 (conflict-free 16x16 fragment reads)
    const uint32_t xfrag_off = (uint32_t)l16 * 64u + (uint32_t)((q + 2 * (l16 >> 2)) & 3) * 16u;
    // LDS-DMA: one instruction = 16 rows x 64 B; lane -> (row lane >> 2, position lane & 3), fetching the k-chunk the read side expects there
    int d_sy[2], d_sx[2];
    uint32_t d_chunk[2];
    int m_sy[2], m_sx[2];                            // the pixel whose squeeze values this lane holds after the K loop (row 32 w + 16 pbl + l16)
#pragma unroll
    for (int pbl = 0; pbl < 2; ++pbl) {
      const int row = 32 * w + 16 * pbl + (lane >> 2);
      d_sy[pbl] = fastdiv(row, p.div_sw);
      d_sx[pbl] = row - d_sy[pbl] * p.SW;
      if (row >= p.SH * p.SW) d_sy[pbl] = 1 << 20;            // rows beyond the tile: always out of frame
      d_chunk[pbl] = (uint32_t)(((lane & 3) - 2 * ((lane >> 2) >> 2)) & 3) * 16u;
      const int mrow = 32 * w + 16 * pbl + l16;
      m_sy[pbl] = mrow < p.SH * p.SW ? fastdiv(mrow, p.div_sw) : (1 << 20);
      m_sx[pbl] = mrow - fastdiv(mrow, p.div_sw) * p.SW;
    }
    // squeeze result -> LDS: the lane's eight channels 32 cgp + 8 q .. + 7 of pixel (row) are the 16-byte chunk 4 cgp + q of the row
    uint32_t s_dst[2];
#pragma unroll
    for (int pbl = 0; pbl < 2; ++pbl) {
      const int row = 32 * w + 16 * pbl + l16;
      s_dst[pbl] = (uint32_t)(OFF_S + row * (MID * 2));
    }
    auto tile_doff = [&](int t, uint32_t (&d)[2]) {            // DMA source offsets of tile slot t (all masked past the last tile)
      int n, y0, x0;
      tile_origin(t, n, y0, x0);
#pragma unroll
      for (int pbl = 0; pbl < 2; ++pbl) {
        const int y = y0 - 1 + d_sy[pbl], x = x0 - 1 + d_sx[pbl];
        const bool ok = t < p.n_tiles && y >= 0 && y < p.H && x >= 0 && x < p.W && !(p.RPR & 4);
        d[pbl] = ok ? (uint32_t)(((long)n * p.H + y) * p.W + x) * (uint32_t)(p.x_ps * 2) + d_chunk[pbl] : kInvalid;
      }
    };
    char* const xring = smem + OFF_X + w * NST * 2048;
    auto issue_x = [&](const uint32_t (&d)[2], int ks, int slot) {
#pragma unroll
      for (int pbl = 0; pbl < 2; ++pbl)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(xring + slot * 2048 + pbl * 1024), 16,
                                                 (int)(d[pbl] == kInvalid ? kInvalid : d[pbl] + (uint32_t)ks * 64u), 0, 0, 0);
    };

    uint32_t d_cur[2], d_nxt[2];
    tile_doff(tile, d_cur);
#pragma unroll
    for (int ks = 0; ks < NST; ++ks) issue_x(d_cur, ks, ks);     // stages 0 .. NST-1 of the first tile into slots 0 .. NST-1
    int slot = 0;                                                // ring slot of the stage the next step multiplies
    // The K loop is ONE software pipeline over all steps of all tiles of this wave: step g multiplies the fragments that step g - 1
    // requested (pixel fragments of stage g, weight fragments of k-step g % 8: 10 ds_read_b128 in registers across the step boundary -
    // and across the tile's two barriers) while it requests those of step g + 1 between its MFMAs, so no LDS latency is exposed.
    // Step g first refills the slot of stage g (its fragments are in registers already) with stage g + NST, then needs stage g + 1:
    // six stages (12 LDS-DMA instructions) stay in flight behind it.
    u32x4 xf[2], wf[8];
    auto read_w = [&](int ks, u32x4 (&dst)[8]) {
#pragma unroll
      for (int f = 0; f < 8; ++f) dst[f] = *reinterpret_cast<const u32x4*>(smem + OFF_W1 + (f * KS1 + ks) * 1024 + lane * 16);
    };
    auto read_x = [&](int sl, u32x4 (&dst)[2]) {
      const char* st = xring + sl * 2048 + xfrag_off;
#pragma unroll
      for (int pbl = 0; pbl < 2; ++pbl) dst[pbl] = *reinterpret_cast<const u32x4*>(st + pbl * 1024);
    };
    // (six masked stores and the first tile's six skip loads: the vector-memory stream in front of the first K loop then has the
    //  shape it has in front of every other one - the counted waits below are compile-time constants)
    u32x4 r_ex[PBI];
#pragma unroll
    for (int i = 0; i < PBI; ++i) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rs_o, (int)kInvalid, 0, 0);
    prefetch_ex(tile, r_ex);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");             // stage 0 of the first tile has landed
    read_x(0, xf);
    read_w(0, wf);

    unsigned long long gs[5] = {0, 0, 0, 0, 0};
    for (; tile < p.n_tiles; tile += gridDim.x) {
      const unsigned long long c0 = __builtin_amdgcn_s_memtime();
      int n, y0, x0;
      tile_origin(tile, n, y0, x0);
      tile_doff(tile + (int)gridDim.x, d_nxt);
      int qt = q, l16t = l16;
      asm volatile("" : "+v"(qt), "+v"(l16t));
      // ---- K loop: s = W1 x + b1 on this wave's 32 pixels ---------------------------------------------------------------------
      f32x4 acc[4][2][2];                                        // [channel group of 32][block b][pixel block]
#pragma unroll
      for (int cgp = 0; cgp < 4; ++cgp)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(smem + OFF_B1 + (32 * cgp + 8 * qt + 4 * b) * 4);
          acc[cgp][b][0] = bv;
          acc[cgp][b][1] = bv;
        }
      if (!(p.RPR & 128)) {
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        // stage 8t+ks+7 = k-step ks-1 of the NEXT tile (ks = 0: k-step 7 of this one) -> the slot whose fragments this step multiplies
        if (ks == 0) issue_x(d_cur, KS1 - 1, slot);
        else issue_x(d_nxt, ks - 1, slot);
        slot = slot + 1 == NST ? 0 : slot + 1;
        // Vector-memory operations of this wave in issue order: ... DMA of steps (t-1, 0..7) | six stores of tile t-1 | six skip loads of
        // tile t | DMA of steps (t, 0..7) ...; two LDS-DMA instructions per step.  Stage 8t+ks+1 was issued six steps ago: for ks <= 5
        // that is in front of the twelve stores / loads (24 younger operations), for ks = 6, 7 behind them (12).
        if (ks <= 5) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        u32x4 xn[2], wn[8];
        read_x(slot, xn);
        read_w((ks + 1) & 7, wn);
        if (!(p.RPR & 2)) {
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
          for (int pbl = 0; pbl < 2; ++pbl) acc[f >> 1][f & 1][pbl] = H16<T>::mfma16(wf[f], xf[pbl], acc[f >> 1][f & 1][pbl]);
        }
        // pin the order: behind every pair of MFMAs one or two of the next step's ten fragment reads
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          if (f < 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        xf[0] = xn[0]; xf[1] = xn[1];
#pragma unroll
        for (int f = 0; f < 8; ++f) wf[f] = wn[f];
      }
      }
      d_cur[0] = d_nxt[0]; d_cur[1] = d_nxt[1];
      const unsigned long long c1 = __builtin_amdgcn_s_memtime();
      wg_barrier();                                              // B1: the D waves have finished with the previous squeeze tile
      const unsigned long long c2 = __builtin_amdgcn_s_memtime();
      // ---- s -> LDS (zero outside the frame: the reference zero-pads the squeeze output) --------------------------------------
      // (a tile whose halo'd footprint lies inside the frame and fills all 128 rows needs no masks; decided per wave, uniformly)
      const bool inner = y0 >= 1 && x0 >= 1 && y0 + p.IH + 1 <= p.H && x0 + p.IW + 1 <= p.W && 32 * w + 32 <= p.SH * p.SW;
#pragma unroll
      for (int pbl = 0; pbl < 2; ++pbl) {
        const int y = y0 - 1 + m_sy[pbl], x = x0 - 1 + m_sx[pbl];
        const bool ok = inner || (y >= 0 && y < p.H && x >= 0 && x < p.W);
        const uint32_t key = (uint32_t)((32 * w + 16 * pbl + l16t) & SWM);
#pragma unroll
        for (int cgp = 0; cgp < 4; ++cgp) {
          u32x4 v;
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            v[2 * b] = okp_pack2<T>(ok ? acc[cgp][b][pbl][0] : 0.f, ok ? acc[cgp][b][pbl][1] : 0.f);
            v[2 * b + 1] = okp_pack2<T>(ok ? acc[cgp][b][pbl][2] : 0.f, ok ? acc[cgp][b][pbl][3] : 0.f);
          }
          *reinterpret_cast<u32x4*>(smem + s_dst[pbl] + ((((uint32_t)(4 * cgp) + (uint32_t)qt) ^ key) << 4)) = v;
        }
      }
      const unsigned long long c3 = __builtin_amdgcn_s_memtime();
      wg_barrier();                                              // B2: the squeeze tile is published
      const unsigned long long c4 = __builtin_amdgcn_s_memtime();
      if (!(p.RPR & (17 | 128))) expand(tile, r_ex);
      else {
#pragma unroll
        for (int i = 0; i < PBI; ++i) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rs_o, (int)kInvalid, 0, 0);
      }
      prefetch_ex(tile + (int)gridDim.x, r_ex);
      const unsigned long long c5 = __builtin_amdgcn_s_memtime();
      gs[0] += c1 - c0; gs[1] += c2 - c1; gs[2] += c3 - c2; gs[3] += c4 - c3; gs[4] += c5 - c4;
      asm volatile("" ::: "memory");
    }
    if (p.dbg && w == 0 && lane == 0) { for (int i = 0; i < 5; ++i) p.dbg[blockIdx.x * 16 + i] = gs[i]; }
  } else {
    // =========================================== D waves ===========================================================================
    const int dt = tid - 64 * NG;
    auto prefetch_dw = [&](int t, u32x4 (&r_dw)[MAXIH]) {
      int n, y0, x0;
      tile_origin(t, n, y0, x0);
      const bool live = t < p.n_tiles && p.skip && !(p.RPR & 64);
      int dtt = dt;
      asm volatile("" : "+v"(dtt));
      const int cg = dtt % CG, ix = dtt / CG;
      uint32_t xo = (uint32_t)(((long)n * p.Ho + y0) * p.Wo + x0 + ix) * (uint32_t)(p.x_ps * 2) + (uint32_t)(HALF + cg * 8) * 2u;
      const bool col_ok = live && ix < p.IW && x0 + ix < p.Wo;
#pragma unroll
      for (int iy = 0; iy < MAXIH; ++iy) {
        r_dw[iy] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)((col_ok && iy < p.IH && y0 + iy < p.Ho) ? xo : kInvalid), 0, 0);
        xo += (uint32_t)(p.W * p.x_ps * 2);
      }
    };
    // y_b = relu(dw3x3(s) + bd (+x)) of one tile from the LDS squeeze tile: thread = (8-channel group cg, column slot); it walks DOWN
    // its column, each squeeze row feeding the three output rows that see it as tap row 2, 1, 0 (the loop of okp_fire2_kernel's phase 2b)
    auto depthwise = [&](int t, const u32x4 (&r_first)[MAXIH]) {
      int n, y0, x0;
      tile_origin(t, n, y0, x0);
      int dtt = dt;
      asm volatile("" : "+v"(dtt));
      const int cg = dtt % CG;
      const float* const wl = reinterpret_cast<const float*>(smem + OFF_WD) + cg * 8;     // [tap][128] fp32, bias at tap 9
      float breg[8];
      {
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(wl + 9 * HALF), u1 = *reinterpret_cast<const f32x4*>(wl + 9 * HALF + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { breg[e] = u0[e]; breg[4 + e] = u1[e]; }
      }
      for (int ix = dtt / CG; ix < ((p.IW + 15) & ~15); ix += 16) {
        u32x4 rr[MAXIH];
        uint32_t oo[MAXIH];
        {
          const int ox = x0 + ix;
          const uint32_t pix = (uint32_t)(((long)n * p.Ho + y0) * p.Wo + ox);
          uint32_t xo = pix * (uint32_t)(p.x_ps * 2) + (uint32_t)(HALF + cg * 8) * 2u, oof = pix * (uint32_t)(p.out_ps * 2) + (uint32_t)(HALF + cg * 8) * 2u;
          const bool col_ok = ix < p.IW && ox < p.Wo;
          const bool first_col = ix < 16;                 // its skip values came with the tile's prefetch; further columns (wide, low tiles) load here
#pragma unroll
          for (int iy = 0; iy < MAXIH; ++iy) {
            const bool ok = col_ok && iy < p.IH && y0 + iy < p.Ho;
            oo[iy] = ok ? oof : kInvalid;
            rr[iy] = r_first[iy];
            if (!first_col) rr[iy] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)((ok && p.skip) ? xo : kInvalid), 0, 0);
            xo += (uint32_t)(p.W * p.x_ps * 2);
            oof += (uint32_t)(p.Wo * p.out_ps * 2);
          }
        }
        const int ixc = ix < p.IW ? ix : 0;
        f32x2 v[MAXIH][4];
#pragma unroll
        for (int iy = 0; iy < MAXIH; ++iy)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[iy][e] = f32x2{breg[2 * e], breg[2 * e + 1]};
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          f32x2 wt[3][4];
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(wl + (dy * 3 + dx) * HALF), u1 = *reinterpret_cast<const f32x4*>(wl + (dy * 3 + dx) * HALF + 4);
            wt[dy][0] = f32x2{u0[0], u0[1]}; wt[dy][1] = f32x2{u0[2], u0[3]}; wt[dy][2] = f32x2{u1[0], u1[1]}; wt[dy][3] = f32x2{u1[2], u1[3]};
          }
          // all eight squeeze rows of the column position requested together, no tile-shape branch (rows below a lower tile repeat
          // its last row and feed output rows that are stored nowhere)
          u32x4 sv[MAXIH + 2];
#pragma unroll
          for (int sr = 0; sr < MAXIH + 2; ++sr) {
            const int sp = (sr < p.IH + 2 ? sr : p.IH + 1) * p.SW + ixc + dx;
            sv[sr] = *reinterpret_cast<const u32x4*>(smem + OFF_S + sp * (MID * 2) + ((cg ^ (sp & SWM)) << 4));
          }
#pragma unroll
          for (int sr = 0; sr < MAXIH + 2; ++sr) {
            f32x2 s2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) s2[e] = f32x2{H16<T>::lo(sv[sr][e]), H16<T>::hi(sv[sr][e])};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
              const int iy = sr - dy;
              if (iy >= 0 && iy < MAXIH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[iy][e] = __builtin_elementwise_fma(s2[e], wt[dy][e], v[iy][e]);
              }
            }
          }
        }
#pragma unroll
        for (int iy = 0; iy < MAXIH; ++iy) {
          u32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = fmaxf(v[iy][e][0] + H16<T>::lo(rr[iy][e]), 0.f);
            const float hi = fmaxf(v[iy][e][1] + H16<T>::hi(rr[iy][e]), 0.f);
            o[e] = okp_pack2<T>(lo, hi);
          }
          __builtin_amdgcn_raw_buffer_store_b128(o, rs_o, (int)((p.RPR & 32) ? kInvalid : oo[iy]), 0, 0);
        }
      }
    };
    // skip values x[:, 128:] of the thread's first column: six 16-byte loads - L2 hits, the tile's LDS-DMA brings the lines in - requested
    // at the end of the previous tile's work, in front of the two barriers: they land while this wave waits there
    u32x4 r_dw[MAXIH];
    prefetch_dw(tile, r_dw);
    unsigned long long ds[5] = {0, 0, 0, 0, 0};
    for (; tile < p.n_tiles; tile += gridDim.x) {
      const unsigned long long c0 = __builtin_amdgcn_s_memtime();
      wg_barrier();                                              // B1: (this wave has finished with the previous squeeze tile)
      const unsigned long long c1 = __builtin_amdgcn_s_memtime();
      wg_barrier();                                              // B2: the squeeze tile of `tile` is published
      const unsigned long long c2 = __builtin_amdgcn_s_memtime();
      if (!(p.RPR & 9)) depthwise(tile, r_dw);
      const unsigned long long c4 = __builtin_amdgcn_s_memtime();
      prefetch_dw(tile + (int)gridDim.x, r_dw);
      const unsigned long long c5 = __builtin_amdgcn_s_memtime();
      ds[0] += c1 - c0; ds[1] += c2 - c1; ds[3] += c4 - c2; ds[4] += c5 - c4;
    }
    if (p.dbg && w == 0 && lane == 0) { for (int i = 0; i < 5; ++i) p.dbg[blockIdx.x * 16 + 8 + i] = ds[i]; }
  }
}

}  // namespace


bool okp_fire3_supported(int cin, int mid, int half, int stride) {
  const char* e = getenv("OKP_FIRE3");        // OKP_FIRE3=0: okp_fire2_kernel instead (A/B runs and the bit-equality test; read per call)
  return !(e && e[0] == '0') && cin == CIN && mid == MID && half == HALF && stride == 1;
}

template <typename T>
static int launch_fire3_t(OkpFire2Params p, hipStream_t stream) {
  // interior rectangle IH x IW as okp_fire2: halo'd footprint <= 128 squeeze pixels, <= 96 interior pixels; fewest tiles, ties -> wider rows
  long best = -1;
  for (int ih = 1; ih <= p.Ho && ih <= MAXIH; ++ih)
    for (int iw = 1; iw <= p.Wo && iw <= 96; ++iw) {
      const int sh = ih + 2, sw = iw + 2;
      if (sh * sw > SP || ih * iw > 16 * PBI) continue;
      const long ty = (p.Ho + ih - 1) / ih, tx = (p.Wo + iw - 1) / iw;
      const long score = ty * tx * 4096 - iw;
      if (best < 0 || score < best) { best = score; p.IH = ih; p.IW = iw; p.SH = sh; p.SW = sw; p.tiles_y = (int)ty; p.tiles_x = (int)tx; }
    }
  p.IP = p.IH * p.IW;
  { const char* e = getenv("OKP_FIRE3_ABL"); p.RPR = e ? atoi(e) : 0; }      // TEMPORARY timing ablations (wrong results)
  const long tiles = (long)p.N * p.tiles_y * p.tiles_x;
  if (tiles >= 0x7FFFFFFFl) { okp_set_error("okp_fire_forward: too many tiles"); return OKP_EINVAL; }
  p.n_tiles = (int)tiles;
  p.div_tiles_frame = okp_fastdiv((uint32_t)(p.tiles_y * p.tiles_x));
  p.div_tiles_x = okp_fastdiv((uint32_t)p.tiles_x);
  p.div_sw = okp_fastdiv((uint32_t)p.SW);
  p.div_iw = okp_fastdiv((uint32_t)p.IW);
  const dim3 grid((unsigned)(p.n_tiles < 256 ? p.n_tiles : 256)), block(512u);      // one workgroup per CU (158 KiB of LDS), persistent
  static unsigned long long* dbg = nullptr;
  const bool want_dbg = getenv("OKP_FIRE3_DBG") != nullptr;
  if (want_dbg && !dbg) (void)hipMalloc((void**)&dbg, 256 * 16 * 8);
  p.dbg = want_dbg ? dbg : nullptr;
  hipLaunchKernelGGL((okp_fire3_kernel<T>), grid, block, 0, stream, p);
  if (want_dbg) {
    static int calls = 0;
    if (++calls == 8) {
      unsigned long long h[256 * 16];
      (void)hipStreamSynchronize(stream);
      (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
      double g[5] = {0, 0, 0, 0, 0}, d[5] = {0, 0, 0, 0, 0};
      const int nwg = (int)grid.x;
      for (int b = 0; b < nwg; ++b) { for (int i = 0; i < 5; ++i) g[i] += (double)h[b * 16 + i]; for (int i = 0; i < 5; ++i) d[i] += (double)h[b * 16 + 8 + i]; }
      const double tiles_per_wg = (double)p.n_tiles / nwg;
      fprintf(stderr, "fire3 clocks per tile (mean over %d workgroups, %.1f tiles each): G: K loop %.0f | wait B1 %.0f | s-write %.0f | wait B2 %.0f | expand+prefetch %.0f   D: wait B1 %.0f | wait B2 %.0f | expand %.0f | depth-wise %.0f | prefetch %.0f\n",
              nwg, tiles_per_wg, g[0] / nwg / tiles_per_wg, g[1] / nwg / tiles_per_wg, g[2] / nwg / tiles_per_wg, g[3] / nwg / tiles_per_wg, g[4] / nwg / tiles_per_wg,
              d[0] / nwg / tiles_per_wg, d[1] / nwg / tiles_per_wg, d[2] / nwg / tiles_per_wg, d[3] / nwg / tiles_per_wg, d[4] / nwg / tiles_per_wg);
    }
  }
  return okp_check_hip(hipGetLastError(), "okp_fire3 launch");
}

int okp_launch_fire3(int dtype, const OkpFire2Params& p, hipStream_t stream) {
  if (dtype == OKP_BF16) return launch_fire3_t<__bf16>(p, stream);
  return launch_fire3_t<_Float16>(p, stream);
}

// A CHAIN of stride-1 fire modules (CIN -> CIN/2 -> CIN, with skip) on maps of at most 16 pixels, one workgroup per
// frame, activations resident in LDS from the first module to the last (bf16, gfx950).
//
// The innermost hourglass level runs six consecutive fire_module(512, 512) on 4x4 maps (low1[1], low2[0..3], low3[0];
// corner_net_lite/core/models/CornerNet_Squeeze.py:10-51 with modules [2,2,2,2,4]).  As separate launches each costs
// about 20 us of launch and memory latency for ~3 MMAC per frame; here the frame's 16 x 512 activation stays in LDS,
// only the weights stream from L2 (390 KB per module and workgroup), and the chain costs one launch.
//
//   per module:  s = W1 x + b1 (no ReLU)            squeeze 1x1            CIN -> MID = CIN/2
//                y[:, :MID] = relu(Wa s + ba + x[:, :MID])                 expand 1x1
//                y[:, MID:] = relu(dw3x3(s) * wd + bd + x[:, MID:])        depth-wise 3x3, zero padding
//
// 16x16x32 MFMAs with the 16 pixels as rows (A operand from LDS) and channels as columns (B operand = weight fragments
// re-laid once per plan into fragment order so that a load is 1 KiB contiguous, prefetched PF k-steps ahead); wave w owns channels
// [32 w, 32 w + 32) of both GEMMs, columns interleaved (2 j + b) so that a lane's two accumulator blocks are adjacent
// channels (one dword per pixel).  The depth-wise branch: thread = (8-channel group, pixel), its 72 weights fetched at
// the start of the module.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <type_traits>

#include "okp_internal.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// EE = true (CIN = 512 only): the chain is framed by the two modules around it at the innermost hourglass level - module 0 is the
// stride-2 ENTRY fire module 384 -> 256 -> 512 reading a map of at most 8 x 8 pixels (make_hg_layer's first module,
// CornerNet_Squeeze.py:20-21 with stride 2: squeeze at full resolution, both branches at stride 2, no skip), the last one the EXIT
// module 512 -> 192 -> 384 (make_layer_revr's last, no skip).  Four more launches of ~10 us each become phases of this one.
template <int CIN, int PXB, bool EE = false>
struct ChainCfg {
  static constexpr int MID = CIN / 2;
  static constexpr int NW = MID / 32;                  // waves
  static constexpr int NT = 64 * NW;
  static constexpr int KS1 = CIN / 32, KS2 = MID / 32;
  static constexpr int NPX = 16 * PXB;                 // pixel rows of an activation buffer (map of at most NPX pixels)
  static constexpr int SIDE = PXB == 1 ? 4 : 8;        // largest map side: 4x4 (one MFMA row block) or 8x8 (four)
  static constexpr int SROWS = (SIDE + 2) * (SIDE + 2);
  static constexpr int XROW = CIN * 2;                 // bytes per pixel row of an activation buffer
  static constexpr int SROW = MID * 2;
  // 16-byte chunks per row are XOR-swizzled with the row inside groups of KEY+1 chunks: the group must divide the row
  static constexpr int XKEY = (CIN / 8) % 16 == 0 ? 15 : 7;
  static constexpr int SKEY = (MID / 8) % 16 == 0 ? 15 : 7;
  static constexpr int ECIN = 384, ENPX = 64, ESIDE = 8, EKS1 = ECIN / 32, EXROW = ECIN * 2;     // entry module: input map, squeeze k-steps
  static constexpr int ESROWS = (ESIDE + 2) * (ESIDE + 2);
  static constexpr int XMID = 192, XKS2 = XMID / 32;   // exit module
  static constexpr int OFF_X = 0;                      // two activation buffers [NPX][CIN]; EE: the entry module's input [64][384] lies over them
  static constexpr int OFF_S = EE ? ENPX * EXROW : 2 * NPX * XROW;   // squeeze tile [(SIDE+2)^2 rows, zero halo][MID]; EE: [(8+2)^2] for the entry module first
  // per module: depth-wise weights [9][MID], bd, b1, ba (fp32), every array padded to whole 1 KiB LDS-DMA instructions
  static constexpr int WD_BYTES = (9 * MID * 4 + 1023) / 1024 * 1024, B_BYTES = (MID * 4 + 1023) / 1024 * 1024;
  static constexpr int CONST_BYTES = WD_BYTES + 3 * B_BYTES;
  static constexpr int OFF_C = OFF_S + (EE ? ESROWS : SROWS) * SROW;   // two such blocks: the next module's constants arrive by LDS-DMA while this one runs
  static constexpr int BYTES = OFF_C + 2 * CONST_BYTES;
  static_assert((CIN / 8) % 8 == 0 && (MID / 8) % 8 == 0 && BYTES <= 160 * 1024, "row swizzle / LDS");
  static_assert(!EE || (CIN == 512 && PXB == 1 && ENPX * EXROW >= 2 * NPX * XROW), "entry / exit modules frame the 512-channel chain");
};

// 16-byte chunk c of row r, XOR-swizzled by the row (the rows and k-groups of a fragment read hit distinct banks)
template <int KEY>
__device__ __forceinline__ uint32_t xoff(int row, int chunk, int rowbytes) { return (uint32_t)row * rowbytes + (uint32_t)((chunk ^ (row & KEY)) << 4); }

#ifdef OKP_FIRE_STAMPS
// Debug build (OKP_EXTRA_CFLAGS=-DOKP_FIRE_STAMPS, printed with OKP_FIRE_STAMPS_PRINT=1): shader-clock stamps of every wave of workgroups 0..15 at the
// phase boundaries of the chain's SECOND chain module, and at kernel entry / before the first chain module / exit: [wg][wave 8][8] u32
#define FC_STAMP(cond, i) do { if ((cond) && lane == 0 && blockIdx.x < 16) { uint64_t t_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); p.dbg[(blockIdx.x * 8 + w) * 8 + (i)] = (uint32_t)t_; } } while (0)
#else
#define FC_STAMP(cond, i) do {} while (0)
#endif

template <typename T, int CIN, int PXB, bool EE>
__global__ __launch_bounds__((ChainCfg<CIN, PXB, EE>::NT)) void okp_fire_chain_kernel(const OkpFireChainParams p) {
  using C = ChainCfg<CIN, PXB, EE>;
  using x2_t = typename H16<T>::x2;
  using x8_t = typename H16<T>::x8;
  constexpr int MID = C::MID, NT = C::NT, KS1 = C::KS1, KS2 = C::KS2, XROW = C::XROW, SROW = C::SROW, NPX = C::NPX;
  constexpr int XK = C::XKEY, SK = C::SKEY;
  // Weight fragments stream through ONE register ring of PF = KS2 k-steps that never drains: the last PF squeeze steps fetch the
  // expand fragments, the expand steps fetch the NEXT module's first squeeze fragments (KS1 = 2 KS2, so a k-step's slot
  // ks % PF is the same in every phase).  A module streams 390 KB per workgroup: with a ring per phase the pipe restarted
  // three times per module and a module took 9.7 us at 40 GB/s per CU.
  constexpr int PF = KS2;
  static_assert(KS1 % PF == 0 && KS2 % PF == 0, "ring slots");
  __shared__ __attribute__((aligned(16))) char smem[C::BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, q = lane >> 4;
  const int n = blockIdx.x;
  const int HW = p.H * p.W;
  FC_STAMP(true, 5);
  const int ch0 = 32 * w + 2 * l16;                    // this lane's channel pair in both GEMMs

  // frame -> LDS (buffer 0), squeeze tile cleared once: its halo stays zero for the whole chain.
  // Prologue order (round 6): everything the first module waits for is requested BEFORE anything is waited for - its constants (LDS-DMA), its
  // first PF weight fragments (registers) and the frame (LDS-DMA).  As a plain loop hipcc compiled the frame copy to
  // `global_load -> s_waitcnt vmcnt(0) -> ds_write` per iteration (6-8 cold reads one behind the other), followed by the constants' round trip and
  // the fragments' round trip: ~10 memory latencies in front of a 25-70 us kernel.
  const int HWe = p.He * p.We;                          // (EE) pixels of the entry module's input map
  auto stage_frame = [&](auto ncin_c, auto npx_c, auto key_c, int n_valid) {
    // by LDS-DMA (no registers, nothing waits): instruction k fills the LDS bytes [1024 k, +1024) of the buffer - row px = i / NCH, chunk POSITION
    // pos = i % NCH for lane index i = 64 k + lane - so the lane fetches the chunk that belongs there, c = pos ^ (px & KEY); pixels beyond the map
    // read zeros through an out-of-range offset
    constexpr int NCH = decltype(ncin_c)::value / 8, TOT = decltype(npx_c)::value * NCH, KEYV = decltype(key_c)::value, NINST = TOT / 64;
    static_assert(TOT % 64 == 0 && NCH % (KEYV + 1) == 0, "whole LDS-DMA instructions, swizzle inside the row");
    const __amdgpu_buffer_rsrc_t rsf = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(p.x) + (size_t)n * n_valid * p.x_ps * 2), 0,
                                                                         n_valid * p.x_ps * 2, 0x00020000);
#pragma unroll
    for (int k0 = 0; k0 < NINST; k0 += C::NW) {
      const int k = k0 + w;
      if (k < NINST) {
        const int i = 64 * k + lane;
        const int px = i / NCH, pos = i % NCH;
        const int c = pos ^ (px & KEYV);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsf, (lds_ptr_t)(smem + C::OFF_X + k * 1024), 16,
                                                 (int)(px < n_valid ? (uint32_t)(px * p.x_ps * 2 + c * 16) : 0x80000000u), 0, 0, 0);
      }
    }
  };
  // interior pixel l16 -> squeeze-tile row (zero halo of one pixel around the H x W map, row pitch W + 2)
  const int SW = p.W + 2;
  auto srow = [&](int px) { const int py = px / p.W; return (py + 1) * SW + (px - py * p.W) + 1; };

  // per-module constants -> LDS by LDS-DMA (no registers, nobody waits until the module that needs them starts):
  // a module then has no dependent global round trips besides its streamed weight fragments
  auto fetch_consts = [&](const OkpFireChainModule& mod, int slot, int mid) {      // mid: squeeze channels of THIS module (exit module: 192)
    char* dst = smem + C::OFF_C + slot * C::CONST_BYTES;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(mod.wd), 0, 9 * mid * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(mod.bd), 0, mid * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(mod.b1), 0, mid * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(mod.ba), 0, mid * 4, 0x00020000);
    // instruction i of an array covers its bytes [1024 i, +1024) (lanes beyond the array read zeros into the padding);
    // the instructions of the four arrays are dealt round-robin over the waves
    int k = 0;
    for (int i = 0; i < C::WD_BYTES / 1024; ++i, ++k)
      if (k % C::NW == w) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + i * 1024), 16, i * 1024 + lane * 16, 0, 0, 0);
    for (int i = 0; i < C::B_BYTES / 1024; ++i, ++k)
      if (k % C::NW == w) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(dst + C::WD_BYTES + i * 1024), 16, i * 1024 + lane * 16, 0, 0, 0);
    for (int i = 0; i < C::B_BYTES / 1024; ++i, ++k)
      if (k % C::NW == w) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr_t)(dst + C::WD_BYTES + C::B_BYTES + i * 1024), 16, i * 1024 + lane * 16, 0, 0, 0);
    for (int i = 0; i < C::B_BYTES / 1024; ++i, ++k)
      if (k % C::NW == w) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(dst + C::WD_BYTES + 2 * C::B_BYTES + i * 1024), 16, i * 1024 + lane * 16, 0, 0, 0);
  };
  fetch_consts(p.mod[0], 0, MID);
  auto frag = [&](const u32x4* base, int ksteps, int ks, int b) { return base[(size_t)(b * ksteps + ks) * 64]; };
  u32x4 wf[PF][2];
  {
    constexpr int FKS1 = EE ? C::EKS1 : KS1;                 // first module: the entry module (EE) or the first chain module
    const u32x4* w1_first = static_cast<const u32x4*>(p.mod[0].w1) + (size_t)w * 2 * FKS1 * 64 + lane;
#pragma unroll
    for (int i = 0; i < PF; ++i) { wf[i][0] = frag(w1_first, FKS1, i, 0); wf[i][1] = frag(w1_first, FKS1, i, 1); }
  }
  if constexpr (EE) {
    stage_frame(std::integral_constant<int, C::ECIN>{}, std::integral_constant<int, C::ENPX>{}, std::integral_constant<int, 15>{}, HWe);
    for (int i = tid; i < C::ESROWS * (MID / 8); i += NT) *reinterpret_cast<u32x4*>(smem + C::OFF_S + i * 16) = u32x4{0u, 0u, 0u, 0u};
  } else {
    stage_frame(std::integral_constant<int, CIN>{}, std::integral_constant<int, NPX>{}, std::integral_constant<int, XK>{}, HW);
    for (int i = tid; i < C::SROWS * (MID / 8); i += NT) *reinterpret_cast<u32x4*>(smem + C::OFF_S + i * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                           // the frame, the cleared squeeze tile and the first module's constants are in LDS
  int cur = 0;

  // ---- EE: entry module (p.mod[0]): s8 = W1 x8 + b1 on the He x We map; y[:, :256] = relu(Wa s8[::2, ::2] + ba);
  //      y[:, 256:] = relu(dw3x3/s2(s8) * wd + bd) -> activation buffer 0 (H x W = ceil(He / 2) x ceil(We / 2)) -------------------
  if constexpr (EE) {
    constexpr int EKS1 = C::EKS1;
    static_assert(PF == 8 && EKS1 == 12, "entry ring schedule");
    const OkpFireChainModule mod = p.mod[0];
    const float* cst = reinterpret_cast<const float*>(smem + C::OFF_C);
    const float* cbd = cst + C::WD_BYTES / 4;
    const float* cb1 = cbd + C::B_BYTES / 4;
    const float* cba = cb1 + C::B_BYTES / 4;
    fetch_consts(p.mod[1], 1, MID);                          // (the module behind the entry module is a chain module: mid = MID)
    const u32x4* w1_lane = static_cast<const u32x4*>(mod.w1) + (size_t)w * 2 * EKS1 * 64 + lane;
    const u32x4* wa_lane = static_cast<const u32x4*>(mod.wa) + (size_t)w * 2 * KS2 * 64 + lane;
    const u32x4* w1_next = static_cast<const u32x4*>(p.mod[1].w1) + (size_t)w * 2 * KS1 * 64 + lane;
    const int SWe = p.We + 2;
    const char* x8 = smem + C::OFF_X;
    {
      const float b0 = cb1[ch0], b1 = cb1[ch0 + 1];
      f32x4 acc0[4], acc1[4];
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) { acc0[pb] = f32x4{b0, b0, b0, b0}; acc1[pb] = f32x4{b1, b1, b1, b1}; }
      // (the first PF fragment sets were requested in the prologue)
#pragma unroll
      for (int ks = 0; ks < EKS1; ++ks) {
        u32x4 a[4];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) a[pb] = *reinterpret_cast<const u32x4*>(x8 + xoff<15>(16 * pb + l16, 4 * ks + q, C::EXROW));
        const u32x4 f0 = wf[ks % PF][0], f1 = wf[ks % PF][1];
        if (ks + PF < EKS1) { wf[ks % PF][0] = frag(w1_lane, EKS1, ks + PF, 0); wf[ks % PF][1] = frag(w1_lane, EKS1, ks + PF, 1); }
        else { wf[ks % PF][0] = frag(wa_lane, KS2, ks + PF - EKS1, 0); wf[ks % PF][1] = frag(wa_lane, KS2, ks + PF - EKS1, 1); }   // expand step ks - 4 -> slot (ks - 4 + 4) % 8
#ifndef OKP_CHAIN_NO_PIN
        __builtin_amdgcn_sched_barrier(0);     // the refill is REQUESTED here, PF steps ahead of its use (hipcc otherwise sinks the loads down to their MFMAs)
#endif
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
          acc0[pb] = H16<T>::mfma16(a[pb], f0, acc0[pb]);
          acc1[pb] = H16<T>::mfma16(a[pb], f1, acc1[pb]);
        }
      }
#pragma unroll
      for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int px = 16 * pb + 4 * q + r;
          if (px < HWe) {
            x2_t v;
            v[0] = (T)acc0[pb][r]; v[1] = (T)acc1[pb][r];
            const int py = px / p.We;
            const int row = (py + 1) * SWe + (px - py * p.We) + 1;
            *reinterpret_cast<x2_t*>(smem + C::OFF_S + xoff<SK>(row, (ch0 * 2) >> 4, SROW) + ((ch0 * 2) & 15)) = v;
          }
        }
    }
    __syncthreads();                                          // s8 complete; x8 is dead: activation buffer 0 (over it) may be written
    char* xn = smem + C::OFF_X;                               // buffer 0 = the first chain module's input
    {
      const float b0 = cba[ch0], b1 = cba[ch0 + 1];
      f32x4 acc0 = {b0, b0, b0, b0}, acc1 = {b1, b1, b1, b1};
      const int o = l16 < HW ? l16 : 0;
      const int oy = o / p.W, ox = o - oy * p.W;
      const int arow = (2 * oy + 1) * SWe + 2 * ox + 1;       // the 1x1 branch samples s8 at (2 oy, 2 ox)
#pragma unroll
      for (int e = 0; e < KS2; ++e) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(smem + C::OFF_S + xoff<SK>(arow, 4 * e + q, SROW));
        const u32x4 f0 = wf[(e + 4) % PF][0], f1 = wf[(e + 4) % PF][1];
        // the slot is free: the first chain module's squeeze step with that slot ((e + 4) % 8; the chain ring keeps step k in slot k % 8)
        wf[(e + 4) % PF][0] = frag(w1_next, KS1, (e + 4) % PF, 0); wf[(e + 4) % PF][1] = frag(w1_next, KS1, (e + 4) % PF, 1);
#ifndef OKP_CHAIN_NO_PIN
        __builtin_amdgcn_sched_barrier(0);     // the refill is REQUESTED here, PF steps ahead of its use (hipcc otherwise sinks the loads down to their MFMAs)
#endif
        acc0 = H16<T>::mfma16(a, f0, acc0);
        acc1 = H16<T>::mfma16(a, f1, acc1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int px = 4 * q + r;
        x2_t v;
        v[0] = (T)fmaxf(acc0[r], 0.f); v[1] = (T)fmaxf(acc1[r], 0.f);
        if (px >= HW) { v[0] = (T)0.f; v[1] = (T)0.f; }
        *reinterpret_cast<x2_t*>(xn + xoff<XK>(px, (ch0 * 2) >> 4, XROW) + ((ch0 * 2) & 15)) = v;
      }
    }
    {
      const int cg = tid % (MID / 8), dpx = tid / (MID / 8);   // 32 channel groups x 16 output pixels
      float v[8];
      {
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(cbd + cg * 8), u1 = *reinterpret_cast<const f32x4*>(cbd + cg * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = u0[e]; v[4 + e] = u1[e]; }
      }
      const int px = dpx < HW ? dpx : 0;
      const int oy = px / p.W, ox = px - oy * p.W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int row = (2 * oy + t / 3) * SWe + 2 * ox + t % 3;     // (2 oy + 1 + dy) * SWe + 2 ox + 1 + dx, dy, dx in -1..1
        const x8_t sv = *reinterpret_cast<const x8_t*>(smem + C::OFF_S + xoff<SK>(row, cg, SROW));
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(cst + t * MID + cg * 8), w1 = *reinterpret_cast<const f32x4*>(cst + t * MID + cg * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = fmaf((float)sv[e], w0[e], v[e]); v[4 + e] = fmaf((float)sv[4 + e], w1[e], v[4 + e]); }
      }
      x8_t out;
#pragma unroll
      for (int e = 0; e < 8; ++e) out[e] = (T)(dpx < HW ? fmaxf(v[e], 0.f) : 0.f);
      *reinterpret_cast<x8_t*>(xn + xoff<XK>(dpx, (MID * 2) / 16 + cg, XROW)) = out;
    }
    __syncthreads();                                          // every reader of s8 is through
    // the chain's squeeze tile (H + 2) x (W + 2) rows lies at the head of the s8 region: clear it, its halo must be zero
    for (int i = tid; i < C::SROWS * (MID / 8); i += NT) *reinterpret_cast<u32x4*>(smem + C::OFF_S + i * 16) = u32x4{0u, 0u, 0u, 0u};
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");        // module 1's constants have landed; its 16 squeeze fragments may stay in flight
    __syncthreads();
  }

  const int m_first = EE ? 1 : 0, m_end = EE ? p.n_modules - 1 : p.n_modules;      // the chain modules proper
  FC_STAMP(true, 6);
  for (int m = m_first; m < m_end; ++m) {
    const OkpFireChainModule mod = p.mod[m];
#ifdef OKP_FIRE_STAMPS
    const bool second = m == m_first + 1;
#endif
    FC_STAMP(second, 0);
    const float* cst = reinterpret_cast<const float*>(smem + C::OFF_C + (m & 1) * C::CONST_BYTES);   // [9][MID] depth-wise weights
    const float* cbd = cst + C::WD_BYTES / 4;              // then bd, b1, ba
    const float* cb1 = cbd + C::B_BYTES / 4;
    const float* cba = cb1 + C::B_BYTES / 4;
    if (m + 1 < p.n_modules) fetch_consts(p.mod[m + 1], (m + 1) & 1, EE && m + 1 == m_end ? C::XMID : MID);     // lands during this module; waited at its end
    // weights in fragment order [wave][block b][k-step][lane][16 B]: one load = 1 KiB contiguous per wave
    const u32x4* w1_lane = static_cast<const u32x4*>(mod.w1) + (size_t)w * 2 * KS1 * 64 + lane;
    const u32x4* wa_lane = static_cast<const u32x4*>(mod.wa) + (size_t)w * 2 * KS2 * 64 + lane;
    const int cg = tid % (MID / 8);                          // depth-wise branch: this thread's 8-channel group
    // does this wave fetch the next module's first squeeze fragments during its expand phase?  (EE: the module behind the last chain
    // module is the exit module, 512 -> 192: six waves' worth of fragments)
    const bool more = m + 1 < p.n_modules && (!(EE && m + 1 == m_end) || w < C::XMID / 32);
    const char* xc = smem + C::OFF_X + cur * NPX * XROW;
    char* xn = smem + C::OFF_X + (cur ^ 1) * NPX * XROW;

    // ---- squeeze: s[px][ch] for the NPX pixel rows, weights streamed PF k-steps ahead ------------------------------
    {
      const float b0 = cb1[ch0], b1 = cb1[ch0 + 1];
      f32x4 acc0[PXB], acc1[PXB];
#pragma unroll
      for (int pb = 0; pb < PXB; ++pb) { acc0[pb] = f32x4{b0, b0, b0, b0}; acc1[pb] = f32x4{b1, b1, b1, b1}; }
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        u32x4 a[PXB];
#pragma unroll
        for (int pb = 0; pb < PXB; ++pb) a[pb] = *reinterpret_cast<const u32x4*>(xc + xoff<XK>(16 * pb + l16, 4 * ks + q, XROW));
        const u32x4 f0 = wf[ks % PF][0], f1 = wf[ks % PF][1];
        if (ks + PF < KS1) { wf[ks % PF][0] = frag(w1_lane, KS1, ks + PF, 0); wf[ks % PF][1] = frag(w1_lane, KS1, ks + PF, 1); }
        else { wf[ks % PF][0] = frag(wa_lane, KS2, ks + PF - KS1, 0); wf[ks % PF][1] = frag(wa_lane, KS2, ks + PF - KS1, 1); }   // expand step ks + PF - KS1
#ifndef OKP_CHAIN_NO_PIN
        __builtin_amdgcn_sched_barrier(0);     // the refill is REQUESTED here, PF steps ahead of its use (hipcc otherwise sinks the loads down to their MFMAs)
#endif
#pragma unroll
        for (int pb = 0; pb < PXB; ++pb) {
          acc0[pb] = H16<T>::mfma16(a[pb], f0, acc0[pb]);
          acc1[pb] = H16<T>::mfma16(a[pb], f1, acc1[pb]);
        }
      }
      // accumulator register r of block pb is pixel 16 pb + 4 q + r; the channel pair is one dword of the squeeze tile's interior
#pragma unroll
      for (int pb = 0; pb < PXB; ++pb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int px = 16 * pb + 4 * q + r;
          if (px < HW) {
            x2_t v;
            v[0] = (T)acc0[pb][r]; v[1] = (T)acc1[pb][r];
            const int row = srow(px);
            *reinterpret_cast<x2_t*>(smem + C::OFF_S + xoff<SK>(row, (ch0 * 2) >> 4, SROW) + ((ch0 * 2) & 15)) = v;
          }
        }
    }
    FC_STAMP(second, 1);
    __syncthreads();
    FC_STAMP(second, 2);

    // ---- expand: y[:, :MID] = relu(Wa s + ba + x[:, :MID]) -> next activation buffer ----------------------------------
    {
      const float b0 = cba[ch0], b1 = cba[ch0 + 1];
      f32x4 acc0[PXB], acc1[PXB];
      int arow[PXB];
#pragma unroll
      for (int pb = 0; pb < PXB; ++pb) {
        acc0[pb] = f32x4{b0, b0, b0, b0}; acc1[pb] = f32x4{b1, b1, b1, b1};
        arow[pb] = srow(16 * pb + l16 < HW ? 16 * pb + l16 : 0);
      }
      const u32x4* w1_next = static_cast<const u32x4*>(p.mod[more ? m + 1 : m].w1) + (size_t)w * 2 * KS1 * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) {
        u32x4 a[PXB];
#pragma unroll
        for (int pb = 0; pb < PXB; ++pb) a[pb] = *reinterpret_cast<const u32x4*>(smem + C::OFF_S + xoff<SK>(arow[pb], 4 * ks + q, SROW));
        const u32x4 f0 = wf[ks % PF][0], f1 = wf[ks % PF][1];
        if (more) { wf[ks % PF][0] = frag(w1_next, KS1, ks, 0); wf[ks % PF][1] = frag(w1_next, KS1, ks, 1); }      // the next module's squeeze step ks
#ifndef OKP_CHAIN_NO_PIN
        __builtin_amdgcn_sched_barrier(0);     // the refill is REQUESTED here, PF steps ahead of its use (hipcc otherwise sinks the loads down to their MFMAs)
#endif
#pragma unroll
        for (int pb = 0; pb < PXB; ++pb) {
          acc0[pb] = H16<T>::mfma16(a[pb], f0, acc0[pb]);
          acc1[pb] = H16<T>::mfma16(a[pb], f1, acc1[pb]);
        }
      }
#pragma unroll
      for (int pb = 0; pb < PXB; ++pb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int px = 16 * pb + 4 * q + r;
          const uint32_t o = xoff<XK>(px, (ch0 * 2) >> 4, XROW) + ((ch0 * 2) & 15);
          const x2_t xv = *reinterpret_cast<const x2_t*>(xc + o);
          x2_t v;
          v[0] = (T)fmaxf(acc0[pb][r] + (float)xv[0], 0.f);
          v[1] = (T)fmaxf(acc1[pb][r] + (float)xv[1], 0.f);
          if (px >= HW) { v[0] = (T)0.f; v[1] = (T)0.f; }
          *reinterpret_cast<x2_t*>(xn + o) = v;
        }
    }

    FC_STAMP(second, 3);
    // ---- depth-wise: y[:, MID:] = relu(dw3x3(s) + bd + x[:, MID:]) -----------------------------------------------------
    // thread = (8-channel group cg, pixel slot); NT / (MID / 8) = 16 pixel slots per pass
    // The NPX / 16 pixels of a thread are worked TOGETHER, tap by tap: a tap's weights are read from LDS once for all of them (pixel by pixel
    // they were re-read for every pixel: 27 LDS reads per pixel, 18 of them weights - 108 per thread and module at 8 x 8), and the pixels'
    // squeeze reads of a tap are in flight together.  Same FMA order per pixel: same bits.
    {
      constexpr int NP = NPX / 16;
      static_assert(NT / (MID / 8) == 16 && NPX % 16 == 0, "pixel slots");
      const int slot = tid / (MID / 8);
      float v[NP][8];
      int base[NP];
      {
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(cbd + cg * 8), u1 = *reinterpret_cast<const f32x4*>(cbd + cg * 8 + 4);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[i][e] = u0[e]; v[i][4 + e] = u1[e]; }
          const int dpx = slot + 16 * i;
          const int px = dpx < HW ? dpx : 0;
          const int py = px / p.W;
          base[i] = py * SW + (px - py * p.W);                  // tap (dy, dx): row base + dy SW + dx  (= (py + 1 + dy - 1) SW + pxx + 1 + dx - 1)
        }
      }
#pragma unroll 1
      for (int t = 0; t < 9; ++t) {                             // (not unrolled: nine taps x NP pixels in one block made hipcc hoist every read and spill 180 registers)
        const int dy = (t * 11) >> 5, dx = t - 3 * dy;          // t / 3, t % 3 for t < 9
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(cst + t * MID + cg * 8), w1 = *reinterpret_cast<const f32x4*>(cst + t * MID + cg * 8 + 4);
        x8_t sv[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) sv[i] = *reinterpret_cast<const x8_t*>(smem + C::OFF_S + xoff<SK>(base[i] + dy * SW + dx, cg, SROW));
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[i][e] = fmaf((float)sv[i][e], w0[e], v[i][e]); v[i][4 + e] = fmaf((float)sv[i][4 + e], w1[e], v[i][4 + e]); }
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int dpx = slot + 16 * i;
        const uint32_t o = xoff<XK>(dpx, (MID * 2) / 16 + cg, XROW);
        const x8_t xv = *reinterpret_cast<const x8_t*>(xc + o);
        x8_t out;
#pragma unroll
        for (int e = 0; e < 8; ++e) out[e] = (T)(dpx < HW ? fmaxf(v[i][e] + (float)xv[e], 0.f) : 0.f);
        *reinterpret_cast<x8_t*>(xn + o) = out;
      }
    }
    // the next module's constants have landed (this wave's share); its first 2 PF squeeze fragments, issued after them, may stay in flight
    if (more) {
      if constexpr (PF == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      static_assert(PF == 8 || PF == 6, "counted wait");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    FC_STAMP(second, 4);
    __syncthreads();
    cur ^= 1;
  }
  FC_STAMP(true, 7);

  // ---- EE: exit module (the last one): 512 -> 192 -> 384, no skip; waves 0..5 own its 192 squeeze / expand channels ----------------
  if constexpr (EE) {
    constexpr int XMID = C::XMID, XKS2 = C::XKS2;
    const OkpFireChainModule mod = p.mod[p.n_modules - 1];
    const float* cst = reinterpret_cast<const float*>(smem + C::OFF_C + ((p.n_modules - 1) & 1) * C::CONST_BYTES);   // [9][192] depth-wise weights
    const float* cbd = cst + C::WD_BYTES / 4;
    const float* cb1 = cbd + C::B_BYTES / 4;
    const float* cba = cb1 + C::B_BYTES / 4;
    const char* xc = smem + C::OFF_X + cur * NPX * XROW;
    char* xn = smem + C::OFF_X + (cur ^ 1) * NPX * XROW;
    const bool act = w < XMID / 32;
    const u32x4* w1_lane = static_cast<const u32x4*>(mod.w1) + (size_t)(act ? w : 0) * 2 * KS1 * 64 + lane;
    const u32x4* wa_lane = static_cast<const u32x4*>(mod.wa) + (size_t)(act ? w : 0) * 2 * XKS2 * 64 + lane;
    if (act) {
      const float b0 = cb1[ch0], b1 = cb1[ch0 + 1];
      f32x4 acc0 = {b0, b0, b0, b0}, acc1 = {b1, b1, b1, b1};
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {                     // squeeze steps 0..7 are in the ring (fetched by the last chain module)
        const u32x4 a = *reinterpret_cast<const u32x4*>(xc + xoff<XK>(l16, 4 * ks + q, XROW));
        const u32x4 f0 = wf[ks % PF][0], f1 = wf[ks % PF][1];
        if (ks + PF < KS1) { wf[ks % PF][0] = frag(w1_lane, KS1, ks + PF, 0); wf[ks % PF][1] = frag(w1_lane, KS1, ks + PF, 1); }
        else if (ks + PF - KS1 < XKS2) { wf[ks % PF][0] = frag(wa_lane, XKS2, ks + PF - KS1, 0); wf[ks % PF][1] = frag(wa_lane, XKS2, ks + PF - KS1, 1); }
#ifndef OKP_CHAIN_NO_PIN
        __builtin_amdgcn_sched_barrier(0);     // the refill is REQUESTED here, PF steps ahead of its use (hipcc otherwise sinks the loads down to their MFMAs)
#endif
        acc0 = H16<T>::mfma16(a, f0, acc0);
        acc1 = H16<T>::mfma16(a, f1, acc1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int px = 4 * q + r;
        if (px < HW) {
          x2_t v;
          v[0] = (T)acc0[r]; v[1] = (T)acc1[r];
          *reinterpret_cast<x2_t*>(smem + C::OFF_S + xoff<SK>(srow(px), (ch0 * 2) >> 4, SROW) + ((ch0 * 2) & 15)) = v;
        }
      }
    }
    __syncthreads();
    if (act) {
      const float b0 = cba[ch0], b1 = cba[ch0 + 1];
      f32x4 acc0 = {b0, b0, b0, b0}, acc1 = {b1, b1, b1, b1};
      const int arow = srow(l16 < HW ? l16 : 0);
#pragma unroll
      for (int e = 0; e < XKS2; ++e) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(smem + C::OFF_S + xoff<SK>(arow, 4 * e + q, SROW));
        acc0 = H16<T>::mfma16(a, wf[e % PF][0], acc0);
        acc1 = H16<T>::mfma16(a, wf[e % PF][1], acc1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int px = 4 * q + r;
        x2_t v;
        v[0] = (T)fmaxf(acc0[r], 0.f); v[1] = (T)fmaxf(acc1[r], 0.f);
        *reinterpret_cast<x2_t*>(xn + xoff<XK>(px, (ch0 * 2) >> 4, XROW) + ((ch0 * 2) & 15)) = v;
      }
    }
    if (tid < 16 * (XMID / 8)) {                              // 24 channel groups x 16 pixels
      const int cg = tid % (XMID / 8), dpx = tid / (XMID / 8);
      float v[8];
      {
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(cbd + cg * 8), u1 = *reinterpret_cast<const f32x4*>(cbd + cg * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = u0[e]; v[4 + e] = u1[e]; }
      }
      const int px = dpx < HW ? dpx : 0;
      const int py = px / p.W, pxx = px - py * p.W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int row = (py + t / 3) * SW + pxx + t % 3;
        const x8_t sv = *reinterpret_cast<const x8_t*>(smem + C::OFF_S + xoff<SK>(row, cg, SROW));
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(cst + t * XMID + cg * 8), w1 = *reinterpret_cast<const f32x4*>(cst + t * XMID + cg * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = fmaf((float)sv[e], w0[e], v[e]); v[4 + e] = fmaf((float)sv[4 + e], w1[e], v[4 + e]); }
      }
      x8_t out;
#pragma unroll
      for (int e = 0; e < 8; ++e) out[e] = (T)fmaxf(v[e], 0.f);
      *reinterpret_cast<x8_t*>(xn + xoff<XK>(dpx, (XMID * 2) / 16 + cg, XROW)) = out;
    }
    __syncthreads();
    cur ^= 1;
  }

  // result -> HBM
  {
    const char* xc = smem + C::OFF_X + cur * NPX * XROW;
    char* dst = static_cast<char*>(p.out) + (size_t)n * HW * p.out_ps * 2;
    constexpr int OUT_CH = EE ? 2 * C::XMID : CIN;
    for (int i = tid; i < HW * (OUT_CH / 8); i += NT) {
      const int px = i / (OUT_CH / 8), c = i % (OUT_CH / 8);
      *reinterpret_cast<u32x4*>(dst + (size_t)px * p.out_ps * 2 + c * 16) = *reinterpret_cast<const u32x4*>(xc + xoff<XK>(px, c, XROW));
    }
  }
}

}  // namespace

// The fragment-order weight copy of a 1x1 plan is built by okp_conv_create (okp_api.hip); a launch only checks that it exists.
int okp_ensure_frags(const okp_conv* plan, hipStream_t) {
  if (plan->frag_dev) return OKP_OK;
  okp_set_error("plan has no fragment-order weights (needs a single-tap, single-source 16-bit plan with cin %% 32 == 0 and cout %% 32 == 0)");
  return OKP_EINVAL;
}


#ifdef OKP_FIRE_STAMPS
static uint32_t* g_chain_dbg = nullptr;
static void chain_stamps_print(const char* what, int nw, hipStream_t stream) {
  if (!getenv("OKP_FIRE_STAMPS_PRINT")) return;
  (void)hipStreamSynchronize(stream);
  uint32_t h[16 * 8 * 8];
  (void)hipMemcpy(h, g_chain_dbg, sizeof(h), hipMemcpyDeviceToHost);
  double sum[5] = {0, 0, 0, 0, 0}, pro = 0, all = 0; int cnt = 0;
  for (int wg = 0; wg < 16; ++wg)
    for (int w = 0; w < nw; ++w) {
      const uint32_t* a = &h[(wg * 8 + w) * 8];
      if (!a[7]) continue;
      sum[0] += a[1] - a[0]; sum[1] += a[2] - a[1]; sum[2] += a[3] - a[2]; sum[3] += a[4] - a[3];
      pro += a[6] - a[5]; all += a[7] - a[5]; ++cnt;
    }
  if (cnt) printf("fire_chain stamps (%s, %d waves): second chain module: squeeze GEMM + s -> LDS %.0f | barrier %.0f | expand GEMM %.0f | depth-wise + constants' wait %.0f clocks;"
                  " entry to first chain module %.0f, entry to behind the last chain module %.0f\n", what, nw, sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, pro / cnt, all / cnt);
}
#define CHAIN_STAMPS_PRINT(what, nw) chain_stamps_print(what, nw, (hipStream_t)stream)
#else
#define CHAIN_STAMPS_PRINT(what, nw) do {} while (0)
#endif

extern "C" int okp_fire_chain_forward(int32_t n_modules, okp_conv* const* squeeze, okp_conv* const* expand,
                                      const float* const* dw_w_dev, const float* const* dw_bias_dev,
                                      int32_t n, const okp_tensor* x, const okp_tensor* out, void* stream) {
  if (!squeeze || !expand || !dw_w_dev || !dw_bias_dev || !x || !out || !x->data || !out->data) { okp_set_error("okp_fire_chain_forward: null argument"); return OKP_EINVAL; }
  if (n_modules < 1 || n_modules > OKP_FIRE_CHAIN_MAX) { okp_set_error("okp_fire_chain_forward: 1..%d modules", OKP_FIRE_CHAIN_MAX); return OKP_EINVAL; }
  if (n < 1) return OKP_OK;
  OkpFireChainParams p;
  std::memset(&p, 0, sizeof(p));
#ifdef OKP_FIRE_STAMPS
  if (!g_chain_dbg) (void)hipMalloc((void**)&g_chain_dbg, 16 * 8 * 8 * 4);
  (void)hipMemsetAsync(g_chain_dbg, 0, 16 * 8 * 8 * 4, (hipStream_t)stream);
  p.dbg = g_chain_dbg;
#endif
  // EE form: module 0 is the stride-2 entry module 384 -> 256 -> 512 (x is the twice larger map), the last one the exit module
  // 512 -> 192 -> 384, at least one fire(512, 512) in between - the whole innermost hourglass level in one launch
  const bool ee = n_modules >= 3 && squeeze[0] && squeeze[n_modules - 1] && squeeze[0]->cin[0] == 384 && squeeze[0]->cout == 256 &&
                  squeeze[n_modules - 1]->cin[0] == 512 && squeeze[n_modules - 1]->cout == 192;
  int cin = 0;
  for (int m = 0; m < n_modules; ++m) {
    okp_conv *sq = squeeze[m], *ex = expand[m];
    if (!sq || !ex || !dw_w_dev[m] || !dw_bias_dev[m]) { okp_set_error("okp_fire_chain_forward: null module %d", m); return OKP_EINVAL; }
    if (!okp_is16(sq->dtype) || ex->dtype != sq->dtype || sq->dtype != squeeze[0]->dtype || sq->n_taps != 1 || ex->n_taps != 1 || sq->n_src != 1 || ex->n_src != 1) {
      okp_set_error("okp_fire_chain_forward: module %d: bf16 / fp16 single-tap 1x1 plans of one type expected", m); return OKP_EINVAL;
    }
    const bool entry = ee && m == 0, exitm = ee && m == n_modules - 1;
    if (!entry && !exitm && cin == 0) cin = sq->cin[0];
    if (entry || exitm) {
      if (ex->cin[0] != sq->cout || ex->cout != sq->cout) { okp_set_error("okp_fire_chain_forward: module %d: expand must be %d -> %d", m, sq->cout, sq->cout); return OKP_EINVAL; }
    } else if (sq->cin[0] != cin || sq->cout * 2 != cin || ex->cin[0] != sq->cout || ex->cout != sq->cout) {
      okp_set_error("okp_fire_chain_forward: module %d is not a %d -> %d -> %d fire module", m, cin, cin / 2, cin); return OKP_EINVAL;
    }
    if (int e = okp_ensure_frags(sq, (hipStream_t)stream)) return e;
    if (int e = okp_ensure_frags(ex, (hipStream_t)stream)) return e;
    p.mod[m].w1 = sq->frag_dev; p.mod[m].w1_cout_pad = sq->cout_pad; p.mod[m].b1 = sq->bias_dev;
    p.mod[m].wa = ex->frag_dev; p.mod[m].wa_cout_pad = ex->cout_pad; p.mod[m].ba = ex->bias_dev;
    p.mod[m].wd = dw_w_dev[m]; p.mod[m].bd = dw_bias_dev[m];
  }
  const bool bf = squeeze[0]->dtype == OKP_BF16;
  if (ee) {
    const int ho = (x->h - 1) / 2 + 1, wo = (x->w - 1) / 2 + 1;
    if (cin != 512 || x->h < 1 || x->w < 1 || x->h > 8 || x->w > 8 || out->h != ho || out->w != wo) {
      okp_set_error("okp_fire_chain_forward: entry/exit form: 384-channel map of at most 8x8 pixels in, %dx%d out, fire(512, 512) modules in between (got %d channels, %dx%d -> %dx%d)",
                    ho, wo, cin, x->h, x->w, out->h, out->w);
      return OKP_EINVAL;
    }
    if (x->pix_stride < 384 || out->pix_stride < 384 || x->pix_stride % 8 || out->pix_stride % 8 || ((uintptr_t)x->data) % 16 || ((uintptr_t)out->data) % 16) {
      okp_set_error("okp_fire_chain_forward: views must be 16-byte aligned with >= 384 channels"); return OKP_EINVAL;
    }
    if ((int64_t)n * x->h * x->w * x->pix_stride * 2 > x->bytes + (int64_t)(x->pix_stride - 384) * 2 ||
        (int64_t)n * ho * wo * out->pix_stride * 2 > out->bytes + (int64_t)(out->pix_stride - 384) * 2) {
      okp_set_error("okp_fire_chain_forward: views too small for %d frames", n); return OKP_EINVAL;
    }
    p.n_modules = n_modules; p.x = x->data; p.out = out->data; p.x_ps = x->pix_stride; p.out_ps = out->pix_stride; p.H = ho; p.W = wo; p.He = x->h; p.We = x->w;
    if (bf) hipLaunchKernelGGL((okp_fire_chain_kernel<__bf16, 512, 1, true>), dim3(n), dim3(ChainCfg<512, 1, true>::NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((okp_fire_chain_kernel<_Float16, 512, 1, true>), dim3(n), dim3(ChainCfg<512, 1, true>::NT), 0, (hipStream_t)stream, p);
    CHAIN_STAMPS_PRINT("512-wide entry / exit form", 8);
    return okp_check_hip(hipGetLastError(), "okp_fire_chain launch");
  }
  const bool small = x->h <= 4 && x->w <= 4;
  if (!((cin == 512 && small) || (cin == 384 && x->h <= 8 && x->w <= 8))) {
    okp_set_error("okp_fire_chain_forward: built for 512-channel chains on <= 4x4 maps and 384-channel chains on <= 8x8 maps (got %d channels, %dx%d)", cin, x->h, x->w);
    return OKP_EINVAL;
  }
  if (x->h != out->h || x->w != out->w || x->h < 1 || x->w < 1) { okp_set_error("okp_fire_chain_forward: same map size in and out"); return OKP_EINVAL; }
  if (x->pix_stride < cin || out->pix_stride < cin || x->pix_stride % 8 || out->pix_stride % 8 || ((uintptr_t)x->data) % 16 || ((uintptr_t)out->data) % 16) {
    okp_set_error("okp_fire_chain_forward: views must be 16-byte aligned with >= %d channels", cin); return OKP_EINVAL;
  }
  if ((int64_t)n * x->h * x->w * x->pix_stride * 2 > x->bytes + (int64_t)(x->pix_stride - cin) * 2 ||
      (int64_t)n * out->h * out->w * out->pix_stride * 2 > out->bytes + (int64_t)(out->pix_stride - cin) * 2) {
    okp_set_error("okp_fire_chain_forward: views too small for %d frames", n); return OKP_EINVAL;
  }
  p.n_modules = n_modules; p.x = x->data; p.out = out->data; p.x_ps = x->pix_stride; p.out_ps = out->pix_stride; p.H = x->h; p.W = x->w;
  if (cin == 512) {
    if (bf) hipLaunchKernelGGL((okp_fire_chain_kernel<__bf16, 512, 1, false>), dim3(n), dim3(ChainCfg<512, 1>::NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((okp_fire_chain_kernel<_Float16, 512, 1, false>), dim3(n), dim3(ChainCfg<512, 1>::NT), 0, (hipStream_t)stream, p);
  } else if (small) {
    if (bf) hipLaunchKernelGGL((okp_fire_chain_kernel<__bf16, 384, 1, false>), dim3(n), dim3(ChainCfg<384, 1>::NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((okp_fire_chain_kernel<_Float16, 384, 1, false>), dim3(n), dim3(ChainCfg<384, 1>::NT), 0, (hipStream_t)stream, p);
  } else {
    if (bf) hipLaunchKernelGGL((okp_fire_chain_kernel<__bf16, 384, 4, false>), dim3(n), dim3(ChainCfg<384, 4>::NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((okp_fire_chain_kernel<_Float16, 384, 4, false>), dim3(n), dim3(ChainCfg<384, 4>::NT), 0, (hipStream_t)stream, p);
  }
  CHAIN_STAMPS_PRINT(cin == 512 ? "512-wide" : "384-wide", cin / 64);
  return okp_check_hip(hipGetLastError(), "okp_fire_chain launch");
}

// Streaming one-launch fire module (bf16 / fp16, gfx950).  At the two high-resolution
// hourglass levels (256 -> 128 -> 256) the unfused module (squeeze launch + expand/depth-wise launch) is bound by
// HBM traffic: x read twice (GEMM input + skip), the squeeze tensor written and re-read.  Here x is read once (+halo,
// mostly L2 hits) and the output written once; the squeeze tile never leaves LDS.
//
//     s   = W1 x + b1                                   squeeze 1x1 (+bn1, no ReLU)      CIN -> MID
//     y_a = relu(Wa s + ba (+ x[:, :MID]))              expand 1x1 (+bn2 half, skip)     MID -> MID
//     y_b = relu(dw3x3(s) * wd + bd (+ x[:, MID:]))     depth-wise 3x3 (+bn2 half, skip) MID -> MID
// (reference: fire_module, corner_net_lite/core/models/CornerNet_Squeeze.py:10-30)
//
// At the low-resolution levels (384 -> 192, 512 -> 256, ...) the same kernel replaces two latency-bound launches by one.
//
// Shape of the kernel (MID/32 waves; MID = 128: 256 threads, ~71 KB LDS -> two workgroups per CU; persistent grid):
//  * A workgroup owns an IH x IW rectangle of output pixels; the squeeze tile is that rectangle plus a one-pixel
//    halo (SH x SW <= 128 pixels).  Halo pixels outside the frame are zero in s (the reference zero-pads s).
//  * Both GEMMs run on 16x16x32 MFMAs with CHANNELS as rows (A operand = weights) and PIXELS as columns (B operand, read
//    from LDS).  Wave w owns 32 channels; row i of block b is channel 32 w + 8 (i >> 2) + 4 b + (i & 3), so lane
//    (column j, row group q) holds in its two accumulator blocks the EIGHT ADJACENT channels 32 w + 8 q .. + 7 of pixel j:
//    16 bytes.  The squeeze result goes to LDS with one ds_write_b128 per 16 pixels, the skip values come and the expand
//    result goes straight to HBM with one 16-byte access per 16 pixels - no transposition (with the pixels as rows a lane
//    held channel PAIRS of four pixels: four times the memory instructions, measured 2-3 us of a 16 us tile).
//  * Squeeze weights stream from L2 in fragment order two k-steps ahead (three rotating fragment sets; the counted waits of the
//    x ring include them), or through a per-wave LDS ring where there is LDS to spare (MID >= 192).  Expand weights are re-fetched per tile.
//  * x streams through a 4-stage LDS ring of 64-byte K-chunks (one MFMA k-step per stage) filled by LDS-DMA; the
//    16-byte chunk position inside a row is rotated by 2*(row>>2) so that the 16x16 fragment reads are conflict-free.
//  * The depth-wise branch reads its 3x3 neighbourhoods from the LDS squeeze tile (runs of 4 pixels share a
//    3 x 6 window), weights in registers for the duration of the phase.
#include <cstdio>
#include <cstring>

#include <cstdlib>

#include "okp_internal.h"

// Timing ablations (wrong results, right timing; experiment builds only): OKP_F2_ABL_NOSTORE drops every output store, OKP_F2_ABL_NOX the
// x requests of every tile but a workgroup's first, OKP_F2_ABL_NOSKIP the skip values, OKP_F2_ABL_NODW the depth-wise branch's MFMAs and reads,
// OKP_F2_ABL_NOWEIGHTS the weight stream.
#ifdef OKP_F2_ABL_NOSTORE
#define F2_STORE(v, rs, off) do { if ((off) == 0x7ffffff0u) __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(off), 0, 0); } while (0)
#else
#define F2_STORE(v, rs, off) __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(off), 0, 0)
#endif


namespace {

constexpr uint32_t kInvalid = 0x80000000u;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int SP = 128;                          // squeeze-tile rows in LDS
constexpr int PBI = 6;                           // interior pixel blocks of 16 (IP <= 96)
constexpr int NST = 4;                           // x ring stages
constexpr int MAXIH = 6;                         // interior rows per tile (depth-wise phase keeps a column's residuals in registers)
// DWM instances: interior rows per tile (16 columns wide).  5: seven squeeze rows of 18 pixels = 126 <= SP.  Four rows (108 squeeze pixels: seven pixel
// blocks instead of eight) divide the 64-, 32- and 16-row maps and give the 512 resident workgroups a whole number of tiles at 64 x 64 (8 each; 5-row
// tiles: 6.5) - measured level at 32 x 32 and 16 x 16 and 5 % slower at 64 x 64 (profiles/r06t_ab_fire2_fixed_geometry.txt): the per-tile fixed
// cost (eight k-step barriers, the phase changes) outweighs the even split
#ifndef OKP_F2_DIH
#define OKP_F2_DIH 5
#endif
constexpr int kDwmIH = OKP_F2_DIH;
static_assert(kDwmIH >= 1 && kDwmIH <= 5, "DWM tile: (rows + 2) x 18 squeeze pixels <= SP");
// bytes per ring stage.  +64: a depth-wise thread takes its skip values from FOUR stages of one ring row (RING_SKIP: stage = channel group / 4);
// with stages a multiple of 256 bytes apart those four 64-byte pieces sit in the same banks (4-way conflict), 64 bytes further each they cover
// the 64 banks once
#ifdef OKP_F2_NO_REMAP
constexpr int XST = SP * 64;
#else
constexpr int XST = SP * 64 + 64;
#endif

template <int MID> struct FireLds {
  static constexpr int OFF_S = 0;                         // [128][MID x 2 B] squeeze tile, 16-B chunks XOR-swizzled by the row
  static constexpr int OFF_X = OFF_S + SP * MID * 2;      // x ring
  static constexpr int OFF_WD = OFF_X + NST * XST;        // [9][MID] fp32 depth-wise weights, then [MID] bias
  static constexpr int OFF_TAB = OFF_WD + 10 * MID * 4;   // interior pixel ip -> byte offset relative to the tile's first pixel: [96] in x, [96] in out
  static constexpr int OFF_MASK = OFF_TAB + 2 * 96 * 4;   // 4 x u32 validity bits of the squeeze pixels
  static constexpr int BYTES = OFF_MASK + 16;
};

// relu(acc + residual) of a lane's eight adjacent channels, rounded to T: packed fp32 adds, round, then the ReLU on the packed 16-bit words
// (bf16 / fp16 read as int16 keep the sign and the order of the positive values: one v_pk_max_i16 per dword instead of a v_max_f32 per value;
// relu(round(x)) == round(relu(x)), so the bits are those of the unpack / add / max / pack form for every non-NaN value)
template <typename T>
__device__ __forceinline__ u32x4 relu_add_pack8(const f32x4& a0, const f32x4& a1, const u32x4& r) {
  u32x4 v;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const f32x4& a = b ? a1 : a0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x2 s2 = f32x2{a[2 * h], a[2 * h + 1]} + f32x2{H16<T>::lo(r[2 * b + h]), H16<T>::hi(r[2 * b + h])};
      v[2 * b + h] = okp_pack2<T>(s2[0], s2[1]);
    }
  }
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  return __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), z));
}

__device__ __forceinline__ int fastdiv(int x, const OkpFastDiv& f) {
  return f.mul ? (int)(__umulhi((uint32_t)x, f.mul) >> f.shift) : x;
}

__device__ __forceinline__ void wait_vm(int n) {   // s_waitcnt vmcnt(n) for a wave-uniform n (the immediate must be a constant)
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
  }
}

#ifdef OKP_FIRE_STAMPS
// Debug build (OKP_EXTRA_CFLAGS=-DOKP_FIRE_STAMPS, printed with OKP_FIRE_STAMPS_PRINT=1; scripts/probe_fire2.py): shader-clock stamps of every
// wave at the phase boundaries of its workgroup's SECOND tile, workgroups 0..15: [wg][wave 8][8] u32.  OKP_FIRE_NOSKIP=1: timing ablation
// without the skip requests (wrong results).
#define F2_STAMP_ANY(i) do { if (lane == 0 && blockIdx.x < 16) { uint64_t t_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); p.dbg[(blockIdx.x * 8 + w) * 8 + (i)] = (uint32_t)t_; } } while (0)
#define F2_STAMP(i) do { if (second && lane == 0 && blockIdx.x < 16) { uint64_t t_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); p.dbg[(blockIdx.x * 8 + w) * 8 + (i)] = (uint32_t)t_; } } while (0)
#else
#define F2_STAMP(i) do {} while (0)
#define F2_STAMP_ANY(i) do {} while (0)
#endif

// DWM: the depth-wise branch on the MATRIX pipe (stride 1, interior tiles 16 pixels wide).  The kernel is bound by vector-ALU issue
// (1 880 VALU instructions per wave and tile beside 176 MFMAs; the depth-wise block alone: 216 packed FMAs + 264 unpacking shifts / masks +
// the epilogue = 716), while the matrix pipe idles three quarters of the time.  A depth-wise 3x3 is, per tap, a DIAGONAL 32 x 32 matrix
// applied to the wave's 32 channels: with the channels-as-rows layout of the two GEMMs the B operand of tap (dy, dx) at pixel p is the very
// fragment the expand GEMM reads, taken from squeeze row p + (dy - 1) SW + (dx - 1), and the A operand of (tap, block b) has ONE non-zero
// element per lane - the tap's weight of the lane's own channel, at a lane-constant position - so it is built from one LDS dword and two
// ANDs.  Nine K-steps of one MFMA per pixel block and accumulator block: 90 MFMAs, 21 fragment reads, no unpacking (the accumulators come
// out as the expand GEMM's do: eight adjacent channels of a pixel per lane, one 16-byte store).  The tap weights are rounded to the
// activation type T on this path (the VALU path multiplies by fp32 weights): one more rounding point of the 16-bit configurations, inside
// their bounds (tests/precision/bounds.py); the products are exact in the fp32 accumulators either way.
template <typename T, int CIN, int MID, int STR, bool DWM = false>
__global__ __launch_bounds__(MID * 2, MID == 128 ? 2 : 1) void okp_fire2_kernel(const OkpFire2Params p) {
  static_assert(STR == 1 || STR == 2, "stride of both branches");
  // DWM at stride 2 (round 6, last): tiles of 3 x 8 OUTPUT pixels, pixel block = 2 rows x 8 columns (lane l16 -> row 2 pb + (l16 >> 3), column
  // l16 & 7; the fourth row does not exist), squeeze tile 7 x 17.  Nothing in the matrix-pipe form needs the 16 pixels of a fragment to be a row:
  // every lane reads its own 16-byte chunk at its own address.  Nine taps x two blocks: 18 fragment reads, 36 MFMAs - where the vector-ALU form
  // walks 7 squeeze rows x 3 positions per thread with half of the column slots idle (8.1 k of a tile's 18.2 k clocks, profiles/r06t_fire2_stamps.txt)
  constexpr bool DWM2 = DWM && STR == 2;
  constexpr int D2H = 3, D2W = 8, D2SW = 2 * (D2W - 1) + 3, D2SH = 2 * (D2H - 1) + 3, D2B = 2;
  static_assert(D2SH * D2SW <= SP && D2B * 16 >= D2H * D2W + 8, "stride-2 matrix-pipe tile");
  constexpr int DIH = kDwmIH;                      // DWM: interior rows per tile, 16 columns wide: (DIH + 2) squeeze rows of DSW = 18 pixels
  constexpr int DSW = 18;
  constexpr int NSB = !DWM ? SP / 16 : STR == 1 ? ((DIH + 2) * DSW + 15) / 16 : (D2SH * D2SW + 15) / 16;   // pixel blocks of the squeeze GEMM (the ring keeps its SP / 16 row blocks)
  constexpr int NW = MID / 32;                     // waves: each owns 32 channels of both GEMMs
  constexpr int NT = 64 * NW;
  constexpr int HALF = MID;
  constexpr int KS1 = CIN / 32;                    // squeeze k-steps
  constexpr int KS2 = MID / 32;                    // expand k-steps
  // Squeeze weights: streamed from L2 in fragment order for every instance (three rotating register sets, or the LDS ring below).  The
  // CIN = 256 instances kept them RESIDENT until round 5 (64 registers per lane for the whole kernel; -DOKP_F2_RESIDENT_W1 builds that
  // form): streaming them frees 40 registers, which the depth-wise phase uses to keep DWB squeeze-row reads in flight before their
  // FMAs - and the kernel no longer sits on 255 registers (the fp16 and stride-2 instances spilled two).  Same bits out; 64 x 64 alone
  // 89 -> 87 us, 32 x 32 31.9 -> 30.9, bf16 step -0.6 % (profiles/r05u_ab_fire2_batch.txt).
#ifdef OKP_F2_RESIDENT_W1
  constexpr bool RES = CIN <= 256;
  constexpr int DWB = 1;
#else
  constexpr bool RES = false;
#ifdef OKP_F2_RESIDENT_WA
  constexpr int DWB = (CIN == 256 && MID == 128) ? OKP_F2_RESIDENT_WA : 1;
#else
  constexpr int DWB = (CIN == 256 && MID == 128) ? 4 : 1;   // (the other instances have no registers to spare: batching spills there)
#endif
#endif
  // Wider inputs stream the squeeze weights.  With >= 192 squeeze channels (one workgroup per CU, LDS to spare) they go through a
  // per-wave LDS ring of DW k-steps filled by LDS-DMA (fragment order: 1 KiB per instruction, read back by the lane that needs it -
  // no barrier, no registers): DW - 1 steps in flight cover the 2-3 us a weight load takes to return under load (hot or cold in L2
  // alike), where the three register sets of the 128-channel instance (two workgroups per CU, no LDS left) cover 2 steps.
  // Expand weights: per tile they were requested behind the squeeze GEMM and waited for in front of the expand GEMM - a weight load takes
  // 2-3 us to return under load, most of the expand phase's 5.3 k clocks (profiles/r05g_fire2_stamps.txt: 48 MFMAs = 768 clocks of pipe).
  // OKP_F2_RESIDENT_WA: the 256 -> 128 instance keeps them in registers for the whole kernel (32 per lane) and gives the depth-wise phase
  // smaller read batches in exchange.
  // (-DOKP_F2_RESWA=1: the matrix-pipe depth-wise 256 -> 128 instance only; 64 x 64 alone 66.5 -> 64.6 us, 32 x 32 and the step level:
  //  profiles/r06t_ab_fire2_prologue_pack.txt; the stride-2 instance loses 7 % with resident expand weights)
#ifndef OKP_F2_RESWA
#define OKP_F2_RESWA 0
#endif
#ifdef OKP_F2_RESIDENT_WA
  constexpr bool RESWA = CIN == 256 && MID == 128;
#else
  constexpr bool RESWA = OKP_F2_RESWA && DWM && CIN == 256 && MID == 128;
#endif
  constexpr bool WRING = !RES && MID >= 192;
  // PIPE (-DOKP_F2_PIPE=1, an experiment that stays in the source): the squeeze k-loop software-pipelined by one step - behind barrier ks come the
  // requests, the fragment reads of step ks and THEN the MFMAs of step ks - 1, which cover the reads' latency (two fragment sets, four rotating
  // weight sets; 234 registers).  Left alone hipcc sinks a step's MFMAs below the next barrier too, but feeds them from two fragment registers
  // (ds_read x 2 -> s_waitcnt -> 4 MFMAs, four times per step).  Measured level to 1.5 % SLOWER (64 x 64 alone 69.9 -> 71.1 us,
  // profiles/r06t_ab_fire2_pipeline_resident_wa.txt): the k-loop does not wait for its fragment reads - the other workgroup of the CU fills the gaps.
#ifndef OKP_F2_PIPE
#define OKP_F2_PIPE 0
#endif
  constexpr bool PIPE = OKP_F2_PIPE && DWM && CIN == 256 && MID == 128 && !RES && !WRING;
  constexpr int NWS = PIPE ? 4 : 3;                // rotating squeeze-weight sets
  constexpr int DW = !WRING ? 0 : MID == 192 ? 5 : 3;
  constexpr int SWM = ((MID / 8) % 16 == 0) ? 15 : 7;   // swizzle key bits (chunks per row must be a multiple of key range)
  constexpr int CG = MID / 8;                      // 8-channel groups of the depth-wise branch; NT / CG = 16 column slots
  constexpr int NDM = (SP / 16 + NW - 1) / NW;     // LDS-DMA instructions per ring step, at most, per wave
  // skip connection (CIN == 2 MID, stride 1): the depth-wise branch adds x[:, MID:], i.e. k-steps KS1/2 .. KS1-1 - with
  // KS1/2 <= NST those are exactly the steps the ring still holds when phase 1 ends: taken from LDS instead of re-read from L2
  constexpr bool RING_SKIP = STR == 1 && CIN == 2 * MID && KS1 / 2 <= NST;
  using L = FireLds<MID>;
  constexpr int OFF_S = L::OFF_S, OFF_X = L::OFF_X, OFF_WD = L::OFF_WD, OFF_TAB = L::OFF_TAB, OFF_MASK = L::OFF_MASK;
  constexpr int OFF_W1 = (L::BYTES + 1023) / 1024 * 1024;   // [DW][wave][block b][1 KiB] squeeze-weight ring (WRING)
  constexpr int LDS_TOTAL = WRING ? OFF_W1 + DW * NW * 2048 : L::BYTES;
  static_assert(NT / CG == 16 && LDS_TOTAL <= 160 * 1024, "thread mapping / LDS");
  __shared__ __attribute__((aligned(16))) char smem[LDS_TOTAL];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, q = lane >> 4;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);

  // ---- once per workgroup: resident squeeze weights, depth-wise constants -> LDS -----------------------------
  // B fragment (column = channel c, k-step ks, k-group q) = 16 bytes at [slice ks/2][row c][64 (ks&1) + 16 q]
  const int chq = 32 * w + 8 * q;                           // this lane's eight channels chq .. chq + 7 (block b, register r: chq + 4 b + r) in both GEMMs
  // weights in fragment order [wave][block b][k-step][lane][16 B] (okp_ensure_frags): one load = 1 KiB contiguous per
  // wave (the packed plan layout would give 16 separate 64-byte segments per load: measured 4x slower to stream)
  const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, NW * 2 * KS1 * 1024, 0x00020000);
  const uint32_t w1_voff = (uint32_t)lane * 16u;
  auto load_w1 = [&](int ks, u32x4 (&dst)[2]) {
#ifdef OKP_F2_ABL_NOWEIGHTS        // timing ablation (wrong results): every k-step multiplies with the weights of k-step 0, requested once per tile
    if (ks != 0) { dst[0] = u32x4{0x3c003c00u, 0u, 0u, 0u}; dst[1] = dst[0]; return; }
#endif
    // (buffer loads: ONE lane offset register + a scalar offset per fragment; as global loads every fragment beyond the 4 KiB immediate range
    //  had its own 64-bit pointer pair, kept in registers across the whole k-loop - a dozen pairs)
#pragma unroll
    for (int b = 0; b < 2; ++b)
      dst[b] = __builtin_amdgcn_raw_buffer_load_b128(rs_w1, (int)w1_voff, (int)(((uint32_t)(w * 2 + b) * KS1 + (uint32_t)ks) * 1024u), 0);
  };
  auto issue_w = [&](int ks) {                                // this wave's two fragment blocks of k-step ks -> ring slot ks % DW
    if constexpr (WRING) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (lds_ptr_t)(smem + OFF_W1 + ((ks % DW) * NW + w) * 2048 + b * 1024), 16,
                                                 (int)(((uint32_t)(w * 2 + b) * KS1 + (uint32_t)ks) * 1024u + (uint32_t)lane * 16u), 0, 0, 0);
    }
  };
  auto first_weights = [&](u32x4 (&f)[RES ? KS1 : NWS][2]) {    // what a tile needs before its k-loop starts
    if constexpr (WRING) {
#pragma unroll
      for (int ks = 0; ks < DW; ++ks) issue_w(ks);
    } else if constexpr (!RES) {
      load_w1(0, f[0]); load_w1(1, f[1]);
    }
  };
  u32x4 w1f[RES ? KS1 : NWS][2];                              // resident: all k-steps; streamed: three (PIPE: four) rotating sets
  if constexpr (RES) {
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) load_w1(ks, w1f[ks]);
  }
  const u32x4* const wa_lane = static_cast<const u32x4*>(p.wa) + (size_t)w * 2 * KS2 * 64 + lane;
  u32x4 waf[2][KS2];                                          // expand weights: resident (RESWA) or re-fetched per tile
  if constexpr (RESWA) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) waf[b][ks] = wa_lane[(size_t)(b * KS2 + ks) * 64];
  }
  const f32x4 b1v0 = *reinterpret_cast<const f32x4*>(p.b1 + chq), b1v1 = *reinterpret_cast<const f32x4*>(p.b1 + chq + 4);
  const f32x4 bav0 = *reinterpret_cast<const f32x4*>(p.ba + chq), bav1 = *reinterpret_cast<const f32x4*>(p.ba + chq + 4);
  // interior pixel block pb, row l16 -> squeeze-tile row (constant over tiles: depends on the tile geometry only)
  uint32_t a_row[PBI], a_key[PBI];
#pragma unroll
  for (int pb = 0; pb < PBI; ++pb) {
    int ip = 16 * pb + l16;
    if (ip >= p.IP) ip = 0;
    const int iy = fastdiv(ip, p.div_iw), ix = ip - iy * p.IW;
    const int sp = (STR * iy + 1) * p.SW + STR * ix + 1;     // the squeeze pixel the 1x1 branch samples
    a_row[pb] = (uint32_t)sp * (MID * 2);
    a_key[pb] = (uint32_t)(sp & SWM);
  }
  // x-ring fragment read: row 16 pb + l16, chunk q sits at position (q + 2 (row >> 2)) & 3 = (q + 2 (l16 >> 2)) & 3
  const uint32_t xfrag_off = (uint32_t)l16 * 64u + (uint32_t)((q + 2 * (l16 >> 2)) & 3) * 16u;
  // squeeze result -> LDS: block pb is tile rows 16 pb + l16; the lane's eight channels are the 16-byte chunk 4 w + q of the row
  const uint32_t s_dst = (uint32_t)(OFF_S + l16 * (MID * 2) + (((4 * w + q) ^ (l16 & SWM)) << 4));
  // LDS-DMA geometry: a ring step is SP/16 = 8 instructions of 16 rows; wave w issues row blocks w, w + NW, ...
  // lane -> (row, position).  The lane FETCHES the 16-byte k-chunk that the read-side rotation expects there.
  constexpr bool UNI = (SP / 16) % NW == 0;                  // every wave issues the same number of instructions
  const int nd = UNI ? NDM : (SP / 16 - w + NW - 1) / NW;    // instructions this wave issues per ring step (wave-uniform)
  int d_sy[NDM], d_sx[NDM];
  uint32_t d_chunk[NDM];
#pragma unroll
  for (int i = 0; i < NDM; ++i) {
    const int row = 16 * (w + NW * i) + (lane >> 2);
    d_sy[i] = fastdiv(row, p.div_sw);
    d_sx[i] = row - d_sy[i] * p.SW;
    if (row >= p.SH * p.SW) d_sy[i] = 1 << 20;              // rows beyond the tile: always out of frame
    d_chunk[i] = (uint32_t)(((lane & 3) - 2 * (row >> 2)) & 3) * 16u;
  }
  // validity of squeeze row `tid` (threads 0..127): same arithmetic, one row per thread
  const int m_sy = tid < p.SH * p.SW ? fastdiv(tid, p.div_sw) : (1 << 20);
  const int m_sx = tid - fastdiv(tid, p.div_sw) * p.SW;

  auto tile_origin = [&](int slot, int& n, int& y0, int& x0) {
    // XCD-aware order (as in okp_igemm_kernel): workgroups are dealt round-robin over the 8 XCDs, so slots with equal
    // slot % 8 share an L2; give each XCD a contiguous range of tiles - vertically adjacent tiles (tile +- tiles_x) then
    // find each other's halo rows in that L2 instead of fetching them into two.  Bijective for any tile count.
    const int xq = p.n_tiles >> 3, xr = p.n_tiles & 7, xcd = slot & 7;
    const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (slot >> 3);
    n = fastdiv(tile, p.div_tiles_frame);
    const int trem = tile - n * p.tiles_y * p.tiles_x;
    const int ty = fastdiv(trem, p.div_tiles_x);
    y0 = ty * p.IH;                                          // OUTPUT coordinates of interior pixel (0, 0)
    x0 = (trem - ty * p.tiles_x) * p.IW;
  };
  uint32_t d_off[NDM];
  auto tile_setup = [&](int tile) {                          // DMA source offsets + validity bits of `tile`
    int n, y0, x0;
    tile_origin(tile, n, y0, x0);
#pragma unroll
    for (int i = 0; i < NDM; ++i) {
      const int y = STR * y0 - 1 + d_sy[i], x = STR * x0 - 1 + d_sx[i];
      const bool ok = y >= 0 && y < p.H && x >= 0 && x < p.W;
      d_off[i] = ok ? (uint32_t)(((long)n * p.H + y) * p.W + x) * (uint32_t)(p.x_ps * 2) + d_chunk[i] : kInvalid;
    }
    if (tid < SP) {
      const int y = STR * y0 - 1 + m_sy, x = STR * x0 - 1 + m_sx;
      const unsigned long long m = __ballot(y >= 0 && y < p.H && x >= 0 && x < p.W);
      if (lane == 0) {
        uint32_t* mk = reinterpret_cast<uint32_t*>(smem + OFF_MASK);
        mk[2 * w] = (uint32_t)m;
        mk[2 * w + 1] = (uint32_t)(m >> 32);
      }
    }
  };
#ifdef OKP_F2_ABL_NOX
  bool abl_started = false;
#endif
  auto issue_x = [&](int ks, int stage) {
#ifdef OKP_F2_ABL_NOX
    if (abl_started) return;
#endif
#pragma unroll
    for (int i = 0; i < NDM; ++i)
      if (UNI || i < nd)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(smem + OFF_X + stage * XST + (w + NW * i) * 1024), 16,
                                                 (int)(d_off[i] == kInvalid ? kInvalid : d_off[i] + (uint32_t)ks * 64u), 0, 0, 0);
  };

  int tile = blockIdx.x;
  if (tile >= p.n_tiles) return;
  F2_STAMP_ANY(5);
  // the first tile's requests go out BEFORE the per-workgroup constants are fetched: both are cold reads (2-3 us each right behind a launch, when
  // every workgroup asks at once), and one behind the other they cost every launch that time twice - 30 launches per network pass
  tile_setup(tile);
  first_weights(w1f);
#pragma unroll
  for (int ks = 0; ks < NST - 1; ++ks) issue_x(ks, ks);
  // ---- once per workgroup: depth-wise constants and the pixel tables -> LDS (first read behind the k-loop's barriers) ----
  {
    // 10 HALF values = 5 per thread (NT = 2 HALF), all five loads in flight before the first is used (as a loop hipcc waited for each by itself:
    // five cold-read latencies in front of every launch's first tile - 6.8 k of the 35 k clocks of a 32 x 32 launch, profiles/r06t_fire2_stamps.txt)
    static_assert(10 * HALF == 5 * NT, "depth-wise constants per thread");
    float cv[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int i = tid + k * NT;
      cv[k] = *(i < 9 * HALF ? p.wd + i : p.bd + (i - 9 * HALF));
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int i = tid + k * NT;
      if constexpr (DWM) {
        // [tap][channel]: the weight rounded to T, in the half of the dword its channel's parity selects (the A operand's element), bias as fp32
        const uint32_t wq = (i & 1) ? okp_pack2<T>(0.f, cv[k]) : okp_pack2<T>(cv[k], 0.f);
        reinterpret_cast<uint32_t*>(smem + OFF_WD)[i] = i < 9 * HALF ? wq : __builtin_bit_cast(uint32_t, cv[k]);
      } else {
        reinterpret_cast<float*>(smem + OFF_WD)[i] = cv[k];
      }
    }
  }
  if (tid < 16 * PBI) {
    const int iy = fastdiv(tid, p.div_iw), ix = tid - iy * p.IW;
    // (the x table serves the skip connection: stride 1, where input and output geometry coincide)
    reinterpret_cast<uint32_t*>(smem + OFF_TAB)[tid] = tid < p.IP ? (uint32_t)(iy * p.W + ix) * (uint32_t)(p.x_ps * 2) : kInvalid;
    reinterpret_cast<uint32_t*>(smem + OFF_TAB)[96 + tid] = tid < p.IP ? (uint32_t)(iy * p.Wo + ix) * (uint32_t)(p.out_ps * 2) : kInvalid;
  }

  __syncthreads();
  F2_STAMP_ANY(6);

  for (; tile < p.n_tiles; tile += gridDim.x) {
    int n, y0, x0;
    tile_origin(tile, n, y0, x0);
#ifdef OKP_FIRE_STAMPS
    const bool second = tile == (int)(blockIdx.x + gridDim.x);
#endif
    F2_STAMP(0);
    // opaque copies: the per-(block, register) pixel arithmetic below is tile-invariant, and hoisting ~100 such
    // values out of the tile loop costs more registers than recomputing them (a multiply-high each)
    int qt = q, tidt = tid;
    asm volatile("" : "+v"(qt), "+v"(tidt));
    // depth-wise phase: thread -> (8-channel group cg, column slot).  ds_read_b128 is served in groups of 16 lanes that are NOT 16 consecutive
    // lanes - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32 (docs/HARDWARE_NOTES.md) - so with cg = tid % 16 a group read half of one squeeze
    // row and half of the next, whose swizzle keys (row & 15) differ in the high chunk bits one time in four: 2-way conflicts, 23 % of the
    // kernel's LDS cycles (SQ_LDS_BANK_CONFLICT, profiles/r05v_sq_counters.json).  With 16 channel groups (MID = 128) the lanes of a hardware
    // group are given ONE row: lane l of a half-wave belongs to group parity(l bits 2, 3, 4), and (l & 3) | ((l >> 1) & 12) numbers its lanes.
#ifdef OKP_F2_NO_REMAP
    const int dw_cg = tidt % CG, dw_slot = tidt / CG;
#else
    const int dw_l = tidt & 31;
    const int dw_cg = CG == 16 ? ((dw_l & 3) | ((dw_l >> 1) & 12)) : tidt % CG;
    const int dw_slot = CG == 16 ? 4 * (tidt >> 6) + 2 * ((tidt >> 5) & 1) + (((dw_l >> 2) ^ (dw_l >> 3) ^ (dw_l >> 4)) & 1) : tidt / CG;
#endif
    // phase 2b residuals of column `ix` (thread = 8-channel group cg, column slot): one 16-byte load per output row
    auto load_col_residuals = [&](int ix, u32x4 (&rr)[MAXIH], uint32_t (&oo)[MAXIH]) {
      const int cg = dw_cg;
      const int ox = x0 + ix;
      const uint32_t pix = (uint32_t)(((long)n * p.Ho + y0) * p.Wo + ox);
      uint32_t xo = pix * (uint32_t)(p.x_ps * 2) + (uint32_t)(HALF + cg * 8) * 2u, oof = pix * (uint32_t)(p.out_ps * 2) + (uint32_t)(HALF + cg * 8) * 2u;
      const bool col_ok = ix < p.IW && ox < p.Wo;
#pragma unroll
      for (int iy = 0; iy < MAXIH; ++iy) {
        const bool ok = col_ok && iy < p.IH && y0 + iy < p.Ho;
        oo[iy] = ok ? oof : kInvalid;
        rr[iy] = u32x4{0u, 0u, 0u, 0u};
        if (p.skip) {
          if constexpr (RING_SKIP) {
            const int row = (iy + 1) * p.SW + (ix < p.IW ? ix : 0) + 1;                   // squeeze-tile row of interior pixel (iy, ix)
            rr[iy] = *reinterpret_cast<const u32x4*>(smem + OFF_X + ((KS1 / 2 + (cg >> 2)) % NST) * XST + row * 64 + (((cg & 3) + 2 * (row >> 2)) & 3) * 16);
          } else {
            rr[iy] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(ok ? xo : kInvalid), 0, 0);
          }
        }
        xo += (uint32_t)(p.W * p.x_ps * 2);
        oof += (uint32_t)(p.Wo * p.out_ps * 2);
      }
    };
    uint32_t o_off[PBI];                                    // phase 2a: output offset and skip values of this lane's pixel of block pb
    u32x4 r_raw[PBI];
    u32x4 rr[MAXIH];                                        // phase 2b: residuals / output offsets of this thread's first column
    uint32_t oo[MAXIH];

    // ---- phase 1: s = W1 x + b1 on the halo'd tile, x through the LDS ring -------------------------------------
    // The first NST-1 ring steps of this tile were issued during the previous tile's phase 2 (or above).
    f32x4 acc[NSB][2];
#pragma unroll
    for (int pb = 0; pb < NSB; ++pb) {
      acc[pb][0] = b1v0;
      acc[pb][1] = b1v1;
    }
    if constexpr (PIPE) {
      // at the top of step ks >= 3 the queue holds, behind x(ks): w(ks), x(ks + 1), w(ks + 1), x(ks + 2) - all of them may stay in flight
      // (the register loads of the weights are waited for by the compiler where their MFMAs are)
      u32x4 a[2][NSB];
#pragma unroll
      for (int ks = 0; ks <= KS1; ++ks) {
        if (ks < KS1) {
          if (ks == 0) wait_vm(0);
          else if (ks >= 3) wait_vm(2 + (ks + 1 < KS1 ? nd + 2 : 0) + (ks + 2 < KS1 ? nd : 0));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (ks + 2 < KS1) load_w1(ks + 2, w1f[(ks + 2) % NWS]);
          if (ks + NST - 1 < KS1) issue_x(ks + NST - 1, (ks + NST - 1) % NST);
          const char* st = smem + OFF_X + (ks % NST) * XST + xfrag_off;
#pragma unroll
          for (int pb = 0; pb < NSB; ++pb) a[ks & 1][pb] = *reinterpret_cast<const u32x4*>(st + pb * 1024);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (ks > 0) {
#pragma unroll
          for (int pb = 0; pb < NSB; ++pb)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              acc[pb][b] = H16<T>::mfma16(w1f[(ks - 1) % NWS][b], a[(ks - 1) & 1][pb], acc[pb][b]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      // ks == 0: everything older (the previous tile's stores, the prefetched steps and weight sets) must be done,
      // because loads and stores share the counter.  Later: ring steps ks+1, ks+2 may stay in flight (nd LDS-DMA each)
      // and, when the squeeze weights are streamed, the weight set of step ks+1 issued between them.
      // WAR on the ring: the barrier below frees stage (ks-1) % NST for the LDS-DMA issued right behind it, so every
      // wave's fragment reads of step ks-1 must have RETURNED before it arrives: the explicit lgkmcnt(0).  The k-loop is
      // fully unrolled and s_barrier orders memory operations only: without that wait hipcc sinks step ks-1's MFMAs - and
      // the lgkmcnt wait in front of them - below this barrier, the reads are still queued in the LDS pipe (8 waves x 8
      // ds_read_b128) when the refill of their stage lands, and the squeeze tile picks up the next k-chunk's bytes.  At N=64
      // (several tiles per workgroup, loaded memory system) that happened in most launches (round-1 bug, found by
      // scripts/probe_kernel_determinism.py / probe_fire2_race.py).  The MFMAs themselves may still sink below the barrier
      // and overlap the next step's reads (pinning them with sched_barrier cost 3 %).
      // (WRING: step j issues the weights of step j - 1 + DW into the slot step j - 1 has just read, then ring step j + 3: behind x(ks)
      //  come w(ks - 3 + DW) - needed itself when DW == 3 -, x(ks + 1), w(ks - 2 + DW), x(ks + 2))
      if (ks == 0) wait_vm(0);
      else if constexpr (WRING) {
        auto nw = [](int j) { return j >= 1 && j - 1 + DW < KS1 ? 2 : 0; };
        auto nx = [](int j) { return j >= 0 && j + NST - 1 < KS1 ? 1 : 0; };
        wait_vm((DW >= 4 ? nw(ks - 2) : 0) + nw(ks - 1) + (nx(ks - 2) + nx(ks - 1)) * nd);
      } else if (ks >= (RES ? NST - 1 : 2))
        wait_vm((ks + 1 < KS1 ? nd + (RES ? 0 : 2) : 0) + (ks + 2 < KS1 ? nd : 0));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if constexpr (WRING) { if (ks >= 1 && ks - 1 + DW < KS1) issue_w(ks - 1 + DW); }
      else if constexpr (!RES) { if (ks + 2 < KS1) load_w1(ks + 2, w1f[(ks + 2) % 3]); }
      if (ks + NST - 1 < KS1) issue_x(ks + NST - 1, (ks + NST - 1) % NST);
      const char* st = smem + OFF_X + (ks % NST) * XST + xfrag_off;
      u32x4 a[NSB], wr[2] = {};
      if constexpr (WRING) {
#pragma unroll
        for (int b = 0; b < 2; ++b) wr[b] = *reinterpret_cast<const u32x4*>(smem + OFF_W1 + ((ks % DW) * NW + w) * 2048 + b * 1024 + lane * 16);
      }
#pragma unroll
      for (int pb = 0; pb < NSB; ++pb) a[pb] = *reinterpret_cast<const u32x4*>(st + pb * 1024);
#ifdef OKP_F2_PIN_READS
      __builtin_amdgcn_sched_barrier(0);       // every fragment read of the step in flight before its first MFMA (hipcc otherwise issues them two at a time)
#endif
#pragma unroll
      for (int pb = 0; pb < NSB; ++pb)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[pb][b] = H16<T>::mfma16(WRING ? wr[b] : w1f[RES ? ks : ks % 3][b], a[pb], acc[pb][b]);
    }
    F2_STAMP(1);
#ifdef OKP_F2_ABL_NOX
    abl_started = true;
#endif
    // expand weights for this tile (dead after phase 2a): issue now, consumed after the barrier
    if constexpr (!RESWA) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
#ifdef OKP_F2_ABL_NOWEIGHTS
          if (b + ks != 0) { waf[b][ks] = u32x4{0x3c003c00u, 0u, 0u, 0u}; continue; }
#endif
          waf[b][ks] = wa_lane[(size_t)(b * KS2 + ks) * 64];
        }
    }
    // s -> LDS (zero outside the frame: the reference zero-pads the squeeze output)
    {
      const uint32_t* mk = reinterpret_cast<const uint32_t*>(smem + OFF_MASK);
      int l16t = l16;
      asm volatile("" : "+v"(l16t));
      uint32_t mq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) mq[i] = mk[i] >> l16t;
#pragma unroll
      for (int pb = 0; pb < NSB; ++pb) {
        const bool ok = (mq[pb >> 1] >> (16 * (pb & 1))) & 1u;
        u32x4 v;
#pragma unroll
        for (int b = 0; b < 2; ++b) {          // (the zero of a pixel outside the frame is selected on the packed words: four selects, not eight)
          const uint32_t p0 = okp_pack2<T>(acc[pb][b][0], acc[pb][b][1]), p1 = okp_pack2<T>(acc[pb][b][2], acc[pb][b][3]);
          v[2 * b] = ok ? p0 : 0u;
          v[2 * b + 1] = ok ? p1 : 0u;
        }
        *reinterpret_cast<u32x4*>(smem + s_dst + pb * 16 * (MID * 2)) = v;
      }
    }
    __syncthreads();
    F2_STAMP(2);
    // ---- phase 2a: y_a = relu(Wa s + ba (+x)) on the interior pixels, whole lines straight to HBM --------------
    uint32_t dwm_o[DIH];                                    // DWM: output offset of this lane's pixel (row iy, column l16), channels chq ..
    uint32_t d2_o[D2B];                                     // DWM2: the same for (block pb, lane)
    if constexpr (DWM2) {
      // pixel (row 2 pb + (l16 >> 3), column l16 & 7): offsets = lane constant + wave-uniform block term; validity per lane (no skip at stride 2)
      int l16t = l16;
      asm volatile("" : "+v"(l16t));
      const int liy = l16t >> 3, lix = l16t & 7;
      const uint32_t pix0 = (uint32_t)(((long)n * p.Ho + y0) * p.Wo + x0);
      const uint32_t lane_o = (uint32_t)(liy * p.Wo + lix) * (uint32_t)(p.out_ps * 2) + (uint32_t)chq * 2u;
      const bool col_ok = lix < p.IW && x0 + lix < p.Wo;
      f32x4 ac2[D2B][2];
#pragma unroll
      for (int pb = 0; pb < D2B; ++pb) {
        const int iy = 2 * pb + liy;
        const bool ok = col_ok && iy < p.IH && y0 + iy < p.Ho;
        d2_o[pb] = ok ? lane_o + (pix0 + (uint32_t)(2 * pb * p.Wo)) * (uint32_t)(p.out_ps * 2) : kInvalid;
        ac2[pb][0] = bav0;
        ac2[pb][1] = bav1;
      }
      // the 1x1 branch samples the squeeze tile at (2 iy + 1, 2 ix + 1): row (4 pb + 2 liy + 1) D2SW + 2 lix + 1
      const uint32_t sp_lane = (uint32_t)((2 * liy + 1) * D2SW + 2 * lix + 1);
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) {
        u32x4 a[D2B];
#pragma unroll
        for (int pb = 0; pb < D2B; ++pb) {
          const uint32_t sp = sp_lane + (uint32_t)(4 * pb * D2SW);
          a[pb] = *reinterpret_cast<const u32x4*>(smem + OFF_S + sp * (uint32_t)(MID * 2) + (((uint32_t)(4 * ks + qt) ^ (sp & (uint32_t)SWM)) << 4));
        }
#pragma unroll
        for (int pb = 0; pb < D2B; ++pb)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            ac2[pb][b] = H16<T>::mfma16(waf[b][ks], a[pb], ac2[pb][b]);
      }
#pragma unroll
      for (int pb = 0; pb < D2B; ++pb) {
        const u32x4 v = relu_add_pack8<T>(ac2[pb][0], ac2[pb][1], u32x4{0u, 0u, 0u, 0u});
        F2_STORE(v, rs_o, d2_o[pb]);
      }
    } else if constexpr (DWM) {
      // Fixed geometry (pixel block = interior row, 16 columns, map width a multiple of 16): a pixel's offsets are a lane constant plus a
      // wave-uniform row term, the only edge is the map's last row (wave-uniform) - no table reads, no per-block branches, and with every
      // trip count known the fragment reads of a k-step are all in flight before its MFMAs (the general form below has a tile-shape
      // condition per block: ds_read -> s_waitcnt -> 2 MFMAs chains, 5.3 k clocks for 768 of matrix pipe: profiles/r05g_fire2_stamps.txt)
      int l16t = l16;
      asm volatile("" : "+v"(l16t));
      const uint32_t pix0 = (uint32_t)(((long)n * p.Ho + y0) * p.Wo + x0);
      const uint32_t lane_x = (uint32_t)l16t * (uint32_t)(p.x_ps * 2) + (uint32_t)chq * 2u, lane_o = (uint32_t)l16t * (uint32_t)(p.out_ps * 2) + (uint32_t)chq * 2u;
      u32x4 rsk[DIH];
      // (branch-free: a row outside the map - or a module without skip - turns into an offset beyond every view: the load returns zeros, the store is dropped)
      const uint32_t rowx = (uint32_t)(p.W * p.x_ps * 2), rowo = (uint32_t)(p.Wo * p.out_ps * 2), noskip = p.skip ? 0u : kInvalid;
      uint32_t sx = pix0 * (uint32_t)(p.x_ps * 2), so = pix0 * (uint32_t)(p.out_ps * 2);
#pragma unroll
      for (int iy = 0; iy < DIH; ++iy) {
        const uint32_t bad = (iy < p.IH && y0 + iy < p.Ho) ? 0u : kInvalid;                 // wave-uniform (valid offsets are < 2 GiB)
        dwm_o[iy] = lane_o + (so | bad);
        rsk[iy] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(lane_x + (sx | bad | noskip)), 0, 0);
        sx += rowx;
        so += rowo;
      }
      f32x4 ac2[DIH][2];
#pragma unroll
      for (int iy = 0; iy < DIH; ++iy) {
        ac2[iy][0] = bav0;
        ac2[iy][1] = bav1;
      }
      // squeeze pixel of (iy, l16): row (iy + 1) DSW + 1 + l16 of the LDS tile
      const uint32_t s_lane = (uint32_t)OFF_S + (uint32_t)(DSW + 1 + l16t) * (uint32_t)(MID * 2);
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) {
        u32x4 a[DIH];
#pragma unroll
        for (int iy = 0; iy < DIH; ++iy) {
          const uint32_t key = (uint32_t)(l16t + (iy + 1) * DSW + 1) & (uint32_t)SWM;
          a[iy] = *reinterpret_cast<const u32x4*>(smem + s_lane + (uint32_t)(iy * DSW * MID * 2) + (((uint32_t)(4 * ks + qt) ^ key) << 4));
        }
#pragma unroll
        for (int iy = 0; iy < DIH; ++iy)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            ac2[iy][b] = H16<T>::mfma16(waf[b][ks], a[iy], ac2[iy][b]);
      }
#pragma unroll
      for (int iy = 0; iy < DIH; ++iy) {
        const u32x4 v = relu_add_pack8<T>(ac2[iy][0], ac2[iy][1], rsk[iy]);
        F2_STORE(v, rs_o, dwm_o[iy]);
      }
    } else {
      // residuals first: their latency hides behind the MFMAs.  Interior tiles take the byte offsets of their
      // pixels from the per-workgroup table (one 16-byte LDS read per block); edge tiles do the arithmetic.
      const bool full = y0 + p.IH <= p.Ho && x0 + p.IW <= p.Wo;
      const uint32_t pix0 = (uint32_t)(((long)n * p.Ho + y0) * p.Wo + x0);
      const uint32_t xb = pix0 * (uint32_t)(p.x_ps * 2) + (uint32_t)chq * 2u, ob = pix0 * (uint32_t)(p.out_ps * 2) + (uint32_t)chq * 2u;
      int l16t = l16;
      asm volatile("" : "+v"(l16t));
#pragma unroll
      for (int pb = 0; pb < PBI; ++pb) {
        uint32_t xr = kInvalid, orr = kInvalid;
        if (16 * pb < p.IP) {
          const int ip = 16 * pb + l16t;
          xr = *reinterpret_cast<const uint32_t*>(smem + OFF_TAB + ip * 4);
          orr = *reinterpret_cast<const uint32_t*>(smem + OFF_TAB + (96 + ip) * 4);
          if (!full) {
            const int iy = fastdiv(ip, p.div_iw), ix = ip - iy * p.IW;
            if (y0 + iy >= p.Ho || x0 + ix >= p.Wo) { xr = kInvalid; orr = kInvalid; }
          }
        }
        o_off[pb] = orr == kInvalid ? kInvalid : ob + orr;
        r_raw[pb] = u32x4{0u, 0u, 0u, 0u};
        if (p.skip && 16 * pb < p.IP) r_raw[pb] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(xr == kInvalid ? kInvalid : xb + xr), 0, 0);
      }
      f32x4 ac2[PBI][2];
#pragma unroll
      for (int pb = 0; pb < PBI; ++pb) {
        ac2[pb][0] = bav0;
        ac2[pb][1] = bav1;
      }
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) {
#pragma unroll
        for (int pb = 0; pb < PBI; ++pb) {
          if (16 * pb < p.IP) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(smem + OFF_S + a_row[pb] + (((uint32_t)(4 * ks + qt) ^ a_key[pb]) << 4));
#pragma unroll
            for (int b = 0; b < 2; ++b)
              ac2[pb][b] = H16<T>::mfma16(waf[b][ks], a, ac2[pb][b]);
          }
        }
      }
#pragma unroll
      for (int pb = 0; pb < PBI; ++pb) {
        if (16 * pb < p.IP) {
          const u32x4 v = relu_add_pack8<T>(ac2[pb][0], ac2[pb][1], r_raw[pb]);
          F2_STORE(v, rs_o, o_off[pb]);
        }
      }
    }
    asm volatile("" ::: "memory");             // keep the stores here (the scheduler otherwise sinks them below phase 2b)
    F2_STAMP(3);

    // ---- phase 2b: y_b = relu(dw3x3(s) + bd (+x)) from the LDS squeeze tile -------------------------------------
    if constexpr (DWM2) {
      int l16t = l16;
      asm volatile("" : "+v"(l16t));
      const int liy = l16t >> 3, lix = l16t & 7;
      {
        const int next = tile + gridDim.x;                   // (the ring is free: every wave is behind the barrier that follows the squeeze tile's writes)
        if (next < p.n_tiles) {
          tile_setup(next);
          first_weights(w1f);
#pragma unroll
          for (int ks = 0; ks < NST - 1; ++ks) issue_x(ks, ks);
        }
      }
      const float* const bl = reinterpret_cast<const float*>(smem + OFF_WD) + 9 * HALF + chq;
      const f32x4 bdv0 = *reinterpret_cast<const f32x4*>(bl), bdv1 = *reinterpret_cast<const f32x4*>(bl + 4);
      f32x4 acd[D2B][2];
#pragma unroll
      for (int pb = 0; pb < D2B; ++pb) { acd[pb][0] = bdv0; acd[pb][1] = bdv1; }
      const uint32_t am = (qt == (l16t >> 2)) ? 0xffffffffu : 0u;                            // A operand: as in the stride-1 form below
      const uint32_t m_lo = (l16t & 2) ? 0u : am, m_hi = (l16t & 2) ? am : 0u;
      const uint32_t wt_lane = (uint32_t)OFF_WD + (uint32_t)(chq + (l16t & 3)) * 4u;
      const uint32_t cx = (uint32_t)(4 * w) + (uint32_t)qt;
      const uint32_t sp_lane = (uint32_t)(2 * liy * D2SW + 2 * lix);                         // tap (dy, dx) of pixel (2 pb + liy, lix): squeeze row (4 pb + 2 liy + dy) D2SW + 2 lix + dx
      u32x4 a0 = {0u, 0u, 0u, 0u}, a1 = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
        const uint32_t v0 = *reinterpret_cast<const uint32_t*>(smem + wt_lane + (tap * HALF) * 4);
        const uint32_t v1 = *reinterpret_cast<const uint32_t*>(smem + wt_lane + (tap * HALF + 4) * 4);
        a0[0] = v0 & m_lo; a0[1] = v0 & m_hi;
        a1[2] = v1 & m_lo; a1[3] = v1 & m_hi;
        u32x4 bf[D2B];
#pragma unroll
        for (int pb = 0; pb < D2B; ++pb) {
          const uint32_t sp = sp_lane + (uint32_t)((4 * pb + dy) * D2SW + dx);
          bf[pb] = *reinterpret_cast<const u32x4*>(smem + OFF_S + sp * (uint32_t)(MID * 2) + ((cx ^ (sp & (uint32_t)SWM)) << 4));
        }
#pragma unroll
        for (int pb = 0; pb < D2B; ++pb) {
          acd[pb][0] = H16<T>::mfma16(a0, bf[pb], acd[pb][0]);
          acd[pb][1] = H16<T>::mfma16(a1, bf[pb], acd[pb][1]);
        }
      }
#pragma unroll
      for (int pb = 0; pb < D2B; ++pb) {
        const u32x4 o = relu_add_pack8<T>(acd[pb][0], acd[pb][1], u32x4{0u, 0u, 0u, 0u});
        F2_STORE(o, rs_o, d2_o[pb] == kInvalid ? kInvalid : d2_o[pb] + (uint32_t)(HALF * 2));
      }
    } else if constexpr (DWM) {
      // On the matrix pipe (see the kernel's head).  Lane (l16 = pixel column ix = row i of the A operand, q): channels chq .. chq + 7.
      int l16t = l16;
      asm volatile("" : "+v"(l16t));
      u32x4 rr2[DIH];
      uint32_t oo2[DIH];
#pragma unroll
      for (int iy = 0; iy < DIH; ++iy) {
        oo2[iy] = dwm_o[iy] + (uint32_t)(HALF * 2);          // the same pixel's second half of the channels
        rr2[iy] = u32x4{0u, 0u, 0u, 0u};
        if (p.skip) {
          if constexpr (RING_SKIP) {      // x[:, MID + chq ..] = k-step KS1/2 + w of the ring, chunk q of squeeze row (iy + 1, ix + 1); rows beyond the tile: never stored
            const uint32_t row = (uint32_t)((iy + 1) * DSW + 1) + (uint32_t)l16t;
            rr2[iy] = *reinterpret_cast<const u32x4*>(smem + OFF_X + ((KS1 / 2 + w) % NST) * XST + row * 64u + (((uint32_t)qt + 2u * (row >> 2)) & 3u) * 16u);
          } else {
            const bool ok = iy < p.IH && y0 + iy < p.Ho;
            const uint32_t pix0 = (uint32_t)(((long)n * p.Ho + y0) * p.Wo + x0);
            const uint32_t sx = ok ? (pix0 + (uint32_t)(iy * p.W)) * (uint32_t)(p.x_ps * 2) : kInvalid;
            rr2[iy] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)((uint32_t)l16t * (uint32_t)(p.x_ps * 2) + (uint32_t)(HALF + chq) * 2u + sx), 0, 0);
          }
        }
      }
      if constexpr (RING_SKIP) {
        if (p.skip) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __syncthreads(); }   // every thread has its skip values: the ring is free
      }
      {
        const int next = tile + gridDim.x;
        if (next < p.n_tiles) {
          tile_setup(next);                                  // (the validity bits are next read behind >= 8 barriers)
          first_weights(w1f);
#pragma unroll
          for (int ks = 0; ks < NST - 1; ++ks) issue_x(ks, ks);
        }
      }
      const float* const bl = reinterpret_cast<const float*>(smem + OFF_WD) + 9 * HALF + chq;
      const f32x4 bdv0 = *reinterpret_cast<const f32x4*>(bl), bdv1 = *reinterpret_cast<const f32x4*>(bl + 4);
      f32x4 acd[DIH][2];
#pragma unroll
      for (int iy = 0; iy < DIH; ++iy) { acd[iy][0] = bdv0; acd[iy][1] = bdv1; }
      // A operand of (tap, block b): lane (i = l16, q) is non-zero only where q == i >> 2, with the weight of channel chq + 4 b + (i & 3) in
      // dword 2 b + ((i & 3) >> 1) (its half is already in the table's dword)
      const uint32_t am = (qt == (l16t >> 2)) ? 0xffffffffu : 0u;
      const uint32_t m_lo = (l16t & 2) ? 0u : am, m_hi = (l16t & 2) ? am : 0u;
      const uint32_t wt_lane = (uint32_t)OFF_WD + (uint32_t)(chq + (l16t & 3)) * 4u;
      const uint32_t cx = (uint32_t)(4 * w) + (uint32_t)qt;
#ifndef OKP_F2_ABL_NODW
      // (all DIH + 2 squeeze rows are walked whatever the tile's height: rows beyond it hold stale - finite or not, it does not matter -
      //  values of the LDS tile, and the accumulator rows they feed are never stored; no branch, so the reads run ahead of the MFMAs)
      u32x4 a0[3], a1[3];                                     // A operands of the three dy taps of a column position: block 0 {lo, hi, 0, 0}, block 1 {0, 0, lo, hi}
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) { a0[dy] = u32x4{0u, 0u, 0u, 0u}; a1[dy] = u32x4{0u, 0u, 0u, 0u}; }
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const uint32_t v0 = *reinterpret_cast<const uint32_t*>(smem + wt_lane + ((dy * 3 + dx) * HALF) * 4);
          const uint32_t v1 = *reinterpret_cast<const uint32_t*>(smem + wt_lane + ((dy * 3 + dx) * HALF + 4) * 4);
          a0[dy][0] = v0 & m_lo; a0[dy][1] = v0 & m_hi;
          a1[dy][2] = v1 & m_lo; a1[dy][3] = v1 & m_hi;
        }
        u32x4 bf[DIH + 2];
#pragma unroll
        for (int r = 0; r < DIH + 2; ++r) {
          const uint32_t sp = (uint32_t)(r * DSW + dx) + (uint32_t)l16t;
          bf[r] = *reinterpret_cast<const u32x4*>(smem + OFF_S + sp * (MID * 2) + ((cx ^ (sp & SWM)) << 4));
        }
#pragma unroll
        for (int r = 0; r < DIH + 2; ++r) {
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int iy = r - dy;
            if (iy >= 0 && iy < DIH) {
              acd[iy][0] = H16<T>::mfma16(a0[dy], bf[r], acd[iy][0]);
              acd[iy][1] = H16<T>::mfma16(a1[dy], bf[r], acd[iy][1]);
            }
          }
        }
      }
#endif
#pragma unroll
      for (int iy = 0; iy < DIH; ++iy) {
        const u32x4 o = relu_add_pack8<T>(acd[iy][0], acd[iy][1], rr2[iy]);
        F2_STORE(o, rs_o, oo2[iy]);
      }
    } else
    // thread = (8-channel group cg, column slot): it walks DOWN its column, each new squeeze row feeding the three
    // output rows that see it as tap row 2, 1, 0; depth-wise weights in registers for the phase.
    {
      const int cg = dw_cg;
      const float* const wl = reinterpret_cast<const float*>(smem + OFF_WD) + cg * 8;     // [tap][128] fp32, bias at tap 9
      float breg[8];
      {
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(wl + 9 * HALF), u1 = *reinterpret_cast<const f32x4*>(wl + 9 * HALF + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { breg[e] = u0[e]; breg[4 + e] = u1[e]; }
      }
      bool prefetched = false;
      for (int ix = dw_slot; ix < ((p.IW + 15) & ~15); ix += 16) {       // uniform trip count: the prefetch sits inside
        // residuals of the whole column first, then the next tile's first ring steps (HBM): those land behind the
        // residuals, while the taps run
        load_col_residuals(ix, rr, oo);
        if (RING_SKIP ? ix + 16 >= ((p.IW + 15) & ~15) : !prefetched) {     // (ring skip: behind the LAST column's reads of the ring)
          prefetched = true;
          if constexpr (RING_SKIP) {
            if (p.skip) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __syncthreads(); }   // every thread has its skip values: the ring is free
          }
          const int next = tile + gridDim.x;
          if (next < p.n_tiles) {
            tile_setup(next);                                // (the validity bits are next read behind >= 8 barriers)
            first_weights(w1f);
#pragma unroll
            for (int ks = 0; ks < NST - 1; ++ks) issue_x(ks, ks);
          }
        }
        const int ixc = ix < p.IW ? ix : 0;
        f32x2 v[MAXIH][4];                                   // one accumulator row per output row of the column
#pragma unroll
        for (int iy = 0; iy < MAXIH; ++iy)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[iy][e] = f32x2{breg[2 * e], breg[2 * e + 1]};
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          f32x2 wt[3][4];                                    // the three taps of this column position
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(wl + (dy * 3 + dx) * HALF), u1 = *reinterpret_cast<const f32x4*>(wl + (dy * 3 + dx) * HALF + 4);
            wt[dy][0] = f32x2{u0[0], u0[1]}; wt[dy][1] = f32x2{u0[2], u0[3]}; wt[dy][2] = f32x2{u1[0], u1[1]}; wt[dy][3] = f32x2{u1[2], u1[3]};
          }
          if constexpr (DWB > 1) {
          // the column's squeeze rows in batches of DWB reads, all of a batch in flight before its first FMA (rows beyond
          // the tile's are read too - the accumulator rows they feed are never stored)
          constexpr int NSR = STR * (MAXIH - 1) + 3, BT = DWB;
#pragma unroll
          for (int s0 = 0; s0 < NSR; s0 += BT) {
            u32x4 svb[BT];
#pragma unroll
            for (int u = 0; u < BT; ++u) {
              if (s0 + u < NSR) {
                const int sp = min((s0 + u) * p.SW + STR * ixc + dx, SP - 1);
                svb[u] = *reinterpret_cast<const u32x4*>(smem + OFF_S + sp * (MID * 2) + ((cg ^ (sp & SWM)) << 4));
              }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < BT; ++u) {
              const int sr = s0 + u;
              if (sr < NSR) {
                f32x2 s2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) s2[e] = f32x2{H16<T>::lo(svb[u][e]), H16<T>::hi(svb[u][e])};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                  const int iy = (sr - dy) / STR;
                  if (sr - dy >= 0 && (sr - dy) % STR == 0 && iy < MAXIH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[iy][e] = __builtin_elementwise_fma(s2[e], wt[dy][e], v[iy][e]);
                  }
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          } else {
#pragma unroll
          for (int sr = 0; sr < STR * (MAXIH - 1) + 3; ++sr) {   // squeeze row sr is tap row dy of output row (sr - dy) / STR
            if (sr < STR * (p.IH - 1) + 3) {
              const int sp = sr * p.SW + STR * ixc + dx;
              const u32x4 sv = *reinterpret_cast<const u32x4*>(smem + OFF_S + sp * (MID * 2) + ((cg ^ (sp & SWM)) << 4));
              f32x2 s2[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) s2[e] = f32x2{H16<T>::lo(sv[e]), H16<T>::hi(sv[e])};
#pragma unroll
              for (int dy = 0; dy < 3; ++dy) {
                const int iy = (sr - dy) / STR;
                if (sr - dy >= 0 && (sr - dy) % STR == 0 && iy < MAXIH) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[iy][e] = __builtin_elementwise_fma(s2[e], wt[dy][e], v[iy][e]);
                }
              }
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
          F2_STORE(o, rs_o, oo[iy]);
        }
      }
    }
    F2_STAMP(4);
    // no barrier here: phase 2 does not read what the next tile's phase 1 writes before its own barriers
  }
  F2_STAMP_ANY(7);
}

}  // namespace


bool okp_fire2_supported(int cin, int mid, int half, int stride) {
  if (half != mid) return false;
  if (stride == 1)
    return (cin == 256 && mid == 128) || (cin == 384 && mid == 192) || (cin == 512 && mid == 256) ||
           (cin == 384 && mid == 128) || (cin == 512 && mid == 192);
  // (256 -> 192 at stride 2, 32x32 -> 16x16, N=64: as a vector-ALU depth-wise instance 40 us against 37.6 us for the two-launch path - left to the
  //  two launches until round 6; with the depth-wise branch on the matrix pipe the one launch is ahead, and it is ONE persistent launch
  //  where the two gather-tile launches were stretched to 102 + 34 us by the side streams' kernels: profiles/r06y_ab_fire2_stride2_dwm.txt)
  return stride == 2 && ((cin == 256 && mid == 128) || (cin == 384 && mid == 192) || (cin == 384 && mid == 256) || (cin == 256 && mid == 192));
}

template <typename T>
static int launch_fire2_t(OkpFire2Params p, int cin, int mid, int stride, hipStream_t stream) {
  // interior rectangle IH x IW: halo'd footprint <= 128 squeeze pixels, <= 96 interior pixels.  First the rounds of the persistent grid; then,
  // with several rounds, the fewest tiles (least halo), ties -> wider rows - and with ONE round the smallest tile, because the launch then takes one
  // tile's time: at 16 x 16 -> 8 x 8 (N = 64) 2 x 8 tiles (256, one per CU; 16.9 us) instead of 3 x 8 (192: a quarter of the CUs idle; 19.0 us).
  // (A cost model - rounds x (squeeze pixels + a fixed part) - chose 4 x 6 at 64 x 64 -> 32 x 32 and lost 6 % there: the depth-wise phase's cost
  //  follows rows per thread and active column slots, not pixels.  profiles/r06w_fire2_stride2_tiles.txt)
  const long resident_wg = 256 * (mid == 128 ? 2 : 1);
  long best = -1;
  for (int ih = 1; ih <= p.Ho && ih <= MAXIH; ++ih)
    for (int iw = 1; iw <= p.Wo && iw <= 96; ++iw) {
      const int sh = stride * (ih - 1) + 3, sw = stride * (iw - 1) + 3;
      if (sh * sw > SP || ih * iw > 16 * PBI) continue;
      const long ty = (p.Ho + ih - 1) / ih, tx = (p.Wo + iw - 1) / iw;
      const long rounds = ((long)p.N * ty * tx + resident_wg - 1) / resident_wg;
      // (one round: squeeze pixels + 8 per idle column slot of the depth-wise phase - 2 x 8 before 4 x 4, measured 16.9 / 17.4 us)
      const long score = (rounds << 40) + (rounds == 1 ? (long)sh * sw + 8 * (16 - (iw < 16 ? iw : 16)) : ty * tx) * 4096 - iw;
      if (best < 0 || score < best) { best = score; p.IH = ih; p.IW = iw; p.SH = sh; p.SW = sw; p.tiles_y = (int)ty; p.tiles_x = (int)tx; }
    }
  // (experiment switch: OKP_F2_S2_TILE="ih,iw" forces the interior rectangle of the stride-2 launches)
  if (stride == 2) {
    static const char* const e = getenv("OKP_F2_S2_TILE");
    int ih = 0, iw = 0;
    if (e && sscanf(e, "%d,%d", &ih, &iw) == 2 && ih >= 1 && ih <= MAXIH && iw >= 1 && (2 * (ih - 1) + 3) * (2 * (iw - 1) + 3) <= SP && ih * iw <= 16 * PBI) {
      p.IH = ih < p.Ho ? ih : p.Ho; p.IW = iw < p.Wo ? iw : p.Wo; p.SH = 2 * (p.IH - 1) + 3; p.SW = 2 * (p.IW - 1) + 3;
      p.tiles_y = (p.Ho + p.IH - 1) / p.IH; p.tiles_x = (p.Wo + p.IW - 1) / p.IW;
    }
  }
  // the 256 -> 128 stride-1 instance on maps whose width is a multiple of 16: tiles of kDwmIH x 16 interior pixels, depth-wise branch on the matrix
  // pipe (okp_fire2_kernel<..., DWM>; OKP_F2_DWM=0 keeps the vector-ALU form for A/B)
  static const bool dwm_on = [] { const char* e = getenv("OKP_F2_DWM"); return !(e && e[0] == '0'); }();
  static const int dwm_mid = [] { const char* e = getenv("OKP_F2_DWM_MID"); return e ? atoi(e) : 256; }();      // (A/B: largest squeeze width that takes it)
  const bool dwm = dwm_on && stride == 1 && p.Wo % 16 == 0 && mid <= dwm_mid;
  if (dwm) {
    p.IH = p.Ho < kDwmIH ? p.Ho : kDwmIH; p.IW = 16; p.SH = p.IH + 2; p.SW = 18;
    p.tiles_y = (p.Ho + p.IH - 1) / p.IH; p.tiles_x = p.Wo / 16;
  }
  // the stride-2 instances on output maps a multiple of 8 wide: tiles of 3 x 8 output pixels (OKP_F2_DWM_S2=0: the vector-ALU form, for A/B) -
  // or 2 x 8 when both fit the resident grid in one round (16 x 16 -> 8 x 8 at N = 64: 256 tiles, one per CU, instead of 192)
  static const bool dwm2_on = [] { const char* e = getenv("OKP_F2_DWM_S2"); return !(e && e[0] == '0'); }();
  const bool dwm2 = dwm_on && dwm2_on && stride == 2 && p.Wo % 8 == 0 && mid <= dwm_mid;
  if (dwm2) {
    const long tx = p.Wo / 8;
    int ih = p.Ho < 3 ? p.Ho : 3;
    if (ih == 3 && (long)p.N * ((p.Ho + 1) / 2) * tx <= resident_wg) ih = 2;
    p.IH = ih; p.IW = 8; p.SH = 2 * (ih - 1) + 3; p.SW = 17;
    p.tiles_y = (p.Ho + ih - 1) / ih; p.tiles_x = (int)tx;
  }
  p.IP = p.IH * p.IW;
  p.RPR = (p.IW + 3) / 4;
  const long tiles = (long)p.N * p.tiles_y * p.tiles_x;
  if (tiles >= 0x7FFFFFFFl) { okp_set_error("okp_fire_forward: too many tiles"); return OKP_EINVAL; }
  p.n_tiles = (int)tiles;
  p.div_tiles_frame = okp_fastdiv((uint32_t)(p.tiles_y * p.tiles_x));
  p.div_tiles_x = okp_fastdiv((uint32_t)p.tiles_x);
  p.div_sw = okp_fastdiv((uint32_t)p.SW);
  p.div_iw = okp_fastdiv((uint32_t)p.IW);
  p.div_rpr = okp_fastdiv((uint32_t)p.RPR);
  const int resident = 256 * (mid == 128 ? 2 : 1);
  const dim3 grid((unsigned)(p.n_tiles < resident ? p.n_tiles : resident)), block((unsigned)(2 * mid));
#ifdef OKP_FIRE_STAMPS
  static uint32_t* dbg = nullptr;
  if (!dbg) (void)hipMalloc((void**)&dbg, 16 * 8 * 8 * 4);
  (void)hipMemsetAsync(dbg, 0, 16 * 8 * 8 * 4, stream);
  p.dbg = dbg;
  struct Print { OkpFire2Params p; int mid; hipStream_t stream; uint32_t* dbg; ~Print() {
    if (!getenv("OKP_FIRE_STAMPS_PRINT")) return;
    (void)hipStreamSynchronize(stream);
    uint32_t h[16 * 8 * 8];
    (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
    const int nw = mid / 32;
    printf("fire2 stamps (tile %d x %d interior, %d tiles, %d waves): clocks per wave of the second tile: squeeze GEMM (ring) | s -> LDS + barrier | expand GEMM + stores | depth-wise branch (+ next tile's requests)\n", p.IH, p.IW, p.n_tiles, nw);
    double sum[4] = {0, 0, 0, 0}; int cnt = 0;
    for (int wg = 0; wg < 16; ++wg)
      for (int w = 0; w < nw; ++w) {
        const uint32_t* a = &h[(wg * 8 + w) * 8];
        if (!a[4]) continue;
        if (wg % 5 == 0 && (w == 0 || w == nw - 1)) printf("wg %2d wave %d: %6u | %6u | %6u | %6u\n", wg, w, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3]);
        for (int i = 0; i < 4; ++i) sum[i] += a[i + 1] - a[i];
        ++cnt;
      }
    if (cnt) printf("mean: %.0f | %.0f | %.0f | %.0f  = %.0f clocks per tile\n", sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, (sum[0] + sum[1] + sum[2] + sum[3]) / cnt);
    double pro = 0, all = 0; int c2 = 0;
    for (int wg = 0; wg < 16; ++wg)
      for (int w = 0; w < nw; ++w) { const uint32_t* a = &h[(wg * 8 + w) * 8]; if (!a[7]) continue; pro += a[6] - a[5]; all += a[7] - a[5]; ++c2; }
    if (c2) printf("whole kernel: prologue (first requests + constants -> LDS + barrier) %.0f clocks, entry to exit %.0f clocks\n", pro / c2, all / c2);
  } } print_at_exit{p, mid, stream, dbg};
#endif
  if (stride == 1) {
    if (dwm && cin == 256 && mid == 128) hipLaunchKernelGGL((okp_fire2_kernel<T, 256, 128, 1, true>), grid, block, 0, stream, p);
    else if (dwm && cin == 384 && mid == 192) hipLaunchKernelGGL((okp_fire2_kernel<T, 384, 192, 1, true>), grid, block, 0, stream, p);
    else if (dwm && cin == 512 && mid == 256) hipLaunchKernelGGL((okp_fire2_kernel<T, 512, 256, 1, true>), grid, block, 0, stream, p);
    else if (dwm && cin == 384 && mid == 128) hipLaunchKernelGGL((okp_fire2_kernel<T, 384, 128, 1, true>), grid, block, 0, stream, p);
    else if (dwm && cin == 512 && mid == 192) hipLaunchKernelGGL((okp_fire2_kernel<T, 512, 192, 1, true>), grid, block, 0, stream, p);
    else if (cin == 256 && mid == 128) hipLaunchKernelGGL((okp_fire2_kernel<T, 256, 128, 1>), grid, block, 0, stream, p);
    else if (cin == 384 && mid == 192) hipLaunchKernelGGL((okp_fire2_kernel<T, 384, 192, 1>), grid, block, 0, stream, p);
    else if (cin == 512 && mid == 256) hipLaunchKernelGGL((okp_fire2_kernel<T, 512, 256, 1>), grid, block, 0, stream, p);
    else if (cin == 384 && mid == 128) hipLaunchKernelGGL((okp_fire2_kernel<T, 384, 128, 1>), grid, block, 0, stream, p);
    else if (cin == 512 && mid == 192) hipLaunchKernelGGL((okp_fire2_kernel<T, 512, 192, 1>), grid, block, 0, stream, p);
    else { okp_set_error("okp_fire_forward: no streaming kernel for %d -> %d", cin, mid); return OKP_EINVAL; }
  } else if (dwm2 && cin == 256 && mid == 128) hipLaunchKernelGGL((okp_fire2_kernel<T, 256, 128, 2, true>), grid, block, 0, stream, p);
  else if (dwm2 && cin == 384 && mid == 192) hipLaunchKernelGGL((okp_fire2_kernel<T, 384, 192, 2, true>), grid, block, 0, stream, p);
  else if (dwm2 && cin == 384 && mid == 256) hipLaunchKernelGGL((okp_fire2_kernel<T, 384, 256, 2, true>), grid, block, 0, stream, p);
  else if (dwm2 && cin == 256 && mid == 192) hipLaunchKernelGGL((okp_fire2_kernel<T, 256, 192, 2, true>), grid, block, 0, stream, p);
  else if (cin == 256 && mid == 192) hipLaunchKernelGGL((okp_fire2_kernel<T, 256, 192, 2>), grid, block, 0, stream, p);
  else if (cin == 256 && mid == 128) hipLaunchKernelGGL((okp_fire2_kernel<T, 256, 128, 2>), grid, block, 0, stream, p);
  else if (cin == 384 && mid == 192) hipLaunchKernelGGL((okp_fire2_kernel<T, 384, 192, 2>), grid, block, 0, stream, p);
  else if (cin == 384 && mid == 256) hipLaunchKernelGGL((okp_fire2_kernel<T, 384, 256, 2>), grid, block, 0, stream, p);
  else { okp_set_error("okp_fire_forward: no streaming kernel for %d -> %d stride %d", cin, mid, stride); return OKP_EINVAL; }
  return okp_check_hip(hipGetLastError(), "okp_fire2 launch");
}

int okp_launch_fire2(int dtype, const OkpFire2Params& p, int cin, int mid, int stride, hipStream_t stream) {
  if (dtype == OKP_BF16) return launch_fire2_t<__bf16>(p, cin, mid, stride, stream);
  return launch_fire2_t<_Float16>(p, cin, mid, stride, stream);
}

extern "C" int okp_fire_forward(const okp_conv* squeeze, const okp_conv* expand, const float* dw_w_dev, const float* dw_bias_dev,
                                const okp_fire_args* a, void* stream) {
  if (!squeeze || !expand || !dw_w_dev || !dw_bias_dev || !a || !a->x.data || !a->out.data) { okp_set_error("okp_fire_forward: null argument"); return OKP_EINVAL; }
  const bool x3 = squeeze->dtype == OKP_F32X3;
  if (!(okp_is16(squeeze->dtype) || x3) || expand->dtype != squeeze->dtype) { okp_set_error("okp_fire_forward: the fused fire kernel takes bf16, fp16 or split-product (OKP_F32X3) plans of one type"); return OKP_EINVAL; }
  if (squeeze->n_taps != 1 || expand->n_taps != 1 || squeeze->n_src != 1 || expand->n_src != 1) { okp_set_error("okp_fire_forward: squeeze/expand must be single-tap 1x1 plans"); return OKP_EINVAL; }
  const int cin = squeeze->cin[0], mid = squeeze->cout, half = expand->cout;
  if (expand->cin[0] != mid || half != mid) { okp_set_error("okp_fire_forward: expects expand cin == squeeze cout == half (sr = 2)"); return OKP_EINVAL; }
  if (cin % 64 || mid % 64 || mid > 256) { okp_set_error("okp_fire_forward: cin %d / mid %d must be multiples of 64, mid <= 256", cin, mid); return OKP_EINVAL; }
  if (a->stride != 1 && a->stride != 2) { okp_set_error("okp_fire_forward: stride %d", a->stride); return OKP_EINVAL; }
  if (a->skip && (a->stride != 1 || cin != 2 * half)) { okp_set_error("okp_fire_forward: skip needs stride 1 and cin == cout"); return OKP_EINVAL; }
  const int ho = (a->x.h - 1) / a->stride + 1, wo = (a->x.w - 1) / a->stride + 1;
  if (a->out.h != ho || a->out.w != wo) { okp_set_error("okp_fire_forward: out is %dx%d, expected %dx%d", a->out.h, a->out.w, ho, wo); return OKP_EINVAL; }
  if (x3) {
    // split-product plans: fp32 tensors; okp_fire_x3.hip has the 256 -> 128 -> 256 module with skip at stride 1
    if (!okp_fire_x3_supported(cin, mid, half, a->stride, a->skip) || squeeze->n_single_slices || expand->n_single_slices || !squeeze->fragT_dev || !expand->fragT_dev) {
      okp_set_error("okp_fire_forward: the split-product one-launch kernel takes 256 -> 128 -> 256 at stride 1 with skip or at stride 2 without (three-term plans); run the squeeze plan and the fused tail instead");
      return OKP_EINVAL;
    }
    if (a->x.pix_stride % 4 || a->out.pix_stride % 4 || a->x.pix_stride < cin || a->out.pix_stride < 2 * half || ((uintptr_t)a->x.data) % 16 || ((uintptr_t)a->out.data) % 16) {
      okp_set_error("okp_fire_forward: views must be 16-byte aligned and wide enough"); return OKP_EINVAL;
    }
    if (a->x.bytes <= 0 || a->x.bytes >= 0x7FFF0000ll || a->out.bytes <= 0 || a->out.bytes >= 0x7FFF0000ll) { okp_set_error("okp_fire_forward: views must be < 2 GiB"); return OKP_EINVAL; }
    OkpFire2Params q;
    memset(&q, 0, sizeof(q));
    q.x = a->x.data; q.x_bytes = (uint32_t)a->x.bytes; q.H = a->x.h; q.W = a->x.w; q.x_ps = a->x.pix_stride;
    q.out = a->out.data; q.out_bytes = (uint32_t)a->out.bytes; q.Ho = ho; q.Wo = wo; q.out_ps = a->out.pix_stride;
    q.N = a->n; q.skip = a->skip;
    q.w1 = squeeze->fragT_dev; q.w1_cout_pad = squeeze->cout_pad; q.b1 = squeeze->bias_dev; q.s1 = squeeze->oscale_dev;
    q.wa = expand->fragT_dev; q.wa_cout_pad = expand->cout_pad; q.ba = expand->bias_dev; q.sa = expand->oscale_dev;
    q.wd = dw_w_dev; q.bd = dw_bias_dev; q.range_flag = squeeze->range_flag;
    return okp_launch_fire_x3(q, a->stride, (hipStream_t)stream);
  }
  if (a->x.pix_stride % 8 || a->out.pix_stride % 8 || a->x.pix_stride < cin || a->out.pix_stride < 2 * half ||
      ((uintptr_t)a->x.data) % 16 || ((uintptr_t)a->out.data) % 16) { okp_set_error("okp_fire_forward: views must be 16-byte aligned and wide enough"); return OKP_EINVAL; }
  if (a->x.bytes <= 0 || a->x.bytes >= 0x7FFF0000ll) { okp_set_error("okp_fire_forward: x spans %lld bytes; views must be < 2 GiB", (long long)a->x.bytes); return OKP_EINVAL; }
  if (!okp_fire2_supported(cin, mid, half, a->stride)) {
    okp_set_error("okp_fire_forward: no one-launch kernel for %d -> %d -> %d at stride %d; run the squeeze plan and the fused tail "
                  "(okp_conv_forward with dw_*) instead", cin, mid, 2 * half, a->stride);
    return OKP_EINVAL;
  }
  {
    if (a->out.bytes <= 0 || a->out.bytes >= 0x7FFF0000ll) { okp_set_error("okp_fire_forward: out spans %lld bytes; views must be < 2 GiB", (long long)a->out.bytes); return OKP_EINVAL; }
    OkpFire2Params q;
    memset(&q, 0, sizeof(q));
    q.x = a->x.data; q.x_bytes = (uint32_t)a->x.bytes; q.H = a->x.h; q.W = a->x.w; q.x_ps = a->x.pix_stride;
    q.out = a->out.data; q.out_bytes = (uint32_t)a->out.bytes; q.Ho = ho; q.Wo = wo; q.out_ps = a->out.pix_stride;
    q.N = a->n; q.skip = a->skip;
#if defined(OKP_FIRE_STAMPS) || defined(OKP_F2_ABL_NOSKIP)
    if (getenv("OKP_FIRE_NOSKIP")) q.skip = 0;       // timing ablation of the debug build (wrong results): no skip values are requested
#endif
    if (int e = okp_ensure_frags(squeeze, (hipStream_t)stream)) return e;
    if (int e = okp_ensure_frags(expand, (hipStream_t)stream)) return e;
    q.w1 = squeeze->fragT_dev; q.w1_cout_pad = squeeze->cout_pad; q.b1 = squeeze->bias_dev;
    q.wa = expand->fragT_dev; q.wa_cout_pad = expand->cout_pad; q.ba = expand->bias_dev;
    q.wd = dw_w_dev; q.bd = dw_bias_dev;
    return okp_launch_fire2(squeeze->dtype, q, cin, mid, a->stride, (hipStream_t)stream);
  }
}

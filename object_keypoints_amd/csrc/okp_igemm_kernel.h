// Tap-list implicit-GEMM convolution for gfx950 (MI355X): fp32, bf16 and fp16 activations / weights, fp32 accumulation.
//
//   D[co][pixel] = sum over K-slices  Wslice[co][128 B of K] . Xslice[pixel][128 B of K]
//
// A K-slice is 128 bytes of the reduction dimension per row (64 bf16 or 32 fp32 channels of one
// tap).  The weight side is pre-packed [slice][cout_pad][128 B]; the pixel side is gathered from
// NHWC activations with one 16-byte buffer_load per lane (hardware range check returns 0 for the
// padding halo, so there is no im2col buffer and no border code).  Both tiles are staged in LDS
// as [row][8 x 16 B] with the 16-B chunk index XOR-swizzled by (row>>1)&7, which makes the
// ds_read_b128 fragment reads of the 32x32 MFMA conflict-free (two 128-B rows share a 256-B
// bank row).  D keeps the output channel on the accumulator rows, so every lane owns 4
// consecutive channels of one pixel per register group; the epilogue adds the bias in registers,
// transposes through LDS and writes whole NHWC pixel rows (16 B per lane, coalesced) after the
// optional residual add and ReLU.
//
// Reference semantics being computed: torch conv2d/conv_transpose2d + eval-mode batch_norm
// (folded into W/bias) + add + relu — see include/okp.h for the file:line list.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "okp_internal.h"


namespace {

constexpr uint32_t kInvalidOff = 0x80000000u;   // >= num_records of every tensor we accept (< 2 GiB)

// fp32 activations and weights whose products run on the fp16 matrix pipe as a three-term split (OKP_F32X3): an fp32
// value x is x_hi + x_lo with x_hi = fp16(x), x_lo = fp16(x - x_hi) (22 significant bits together), and
// x * w = x_hi w_hi + x_lo w_hi + x_hi w_lo (+ x_lo w_lo, 2^-22 relative: dropped), each term exact in the fp32
// accumulator.  3 MFMAs of 32 cycles do the work of eight 64-cycle fp32 MFMAs: 5.3x the fp32 matrix peak at fp32-grade
// results (tests/precision/emulate.py: the heat maps stay within 3e-6 of the reference, like the exact-fp32 kernel).
// Tensors in HBM and LDS stay fp32: the weight blob holds the two fp16 halves pre-split by okp_conv_create
// (per 128-byte K-slice row: [hi k0-15 | lo k0-15 | hi k16-31 | lo k16-31], the bytes of 32 fp32), the activation
// fragments are split in registers.  Operands must be below 65504 in magnitude (as for OKP_F16).
struct F32S { float v; };
static_assert(sizeof(F32S) == 4, "fp32 storage");

// 8 fp32 of K held by a lane (two 16-byte chunks) -> its fp16 hi / lo fragments
__device__ __forceinline__ void okp_split8(const u32x4& r0, const u32x4& r1, u32x4& hi, u32x4& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
  const f32x4 x0 = __builtin_bit_cast(f32x4, r0), x1 = __builtin_bit_cast(f32x4, r1);
  // per pair of elements: v_cvt_pk_f16_f32 (hi), two v_fma_mix_f32 (x - hi, exact, reading the packed half directly), v_cvt_pk_f16_f32 (lo):
  // 4 VALU instructions per 2 elements (hipcc's own lowering of the same arithmetic takes 6: cvt_pk, 2 cvt_f32_f16, 2 sub, cvt_pk)
  f16x2 hp[4], lp[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float xa = q < 2 ? x0[2 * q] : x1[2 * q - 4], xb = q < 2 ? x0[2 * q + 1] : x1[2 * q - 3];
    f16x2 h2; h2[0] = (_Float16)xa; h2[1] = (_Float16)xb;
    const uint32_t hw = __builtin_bit_cast(uint32_t, h2);
    float da, db;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(da) : "v"(hw), "v"(xa));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(db) : "v"(hw), "v"(xb));
    f16x2 l2; l2[0] = (_Float16)da; l2[1] = (_Float16)db;
    hp[q] = h2; lp[q] = l2;
  }
  hi = u32x4{__builtin_bit_cast(uint32_t, hp[0]), __builtin_bit_cast(uint32_t, hp[1]), __builtin_bit_cast(uint32_t, hp[2]), __builtin_bit_cast(uint32_t, hp[3])};
  lo = u32x4{__builtin_bit_cast(uint32_t, lp[0]), __builtin_bit_cast(uint32_t, lp[1]), __builtin_bit_cast(uint32_t, lp[2]), __builtin_bit_cast(uint32_t, lp[3])};
#else
  hi = r0; lo = r1;
#endif
}

// the high halves alone (single-term slices)
__device__ __forceinline__ u32x4 okp_hi8(const u32x4& r0, const u32x4& r1) {
  const f32x4 x0 = __builtin_bit_cast(f32x4, r0), x1 = __builtin_bit_cast(f32x4, r1);
  f16x8 h;
#pragma unroll
  for (int e = 0; e < 4; ++e) { h[e] = (_Float16)x0[e]; h[4 + e] = (_Float16)x1[e]; }
  return __builtin_bit_cast(u32x4, h);
}

template <typename T, int MT> struct Mma;
template <> struct Mma<F32S, 32> {
  using acc_t = f32x16;
  // (the split kernel has its own main loop: operands arrive as hi / lo pairs)
  static __device__ __forceinline__ void run3(const u32x4& ah, const u32x4& al, const u32x4& bh, const u32x4& bl, f32x16& c) {
    c = H16<_Float16>::mfma32(al, bh, c);
    c = H16<_Float16>::mfma32(ah, bl, c);
    c = H16<_Float16>::mfma32(ah, bh, c);
  }
  static __device__ __forceinline__ void run(const u32x4&, const u32x4&, f32x16&) {}
};
template <> struct Mma<float, 32> {
  using acc_t = f32x16;
  // 16 B = 4 fp32 of K per lane: four 32x32x2 MFMAs, lane half h supplies k = 4h+e (K order is
  // the same on both operands, so any permutation of K inside the slice is legal).
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
    const f32x4 fa = __builtin_bit_cast(f32x4, a);
    const f32x4 fb = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], fb[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1], fb[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2], fb[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3], fb[3], c, 0, 0, 0);
  }
};
template <> struct Mma<__bf16, 32> {
  using acc_t = f32x16;
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) { c = H16<__bf16>::mfma32(a, b, c); }
};
template <> struct Mma<_Float16, 32> {
  using acc_t = f32x16;
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) { c = H16<_Float16>::mfma32(a, b, c); }
};
// 16x16 MFMA tiles: lane l supplies row l&15 and the 16-byte chunk (l>>4) of a 64-byte K group, so one k-step covers
// 32 bf16 / fp16 (or 16 fp32) of K.  Same FLOP per cycle as the 32x32 forms; the chip holds a higher clock on this shape
// (MI355X_MICROARCH.md, DVFS give-back item 7), which is why it exists as a variant here.
template <> struct Mma<__bf16, 16> {
  using acc_t = f32x4;
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) { c = H16<__bf16>::mfma16(a, b, c); }
};
template <> struct Mma<_Float16, 16> {
  using acc_t = f32x4;
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) { c = H16<_Float16>::mfma16(a, b, c); }
};
template <> struct Mma<float, 16> {
  using acc_t = f32x4;
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
    const f32x4 fa = __builtin_bit_cast(f32x4, a);
    const f32x4 fb = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], c, 0, 0, 0);
  }
};

template <typename T> struct Io;
template <> struct Io<float> {
  static constexpr int kBytes8 = 32;   // bytes of 8 channels
  static __device__ __forceinline__ void add8(float (&v)[8], const char* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
  }
  static __device__ __forceinline__ void store8(const float (&v)[8], char* p) {
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 16) = b;
  }
};
template <typename T> struct Io16 {
  using x8 = typename H16<T>::x8;
  static constexpr int kBytes8 = 16;
  static __device__ __forceinline__ void add8(float (&v)[8], const char* p) {
    const x8 a = *reinterpret_cast<const x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += (float)a[e];
  }
  static __device__ __forceinline__ void store8(const float (&v)[8], char* p) {
    x8 a;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (T)v[e];
    *reinterpret_cast<x8*>(p) = a;
  }
};
template <> struct Io<F32S> : Io<float> {};
template <> struct Io<__bf16> : Io16<__bf16> {};
template <> struct Io<_Float16> : Io16<_Float16> {};
// vector types of the element type (fp32 never uses the 16-bit ones; they only have to name a type)
template <typename T> struct Vt { using x4 = typename H16<T>::x4; using x8 = typename H16<T>::x8; };
template <> struct Vt<float> { using x4 = bf16x4; using x8 = bf16x8; };
template <> struct Vt<F32S> { using x4 = bf16x4; using x8 = bf16x8; };

__device__ __forceinline__ int fastdiv(int x, const OkpFastDiv& f) {
  return f.mul ? (int)(__umulhi((uint32_t)x, f.mul) >> f.shift) : x;
}

// Byte offset of a 16-B chunk in a [row][KB bytes] LDS tile.  The chunk index is XOR-swizzled with row bits so
// that the 16 rows a ds_read_b128 lane group touches fall on 16 different 16-B slots of the 256-B bank row:
// 128-B rows (2 per bank row) use (row>>1)&7, 64-B rows (4 per bank row) use (row>>2)&3.
template <int KB>
__device__ __forceinline__ uint32_t swz(int row, int chunk) {
  if constexpr (KB == 128) return (uint32_t)row * 128u + (uint32_t)((chunk ^ ((row >> 1) & 7)) << 4);
  else return (uint32_t)row * 64u + (uint32_t)((chunk ^ ((row >> 2) & 3)) << 4);
}

struct SliceMeta {            // per-slice gather constants, built once per workgroup in LDS
  int32_t d_lo, d_hi;         // byte deltas of the two 64-byte halves relative to the pixel's base offset
  uint32_t packed;            // tap_lo | tap_hi << 8 | nvalid << 16 | src << 24
  uint32_t pad;
};
constexpr int kMetaMax = 256;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <typename T, int BCO, int BPX, int WCO, int WPX, int NS, int KB, int MT, int NSRC>
__global__ __launch_bounds__(64 * WCO * WPX, (std::is_same<T, F32S>::value ? 2 : 1)) void okp_igemm_kernel(const OkpIgemmParams p) {     // split-product tiles: two waves per SIMD (<= 256 registers)
  constexpr int NT = 64 * WCO * WPX;
  constexpr int ESZ = (int)sizeof(T);
  static_assert(KB == 128 || KB == 64, "LDS row = one 128-byte K-slice or half of one");
  constexpr int CPR = KB / 16;                // 16-B chunks per LDS row
  constexpr int HPS = 128 / KB;               // ring steps per 128-byte K-slice
  static_assert(MT == 32 || MT == 16, "MFMA tile");
  constexpr int CHK = 64 / MT;                // 16-B chunks of K one k-step consumes per row (lane halves / quarters)
  constexpr int KSTEPS = KB / (16 * CHK);     // MFMA k-steps per ring step
  static_assert(KSTEPS >= 2, "the loop double-buffers fragments over two k-steps");
  constexpr int AREGS = MT * MT / 64;         // accumulator registers per tile
  using acc_t = typename Mma<T, MT>::acc_t;
  constexpr int RPP = NT / CPR;               // tile rows covered by one loader pass
  constexpr int WROWS = BCO / RPP;
  constexpr int XROWS = BPX / RPP;
  constexpr int TCO = BCO / WCO / MT;
  constexpr int TPX = BPX / WPX / MT;
  constexpr int STAGE = (BCO + BPX) * KB;
  constexpr int LDS_BYTES = NS * STAGE;        // NS-deep ring of stages
  static_assert(NS >= 2, "ring depth");
  constexpr int PASSES = (BPX * BCO * ESZ > LDS_BYTES) ? 2 : 1;   // epilogue staging holds the tile in the output type
  static_assert(RPP % 16 == 0, "loader swizzle assumes the pass height keeps (row>>1)&7");
  static_assert(WPX % PASSES == 0, "a two-pass epilogue splits the pixels by wave column");
  constexpr int WPP = WPX / PASSES;           // wave columns staged per epilogue pass
  static_assert(BPX * BCO * ESZ / PASSES <= LDS_BYTES, "epilogue staging must fit");
  constexpr int PX_PER_PASS = BPX / PASSES;
  using x4_t = typename Vt<T>::x4;                 // 4 / 8 elements of the 16-bit types (unused for fp32)
  using x8_t = typename Vt<T>::x8;
  constexpr int PITCH = BCO * ESZ;

  // ONE LDS object (a second __shared__ array makes hipcc drain LDS-DMA before unrelated ds_reads)
  constexpr bool X3 = std::is_same<T, F32S>::value;      // split-product plans (OKP_F32X3): see struct F32S
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES + kMetaMax * (int)sizeof(SliceMeta) + (X3 ? 2048 : 1024)];
  SliceMeta* const meta = reinterpret_cast<SliceMeta*>(smem + LDS_BYTES);
  char* const bias_lds = smem + LDS_BYTES + kMetaMax * (int)sizeof(SliceMeta);     // this tile's BCO biases (fp32); X3: + BCO output scales
  static_assert(BCO <= 256, "bias staging is one 1 KiB LDS-DMA");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave / WPX, wpx = wave % WPX;
  const int HoWo = p.Ho * p.Wo;
  const int P = p.N * HoWo;

  // Loader geometry.  One `buffer_load_dwordx4 ... lds` writes 64 lanes x 16 B = 8 tile rows linearly into LDS,
  // so lane (row r0 = tid>>3, position tid&7) must FETCH the logical chunk that the read-side swizzle expects
  // at that position: c = pos ^ ((row>>1)&7)  (swizzle on the source address, linear destination).
  const int r0 = tid / CPR;
  const int c = (KB == 128) ? ((tid & 7) ^ ((r0 >> 1) & 7)) : ((tid & 3) ^ ((r0 >> 2) & 3));

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.weights), 0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src[0]), 0, (int)p.src_bytes[0], 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src[NSRC - 1]), 0, (int)p.src_bytes[NSRC - 1], 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.n_co_tiles * BCO * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X3 ? p.oscale : p.bias), 0, p.n_co_tiles * BCO * 4, 0x00020000);

  // ---- slice constants -> LDS (one thread per slice) -------------------------------------------
  for (int s = tid; s < p.n_slices; s += NT) {
    const OkpSlice sl = p.slices[s];
    const int src = (NSRC == 1) ? 0 : (int)sl.src;
    const int W = src ? p.srcW[NSRC - 1] : p.srcW[0];
    const int ps = src ? p.src_pix_stride[NSRC - 1] : p.src_pix_stride[0];
    SliceMeta m;
    m.d_lo = ((p.taps[sl.tap_lo].dy * W + p.taps[sl.tap_lo].dx) * ps + sl.c0_lo) * ESZ;
    m.d_hi = ((p.taps[sl.tap_hi].dy * W + p.taps[sl.tap_hi].dx) * ps + sl.c0_hi) * ESZ;
    m.packed = (uint32_t)sl.tap_lo | ((uint32_t)sl.tap_hi << 8) | ((uint32_t)sl.nvalid << 16) | ((uint32_t)src << 24);
    m.pad = (uint32_t)sl.pad;                    // split-product plans: 1 = single-term slice
    meta[s] = m;
  }

  const int fr = lane & (MT - 1), fh = lane / MT;
  __syncthreads();                               // slice constants visible

  // ---- persistent tile loop: the grid is sized to the resident workgroups; a workgroup walks tiles
  // blockIdx.x, +gridDim.x, ...  The epilogue's stores are fire-and-forget, so they drain to HBM while the
  // next tile's gather and MFMAs run (a one-tile-per-workgroup grid leaves the matrix cores idle during the
  // chip-wide write burst, and pays prologue + first-DMA latency once per tile).
  for (int slot = blockIdx.x; slot < p.n_tiles; slot += gridDim.x) {
  // XCD-aware tile order (speed only): workgroups are dealt round-robin over the 8 XCDs, so slots with equal
  // slot % 8 share an L2.  Give each XCD a contiguous range of tiles: spatially adjacent pixel tiles then re-use
  // each other's halo rows (and the same weight slices) out of that XCD's L2.  Bijective for any tile count.
  const int xq = p.n_tiles >> 3, xr = p.n_tiles & 7, xcd = slot & 7;
  const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (slot >> 3);
  const int cls = tile / p.tiles_per_class;                  // sub-pixel class (0 unless n_classes == 4)
  const int tile_c = tile - cls * p.tiles_per_class;
  const int sbase = cls * p.slices_per_class;                // this class's K-slices
  const int co_tile = tile_c % p.n_co_tiles;
  const int px_tile = tile_c / p.n_co_tiles;
  const int co0 = co_tile * BCO, px0 = px_tile * BPX;
  const int out_oy = p.out_oy + (cls >> 1), out_ox = p.out_ox + (cls & 1);

  // this tile's biases -> LDS by one LDS-DMA (wave 0), consumed only in the epilogue: the load is never waited for
  if (wave == 0)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)bias_lds, 16,
                                             (int)(lane * 4 < BCO ? (uint32_t)(co0 + lane * 4) * 4u : kInvalidOff), 0, 0, 0);
  if (X3 && wave == 1)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, (lds_ptr_t)(bias_lds + 1024), 16,
                                             (int)(lane * 4 < BCO ? (uint32_t)(co0 + lane * 4) * 4u : kInvalidOff), 0, 0, 0);

  // ---- per-thread row state -------------------------------------------------------------
  uint32_t wbase[WROWS];
#pragma unroll
  for (int i = 0; i < WROWS; ++i) {
    const int wrow = r0 + i * RPP;                            // LDS row of the weight tile
    const int co = co0 + wrow;
    wbase[i] = (co < p.cout_pad) ? (uint32_t)co * 128u + (uint32_t)c * 16u : kInvalidOff;
  }
  uint32_t xbase[NSRC][XROWS];
  uint32_t xmask[XROWS];
  int row_ho[XROWS], row_wo[XROWS];
#pragma unroll
  for (int i = 0; i < XROWS; ++i) {
    const int pix = px0 + r0 + i * RPP;
    const bool valid = pix < P;
    const int pp = valid ? pix : 0;
    const int n = fastdiv(pp, p.div_howo);
    const int rem = pp - n * HoWo;
    const int ho = fastdiv(rem, p.div_wo);
    const int wo = rem - ho * p.Wo;
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
      const int hi0 = ho * p.conv_stride[s], wi0 = wo * p.conv_stride[s];
      xbase[s][i] = (uint32_t)(((n * p.srcH[s] + hi0) * p.srcW[s] + wi0) * p.src_pix_stride[s]) * (uint32_t)ESZ + (uint32_t)(c & 3) * 16u;   // KB == 64: c < 4
    }
    xmask[i] = 0;
    row_ho[i] = valid ? ho : -0x40000000;        // rows beyond the problem fail every bounds test below
    row_wo[i] = wo;
  }
  // tap loop outermost: one scalar load of the tap per iteration, the rows unrolled underneath it
  for (int t = 0; t < p.n_taps; ++t) {
    const int s = (NSRC == 1) ? 0 : p.taps[t].src;
    const int dy = p.taps[t].dy, dx = p.taps[t].dx;
    const int cs = s ? p.conv_stride[NSRC - 1] : p.conv_stride[0];       // selects, not indexing: no scratch copy
    const int H = s ? p.srcH[NSRC - 1] : p.srcH[0], W = s ? p.srcW[NSRC - 1] : p.srcW[0];
#pragma unroll
    for (int i = 0; i < XROWS; ++i) {
      const int hi = row_ho[i] * cs + dy, wi = row_wo[i] * cs + dx;
      const bool ok = hi >= 0 && hi < H && wi >= 0 && wi < W;
      xmask[i] |= (ok ? 1u : 0u) << t;
    }
  }

  acc_t acc[TCO][TPX];
#pragma unroll
  for (int i = 0; i < TCO; ++i)
#pragma unroll
    for (int j = 0; j < TPX; ++j)
#pragma unroll
      for (int e = 0; e < AREGS; ++e) acc[i][j][e] = 0.f;

  // LDS-DMA gather of slice `s` (constants `m`) into `stage`, in two parts so that the issue slots can be
  // interleaved with the MFMAs of the slice being computed.  Masked lanes (padding halo, rows beyond the
  // problem, chunks beyond Cin) use an offset past the buffer: the hardware range check then writes zeros
  // to LDS (verified by scripts/hwtests/dma_oob.hip).
  auto issue_w = [&](int t, int stage) {       // t = ring step = slice * HPS + half
    const uint32_t wslice = (uint32_t)(sbase + t / HPS) * (uint32_t)p.cout_pad * 128u + (uint32_t)(t % HPS) * 64u;
    char* const wt = smem + stage * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < WROWS; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(wt + i * RPP * KB), 16, (int)(wbase[i] + wslice), 0, 0, 0);
  };
  auto issue_x = [&](int t, const SliceMeta& m, int stage) {
    const int half = t % HPS;                                    // KB == 64: which 64-byte half of the slice
    const bool hi_half = (KB == 128) ? (c >= 4) : (half != 0);
    const uint32_t delta = (uint32_t)(hi_half ? m.d_hi : m.d_lo);
    const uint32_t tap = hi_half ? ((m.packed >> 8) & 0xffu) : (m.packed & 0xffu);
    const uint32_t cidx = (KB == 128) ? (uint32_t)c : (uint32_t)(half * 4 + c);
    const uint32_t chunk_ok = (cidx < ((m.packed >> 16) & 0xffu)) ? 1u : 0u;
    const int src = (NSRC == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)(m.packed >> 24));
    char* const xt = smem + stage * STAGE + BCO * KB + wave * 1024;
    if (NSRC == 1 || src == 0) {
#pragma unroll
      for (int i = 0; i < XROWS; ++i) {
        const uint32_t ok = chunk_ok & (xmask[i] >> tap) & 1u;
        const uint32_t off = ok ? xbase[0][i] + delta : kInvalidOff;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x0, (lds_ptr_t)(xt + i * RPP * KB), 16, (int)off, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < XROWS; ++i) {
        const uint32_t ok = chunk_ok & (xmask[i] >> tap) & 1u;
        const uint32_t off = ok ? xbase[NSRC - 1][i] + delta : kInvalidOff;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x1, (lds_ptr_t)(xt + i * RPP * KB), 16, (int)off, 0, 0, 0);
      }
    }
  };

  auto load_frag = [&](int stage, int kk, u32x4 (&a)[TCO], u32x4 (&b)[TPX]) {
    const char* wt = smem + stage * STAGE;
    const char* xt = wt + BCO * KB;
#pragma unroll
    for (int i = 0; i < TCO; ++i) a[i] = *reinterpret_cast<const u32x4*>(wt + swz<KB>((wco * TCO + i) * MT + fr, CHK * kk + fh));
#pragma unroll
    for (int j = 0; j < TPX; ++j) b[j] = *reinterpret_cast<const u32x4*>(xt + swz<KB>((wpx * TPX + j) * MT + fr, CHK * kk + fh));
  };
  auto mma_step = [&](const u32x4 (&a)[TCO], const u32x4 (&b)[TPX]) {
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
      for (int j = 0; j < TPX; ++j) {
        Mma<T, MT>::run(a[i], b[j], acc[i][j]);
      }
  };

  // ---- main loop: NS-deep LDS ring filled by LDS-DMA, one barrier per ring step -------------------------
  // A ring step is KB bytes of K per row (a whole 128-byte slice, or one 64-byte half).  NS-1 steps are in flight
  // ahead of the MFMAs.  Step t: wait (counted vmcnt) until this wave's DMA of step t has landed, barrier (=> every
  // wave's part has landed AND every wave is past step t-1, so its stage is free), issue step t+NS-1 into that
  // stage in the shadow of the first MFMA groups, compute step t.  Fragment registers are double-buffered.
  constexpr int NDMA = WROWS + XROWS;            // LDS-DMA instructions per ring step per wave
  const int T_ = p.slices_per_class * HPS;
#pragma unroll
  for (int j = 0; j < NS - 1; ++j) {
    if (j < T_) {
      const SliceMeta mj = meta[sbase + j / HPS];
      issue_w(j, j);
      issue_x(j, mj, j);
    }
  }
  // XB (cross-barrier pipelining, used where two fragment sets fit the register file): the LAST k-step of ring
  // step t is computed AFTER the barrier that opens step t+1, so its MFMAs cover the latency of step t+1's first
  // fragment reads and the issue slots of the next LDS-DMA.  Without it every wave leaves the barrier with empty
  // fragment registers and the matrix cores idle until the first ds_reads return (all waves in lock step).
  constexpr bool XB = MT == 32 || WCO * WPX == 4;
  if constexpr (X3) {
    // Split-product loop: a ring step holds KB / 64 sub-steps of K = 16.  Per sub-step a lane reads its weight fragments
    // ready-made (hi: chunk 4m + h, lo: chunk 4m + 2 + h of the row) and two fp32 chunks of every pixel fragment
    // (k = 16m + 8h .. + 7, the same K order as the weight fragment), splits the latter in registers and issues
    // three 32x32x16 fp16 MFMAs per tile pair.  One fragment set: with two waves per SIMD the partner's MFMAs cover
    // this wave's reads and conversions.
    static_assert(MT == 32, "the split product runs on 32x32x16 MFMAs");
    constexpr int SUB = KB / 64;
    int st_c = 0, st_i = NS - 1;
    SliceMeta m = meta[sbase + (NS - 1 < T_ ? NS - 1 : 0) / HPS];
    // One ring step.  SINGLE (okp_conv_create_x3, tap_terms = 1): only x_hi * w_hi - no low halves are read, split or multiplied.
    // The plan puts its single-term slices first, so the K loop is two loops over the same ring, each with one straight body
    // (a per-step branch between the two bodies made hipcc spill 450 registers).
    auto ring_step = [&](int t, auto single_tag) {
      constexpr bool SINGLE = decltype(single_tag)::value;
      const int nxt = t + NS - 1;
      const bool more = nxt < T_;
      if (NS > 2 && t + NS - 2 < T_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const char* wt = smem + st_c * STAGE;
      const char* xt = wt + BCO * KB;
#pragma unroll
      for (int mm = 0; mm < SUB; ++mm) {
        // pixel fragments one at a time (raw fp32 of fragment j + 1 in flight under the MFMAs of fragment j): the split
        // needs 16 registers per fragment on top of the 128 accumulators, so only two fragments are ever live
        u32x4 ah[TCO], al[TCO], r0[2], r1[2];
        auto read_b = [&](int j, int slot) {
          const int row = (wpx * TPX + j) * MT + fr;
          r0[slot] = *reinterpret_cast<const u32x4*>(xt + swz<KB>(row, 4 * mm + 2 * fh));
          r1[slot] = *reinterpret_cast<const u32x4*>(xt + swz<KB>(row, 4 * mm + 2 * fh + 1));
        };
        read_b(0, 0);
#pragma unroll
        for (int i = 0; i < TCO; ++i) {
          const int row = (wco * TCO + i) * MT + fr;
          ah[i] = *reinterpret_cast<const u32x4*>(wt + swz<KB>(row, 4 * mm + fh));
          if constexpr (!SINGLE) al[i] = *reinterpret_cast<const u32x4*>(wt + swz<KB>(row, 4 * mm + 2 + fh));
        }
        if (mm == 0 && more) issue_w(nxt, st_i);
        if (mm == SUB - 1 && more) issue_x(nxt, m, st_i);
#pragma unroll
        for (int j = 0; j < TPX; ++j) {
          if (j + 1 < TPX) read_b(j + 1, (j + 1) & 1);
          if constexpr (SINGLE) {
            const u32x4 bh = okp_hi8(r0[j & 1], r1[j & 1]);
#pragma unroll
            for (int i = 0; i < TCO; ++i) acc[i][j] = H16<_Float16>::mfma32(ah[i], bh, acc[i][j]);
          } else {
            u32x4 bh, bl;
            okp_split8(r0[j & 1], r1[j & 1], bh, bl);
#pragma unroll
            for (int i = 0; i < TCO; ++i) Mma<F32S, 32>::run3(ah[i], al[i], bh, bl, acc[i][j]);
          }
        }
      }
      m = meta[sbase + (nxt + 1 < T_ ? nxt + 1 : T_ - 1) / HPS];
      st_c = (st_c + 1 == NS) ? 0 : st_c + 1;
      st_i = (st_i + 1 == NS) ? 0 : st_i + 1;
    };
    const int T1 = p.n_single_slices * HPS;          // ring steps of the leading single-term slices (0 for all-three-term plans)
    int t = 0;
    for (; t < T1; ++t) ring_step(t, std::integral_constant<bool, true>{});
    for (; t < T_; ++t) ring_step(t, std::integral_constant<bool, false>{});
  } else if constexpr (XB) {
    auto mma_half = [&](const u32x4 (&a)[TCO], const u32x4 (&b)[TPX], auto half) {     // first / second half of the rows
      constexpr int I0 = decltype(half)::value ? (TCO + 1) / 2 : 0;
      constexpr int I1 = decltype(half)::value ? TCO : (TCO + 1) / 2;
#pragma unroll
      for (int i = I0; i < I1; ++i)
#pragma unroll
        for (int j = 0; j < TPX; ++j) Mma<T, MT>::run(a[i], b[j], acc[i][j]);
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    u32x4 fa[2][TCO], fb[2][TPX];
    int st_c = 0;
    if (NS > 2 && NS - 2 < T_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_frag(0, 0, fa[0], fb[0]);
    if (NS - 1 < T_) {
      const SliceMeta m0 = meta[sbase + (NS - 1) / HPS];
      issue_w(NS - 1, NS - 1);
      issue_x(NS - 1, m0, NS - 1);
    }
    SliceMeta m = meta[sbase + (NS < T_ ? NS : T_ - 1) / HPS];
    constexpr int L = (KSTEPS - 1) & 1;          // fragment set of the last k-step
    static_assert(KSTEPS % 2 == 0, "the first k-step of every ring step uses set 0");
    for (int t = 0; t + 1 < T_; ++t) {
#pragma unroll
      for (int kk = 0; kk + 1 < KSTEPS; ++kk) {
        load_frag(st_c, kk + 1, fa[(kk + 1) & 1], fb[(kk + 1) & 1]);
        mma_step(fa[kk & 1], fb[kk & 1]);
      }
      const int st_n = (st_c + 1 == NS) ? 0 : st_c + 1;
      const int nxt = t + NS;
      const bool more = nxt < T_;
      // my reads of stage st_c have returned (it is overwritten after the barrier); step t+1 has landed
      __builtin_amdgcn_sched_barrier(0);           // the scheduler may not hoist the last k-step's MFMAs above the barrier
      if (NS > 2 && t + NS - 1 < T_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      load_frag(st_n, 0, fa[0], fb[0]);
      __builtin_amdgcn_sched_barrier(0);           // ... nor sink the next step's first reads below them
      mma_half(fa[L], fb[L], H0{});
      if (more) issue_w(nxt, st_c);
      mma_half(fa[L], fb[L], H1{});
      if (more) issue_x(nxt, m, st_c);
      m = meta[sbase + (nxt + 1 < T_ ? nxt + 1 : T_ - 1) / HPS];
      st_c = st_n;
    }
#pragma unroll
    for (int kk = 0; kk + 1 < KSTEPS; ++kk) {      // last ring step: nothing left to prefetch
      load_frag(st_c, kk + 1, fa[(kk + 1) & 1], fb[(kk + 1) & 1]);
      mma_step(fa[kk & 1], fb[kk & 1]);
    }
    mma_step(fa[L], fb[L]);
  } else {
    int st_c = 0, st_i = NS - 1;                   // stage being computed / stage being filled
    SliceMeta m = meta[sbase + (NS - 1 < T_ ? NS - 1 : 0) / HPS];
    for (int t = 0; t < T_; ++t) {
      const int nxt = t + NS - 1;
      const bool more = nxt < T_;
      // steps t+1 .. t+NS-2 may stay in flight; in the tail fewer were issued, so drain completely
      // (lgkmcnt(0): this wave's fragment reads of step t-1 have returned before the barrier frees their stage)
      if (NS > 2 && t + NS - 2 < T_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if constexpr (MT == 16) {
        // 16x16 tiles: a k-step's fragments are 12 registers x 4, so only ONE set is live (the second set would spill
        // at 256 VGPRs); the B fragments are split in halves so the first MFMAs start after 8 of the 12 reads
        u32x4 a[TCO], b[TPX];
  #pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
          load_frag(st_c, kk, a, b);
          if (kk == 0 && more) issue_w(nxt, st_i);
          if (kk == 1 && more) issue_x(nxt, m, st_i);     // (issuing it with the weights, a k-step earlier, measured 2 % slower)
          mma_step(a, b);
        }
      } else {
        u32x4 a0[TCO], b0[TPX], a1[TCO], b1[TPX];
        load_frag(st_c, 0, a0, b0);
        load_frag(st_c, 1, a1, b1);
        if (more) issue_w(nxt, st_i);
        mma_step(a0, b0);
        if constexpr (KSTEPS > 2) load_frag(st_c, 2, a0, b0);
        if (more) issue_x(nxt, m, st_i);
        mma_step(a1, b1);
        if constexpr (KSTEPS > 2) {
          load_frag(st_c, 3, a1, b1);
          mma_step(a0, b0);
          mma_step(a1, b1);
        }
      }
      m = meta[sbase + (nxt + 1 < T_ ? nxt + 1 : T_ - 1) / HPS];   // constants for the next issue, read a step early
      st_c = (st_c + 1 == NS) ? 0 : st_c + 1;
      st_i = (st_i + 1 == NS) ? 0 : st_i + 1;
    }
  }
  __syncthreads();                               // all waves done with the last stage before it is reused

  // ---- epilogue: bias in registers, transpose through LDS in the OUTPUT type, coalesced NHWC rows -----------
  // Every lane owns 4 consecutive channels of one pixel per register group; it adds the bias, rounds to T and
  // writes them to a [pixel][channel] LDS image (16-B chunks XOR-swizzled by the pixel row).  With bf16 the whole
  // 256x256 tile fits the ring's LDS, so all waves stage at once and the tile leaves in ONE pass; the residual is
  // added after the read-back (bf16 + residual in fp32, ReLU, one more rounding).
  const bool dense_out = p.out_step == 1 && p.OH == p.Ho && p.OW == p.Wo;   // output pixel index == GEMM pixel index
  {
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
    if (PASSES == 1 || wpx / WPP == pass) {
#pragma unroll
      for (int i = 0; i < TCO; ++i) {
#pragma unroll
        for (int g = 0; g < AREGS / 4; ++g) {
          // accumulator rows: 32x32 -> 8g + 4*(lane>>5) + e ; 16x16 -> 4*(lane>>4) + e  (4 consecutive channels per lane)
          const int co_l = (wco * TCO + i) * MT + (MT == 32 ? 8 * g + 4 * fh : 4 * fh);
          const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_lds + co_l * 4);
          f32x4 sv = {1.f, 1.f, 1.f, 1.f};
          if constexpr (X3) sv = *reinterpret_cast<const f32x4*>(bias_lds + 1024 + co_l * 4);
#pragma unroll
          for (int j = 0; j < TPX; ++j) {
            const int prow = ((wpx % WPP) * TPX + j) * MT + fr;
            char* dst = smem + prow * PITCH + ((((co_l * ESZ) >> 4) ^ (prow & 7)) << 4) + ((co_l * ESZ) & 15);
            if constexpr (ESZ == 2) {
              x4_t o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = (T)(acc[i][j][4 * g + e] + bv[e]);
              *reinterpret_cast<x4_t*>(dst) = o;
            } else {
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = X3 ? __builtin_fmaf(acc[i][j][4 * g + e], sv[e], bv[e]) : acc[i][j][4 * g + e] + bv[e];
              *reinterpret_cast<f32x4*>(dst) = v;
            }
          }
        }
      }
    }
    __syncthreads();
    constexpr int GROUPS = BCO / 8;
    constexpr int ITEMS = PX_PER_PASS * GROUPS;
    constexpr int U = ITEMS / NT;                  // 8-channel output items per thread and pass
    constexpr int UB = U >= 2 ? 2 : 1;             // items per batch: their LDS reads are in flight together
    constexpr int CH8 = 8 * ESZ / 16;              // 16-B chunks per item (1 for bf16, 2 for fp32)
    static_assert(ITEMS % NT == 0 && U % UB == 0 && (U < 8 || (U / 4) % UB == 0 || U < 16), "epilogue work split");
    // Residual vectors: requested UH items at a time (the accumulators are dead - they were staged above - so registers
    // are available), instead of one exposed memory round trip per batch of two items.
    constexpr int UH = U >= 16 ? U / 4 : (U >= 8 ? U / 2 : U);
    float range_m = 0.f;                           // split-product plans: largest |result| of this pass (okp_range_max); local to the epilogue, so that
                                                   // nothing of the guard is live across the K loop
#pragma unroll 1
    for (int ub = 0; ub < U; ub += UH) {
    u32x4 rres[UH][CH8];
    if (p.res) {
#pragma unroll
      for (int u = 0; u < UH; ++u) {
        const int it = tid + (ub + u) * NT;
        const int q = it % GROUPS, prow = it / GROUPS;
        const int pix = px0 + pass * PX_PER_PASS + prow;
        const int co = co0 + q * 8;
        const bool ok = pix < P && co < p.cout;
        size_t opix = (size_t)(ok ? pix : 0);
        if (!dense_out) {
          const int pp = ok ? pix : 0;
          const int n = fastdiv(pp, p.div_howo);
          const int rem = pp - n * HoWo;
          const int ho = fastdiv(rem, p.div_wo);
          const int wo = rem - ho * p.Wo;
          opix = ((size_t)n * p.OH + (size_t)(ho * p.out_step + out_oy)) * p.OW + (size_t)(wo * p.out_step + out_ox);
        }
        if (X3 && p.res16) {        // fp16 residual of an fp32 plan (the branch output of a residual block in the mixed configuration)
          rres[u][0] = *reinterpret_cast<const u32x4*>(static_cast<const char*>(p.res) + (opix * p.res_pix_stride + (ok ? co : 0)) * 2);
        } else {
          const char* rp = static_cast<const char*>(p.res) + (opix * p.res_pix_stride + (ok ? co : 0)) * ESZ;
#pragma unroll
          for (int h = 0; h < CH8; ++h) rres[u][h] = *reinterpret_cast<const u32x4*>(rp + 16 * h);
        }
      }
    }
#pragma unroll
    for (int u0 = 0; u0 < UH; u0 += UB) {
      u32x4 raw[UB][CH8];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int it = tid + (ub + u0 + u) * NT;
        const int q = it % GROUPS, prow = it / GROUPS;
#pragma unroll
        for (int h = 0; h < CH8; ++h)
          raw[u][h] = *reinterpret_cast<const u32x4*>(smem + prow * PITCH + (((CH8 * q + h) ^ (prow & 7)) << 4));
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int it = tid + (ub + u0 + u) * NT;
        const int q = it % GROUPS, prow = it / GROUPS;
        const int pix = px0 + pass * PX_PER_PASS + prow;
        const int co = co0 + q * 8;
        if (pix < P && co < p.cout) {
          size_t opix = (size_t)pix;
          if (!dense_out) {
            const int n = fastdiv(pix, p.div_howo);
            const int rem = pix - n * HoWo;
            const int ho = fastdiv(rem, p.div_wo);
            const int wo = rem - ho * p.Wo;
            opix = ((size_t)n * p.OH + (size_t)(ho * p.out_step + out_oy)) * p.OW + (size_t)(wo * p.out_step + out_ox);
          }
          size_t opix32 = opix;                    // pixel index in the fp32 output (differs from the fp16 copy's under out_sub2)
          bool st32 = true;
          if constexpr (X3) {
            st32 = p.out != nullptr;
            if (p.out_sub2) {                      // fp32 result kept at even rows / columns only, in a tensor of half the size
              const int n = fastdiv(pix, p.div_howo);
              const int rem = pix - n * HoWo;
              const int ho = fastdiv(rem, p.div_wo);
              const int wo = rem - ho * p.Wo;
              st32 = st32 && !((ho | wo) & 1);
              opix32 = ((size_t)n * p.OH2 + (size_t)(ho >> 1)) * p.OW2 + (size_t)(wo >> 1);
            }
          }
          char* op = static_cast<char*>(p.out) + (opix32 * p.out_pix_stride + co) * ESZ;
          if (!p.res && ESZ == 2) {                // no residual: ReLU on the packed 16-bit words, store as read
            u32x4 w = raw[u][0];
            if (p.act == OKP_ACT_RELU) {
              // bf16 / fp16 read as int16 keep the sign and the order of positive values: max(x, 0) clears negatives
              // (one v_pk_max_i16 per register instead of unpack / compare / select)
              typedef short s16x8 __attribute__((ext_vector_type(8)));
              const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
              w = __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8, w), z));
            }
            *reinterpret_cast<u32x4*>(op) = w;
          } else {
            float v[8];
            if constexpr (ESZ == 2) {
              const x8_t s8 = __builtin_bit_cast(x8_t, raw[u][0]);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (float)s8[e];
            } else {
              const f32x4 a4 = __builtin_bit_cast(f32x4, raw[u][0]), b4 = __builtin_bit_cast(f32x4, raw[u][CH8 - 1]);
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[e] = a4[e]; v[4 + e] = b4[e]; }
            }
            if (p.res) {
              if constexpr (ESZ == 2) {
                const x8_t r8 = __builtin_bit_cast(x8_t, rres[u0 + u][0]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
              } else if (X3 && p.res16) {
                const f16x8 r8 = __builtin_bit_cast(f16x8, rres[u0 + u][0]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
              } else {
                const f32x4 ra = __builtin_bit_cast(f32x4, rres[u0 + u][0]), rb = __builtin_bit_cast(f32x4, rres[u0 + u][CH8 - 1]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += ra[e]; v[4 + e] += rb[e]; }
              }
            }
            if (p.act == OKP_ACT_RELU) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if constexpr (X3) {
#pragma unroll
              for (int e = 0; e < 8; ++e) range_m = okp_range_max(range_m, v[e]);
            }
            if (st32) Io<T>::store8(v, op);
            if constexpr (X3) {
              if (p.out16) {             // fp16 shadow of the result (what a single-term consumer on the fp16 kernels reads)
                f16x8 o16;
#pragma unroll
                for (int e = 0; e < 8; ++e) o16[e] = (_Float16)v[e];
                *reinterpret_cast<f16x8*>(static_cast<char*>(p.out16) + (opix * p.out16_pix_stride + co) * 2) = o16;
              }
            }
          }
        }
      }
    }
    }  // ub
    if constexpr (X3) okp_raise_range_flag(p.range_flag, okp_range_exceeded(range_m));
    __syncthreads();       // staging is free again (next pass, or the next tile's LDS-DMA)
  }
  }

  // ---- optional fused depth-wise 3x3 branch over the same pixels / channel range (fire-module tail) -------
  // A thread owns one 16-byte channel group (its 9 x VN weights live in registers) and SEG consecutive pixels,
  // processed in groups of G: with stride 1 and a group inside one image row the 3 x (G+2) taps are loaded once
  // and slid across the G outputs (18 loads for 4 pixels instead of 36).  Branch-free zero padding: a tap in the
  // halo gets an out-of-range buffer offset, so all taps of a group are in flight together (L2-hot: the GEMM just
  // streamed this tensor).
  if (p.dw_w) {
    constexpr int VN = 16 / ESZ;
    constexpr int CG = BCO / VN;
    constexpr int PSEG = NT / CG;
    constexpr int SEG = BPX / PSEG;
    constexpr int G = SEG >= 4 ? 4 : SEG;
    static_assert(NT % CG == 0 && BPX % PSEG == 0 && SEG % G == 0, "depth-wise work split");
    const int cq = tid % CG, pl = tid / CG;
    const int ch = co0 + cq * VN;
    float dw_range_m = 0.f;
    if (ch < p.cout) {
      float wreg[9][VN], breg[VN];
#pragma unroll
      for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < VN; ++e) wreg[k][e] = p.dw_w[k * p.cout + ch + e];
#pragma unroll
      for (int e = 0; e < VN; ++e) breg[e] = p.dw_bias[ch + e];
      const int cs = p.conv_stride[0], H = p.srcH[0], W = p.srcW[0], ps = p.src_pix_stride[0];
      const bool slide = cs == 1 && (p.Wo % G) == 0;     // groups start at multiples of G: never cross a row
      auto tap_off = [&](int n, int hi, int wi) -> uint32_t {
        const bool ok = hi >= 0 && hi < H && wi >= 0 && wi < W;
        return ok ? (uint32_t)(((n * H + hi) * W + wi) * ps + ch) * (uint32_t)ESZ : kInvalidOff;
      };
      auto to_f = [&](const u32x4& raw, float (&x)[VN]) {
        if constexpr (ESZ == 2) {
          const x8_t xv = __builtin_bit_cast(x8_t, raw);
#pragma unroll
          for (int e = 0; e < VN; ++e) x[e] = (float)xv[e];
        } else {
          const f32x4 xv = __builtin_bit_cast(f32x4, raw);
#pragma unroll
          for (int e = 0; e < VN; ++e) x[e] = xv[e];
        }
      };
      auto finish = [&](float (&v)[VN], int n, int ho, int wo) {
        const size_t opix = ((size_t)n * p.OH + (size_t)(ho + out_oy)) * p.OW + (size_t)(wo + out_ox);
        if (p.dw_res) {
          float r[VN];
          to_f(*reinterpret_cast<const u32x4*>(static_cast<const char*>(p.dw_res) + (opix * p.dw_res_pix_stride + ch) * ESZ), r);
#pragma unroll
          for (int e = 0; e < VN; ++e) v[e] += r[e];
        }
        if (p.act == OKP_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < VN; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (X3) {
#pragma unroll
          for (int e = 0; e < VN; ++e) dw_range_m = okp_range_max(dw_range_m, v[e]);
        }
        char* op = static_cast<char*>(p.dw_out) + (opix * p.dw_out_pix_stride + ch) * ESZ;
        if constexpr (ESZ == 2) {
          x8_t o;
#pragma unroll
          for (int e = 0; e < VN; ++e) o[e] = (T)v[e];
          *reinterpret_cast<x8_t*>(op) = o;
        } else {
          f32x4 o = {v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(op) = o;
        }
      };
      for (int g0 = 0; g0 < SEG; g0 += G) {
        const int pix0 = px0 + pl * SEG + g0;
        if (slide) {
          if (pix0 < P) {                                  // Wo % G == 0  =>  the whole group is in range
            const int n = fastdiv(pix0, p.div_howo);
            const int rem = pix0 - n * HoWo;
            const int ho = fastdiv(rem, p.div_wo);
            const int wo0 = rem - ho * p.Wo;
            u32x4 win[3][G + 2];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
              for (int cc = 0; cc < G + 2; ++cc)
                win[r][cc] = __builtin_amdgcn_raw_buffer_load_b128(rs_x0, (int)tap_off(n, ho + r - 1, wo0 + cc - 1), 0, 0);
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
                  for (int e = 0; e < VN; ++e) v[e] = fmaf(x[e], wreg[r * 3 + c3][e], v[e]);
                }
              finish(v, n, ho, wo0 + k);
            }
          }
        } else {
          for (int k = 0; k < G; ++k) {
            const int pix = pix0 + k;
            if (pix >= P) break;
            const int n = fastdiv(pix, p.div_howo);
            const int rem = pix - n * HoWo;
            const int ho = fastdiv(rem, p.div_wo);
            const int wo = rem - ho * p.Wo;
            u32x4 tapv[9];
#pragma unroll
            for (int t = 0; t < 9; ++t)
              tapv[t] = __builtin_amdgcn_raw_buffer_load_b128(rs_x0, (int)tap_off(n, ho * cs + t / 3 - 1, wo * cs + t % 3 - 1), 0, 0);
            float v[VN];
#pragma unroll
            for (int e = 0; e < VN; ++e) v[e] = breg[e];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
              float x[VN];
              to_f(tapv[t], x);
#pragma unroll
              for (int e = 0; e < VN; ++e) v[e] = fmaf(x[e], wreg[t][e], v[e]);
            }
            finish(v, n, ho, wo);
          }
        }
      }
    }
    if constexpr (X3) okp_raise_range_flag(p.range_flag, okp_range_exceeded(dw_range_m));
  }
  }  // tile loop
}

template <typename T, int BCO, int BPX, int WCO, int WPX, int NS, int KB, int MT = 32>
int launch_cfg(const okp_conv* plan, OkpIgemmParams p, hipStream_t stream) {
  const int P = p.N * p.Ho * p.Wo;
  p.n_co_tiles = (p.cout_pad + BCO - 1) / BCO;
  const int n_px_tiles = (P + BPX - 1) / BPX;
  p.tiles_per_class = p.n_co_tiles * n_px_tiles;
  p.n_tiles = p.tiles_per_class * p.n_classes;
  constexpr int kLds = NS * (BCO + BPX) * KB + kMetaMax * (int)sizeof(SliceMeta) + 1024;
  constexpr int kPerCu = (160 * 1024 / kLds) < 1 ? 1 : (160 * 1024 / kLds > 4 ? 4 : 160 * 1024 / kLds);
  const int resident = 256 * kPerCu;              // MI355X: 256 CUs
  const dim3 grid((unsigned)(p.n_tiles < resident ? p.n_tiles : resident));
  const dim3 block(64 * WCO * WPX);
  if (plan->n_src == 1)
    hipLaunchKernelGGL((okp_igemm_kernel<T, BCO, BPX, WCO, WPX, NS, KB, MT, 1>), grid, block, 0, stream, p);
  else
    hipLaunchKernelGGL((okp_igemm_kernel<T, BCO, BPX, WCO, WPX, NS, KB, MT, 2>), grid, block, 0, stream, p);
  return okp_check_hip(hipGetLastError(), "okp_igemm launch");
}

}  // namespace

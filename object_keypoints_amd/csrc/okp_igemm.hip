// Tap-list implicit-GEMM convolution for gfx950 (MI355X), fp32 and bf16.
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
#include "okp_internal.h"

namespace {

constexpr uint32_t kInvalidOff = 0x80000000u;   // >= num_records of every tensor we accept (< 2 GiB)

template <typename T> struct Mma;
template <> struct Mma<float> {
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
template <> struct Mma<__bf16> {
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
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
template <> struct Io<__bf16> {
  static constexpr int kBytes8 = 16;
  static __device__ __forceinline__ void add8(float (&v)[8], const char* p) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += (float)a[e];
  }
  static __device__ __forceinline__ void store8(const float (&v)[8], char* p) {
    bf16x8 a;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x8*>(p) = a;
  }
};

__device__ __forceinline__ int fastdiv(int x, const OkpFastDiv& f) {
  return f.mul ? (int)(__umulhi((uint32_t)x, f.mul) >> f.shift) : x;
}

__device__ __forceinline__ uint32_t swz(int row, int chunk) {   // byte offset of a 16-B chunk in a [row][128 B] tile
  return (uint32_t)row * 128u + (uint32_t)((chunk ^ ((row >> 1) & 7)) << 4);
}

struct SliceMeta {            // per-slice gather constants, built once per workgroup in LDS
  int32_t d_lo, d_hi;         // byte deltas of the two 64-byte halves relative to the pixel's base offset
  uint32_t packed;            // tap_lo | tap_hi << 8 | nvalid << 16 | src << 24
  uint32_t pad;
};
constexpr int kMetaMax = 256;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <typename T, int BCO, int BPX, int WCO, int WPX, int NSRC>
__global__ __launch_bounds__(64 * WCO * WPX) void okp_igemm_kernel(const OkpIgemmParams p) {
  constexpr int NT = 64 * WCO * WPX;
  constexpr int ESZ = (int)sizeof(T);
  constexpr int RPP = NT / 8;                 // tile rows covered by one loader pass (8 chunks per row)
  constexpr int WROWS = BCO / RPP;
  constexpr int XROWS = BPX / RPP;
  constexpr int TCO = BCO / WCO / 32;
  constexpr int TPX = BPX / WPX / 32;
  constexpr int STAGE = (BCO + BPX) * 128;
  constexpr int LDS_BYTES = 2 * STAGE;
  constexpr int PASSES = (BPX * BCO * 4 > LDS_BYTES) ? 2 : 1;
  static_assert(RPP % 16 == 0, "loader swizzle assumes the pass height keeps (row>>1)&7");
  static_assert(PASSES == 1 || WPX == 2, "two-pass epilogue splits pixels by wave column");
  static_assert(BPX * BCO * 4 / PASSES <= LDS_BYTES, "epilogue staging must fit");
  constexpr int PX_PER_PASS = BPX / PASSES;
  constexpr int PITCH = BCO * 4;

  // ONE LDS object (a second __shared__ array makes hipcc drain LDS-DMA before unrelated ds_reads)
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES + kMetaMax * (int)sizeof(SliceMeta)];
  SliceMeta* const meta = reinterpret_cast<SliceMeta*>(smem + LDS_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave / WPX, wpx = wave % WPX;
  const int co_tile = blockIdx.x % p.n_co_tiles;
  const int px_tile = blockIdx.x / p.n_co_tiles;
  const int co0 = co_tile * BCO, px0 = px_tile * BPX;
  const int HoWo = p.Ho * p.Wo;
  const int P = p.N * HoWo;

  // Loader geometry.  One `buffer_load_dwordx4 ... lds` writes 64 lanes x 16 B = 8 tile rows linearly into LDS,
  // so lane (row r0 = tid>>3, position tid&7) must FETCH the logical chunk that the read-side swizzle expects
  // at that position: c = pos ^ ((row>>1)&7)  (swizzle on the source address, linear destination).
  const int r0 = tid >> 3;
  const int c = (tid & 7) ^ ((r0 >> 1) & 7);

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.weights), 0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src[0]), 0, (int)p.src_bytes[0], 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src[NSRC - 1]), 0, (int)p.src_bytes[NSRC - 1], 0x00020000);

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
    m.pad = 0;
    meta[s] = m;
  }

  // ---- per-thread row state -------------------------------------------------------------
  uint32_t wbase[WROWS];
#pragma unroll
  for (int i = 0; i < WROWS; ++i) {
    const int co = co0 + r0 + i * RPP;
    wbase[i] = (co < p.cout_pad) ? (uint32_t)co * 128u + (uint32_t)c * 16u : kInvalidOff;
  }
  uint32_t xbase[NSRC][XROWS];
  uint32_t xmask[XROWS];
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
      xbase[s][i] = (uint32_t)(((n * p.srcH[s] + hi0) * p.srcW[s] + wi0) * p.src_pix_stride[s]) * (uint32_t)ESZ + (uint32_t)(c & 3) * 16u;
    }
    uint32_t m = 0;
    for (int t = 0; t < p.n_taps; ++t) {
      const int s = (NSRC == 1) ? 0 : p.taps[t].src;
      const int hi = ho * p.conv_stride[s] + p.taps[t].dy;
      const int wi = wo * p.conv_stride[s] + p.taps[t].dx;
      const bool ok = valid && hi >= 0 && hi < p.srcH[s] && wi >= 0 && wi < p.srcW[s];
      m |= (ok ? 1u : 0u) << t;
    }
    xmask[i] = m;
  }

  f32x16 acc[TCO][TPX];
#pragma unroll
  for (int i = 0; i < TCO; ++i)
#pragma unroll
    for (int j = 0; j < TPX; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // LDS-DMA gather of slice `s` (constants `m`) into `stage`, in two parts so that the issue slots can be
  // interleaved with the MFMAs of the slice being computed.  Masked lanes (padding halo, rows beyond the
  // problem, chunks beyond Cin) use an offset past the buffer: the hardware range check then writes zeros
  // to LDS (verified by scripts/hwtests/dma_oob.hip).
  auto issue_w = [&](int s, int stage) {
    const uint32_t wslice = (uint32_t)s * (uint32_t)p.cout_pad * 128u;
    char* const wt = smem + stage * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < WROWS; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(wt + i * RPP * 128), 16, (int)(wbase[i] + wslice), 0, 0, 0);
  };
  auto issue_x = [&](const SliceMeta& m, int stage) {
    const bool hi_half = c >= 4;
    const uint32_t delta = (uint32_t)(hi_half ? m.d_hi : m.d_lo);
    const uint32_t tap = hi_half ? ((m.packed >> 8) & 0xffu) : (m.packed & 0xffu);
    const uint32_t chunk_ok = ((uint32_t)c < ((m.packed >> 16) & 0xffu)) ? 1u : 0u;
    const int src = (NSRC == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)(m.packed >> 24));
    char* const xt = smem + stage * STAGE + BCO * 128 + wave * 1024;
    if (NSRC == 1 || src == 0) {
#pragma unroll
      for (int i = 0; i < XROWS; ++i) {
        const uint32_t ok = chunk_ok & (xmask[i] >> tap) & 1u;
        const uint32_t off = ok ? xbase[0][i] + delta : kInvalidOff;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x0, (lds_ptr_t)(xt + i * RPP * 128), 16, (int)off, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < XROWS; ++i) {
        const uint32_t ok = chunk_ok & (xmask[i] >> tap) & 1u;
        const uint32_t off = ok ? xbase[NSRC - 1][i] + delta : kInvalidOff;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x1, (lds_ptr_t)(xt + i * RPP * 128), 16, (int)off, 0, 0, 0);
      }
    }
  };

  const int fr = lane & 31, fh = lane >> 5;
  auto load_frag = [&](int stage, int kk, u32x4 (&a)[TCO], u32x4 (&b)[TPX]) {
    const char* wt = smem + stage * STAGE;
    const char* xt = wt + BCO * 128;
#pragma unroll
    for (int i = 0; i < TCO; ++i) a[i] = *reinterpret_cast<const u32x4*>(wt + swz((wco * TCO + i) * 32 + fr, 2 * kk + fh));
#pragma unroll
    for (int j = 0; j < TPX; ++j) b[j] = *reinterpret_cast<const u32x4*>(xt + swz((wpx * TPX + j) * 32 + fr, 2 * kk + fh));
  };
  auto mma_step = [&](const u32x4 (&a)[TCO], const u32x4 (&b)[TPX]) {
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
      for (int j = 0; j < TPX; ++j) Mma<T>::run(a[i], b[j], acc[i][j]);
  };

  // ---- main loop: two LDS stages filled by LDS-DMA, one barrier per K-slice ---------------------------
  // iteration s:  [own DMA of slice s drained] barrier -> fragments of k-steps 0/1 -> DMA of slice s+1 into the
  // other stage (free: every wave is past slice s-1), issued in the shadow of the first MFMA groups -> remaining
  // k-steps.  Fragment registers are double-buffered so LDS latency hides behind the previous k-step's MFMAs.
  const int S = p.n_slices;
  __syncthreads();                               // slice constants visible
  SliceMeta m = meta[0];
  issue_w(0, 0);
  issue_x(m, 0);
  m = meta[S > 1 ? 1 : 0];
  for (int s = 0; s < S; ++s) {
    const int stage = s & 1;
    const bool more = (s + 1 < S) && !(p.debug & 1);
    __syncthreads();                             // hipcc drains vmcnt(0) here because LDS-DMA is in flight
    if (p.debug & 2) {
      if (more) { issue_w(s + 1, stage ^ 1); issue_x(m, stage ^ 1); m = meta[s + 2 < S ? s + 2 : s + 1]; }
      continue;
    }
    u32x4 a0[TCO], b0[TPX], a1[TCO], b1[TPX];
    load_frag(stage, 0, a0, b0);
    load_frag(stage, 1, a1, b1);
    if (more) issue_w(s + 1, stage ^ 1);
    mma_step(a0, b0);
    load_frag(stage, 2, a0, b0);
    if (more) issue_x(m, stage ^ 1);
    mma_step(a1, b1);
    load_frag(stage, 3, a1, b1);
    if (more) m = meta[s + 2 < S ? s + 2 : s + 1];   // constants for the next issue, read a full iteration early
    mma_step(a0, b0);
    mma_step(a1, b1);
  }
  __syncthreads();                               // all waves done with the last stage before it is reused

  // ---- epilogue: bias in registers, transpose through LDS, coalesced NHWC rows ----------------
  if (p.debug & 4) return;
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
    if (PASSES == 1 || wpx == pass) {
#pragma unroll
      for (int i = 0; i < TCO; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co_l = (wco * TCO + i) * 32 + 8 * g + 4 * fh;
          const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + co0 + co_l);   // bias is padded to n_co_tiles*BCO
#pragma unroll
          for (int j = 0; j < TPX; ++j) {
            const int prow = ((PASSES == 1 ? wpx * TPX : 0) + j) * 32 + fr;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + bv[e];
            *reinterpret_cast<f32x4*>(smem + prow * PITCH + (((co_l >> 2) ^ (prow & 7)) << 4)) = v;
          }
        }
      }
    }
    __syncthreads();
    constexpr int GROUPS = BCO / 8;
    constexpr int ITEMS = PX_PER_PASS * GROUPS;
    for (int it = tid; it < ITEMS; it += NT) {
      const int q = it % GROUPS;
      const int prow = it / GROUPS;
      const int pix = px0 + pass * PX_PER_PASS + prow;
      const int co = co0 + q * 8;
      if (pix < P && co < p.cout) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(smem + prow * PITCH + (((2 * q) ^ (prow & 7)) << 4));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(smem + prow * PITCH + (((2 * q + 1) ^ (prow & 7)) << 4));
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const int n = fastdiv(pix, p.div_howo);
        const int rem = pix - n * HoWo;
        const int ho = fastdiv(rem, p.div_wo);
        const int wo = rem - ho * p.Wo;
        const size_t opix = ((size_t)n * p.OH + (size_t)(ho * p.out_step + p.out_oy)) * p.OW + (size_t)(wo * p.out_step + p.out_ox);
        if (p.res) Io<T>::add8(v, static_cast<const char*>(p.res) + (opix * p.res_pix_stride + co) * ESZ);
        if (p.act == OKP_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        Io<T>::store8(v, static_cast<char*>(p.out) + (opix * p.out_pix_stride + co) * ESZ);
      }
    }
    if (pass + 1 < PASSES) __syncthreads();
  }
}

template <typename T, int BCO, int BPX, int WCO, int WPX>
int launch_cfg(const okp_conv* plan, OkpIgemmParams p, hipStream_t stream) {
  const int P = p.N * p.Ho * p.Wo;
  p.n_co_tiles = (p.cout_pad + BCO - 1) / BCO;
  const int n_px_tiles = (P + BPX - 1) / BPX;
  const dim3 grid((unsigned)(p.n_co_tiles * n_px_tiles));
  const dim3 block(64 * WCO * WPX);
  if (plan->n_src == 1)
    hipLaunchKernelGGL((okp_igemm_kernel<T, BCO, BPX, WCO, WPX, 1>), grid, block, 0, stream, p);
  else
    hipLaunchKernelGGL((okp_igemm_kernel<T, BCO, BPX, WCO, WPX, 2>), grid, block, 0, stream, p);
  return okp_check_hip(hipGetLastError(), "okp_igemm launch");
}

template <typename T>
int launch_tile(const okp_conv* plan, const OkpIgemmParams& p, int tile, hipStream_t stream) {
  switch (tile) {
    case 3: return launch_cfg<T, 256, 256, 4, 2>(plan, p, stream);
    case 2: return launch_cfg<T, 128, 128, 2, 2>(plan, p, stream);
    default: return launch_cfg<T, 64, 64, 2, 2>(plan, p, stream);
  }
}

}  // namespace

int okp_select_tile(int cout_pad, long P) {
  // Fill the 256 CUs first; only then grow the tile (bigger tiles re-read less from L2).
  auto tiles = [&](int b) { return ((P + b - 1) / b) * ((cout_pad + b - 1) / b); };
  if (tiles(256) >= 256 && cout_pad >= 192) return 3;
  if (tiles(128) >= 256) return 2;
  return 1;
}

int okp_launch_igemm(const okp_conv* plan, const OkpIgemmParams& p, int tile, hipStream_t stream) {
  if (tile == 0) tile = okp_select_tile(p.cout_pad, (long)p.N * p.Ho * p.Wo);
  if (plan->dtype == OKP_BF16) return launch_tile<__bf16>(plan, p, tile, stream);
  return launch_tile<float>(plan, p, tile, stream);
}

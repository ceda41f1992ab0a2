// Patch-resident split-product implicit GEMM: the 3x3 convolutions (and the 4x4/s2 transposed convolutions) of OKP_F32X3 plans
// with 256 output channels - fp32 tensors, every product as the three-term fp16 split x_hi w_hi + x_lo w_hi + x_hi w_lo on the
// fp16 matrix pipe (struct F32S in okp_igemm_kernel.h: what the split computes and why it is fp32-grade).
//
// okp_igemm_kernel<F32S> gathers the fp32 pixel operand of EVERY K-slice from L2 and splits it in registers: a 3x3 convolution
// moves each input line nine times into LDS and converts it nine times on the vector ALUs (4 instructions per 2 elements, in
// the K loop, beside the MFMAs).  Here the structure of okp_igemm_patch.hip is used instead: a workgroup owns 256 output
// channels x one 16x16-pixel block of one frame; per 32-CHANNEL chunk of a source the block's input patch including its halo
// (18x18 pixels x 128 bytes of fp32 = 40.5 KiB) is copied to LDS once by LDS-DMA while the previous chunk is being multiplied,
// and it is split ONCE, in place: the 32 bytes that held k = 8q .. 8q+7 as fp32 then hold [hi(k) | lo(k)] as 2 x 8 fp16 - the
// same 128 bytes per pixel, now in the order a 16x16x32 MFMA fragment wants them (lane k-group q reads one 16-byte chunk of
// each half).  The nine taps of the chunk are nine K-steps that read ready-made x_hi / x_lo fragments at a tap offset: no
// vector-ALU work and no second gather in the loop, three MFMAs per four fragment reads.  Weights stream through the same
// 2 x 32 KiB ring as in the 16-bit kernel, from the split-product blob okp_conv_create builds ([hi k0-15 | lo k0-15 | hi k16-31
// | lo k16-31] per 128-byte row of 32 k), unchanged.
//
// WHEN a patch is split is written into the step table at plan creation (OkpPatchStep.pbuf bits 1-2, okp_api.hip):
//   * a group (the K-steps of one chunk and geometry) of n >= 2 steps requests the next group's patch in its steps 0 .. n-2 and
//     splits it during step n-1 (OKP_PSTEP_CVT_NEXT): every request has landed at that step's barrier, the split is visible at
//     the next one;
//   * behind a one-step group (the projected 1x1 skip, the single-tap parity class of a stride-2 3x3) and at the first step of a
//     tile the patch is split at the start of its own first step, followed by a second barrier (OKP_PSTEP_CVT_SELF).
// Geometries, step table, sub-pixel classes, tile order and the per-tile offset table are those of okp_igemm_patch.hip.
//
// LDS: weights ring 2 x 32 KiB | patch buffers 2 x 41 KiB | step table 4 KiB | bias + output scale 2 KiB | offset tables 7.7 KiB
// = 159.7 KiB, one workgroup (8 waves, 4 x 2, wave tile 64 channels x 128 pixels on 16x16x32 fp16 MFMAs) per CU; the fp32 tile
// (256 KiB) leaves through the same LDS in two passes of 128 pixels, as whole 1 KiB pixel rows.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "okp_igemm_kernel.h"

namespace {

// Swizzle key of a patch pixel's eight 16-byte chunks, a function of its patch column pair.  A ds_read_b128 is served in four
// groups of 16 lanes - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) - so a group mixes
// two k-groups of a 16-row fragment: columns 0-3 / 12-15 of k-group 2H with columns 4-11 of k-group 2H + 1.  Lane k-group q reads
// chunk 2q (hi) or 2q + 1 (lo) here, where the 16-bit kernel reads chunk q: its key table with bits 0 and 1 of every entry
// exchanged (0,2,5,3,4,6,5,3,0 per column pair) is conflict-free for the column offsets 0, 1 and 2 of the taps, for both halves
// (checked against the lane groups above for every offset).
constexpr uint32_t kPatchKeysX3 = 0u | (2u << 3) | (5u << 6) | (3u << 9) | (4u << 12) | (6u << 15) | (5u << 18) | (3u << 21) | (0u << 24);
__device__ __forceinline__ uint32_t patch_key_x3(int col) { return (kPatchKeysX3 >> (3 * (col >> 1))) & 7u; }

constexpr int kWStage = 256 * 128;                 // one K-step of weights: 256 rows x (32 k as fp16 hi + lo = 128 B)
constexpr int kPitch = 18;                         // patch row pitch in pixels
constexpr int kPatchBuf = 41 * 1024;               // 18 x 18 px x 128 B, rounded up to whole 1 KiB LDS-DMA blocks
constexpr int kLdsPatch = 2 * kWStage;
constexpr int kLdsSteps = kLdsPatch + 2 * kPatchBuf;
constexpr int kLdsBias = kLdsSteps + 256 * (int)sizeof(OkpPatchStep);     // 256 biases, then 256 output scales
constexpr int kGeoEntries = 328;                   // 18 x 18 pixels, rounded up to whole 8-pixel LDS-DMA blocks
constexpr int kLdsGeo = kLdsBias + 2048;
constexpr int kLdsTotal = kLdsGeo + OKP_PATCH_MAX_GEOM * kGeoEntries * 4;
constexpr int kStagePx = 128;                      // pixels per epilogue pass (x 256 channels x 4 B = 128 KiB)
static_assert(sizeof(OkpPatchStep) == 16, "step table entries are read as one 16-byte vector");
static_assert(kStagePx * 1024 <= kLdsSteps, "epilogue staging must not reach the step table");
static_assert(kLdsTotal <= 160 * 1024, "one workgroup's LDS is at most the CU's 160 KiB");

// OUT_PAIRS: the output is written in pair format (okp_conv_args.out_pairs; an instantiation of its own: the two store loops in one kernel spill)
template <bool OUT_PAIRS>
__global__ __launch_bounds__(512) void okp_igemm_patch_x3_kernel(const OkpPatchParams p) {
  constexpr int TCO = 4, TPX = 8;                  // 16x16 accumulator tiles per wave: 64 channels x 128 pixels
  __shared__ __attribute__((aligned(16))) char smem[kLdsTotal];
  char* const steps_lds = smem + kLdsSteps;
  char* const bias_lds = smem + kLdsBias;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave >> 1, wpx = wave & 1;
  const int fr = lane & 15, fh = lane >> 4;
  bool range_bad = false;                          // some result of this launch leaves the fp16 range (okp_unsplittable: exact, a lane mask in scalar registers -
                                                   // measured free here, where a running maximum in a vector register cost 1.4 %)

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.weights), 0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.n_co_tiles * 256 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_sc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.oscale), 0, p.n_co_tiles * 256 * 4, 0x00020000);
  // (the sources' buffer resources are built per tile from the tile's FRAME - 64-bit base, frame-relative 32-bit offsets with bit 31 as the
  //  "outside" flag - so a source may span more than 2 GiB: the fp32 / pair-format stem output of a 64-frame batch is 2.1 GB)

  if (tid < p.n_steps) reinterpret_cast<u32x4*>(steps_lds)[tid] = reinterpret_cast<const u32x4*>(p.steps)[tid];
  __syncthreads();

  // weights loader: lane (row r0 = tid >> 3, position tid & 7) fetches the chunk the read-side swizzle expects there
  const int r0 = tid >> 3;
  const int wc = (tid & 7) ^ ((r0 >> 1) & 7);
  // weight fragment chunks of lane k-group fh (k = 8 fh .. 8 fh + 7 of the step's 32): hi at (fh & 1) + 4 (fh >> 1), lo two chunks on
  const int wch = (fh & 1) + 4 * (fh >> 1);

  for (int slot = blockIdx.x; slot < p.n_tiles; slot += gridDim.x) {
    // XCD-aware, class-minor tile order: see okp_igemm_patch.hip
    const int xq = p.n_tiles >> 3, xr = p.n_tiles & 7, xcd = slot & 7;
    const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (slot >> 3);
    const int cls = p.n_classes == 4 ? (tile & 3) : 0;
    const int tile_c = p.n_classes == 4 ? (tile >> 2) : tile;
    const int t0 = cls * p.steps_per_class, t1 = t0 + p.steps_per_class;      // this class's K-steps
    const int co_tile = tile_c % p.n_co_tiles;
    const int px_tile = tile_c / p.n_co_tiles;
    const int n = fastdiv(px_tile, p.div_tiles_frame);
    const int trem = px_tile - n * p.tiles_y * p.tiles_x;
    const int tyi = fastdiv(trem, p.div_tiles_x);
    const int y0 = tyi * 16, x0 = (trem - tyi * p.tiles_x) * 16;
    const int co0 = co_tile * 256;
    auto frame_rsrc = [&](int s_) {
      const int64_t off = (int64_t)n * p.src_frame_bytes[s_], left = p.src_total_bytes[s_] - off;
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(p.src_data[s_])) + off, 0, (int)(left < 0x7FFF0000ll ? left : 0x7FFF0000ll), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rs_s0 = frame_rsrc(0), rs_s1 = frame_rsrc(1);

    if (wave == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)bias_lds, 16, (int)((uint32_t)(co0 + lane * 4) * 4u), 0, 0, 0);
    if (wave == 1)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_sc, (lds_ptr_t)(bias_lds + 1024), 16, (int)((uint32_t)(co0 + lane * 4) * 4u), 0, 0, 0);

    uint32_t wbase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = co0 + r0 + i * 64;
      wbase[i] = (co < p.cout_pad) ? (uint32_t)co * 128u + (uint32_t)wc * 16u : kInvalidOff;
    }

    auto issue_w = [&](int t, int stage) {
      const uint32_t wslice = (uint32_t)t * (uint32_t)p.cout_pad * 128u;
      char* const wt = smem + stage * kWStage + wave * 1024;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(wt + i * 8192), 16, (int)(wbase[i] + wslice), 0, 0, 0);
    };
    // source offset (first channel, bytes) | column key of patch pixel idx of geometry G, or bit 31
    auto patch_pixel = [&](const OkpPatchGeom& G, int idx) -> uint32_t {
      const int i = idx / kPitch, j = idx - i * kPitch;
      const int ys = G.conv_stride * y0 + G.oy + i * G.step, xs = G.conv_stride * x0 + G.ox + j * G.step;
      const bool ok = idx < G.npx && j < G.PW && ys >= 0 && ys < G.H && xs >= 0 && xs < G.W;
      return ok ? ((uint32_t)(ys * G.W + xs) * (uint32_t)(G.pix_stride * 4)) | patch_key_x3(j) : kInvalidOff;      // (relative to frame n)
    };
    // passes [k0, k1) of a patch: pass k = 1 KiB blocks 8 k .. 8 k + 7 (one per wave) = patch pixels 64 k .. 64 k + 63; addresses
    // from the tile's offset table (no scalar loads in the loop)
    auto issue_patch_tab = [&](int geom, uint32_t c0b, int k0, int k1, int buf) {
      const int npx = 18 * (int)((p.geom_ph >> (5 * geom)) & 31u);
      const __amdgpu_buffer_rsrc_t rs_x = ((p.geom_src >> geom) & 1) ? rs_s1 : rs_s0;
      const uint32_t* const tab = reinterpret_cast<const uint32_t*>(smem + kLdsGeo) + geom * kGeoEntries + (lane >> 3);
      for (int k = k0; k < k1; ++k) {
        const int blk = k * 8 + wave;
        if (blk * 8 >= npx) continue;                           // wave-uniform: nothing of this block is inside the patch
        const uint32_t e = tab[blk * 8];
        const uint32_t off = (e & kInvalidOff) ? kInvalidOff : (e & ~7u) + c0b + ((((uint32_t)lane & 7u) ^ (e & 7u)) << 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(smem + kLdsPatch + buf * kPatchBuf + blk * 1024), 16, (int)off, 0, 0, 0);
      }
    };
    // the tile's first patch: addresses worked out in registers (the table is being written by other threads right now)
    auto issue_patch = [&](int geom, uint32_t c0b, int k0, int k1, int buf) {
      const OkpPatchGeom& G = p.g[geom];                         // uniform index into the kernel arguments: scalar loads
      const __amdgpu_buffer_rsrc_t rs_x = ((p.geom_src >> geom) & 1) ? rs_s1 : rs_s0;
      for (int k = k0; k < k1; ++k) {
        const int blk = k * 8 + wave;
        if (blk * 8 >= G.npx) continue;
        const uint32_t e = patch_pixel(G, blk * 8 + (lane >> 3));
        const uint32_t off = (e & kInvalidOff) ? kInvalidOff : (e & ~7u) + c0b + ((((uint32_t)lane & 7u) ^ (e & 7u)) << 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(smem + kLdsPatch + buf * kPatchBuf + blk * 1024), 16, (int)off, 0, 0, 0);
      }
    };
    // In-place split of a landed patch (npx = 18 x rows pixels): a thread owns an aligned 32-byte pair of a pixel's 128 bytes - the fp32
    // values k = 8q .. 8q+7 of the chunk, in the physical order the swizzle left them - and writes hi(k) over the logical even chunk
    // of the pair and lo(k) over the odd one.  Three items per thread for an 18 x 18 patch, all six reads in flight together.
    auto split_patch = [&](int buf, int npx) {
      char* const base = smem + kLdsPatch + buf * kPatchBuf;
      u32x4 ra[3], rb[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int it = tid + u * 512;
        if (it < npx * 4) {
          ra[u] = *reinterpret_cast<const u32x4*>(base + it * 32);
          rb[u] = *reinterpret_cast<const u32x4*>(base + it * 32 + 16);
        }
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int it = tid + u * 512;
        if (it < npx * 4) {
          const int idx = it >> 2;
          const int col = idx - (int)(((uint32_t)idx * 3641u) >> 16) * kPitch;     // idx % 18 for idx < 328
          const bool odd = patch_key_x3(col) & 1u;                                  // odd key: the pair's two chunks sit exchanged
          u32x4 hi, lo;
          okp_split8(odd ? rb[u] : ra[u], odd ? ra[u] : rb[u], hi, lo);
          *reinterpret_cast<u32x4*>(base + it * 32) = odd ? lo : hi;
          *reinterpret_cast<u32x4*>(base + it * 32 + 16) = odd ? hi : lo;
        }
      }
    };

    f32x4 acc[TCO][TPX];
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
      for (int j = 0; j < TPX; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this tile's offset tables (read by the in-loop requests, all of them behind the first K-step's barrier)
    for (int g = 0; g < p.n_geom; ++g)
      if (tid < kGeoEntries) reinterpret_cast<uint32_t*>(smem + kLdsGeo)[g * kGeoEntries + tid] = patch_pixel(p.g[g], tid);
    {                                              // the first patch of the class: geometry / chunk / buffer of step t0
      const u32x4 s0 = *reinterpret_cast<const u32x4*>(steps_lds + t0 * 16);
      const uint32_t w3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s0[3]);
      const int g0 = w3 & 0xff;
      issue_patch(g0, ((w3 >> 16) & 0xffu) * 128u, 0, (p.g[g0].npx + 63) >> 6, __builtin_amdgcn_readfirstlane((int)s0[2]) & 1);
    }
    issue_w(t0, t0 & 1);

    for (int t = t0; t < t1; ++t) {
      const u32x4 sv = *reinterpret_cast<const u32x4*>(steps_lds + t * 16);
      const uint32_t tap_bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)sv[0]);
      const uint32_t nx_c0b = (uint32_t)__builtin_amdgcn_readfirstlane((int)sv[1]);
      const uint32_t pk = (uint32_t)__builtin_amdgcn_readfirstlane((int)sv[2]);
      const int pbuf = pk & 1, nx_k0 = (pk >> 8) & 0xff, nx_k1 = (pk >> 16) & 0xff, nx_geom = pk >> 24;
      const uint32_t w3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sv[3]);
      const int dxo = (w3 >> 8) & 0xff;                            // column offset of this step's tap inside the patch
      const bool more = t + 1 < t1;
      const bool next_in_class = (int)(w3 >> 24) + 1 < t1;         // the next group still belongs to this class
      const bool next_patch = nx_k1 > nx_k0 && next_in_class;
      // (a source in pair format arrives split: okp_conv_args.src_pairs)
      const bool in_pairs = (p.pairs >> ((p.geom_src >> (w3 & 0xff)) & 1u)) & 1u, nx_pairs = (p.pairs >> ((p.geom_src >> nx_geom) & 1u)) & 1u;
      const bool cvt_self = ((pk & OKP_PSTEP_CVT_SELF) || t == t0) && !in_pairs;   // (a class's first patch was requested by the tile's prologue)
      const bool cvt_next = (pk & OKP_PSTEP_CVT_NEXT) && next_in_class && !nx_pairs;

      // my part of step t's weights (and of the patches requested so far) has landed, and my fragment reads of step t-1 have
      // returned (the barrier frees their stage / patch buffer for the next LDS-DMA)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                // ... everyone's has; stage (t+1)&1 and the other patch buffer are free
#ifdef OKP_X3_ABL_NOCVT                          // timing ablation (wrong results): patches are never split
      if (false) {
#else
      if (cvt_self) {
#endif
        split_patch(pbuf, 18 * (int)((p.geom_ph >> (5 * (w3 & 0xff))) & 31u));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }

      const char* const wt = smem + (t & 1) * kWStage;
      const char* const pbase = smem + kLdsPatch + pbuf * kPatchBuf;
      // Fragment addresses: a pixel fragment j of the wave is patch row (8 wpx + j + dy), columns fr + dx; with the swizzle keyed on
      // the COLUMN its term is the same for all eight j, which are then reached by immediate offsets of one row pitch.
      const char* const bb = pbase + tap_bytes + (uint32_t)((wpx * TPX * kPitch + fr) * 128);
      const uint32_t bhi = ((uint32_t)(2 * fh) ^ patch_key_x3(fr + dxo)) << 4;     // this lane's hi chunk; its lo chunk is the other one of the pair
      auto lda = [&](int i, int lo) {
        return *reinterpret_cast<const u32x4*>(wt + swz<128>((wco * TCO + i) * 16 + fr, wch + 2 * lo));
      };
      auto ldb = [&](int j, int lo) {
        return *reinterpret_cast<const u32x4*>(bb + j * (kPitch * 128) + (bhi ^ (uint32_t)(16 * lo)));
      };
      // twelve MFMAs of pixel fragment j: per accumulator w_lo x_hi, w_hi x_lo, w_hi x_hi (small terms first), the four channel
      // tiles interleaved so that an accumulator's three MFMAs are four instructions apart
      auto mma12 = [&](const u32x4 (&ah)[TCO], const u32x4 (&al)[TCO], const u32x4& xh, const u32x4& xl, auto jt) {
        constexpr int J = decltype(jt)::value;
#pragma unroll
        for (int i = 0; i < TCO; ++i) acc[i][J] = H16<_Float16>::mfma16(al[i], xh, acc[i][J]);
#pragma unroll
        for (int i = 0; i < TCO; ++i) acc[i][J] = H16<_Float16>::mfma16(ah[i], xl, acc[i][J]);
#pragma unroll
        for (int i = 0; i < TCO; ++i) acc[i][J] = H16<_Float16>::mfma16(ah[i], xh, acc[i][J]);
      };
      u32x4 ah[TCO], al[TCO], xh[2], xl[2];
#pragma unroll
      for (int i = 0; i < TCO; ++i) { ah[i] = lda(i, 0); al[i] = lda(i, 1); }
      xh[0] = ldb(0, 0); xl[0] = ldb(0, 1);
      if (more) issue_w(t + 1, (t + 1) & 1);
      // Fragment pipeline: the pixel fragments of j + 1 are read under the twelve MFMAs of j
#define OKP_X3_STAGE(J)                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                  \
      if (J + 1 < TPX) { xh[(J + 1) & 1] = ldb(J + 1, 0); xl[(J + 1) & 1] = ldb(J + 1, 1); } \
      if (J == 2 && next_patch) issue_patch_tab(nx_geom, nx_c0b, nx_k0, nx_k1, pbuf ^ 1);  \
      __builtin_amdgcn_sched_barrier(0);                                                  \
      mma12(ah, al, xh[J & 1], xl[J & 1], std::integral_constant<int, J>{});
      OKP_X3_STAGE(0) OKP_X3_STAGE(1) OKP_X3_STAGE(2) OKP_X3_STAGE(3)
      OKP_X3_STAGE(4) OKP_X3_STAGE(5) OKP_X3_STAGE(6) OKP_X3_STAGE(7)
#undef OKP_X3_STAGE
      __builtin_amdgcn_sched_barrier(0);
      // the next group's patch: every request was issued before this step and has landed (this step's barrier); nobody reads that
      // buffer before the next barrier, in front of which every wave waits for its own LDS writes
#ifndef OKP_X3_ABL_NOCVT
      if (cvt_next) split_patch(pbuf ^ 1, 18 * (int)((p.geom_ph >> (5 * nx_geom)) & 31u));
#endif
    }
    __syncthreads();                               // all waves done with the last stage and patch before LDS is reused

    // ---- epilogue: accumulator x 1/scale + bias in fp32, transposition through LDS in two passes of 128 pixels (the wave column wpx
    // of a pass holds them), residual + ReLU on the way out, whole 1 KiB pixel rows per wave ----
    const bool relu = p.act == OKP_ACT_RELU;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      if (wpx == pass) {
#pragma unroll
        for (int i = 0; i < TCO; ++i) {
          const int co_l = (wco * TCO + i) * 16 + 4 * fh;
          const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_lds + co_l * 4);
          const f32x4 sv = *reinterpret_cast<const f32x4*>(bias_lds + 1024 + co_l * 4);
#pragma unroll
          for (int j = 0; j < TPX; ++j) {
            const int prow = j * 16 + fr;            // pixel row of the pass
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(acc[i][j][e], sv[e], bv[e]);
            *reinterpret_cast<f32x4*>(smem + prow * 1024 + ((((co_l >> 2)) ^ (prow & 7)) << 4)) = v;
          }
        }
      }
      __syncthreads();
      if constexpr (OUT_PAIRS) {
        // pair-format output (okp_conv_args.out_pairs): a thread takes 8 adjacent channels of a pixel - two 16-byte chunks of the staged
        // row - and writes [hi | lo] as one 32-byte piece; a wave = two pixel rows
        constexpr int U2 = kStagePx * 32 / 512, UH2 = 4;
#pragma unroll 1
        for (int ub = 0; ub < U2; ub += UH2) {
          u32x4 rra[UH2], rrb[UH2];
          uint32_t ooff[UH2];
#pragma unroll
          for (int u = 0; u < UH2; ++u) {
            const int it = tid + (ub + u) * 512;
            const int q8 = it & 31, prow = pass * kStagePx + (it >> 5);
            const int co = co0 + q8 * 8;
            const uint32_t opix = (uint32_t)((n * p.OH + (y0 + (prow >> 4)) * p.out_step + p.out_oy + (cls >> 1)) * p.OW + (x0 + (prow & 15)) * p.out_step + p.out_ox + (cls & 1));
            ooff[u] = co < p.cout ? opix : kInvalidOff;
          }
          if (p.res) {
#pragma unroll
            for (int u = 0; u < UH2; ++u) {
              const int q8 = (tid + (ub + u) * 512) & 31;
              const int co = co0 + q8 * 8;
              const uint32_t opix = ooff[u] == kInvalidOff ? 0u : ooff[u];
              const char* const rp = static_cast<const char*>(p.res) + ((size_t)opix * p.res_pix_stride + (co < p.cout ? co : 0)) * 4;
              rra[u] = *reinterpret_cast<const u32x4*>(rp); rrb[u] = *reinterpret_cast<const u32x4*>(rp + 16);
            }
          }
#pragma unroll
          for (int u = 0; u < UH2; ++u) {
            const int it = tid + (ub + u) * 512;
            const int q8 = it & 31, lrow = it >> 5;
            f32x4 va = *reinterpret_cast<const f32x4*>(smem + lrow * 1024 + (((2 * q8) ^ (lrow & 7)) << 4));
            f32x4 vb = *reinterpret_cast<const f32x4*>(smem + lrow * 1024 + (((2 * q8 + 1) ^ (lrow & 7)) << 4));
            if (ooff[u] == kInvalidOff) continue;
            if (p.res) {
              const f32x4 ra = __builtin_bit_cast(f32x4, rra[u]), rb = __builtin_bit_cast(f32x4, rrb[u]);
#pragma unroll
              for (int e = 0; e < 4; ++e) { va[e] += ra[e]; vb[e] += rb[e]; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) range_bad |= okp_unsplittable(va[e]) || okp_unsplittable(vb[e]);      // (before the ReLU: fmaxf turns a NaN into 0)
            if (relu) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { va[e] = fmaxf(va[e], 0.f); vb[e] = fmaxf(vb[e], 0.f); }
            }
            u32x4 hi, lo;
            okp_split8(__builtin_bit_cast(u32x4, va), __builtin_bit_cast(u32x4, vb), hi, lo);
            char* const op = static_cast<char*>(p.out) + ((size_t)ooff[u] * p.out_pix_stride + co0 + q8 * 8) * 4;
            *reinterpret_cast<u32x4*>(op) = hi;
            *reinterpret_cast<u32x4*>(op + 16) = lo;
          }
        }
      } else {
      constexpr int U = kStagePx * 64 / 512;         // 16-byte items (4 channels of one pixel) per thread and pass: a wave = one pixel row
        constexpr int UH = 8;                          // residual vectors in flight together
#pragma unroll 1
        for (int ub = 0; ub < U; ub += UH) {
          u32x4 rres[UH];
          uint32_t ooff[UH];
#pragma unroll
          for (int u = 0; u < UH; ++u) {
            const int it = tid + (ub + u) * 512;
            const int q = it & 63, prow = pass * kStagePx + (it >> 6);
            const int co = co0 + q * 4;
            const uint32_t opix = (uint32_t)((n * p.OH + (y0 + (prow >> 4)) * p.out_step + p.out_oy + (cls >> 1)) * p.OW + (x0 + (prow & 15)) * p.out_step + p.out_ox + (cls & 1));
            ooff[u] = co < p.cout ? opix : kInvalidOff;
          }
          if (p.res) {                                 // one uniform branch, unconditional loads (clamped): all UH in flight together
#pragma unroll
            for (int u = 0; u < UH; ++u) {
              const int q = (tid + (ub + u) * 512) & 63;
              const int co = co0 + q * 4;
              const uint32_t opix = ooff[u] == kInvalidOff ? 0u : ooff[u];
              rres[u] = *reinterpret_cast<const u32x4*>(static_cast<const char*>(p.res) + ((size_t)opix * p.res_pix_stride + (co < p.cout ? co : 0)) * 4);
            }
          }
#pragma unroll
          for (int u = 0; u < UH; ++u) {
            const int it = tid + (ub + u) * 512;
            const int q = it & 63, lrow = it >> 6;
            f32x4 v = *reinterpret_cast<const f32x4*>(smem + lrow * 1024 + ((q ^ (lrow & 7)) << 4));
            if (ooff[u] == kInvalidOff) continue;
            if (p.res) {
              const f32x4 r = __builtin_bit_cast(f32x4, rres[u]);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += r[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) range_bad |= okp_unsplittable(v[e]);
            if (relu) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            *reinterpret_cast<f32x4*>(static_cast<char*>(p.out) + ((size_t)ooff[u] * p.out_pix_stride + co0 + q * 4) * 4) = v;
          }
        }
      }
      __syncthreads();                               // staging is free again: the next pass / the next tile's LDS-DMA may overwrite it
    }
  }
  okp_raise_range_flag(p.range_flag, range_bad);
}

}  // namespace

int okp_launch_igemm_patch_x3(const okp_conv* plan, const OkpPatchParams& p, hipStream_t stream) {
  if (plan->dtype != OKP_F32X3 || !p.oscale) { okp_set_error("okp_conv_forward: the split-product patch kernel takes OKP_F32X3 plans"); return OKP_EINVAL; }
  const dim3 grid((unsigned)(p.n_tiles < 256 ? p.n_tiles : 256)), block(512);
  if (p.pairs & 4u) hipLaunchKernelGGL(okp_igemm_patch_x3_kernel<true>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(okp_igemm_patch_x3_kernel<false>, grid, block, 0, stream, p);
  return okp_check_hip(hipGetLastError(), "okp_igemm_patch_x3 launch");
}

// Patch-resident implicit GEMM for the 3x3 convolutions of the pre-stage and the hourglass (bf16 / fp16, 256 output channels).
//
// okp_igemm_kernel gathers the pixel operand of EVERY K-slice from L2: a 3x3 convolution moves each input line nine
// times into LDS (32 KiB of pixels + 32 KiB of weights per slice and workgroup).  Timing ablations of that kernel
// (round 1; conv 256->256 at 64x64, N=64) gave 269 us complete, 185 us without the LDS-DMA and
// 194 us without the MFMAs: the L2 -> LDS stream costs as much as the matrix work and the two overlap poorly.
//
// Here a workgroup owns 256 output channels x one 16x16-pixel block of ONE frame.  For each 64-channel chunk of a source
// the block's input patch INCLUDING its halo (18x18 pixels x 128 bytes = 40.5 KiB) is copied to LDS once, by LDS-DMA,
// while the previous chunk is being multiplied; the nine taps of the chunk are then nine K-steps whose pixel fragments
// are read from that patch at a tap-dependent offset.  Per K-step the workgroup now streams 32 KiB of weights + 4.5 KiB
// of pixels instead of 64 KiB (4.7 instead of 8 LDS-DMA instructions per wave: each costs the SIMD ~70 clocks, whether or
// not it is waited for).  Out-of-image patch pixels get an out-of-range buffer offset: the LDS-DMA writes zeros (padding).
//
// A patch is described by a GEOMETRY (OkpPatchGeom: source, origin, rows x columns, pixel step), a K-step by its tap's
// offset inside the patch (OkpPatchStep, built at plan creation in okp_api.hip):
//   * stride-1 3x3: one 18x18 geometry per source;
//   * a strided single-tap source (the projected 1x1 skip of `residual`): 16x16 with the conv stride as pixel step;
//   * stride-2 3x3: four geometries = the parity classes of (dy, dx), 17x17 / 17x16 / 16x17 / 16x16 with pixel step 2;
//   * 4x4/s2 transposed convolution + merge add: four sub-pixel CLASSES (tile = class x block), each 2x2 taps of the same
//     18x18 geometry, written at output pixel (2 y + cy, 2 x + cx) with `up1` as the residual.
// K order and MFMA shape are those of the 256x256 gather tile (code 6), so the results are bit-identical to it.
//
// LDS: weights ring 2 x 32 KiB | patch buffers 2 x 41 KiB | step table 4 KiB | bias 1 KiB  = 151 KiB, one workgroup
// (8 waves, 4 x 2, wave tile 64 channels x 128 pixels on 16x16x32 MFMAs) per CU; the epilogue stages the bf16 tile in
// the same LDS and writes whole 512-byte pixel rows.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "okp_igemm_kernel.h"


namespace {

// Swizzle key of a patch pixel's eight 16-byte chunks, a function of its patch COLUMN pair: h(col >> 1), h = 0,1,6,3,4,5,6,3,0.
// gfx950 serves a ds_read_b128 in four passes of 16 lanes, and a pass is NOT lanes 16 p .. 16 p + 15: it takes eight lanes of
// k-group 2H and eight of k-group 2H + 1, from complementary pairs of lane quads (scripts/hwtests/lds_pattern.hip and the fit in
// DESIGN.md).  The plain key (col >> 1) & 7 of the gather tile is conflict-free for fragments that start at an even multiple
// of 16 pixels only; for the taps with column offset 1 and 2 every pixel-fragment read took two passes per 256 bytes
// (SQ_LDS_BANK_CONFLICT = 30 percent of SQ_LDS_IDX_ACTIVE).  This table is conflict-free for column offsets 0, 1 and 2.
constexpr uint32_t kPatchKeys = 0x7ac788u;
// MT = 32 (the 32x32x16 instantiation, tile 14: the MFMA-shape experiment of round 5): a 32-row fragment is two patch rows of 16 columns and
// lane bit 4 the row, so every lane group holds each column once with one k-group: the plain key (col >> 1) & 7 is conflict-free there.
template <int MT> __device__ __forceinline__ uint32_t patch_key(int col) {
  return MT == 16 ? (kPatchKeys >> (3 * (col >> 1))) & 7u : (uint32_t)(col >> 1) & 7u;
}

constexpr int kWStage = 256 * 128;                 // one K-step of weights: 256 rows x 128 B
constexpr int kPitch = 18;                         // patch row pitch in pixels (16-wide patches leave two columns unused)
constexpr int kPatchBuf = 41 * 1024;               // 18 x 18 px x 128 B, rounded up to whole 1 KiB LDS-DMA blocks
constexpr int kLdsPatch = 2 * kWStage;
constexpr int kLdsSteps = kLdsPatch + 2 * kPatchBuf;
constexpr int kLdsBias = kLdsSteps + 256 * (int)sizeof(OkpPatchStep);
#ifdef OKP_PATCH_STAMPS
// Debug build (OKP_EXTRA_CFLAGS=-DOKP_PATCH_STAMPS, scripts/run_patch_stamps.sh): shader-clock stamps of every K-step of a workgroup's first
// tile - before / after the LDS-DMA wait, after the barrier, after the first MFMA group was issued - kept in LDS and dumped for workgroups
// 0..15.  Perturbs the loop by three scalar-memory round trips per step.
constexpr int kStampBytes = 1024;                  // 8 steps x 8 waves x 16 B (what is left of the LDS)
__device__ __forceinline__ uint64_t stamp() { uint64_t t; asm volatile("s_memtime %0" : "=s"(t)); return t; }
#else
constexpr int kStampBytes = 0;
#endif
// Source offsets of the patch pixels, per geometry, rebuilt for every tile: [geometry][328] u32 = byte offset of the pixel's first channel
// | swizzle key of its column (bits 0-2), or bit 31 for a pixel outside the image / the patch.  The in-loop patch requests read their
// addresses here instead of working them out (scalar loads of the geometry from the kernel arguments + ~30 vector instructions per
// LDS-DMA: measured 4-6 % of a 3x3 launch and ~10 % of a stride-2 one, profiles/r04l_ablation_patch_issue_parts.txt).
constexpr int kGeoEntries = 328;                   // 18 x 18 pixels, rounded up to whole 8-pixel LDS-DMA blocks
constexpr int kLdsGeo = kLdsBias + 1024;
constexpr int kLdsStamps = kLdsGeo + OKP_PATCH_MAX_GEOM * kGeoEntries * 4;
constexpr int kLdsTotal = kLdsStamps + kStampBytes;
static_assert(sizeof(OkpPatchStep) == 16, "step table entries are read as one 16-byte vector");
static_assert(256 * 512 <= kLdsSteps, "epilogue staging (256 px x 256 ch bf16) must not reach the step table");
static_assert(kLdsTotal <= 160 * 1024, "one workgroup's LDS is at most the CU's 160 KiB");

template <typename T, int MT = 16>
__global__ __launch_bounds__(512) void okp_igemm_patch_kernel(const OkpPatchParams p) {
  using x4_t = typename H16<T>::x4;
  using x8_t = typename H16<T>::x8;
  static_assert(MT == 16 || MT == 32, "MFMA tile");
  constexpr int TCO = 64 / MT, TPX = 128 / MT;     // accumulator tiles per wave: 64 channels x 128 pixels (16x16: 4 x 8, 32x32: 2 x 4)
  constexpr int AREGS = MT * MT / 64;
  using acc_t = typename Mma<T, MT>::acc_t;
  __shared__ __attribute__((aligned(16))) char smem[kLdsTotal];
  char* const steps_lds = smem + kLdsSteps;
  char* const bias_lds = smem + kLdsBias;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave >> 1, wpx = wave & 1;
  const int fr = lane & 15, fh = lane >> 4;

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.weights), 0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.n_co_tiles * 256 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src_data[0]), 0, (int)p.src_bytes[0], 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src_data[1]), 0, (int)p.src_bytes[1], 0x00020000);

  if (tid < p.n_steps) reinterpret_cast<u32x4*>(steps_lds)[tid] = reinterpret_cast<const u32x4*>(p.steps)[tid];
  __syncthreads();

  // weights loader: lane (row r0 = tid >> 3, position tid & 7) fetches the chunk the read-side swizzle expects there
  const int r0 = tid >> 3;
  const int wc = (tid & 7) ^ ((r0 >> 1) & 7);

  for (int slot = blockIdx.x; slot < p.n_tiles; slot += gridDim.x) {
    // XCD-aware order as in okp_igemm_kernel: each XCD walks a contiguous range of tiles (neighbouring blocks share halos)
    const int xq = p.n_tiles >> 3, xr = p.n_tiles & 7, xcd = slot & 7;
    const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (slot >> 3);
    // class-minor order: the four sub-pixel classes of a pixel block are consecutive tiles, i.e. workgroups of ONE XCD in the
    // same round - they share the block's input patch in that L2, and their interleaved output pixels (class (cy, cx) writes
    // pixel (2y + cy, 2x + cx)) and residual reads meet in the same DRAM pages at the same time instead of in four passes
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

    if (wave == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)bias_lds, 16, (int)((uint32_t)(co0 + lane * 4) * 4u), 0, 0, 0);

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
    // passes [k0, k1) of a patch: pass k = 1 KiB blocks 8 k .. 8 k + 7 (one per wave) = patch pixels 64 k .. 64 k + 63
    // source offset (first channel) | column key of patch pixel idx of geometry G, or bit 31
    auto patch_pixel = [&](const OkpPatchGeom& G, int idx) -> uint32_t {
      const int i = idx / kPitch, j = idx - i * kPitch;
      const int ys = G.conv_stride * y0 + G.oy + i * G.step, xs = G.conv_stride * x0 + G.ox + j * G.step;
      const bool ok = idx < G.npx && j < G.PW && ys >= 0 && ys < G.H && xs >= 0 && xs < G.W;
      return ok ? ((uint32_t)((n * G.H + ys) * G.W + xs) * (uint32_t)(G.pix_stride * 2)) | patch_key<MT>(j) : kInvalidOff;
    };
    // the in-loop form: addresses from the tile's offset table, everything else from registers (no scalar loads)
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
    auto issue_patch = [&](int geom, uint32_t c0b, int k0, int k1, int buf) {
      const OkpPatchGeom& G = p.g[geom];                         // uniform index into the kernel arguments: scalar loads
      const int PW = G.PW, npx = G.npx;                          // valid columns; rows x kPitch
      const int H = G.H, W = G.W, ps2 = G.pix_stride * 2, step = G.step;
      const int yb = G.conv_stride * y0 + G.oy, xb = G.conv_stride * x0 + G.ox;
      const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(G.data), 0, (int)G.bytes, 0x00020000);
      for (int k = k0; k < k1; ++k) {
        const int blk = k * 8 + wave;
        if (blk * 8 >= npx) continue;                           // wave-uniform: nothing of this block is inside the patch
        const int idx = blk * 8 + (lane >> 3);
        const int i = idx / kPitch, j = idx - i * kPitch;
        const int ch = (lane & 7) ^ (int)patch_key<MT>(j);      // swizzle by the patch COLUMN: the same for every tap row
        const int ys = yb + i * step, xs = xb + j * step;
        const bool ok = idx < npx && j < PW && ys >= 0 && ys < H && xs >= 0 && xs < W;
        const uint32_t off = ok ? (uint32_t)((n * H + ys) * W + xs) * (uint32_t)ps2 + c0b + (uint32_t)ch * 16u : kInvalidOff;
        char* const dst = smem + kLdsPatch + buf * kPatchBuf + blk * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)dst, 16, (int)off, 0, 0, 0);
      }
    };

    acc_t acc[TCO][TPX];
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
      for (int j = 0; j < TPX; ++j)
#pragma unroll
        for (int e = 0; e < AREGS; ++e) acc[i][j][e] = 0.f;

    // this tile's offset tables (read by the in-loop requests, all of them behind the first K-step's barrier)
    for (int g = 0; g < p.n_geom; ++g)
      if (tid < kGeoEntries) reinterpret_cast<uint32_t*>(smem + kLdsGeo)[g * kGeoEntries + tid] = patch_pixel(p.g[g], tid);
    {                                              // the first patch of the class: geometry / chunk / buffer of step t0
      const u32x4 s0 = *reinterpret_cast<const u32x4*>(steps_lds + t0 * 16);
      const uint32_t w3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s0[3]);
      const int g0 = w3 & 0xff;
      issue_patch(g0, ((w3 >> 16) & 0xffu) * 128u, 0, (p.g[g0].npx + 63) >> 6, __builtin_amdgcn_readfirstlane((int)s0[2]) & 0xff);
    }
    issue_w(t0, t0 & 1);

    for (int t = t0; t < t1; ++t) {
      const u32x4 sv = *reinterpret_cast<const u32x4*>(steps_lds + t * 16);
      const uint32_t tap_bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)sv[0]);
      const uint32_t nx_c0b = (uint32_t)__builtin_amdgcn_readfirstlane((int)sv[1]);
      const uint32_t pk = (uint32_t)__builtin_amdgcn_readfirstlane((int)sv[2]);
      const int pbuf = pk & 0xff, nx_k0 = (pk >> 8) & 0xff, nx_k1 = (pk >> 16) & 0xff, nx_geom = pk >> 24;
      const uint32_t w3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sv[3]);
      const int dxo = (w3 >> 8) & 0xff;                            // column offset of this step's tap inside the patch
      const bool more = t + 1 < t1;
#ifdef OKP_PATCH_ABL_NOPATCH                    // timing ablation (wrong results): no patch but the first of a tile is ever requested
      const bool next_patch = false; (void)nx_k0; (void)nx_k1; (void)nx_geom; (void)nx_c0b;
#else
      const bool next_patch = nx_k1 > nx_k0 && (int)(w3 >> 24) + 1 < t1;   // the next group still belongs to this class
#endif

      // my part of step t's weights (and of its patch) has landed, and my fragment reads of step t-1 have returned (the
      // barrier frees their stage / patch buffer for the next LDS-DMA)
#ifdef OKP_PATCH_STAMPS
      uint64_t T0 = stamp();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      uint64_t T1 = stamp();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      uint64_t T2 = stamp();
#else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                // ... everyone's has; stage (t+1)&1 and the other patch buffer are free
#endif

      const char* const wt = smem + (t & 1) * kWStage;
      const char* const pbase = smem + kLdsPatch + pbuf * kPatchBuf;
      if constexpr (MT == 32) {
        // 32x32x16 MFMAs: a K-step is four sub-steps of K = 16; a pixel fragment = two patch rows of 16 columns (lane bit 4 = the row),
        // lane half h32 supplies k = 16 s + 8 h32 .. + 7 = chunk 2 s + h32.  Fragments double-buffered over the sub-steps.
        const int r32 = lane & 31, h32 = lane >> 5;
        const char* const bb = pbase + tap_bytes + (uint32_t)(((wpx * 8 + (r32 >> 4)) * kPitch + (r32 & 15)) * 128);
        const uint32_t bsw = patch_key<32>((r32 & 15) + dxo);
        auto lda = [&](int sub, int i) { return *reinterpret_cast<const u32x4*>(wt + swz<128>((wco * TCO + i) * 32 + r32, 2 * sub + h32)); };
        auto ldb = [&](int sub, int j) { return *reinterpret_cast<const u32x4*>(bb + j * (2 * kPitch * 128) + ((((uint32_t)(2 * sub + h32)) ^ bsw) << 4)); };
        u32x4 fa[2][TCO], fb[2][TPX];
#pragma unroll
        for (int i = 0; i < TCO; ++i) fa[0][i] = lda(0, i);
#pragma unroll
        for (int j = 0; j < TPX; ++j) fb[0][j] = ldb(0, j);
        if (more) issue_w(t + 1, (t + 1) & 1);
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          __builtin_amdgcn_sched_barrier(0);
          if (sub + 1 < 4) {
#pragma unroll
            for (int i = 0; i < TCO; ++i) fa[(sub + 1) & 1][i] = lda(sub + 1, i);
#pragma unroll
            for (int j = 0; j < TPX; ++j) fb[(sub + 1) & 1][j] = ldb(sub + 1, j);
          }
          if (sub == 1 && next_patch) issue_patch_tab(nx_geom, nx_c0b, nx_k0, nx_k1, pbuf ^ 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int j = 0; j < TPX; ++j) acc[i][j] = H16<T>::mfma32(fa[sub & 1][i], fb[sub & 1][j], acc[i][j]);
        }
      } else {
      // Fragment addresses: row index and swizzle of the weights depend on the lane only; a pixel fragment j of the
      // wave is patch row (8 wpx + j + dy), columns fr + dx - with the swizzle keyed on the COLUMN its term is the same
      // for all eight j, which are then reached by immediate offsets of one row pitch (kPitch * 128 bytes).
      const char* const bb = pbase + tap_bytes + (uint32_t)((wpx * TPX * kPitch + fr) * 128);
      const uint32_t bsw = patch_key<16>(fr + dxo);
      // Fragment pipeline: the pixel fragments are read two at a time, one pair ahead of the eight MFMAs that use
      // them, and the second k-step's weight fragments during the first k-step - so the LDS reads of a wave run
      // under its own MFMAs instead of in a burst after the barrier that all eight waves issue (and wait for) together.
      auto lda = [&](int kk, int i) {
        return *reinterpret_cast<const u32x4*>(wt + swz<128>((wco * TCO + i) * 16 + fr, 4 * kk + fh));
      };
      auto ldb = [&](int kk, int j) {
        return *reinterpret_cast<const u32x4*>(bb + j * (kPitch * 128) + ((((uint32_t)(4 * kk + fh)) ^ bsw) << 4));
      };
      auto mma8 = [&](const u32x4 (&a)[TCO], const u32x4 (&bq)[2], auto s) {
        constexpr int S = decltype(s)::value;
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            acc[i][2 * S + h] = H16<T>::mfma16(a[i], bq[h], acc[i][2 * S + h]);
      };
      using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
      using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
      u32x4 a0[TCO], a1[TCO], bq0[2], bq1[2];
#pragma unroll
      for (int i = 0; i < TCO; ++i) a0[i] = lda(0, i);
      bq0[0] = ldb(0, 0); bq0[1] = ldb(0, 1);
      bq1[0] = ldb(0, 2); bq1[1] = ldb(0, 3);
      if (more) issue_w(t + 1, (t + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      mma8(a0, bq0, S0{});
      __builtin_amdgcn_sched_barrier(0);
#ifdef OKP_PATCH_STAMPS
      {
        uint64_t T3 = stamp();
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(T0), "+s"(T1), "+s"(T2), "+s"(T3) :: "memory");
        if (slot == (int)blockIdx.x && lane == 0 && t - t0 < kStampBytes / 128)
          *reinterpret_cast<u32x4*>(smem + kLdsStamps + ((t - t0) * 8 + wave) * 16) = u32x4{(uint32_t)T0, (uint32_t)T1, (uint32_t)T2, (uint32_t)T3};
      }
      __builtin_amdgcn_sched_barrier(0);
#endif
      bq0[0] = ldb(0, 4); bq0[1] = ldb(0, 5); a1[0] = lda(1, 0); a1[1] = lda(1, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma8(a0, bq1, S1{});
      __builtin_amdgcn_sched_barrier(0);
      bq1[0] = ldb(0, 6); bq1[1] = ldb(0, 7); a1[2] = lda(1, 2); a1[3] = lda(1, 3);
      __builtin_amdgcn_sched_barrier(0);
      mma8(a0, bq0, S2{});
      __builtin_amdgcn_sched_barrier(0);
      bq0[0] = ldb(1, 0); bq0[1] = ldb(1, 1);
      if (next_patch) issue_patch_tab(nx_geom, nx_c0b, nx_k0, nx_k1, pbuf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      mma8(a0, bq1, S3{});
      __builtin_amdgcn_sched_barrier(0);
      bq1[0] = ldb(1, 2); bq1[1] = ldb(1, 3);
      __builtin_amdgcn_sched_barrier(0);
      mma8(a1, bq0, S0{});
      __builtin_amdgcn_sched_barrier(0);
      bq0[0] = ldb(1, 4); bq0[1] = ldb(1, 5);
      __builtin_amdgcn_sched_barrier(0);
      mma8(a1, bq1, S1{});
      __builtin_amdgcn_sched_barrier(0);
      bq1[0] = ldb(1, 6); bq1[1] = ldb(1, 7);
      __builtin_amdgcn_sched_barrier(0);
      mma8(a1, bq0, S2{});
      mma8(a1, bq1, S3{});
          }
    }
#ifdef OKP_PATCH_STAMPS
    {
      uint64_t TE = stamp();
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(TE) :: "memory");
      if (slot == (int)blockIdx.x && lane == 0 && t1 - t0 < kStampBytes / 128)
        *reinterpret_cast<u32x4*>(smem + kLdsStamps + ((t1 - t0) * 8 + wave) * 16) = u32x4{(uint32_t)TE, 0u, 0u, 0u};
    }
    __syncthreads();
    if (slot == (int)blockIdx.x && blockIdx.x < 16 && p.dbg)
      for (int i = tid; i < kStampBytes / 4; i += 512) p.dbg[blockIdx.x * (kStampBytes / 4) + i] = reinterpret_cast<const uint32_t*>(smem + kLdsStamps)[i];
#endif
    __syncthreads();                               // all waves done with the last stage and patch before LDS is reused

    // ---- epilogue: bias, bf16, transposition through LDS, residual + ReLU on the way out, 512-byte pixel rows ----
#pragma unroll
    for (int i = 0; i < TCO; ++i) {
#pragma unroll
      for (int g = 0; g < AREGS / 4; ++g) {
        // accumulator rows: 16x16 -> 4 (lane >> 4) + e; 32x32 -> 8 g + 4 (lane >> 5) + e (four consecutive channels per lane and group)
        const int co_l = (wco * TCO + i) * MT + (MT == 32 ? 8 * g + 4 * (lane >> 5) : 4 * fh);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_lds + co_l * 4);
#pragma unroll
        for (int j = 0; j < TPX; ++j) {
          const int prow = (wpx * TPX + j) * MT + (lane & (MT - 1));
          char* dst = smem + prow * 512 + ((((co_l * 2) >> 4) ^ (prow & 7)) << 4) + ((co_l * 2) & 15);
          x4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (T)(acc[i][j][4 * g + e] + bv[e]);
          *reinterpret_cast<x4_t*>(dst) = o;
        }
      }
    }
    __syncthreads();
    constexpr int U = 256 * 32 / 512;              // 16-byte items (8 channels of one pixel) per thread
    constexpr int UH = 8;                          // residual vectors in flight together (4: -1 %, 16: spills, +12 %)
    const bool relu = p.act == OKP_ACT_RELU;
#pragma unroll 1
    for (int ub = 0; ub < U; ub += UH) {
      u32x4 rres[UH];
      uint32_t ooff[UH];
#pragma unroll
      for (int u = 0; u < UH; ++u) {
        const int it = tid + (ub + u) * 512;
        const int q = it & 31, prow = it >> 5;
        const int co = co0 + q * 8;
        const uint32_t opix = (uint32_t)((n * p.OH + (y0 + (prow >> 4)) * p.out_step + p.out_oy + (cls >> 1)) * p.OW + (x0 + (prow & 15)) * p.out_step + p.out_ox + (cls & 1));
        ooff[u] = co < p.cout ? opix : kInvalidOff;
      }
      if (p.res) {                                 // one uniform branch, unconditional loads (clamped): all UH in flight together
#pragma unroll
        for (int u = 0; u < UH; ++u) {
          const int q = (tid + (ub + u) * 512) & 31;
          const int co = co0 + q * 8;
          const uint32_t opix = ooff[u] == kInvalidOff ? 0u : ooff[u];
          rres[u] = *reinterpret_cast<const u32x4*>(static_cast<const char*>(p.res) + ((size_t)opix * p.res_pix_stride + (co < p.cout ? co : 0)) * 2);
        }
      }
#pragma unroll
      for (int u = 0; u < UH; ++u) {
        const int it = tid + (ub + u) * 512;
        const int q = it & 31, prow = it >> 5;
        u32x4 w = *reinterpret_cast<const u32x4*>(smem + prow * 512 + ((q ^ (prow & 7)) << 4));
        if (ooff[u] == kInvalidOff) continue;
        char* op = static_cast<char*>(p.out) + ((size_t)ooff[u] * p.out_pix_stride + co0 + q * 8) * 2;
        if (!p.res) {
          if (relu) {
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            w = __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8, w), z));
          }
          *reinterpret_cast<u32x4*>(op) = w;
        } else {
          const x8_t s8 = __builtin_bit_cast(x8_t, w), r8 = __builtin_bit_cast(x8_t, rres[u]);
          x8_t o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float v = (float)s8[e] + (float)r8[e];
            if (relu) v = fmaxf(v, 0.f);
            o[e] = (T)v;
          }
          *reinterpret_cast<x8_t*>(op) = o;
        }
      }
    }
    __syncthreads();                               // staging is free again: the next tile's LDS-DMA may overwrite it
  }
}

}  // namespace

bool okp_patch_supported(const okp_conv* plan, const OkpIgemmParams& p) {
  const bool x3 = plan->dtype == OKP_F32X3;
  if (!plan->patch_steps_dev || !(okp_is16(plan->dtype) || x3)) return false;
  if ((p.n_classes != 1 && p.n_classes != 4) || p.dw_w) return false;     // (okp_conv_forward has checked that the output grid fits)
  if (p.Ho % 16 || p.Wo % 16) return false;
  // split-product plans: the fp16 side output, the fp16 residual and the subsampled fp32 output stay with the gather tiles
  if (x3 && (p.out16 || p.res16 || p.out_sub2 || !p.out || plan->n_single_slices)) return false;
  for (int s = 0; s < plan->n_src; ++s) {
    // the patch of output block (y0, x0) starts at conv_stride * (y0, x0) + (oy, ox) in the source: the source must be
    // the map the stride implies, or the zero padding at its lower/right edge would differ from the gather kernel's
    if (p.src_pix_stride[s] < plan->cin[s]) return false;
    // a table entry is (byte offset of the pixel | 3-bit column key): pixels must start on 8-byte boundaries
    if ((p.src_pix_stride[s] * okp_esz(plan->dtype)) % 8) return false;
  }
  return true;
}

// OkpIgemmParams of a launch -> the patch kernels' parameter block (both element types)
int okp_fill_patch_params(const okp_conv* plan, const OkpIgemmParams& q, OkpPatchParams& p) {
  std::memset(&p, 0, sizeof(p));
  for (int gi = 0; gi < plan->patch_n_geom; ++gi) {
    const int ss = plan->patch_src[gi];
    OkpPatchGeom& g = p.g[gi];
    g.data = q.src[ss]; g.bytes = q.src_bytes[ss]; g.H = q.srcH[ss]; g.W = q.srcW[ss]; g.pix_stride = q.src_pix_stride[ss];
    g.PW = plan->patch_PW[gi]; g.npx = 18 * plan->patch_PH[gi];
    g.oy = plan->patch_oy[gi]; g.ox = plan->patch_ox[gi]; g.step = plan->patch_step[gi]; g.conv_stride = plan->conv_stride[ss];
  }
  p.n_geom = plan->patch_n_geom;
  for (int gi = 0; gi < plan->patch_n_geom; ++gi) {
    const int ss = plan->patch_src[gi];
    // geom_src holds one bit per geometry (sources 0 / 1), geom_ph five (<= 18 rows); bit 31 of a table entry is the
    // "outside" flag, which views below 2 GiB (check_view) leave free
    // (split-product plans address frame by frame: one FRAME of the source must leave bit 31 of a table entry free, not the whole view)
    const int64_t span = plan->dtype == OKP_F32X3 ? (int64_t)q.srcH[ss] * q.srcW[ss] * q.src_pix_stride[ss] * 4 : (q.src_bytes64[ss] ? q.src_bytes64[ss] : (int64_t)q.src_bytes[ss]);
    if (ss < 0 || ss > 1 || plan->patch_PH[gi] > 18 || span >= 0x7FFF0000ll) { okp_set_error("okp_conv_forward: patch geometry %d outside what the in-loop tables encode", gi); return OKP_EINVAL; }
    p.geom_ph |= (uint32_t)plan->patch_PH[gi] << (5 * gi);
    p.geom_src |= (uint32_t)ss << gi;
    p.src_data[ss] = q.src[ss]; p.src_bytes[ss] = q.src_bytes[ss];
    p.src_total_bytes[ss] = q.src_bytes64[ss] ? q.src_bytes64[ss] : (int64_t)q.src_bytes[ss];
    p.src_frame_bytes[ss] = (int64_t)q.srcH[ss] * q.srcW[ss] * q.src_pix_stride[ss] * okp_esz(plan->dtype);
  }
  if (!p.src_data[1]) { p.src_data[1] = p.src_data[0]; p.src_bytes[1] = p.src_bytes[0]; p.src_total_bytes[1] = p.src_total_bytes[0]; p.src_frame_bytes[1] = p.src_frame_bytes[0]; }
  p.weights = q.weights; p.w_bytes = q.w_bytes; p.cout_pad = q.cout_pad; p.cout = q.cout; p.bias = q.bias; p.oscale = q.oscale;
  p.steps = plan->patch_steps_dev; p.n_steps = plan->n_slices;
  p.n_classes = q.n_classes; p.steps_per_class = plan->n_slices / q.n_classes;
  p.OH = q.OH; p.OW = q.OW; p.out_step = q.out_step; p.out_oy = q.out_oy; p.out_ox = q.out_ox;
  p.N = q.N; p.H = q.Ho; p.W = q.Wo; p.tiles_y = q.Ho / 16; p.tiles_x = q.Wo / 16;
  p.div_tiles_frame = okp_fastdiv((uint32_t)(p.tiles_y * p.tiles_x)); p.div_tiles_x = okp_fastdiv((uint32_t)p.tiles_x);
  p.out = q.out; p.out_bytes = q.out_bytes; p.out_pix_stride = q.out_pix_stride;
  p.res = q.res; p.res_bytes = q.res_bytes; p.res_pix_stride = q.res_pix_stride; p.act = q.act;
  p.pairs = ((uint32_t)q.src_pairs & 3u) | (q.out_pairs ? 4u : 0u);
  p.range_flag = q.range_flag;
  p.n_co_tiles = q.cout_pad / 256;
  p.tiles_per_class = p.n_co_tiles * p.N * p.tiles_y * p.tiles_x;
  p.n_tiles = p.tiles_per_class * p.n_classes;
  return OKP_OK;
}

int okp_launch_igemm_patch(const okp_conv* plan, const OkpIgemmParams& q, hipStream_t stream) {
  if (!okp_patch_supported(plan, q)) { okp_set_error("okp_conv_forward: tile 13 (patch-resident kernel) does not apply to this plan / problem"); return OKP_EINVAL; }
  OkpPatchParams p;
  if (int e = okp_fill_patch_params(plan, q, p)) return e;
  if (plan->dtype == OKP_F32X3) return okp_launch_igemm_patch_x3(plan, p, stream);
#ifdef OKP_PATCH_STAMPS
  static uint32_t* dbg = nullptr;
  if (!dbg) (void)hipMalloc((void**)&dbg, 16 * kStampBytes);
  (void)hipMemsetAsync(dbg, 0, 16 * kStampBytes, stream);
  p.dbg = dbg;
#endif
  const dim3 grid((unsigned)(p.n_tiles < 256 ? p.n_tiles : 256)), block(512);
  if (q.mfma32) {         // tile 14: the 32x32x16 instantiation (same wave tile; kept for the MFMA-shape A/B, never the heuristic's choice)
    if (plan->dtype == OKP_BF16) hipLaunchKernelGGL((okp_igemm_patch_kernel<__bf16, 32>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((okp_igemm_patch_kernel<_Float16, 32>), grid, block, 0, stream, p);
  } else if (plan->dtype == OKP_BF16) hipLaunchKernelGGL((okp_igemm_patch_kernel<__bf16, 16>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((okp_igemm_patch_kernel<_Float16, 16>), grid, block, 0, stream, p);
#ifdef OKP_PATCH_STAMPS
  if (getenv("OKP_PATCH_STAMPS_PRINT")) {
    (void)hipStreamSynchronize(stream);
    std::vector<uint32_t> h(16 * kStampBytes / 4);
    (void)hipMemcpy(h.data(), dbg, 16 * kStampBytes, hipMemcpyDeviceToHost);
    const int ns = std::min(p.steps_per_class, kStampBytes / 128 - 1);
    printf("patch stamps: %d K-steps per tile; clocks per step: wait for the LDS-DMA | barrier | until the first MFMA group was issued | rest of the body\n", p.steps_per_class);
    for (int wg = 0; wg < 16; wg += 5)
      for (int t = 0; t < ns; ++t) {
        printf("wg %2d step %2d:", wg, t);
        for (int w = 0; w < 8; w += 7) {
          const uint32_t* a = &h[(size_t)wg * (kStampBytes / 4) + (t * 8 + w) * 4];
          printf("   wave %d: %5u | %5u | %5u | %5u", w, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[32] - a[3]);
        }
        printf("\n");
      }
    double sum[4] = {0, 0, 0, 0};
    int cnt = 0;
    for (int wg = 0; wg < 16; ++wg)
      for (int t = 1; t < ns; ++t)
        for (int w = 0; w < 8; ++w) {
          const uint32_t* a = &h[(size_t)wg * (kStampBytes / 4) + (t * 8 + w) * 4];
          sum[0] += a[1] - a[0]; sum[1] += a[2] - a[1]; sum[2] += a[3] - a[2]; sum[3] += a[32] - a[3]; ++cnt;
        }
    printf("mean over 16 workgroups, all waves, steps 1..: wait %.0f | barrier %.0f | first MFMAs %.0f | rest %.0f  = %.0f clocks per step\n",
           sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, (sum[0] + sum[1] + sum[2] + sum[3]) / cnt);
  }
#endif
  return okp_check_hip(hipGetLastError(), "okp_igemm_patch launch");
}

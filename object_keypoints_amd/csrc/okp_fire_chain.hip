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
#include <cstring>

#include "okp_internal.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int CIN, int PXB>
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
  static constexpr int OFF_X = 0;                      // two activation buffers [NPX][CIN]
  static constexpr int OFF_S = 2 * NPX * XROW;         // squeeze tile [(SIDE+2)^2 rows, zero halo][MID]
  // per module: depth-wise weights [9][MID], bd, b1, ba (fp32), every array padded to whole 1 KiB LDS-DMA instructions
  static constexpr int WD_BYTES = (9 * MID * 4 + 1023) / 1024 * 1024, B_BYTES = (MID * 4 + 1023) / 1024 * 1024;
  static constexpr int CONST_BYTES = WD_BYTES + 3 * B_BYTES;
  static constexpr int OFF_C = OFF_S + SROWS * SROW;   // two such blocks: the next module's constants arrive by LDS-DMA while this one runs
  static constexpr int BYTES = OFF_C + 2 * CONST_BYTES;
  static_assert((CIN / 8) % 8 == 0 && (MID / 8) % 8 == 0 && BYTES <= 160 * 1024, "row swizzle / LDS");
};

// 16-byte chunk c of row r, XOR-swizzled by the row (the rows and k-groups of a fragment read hit distinct banks)
template <int KEY>
__device__ __forceinline__ uint32_t xoff(int row, int chunk, int rowbytes) { return (uint32_t)row * rowbytes + (uint32_t)((chunk ^ (row & KEY)) << 4); }

template <typename T, int CIN, int PXB>
__global__ __launch_bounds__((ChainCfg<CIN, PXB>::NT)) void okp_fire_chain_kernel(const OkpFireChainParams p) {
  using C = ChainCfg<CIN, PXB>;
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
  const int ch0 = 32 * w + 2 * l16;                    // this lane's channel pair in both GEMMs

  // frame -> LDS (buffer 0), squeeze tile cleared once: its halo stays zero for the whole chain
  {
    const char* src = static_cast<const char*>(p.x) + (size_t)n * HW * p.x_ps * 2;
    for (int i = tid; i < NPX * (CIN / 8); i += NT) {
      const int px = i / (CIN / 8), c = i % (CIN / 8);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (px < HW) v = *reinterpret_cast<const u32x4*>(src + (size_t)px * p.x_ps * 2 + c * 16);
      *reinterpret_cast<u32x4*>(smem + C::OFF_X + xoff<XK>(px, c, XROW)) = v;
    }
    for (int i = tid; i < C::SROWS * (MID / 8); i += NT) *reinterpret_cast<u32x4*>(smem + C::OFF_S + i * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  // interior pixel l16 -> squeeze-tile row (zero halo of one pixel around the H x W map, row pitch W + 2)
  const int SW = p.W + 2;
  auto srow = [&](int px) { const int py = px / p.W; return (py + 1) * SW + (px - py * p.W) + 1; };
  __syncthreads();

  // per-module constants -> LDS by LDS-DMA (no registers, nobody waits until the module that needs them starts):
  // a module then has no dependent global round trips besides its streamed weight fragments
  auto fetch_consts = [&](const OkpFireChainModule& mod, int slot) {
    char* dst = smem + C::OFF_C + slot * C::CONST_BYTES;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(mod.wd), 0, 9 * MID * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(mod.bd), 0, MID * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(mod.b1), 0, MID * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(mod.ba), 0, MID * 4, 0x00020000);
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
  fetch_consts(p.mod[0], 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  u32x4 wf[PF][2];
  {
    const u32x4* w1_first = static_cast<const u32x4*>(p.mod[0].w1) + (size_t)w * 2 * KS1 * 64 + lane;
#pragma unroll
    for (int i = 0; i < PF; ++i) { wf[i][0] = w1_first[(size_t)i * 64]; wf[i][1] = w1_first[(size_t)(KS1 + i) * 64]; }
  }
  int cur = 0;
  for (int m = 0; m < p.n_modules; ++m) {
    const OkpFireChainModule mod = p.mod[m];
    const float* cst = reinterpret_cast<const float*>(smem + C::OFF_C + (m & 1) * C::CONST_BYTES);   // [9][MID] depth-wise weights
    const float* cbd = cst + C::WD_BYTES / 4;              // then bd, b1, ba
    const float* cb1 = cbd + C::B_BYTES / 4;
    const float* cba = cb1 + C::B_BYTES / 4;
    if (m + 1 < p.n_modules) fetch_consts(p.mod[m + 1], (m + 1) & 1);     // lands during this module; waited at its end
    // weights in fragment order [wave][block b][k-step][lane][16 B]: one load = 1 KiB contiguous per wave
    const u32x4* w1_lane = static_cast<const u32x4*>(mod.w1) + (size_t)w * 2 * KS1 * 64 + lane;
    const u32x4* wa_lane = static_cast<const u32x4*>(mod.wa) + (size_t)w * 2 * KS2 * 64 + lane;
    auto frag = [&](const u32x4* base, int ksteps, int ks, int b) { return base[(size_t)(b * ksteps + ks) * 64]; };
    const int cg = tid % (MID / 8);                          // depth-wise branch: this thread's 8-channel group
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
    __syncthreads();

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
      const bool more = m + 1 < p.n_modules;
      const u32x4* w1_next = static_cast<const u32x4*>(p.mod[more ? m + 1 : m].w1) + (size_t)w * 2 * KS1 * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) {
        u32x4 a[PXB];
#pragma unroll
        for (int pb = 0; pb < PXB; ++pb) a[pb] = *reinterpret_cast<const u32x4*>(smem + C::OFF_S + xoff<SK>(arow[pb], 4 * ks + q, SROW));
        const u32x4 f0 = wf[ks % PF][0], f1 = wf[ks % PF][1];
        if (more) { wf[ks % PF][0] = frag(w1_next, KS1, ks, 0); wf[ks % PF][1] = frag(w1_next, KS1, ks, 1); }      // the next module's squeeze step ks
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

    // ---- depth-wise: y[:, MID:] = relu(dw3x3(s) + bd + x[:, MID:]) -----------------------------------------------------
    // thread = (8-channel group cg, pixel slot); NT / (MID / 8) = 16 pixel slots per pass
    for (int dpx = tid / (MID / 8); dpx < NPX; dpx += 16) {
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
        const int row = (py + t / 3) * SW + pxx + t % 3;         // (py + 1 + dy) * SW + pxx + 1 + dx with dy, dx in -1..1
        const x8_t sv = *reinterpret_cast<const x8_t*>(smem + C::OFF_S + xoff<SK>(row, cg, SROW));
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(cst + t * MID + cg * 8), w1 = *reinterpret_cast<const f32x4*>(cst + t * MID + cg * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = fmaf((float)sv[e], w0[e], v[e]); v[4 + e] = fmaf((float)sv[4 + e], w1[e], v[4 + e]); }
      }
      const uint32_t o = xoff<XK>(dpx, (MID * 2) / 16 + cg, XROW);
      const x8_t xv = *reinterpret_cast<const x8_t*>(xc + o);
      x8_t out;
#pragma unroll
      for (int e = 0; e < 8; ++e) out[e] = (T)(dpx < HW ? fmaxf(v[e] + (float)xv[e], 0.f) : 0.f);
      *reinterpret_cast<x8_t*>(xn + o) = out;
    }
    // the next module's constants have landed (this wave's share); its first 2 PF squeeze fragments, issued after them, may stay in flight
    if (m + 1 < p.n_modules) {
      if constexpr (PF == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      static_assert(PF == 8 || PF == 6, "counted wait");
    }
    __syncthreads();
    cur ^= 1;
  }

  // result -> HBM
  {
    const char* xc = smem + C::OFF_X + cur * NPX * XROW;
    char* dst = static_cast<char*>(p.out) + (size_t)n * HW * p.out_ps * 2;
    for (int i = tid; i < HW * (CIN / 8); i += NT) {
      const int px = i / (CIN / 8), c = i % (CIN / 8);
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


extern "C" int okp_fire_chain_forward(int32_t n_modules, okp_conv* const* squeeze, okp_conv* const* expand,
                                      const float* const* dw_w_dev, const float* const* dw_bias_dev,
                                      int32_t n, const okp_tensor* x, const okp_tensor* out, void* stream) {
  if (!squeeze || !expand || !dw_w_dev || !dw_bias_dev || !x || !out || !x->data || !out->data) { okp_set_error("okp_fire_chain_forward: null argument"); return OKP_EINVAL; }
  if (n_modules < 1 || n_modules > OKP_FIRE_CHAIN_MAX) { okp_set_error("okp_fire_chain_forward: 1..%d modules", OKP_FIRE_CHAIN_MAX); return OKP_EINVAL; }
  if (n < 1) return OKP_OK;
  OkpFireChainParams p;
  std::memset(&p, 0, sizeof(p));
  int cin = 0;
  for (int m = 0; m < n_modules; ++m) {
    okp_conv *sq = squeeze[m], *ex = expand[m];
    if (!sq || !ex || !dw_w_dev[m] || !dw_bias_dev[m]) { okp_set_error("okp_fire_chain_forward: null module %d", m); return OKP_EINVAL; }
    if (!okp_is16(sq->dtype) || ex->dtype != sq->dtype || sq->dtype != squeeze[0]->dtype || sq->n_taps != 1 || ex->n_taps != 1 || sq->n_src != 1 || ex->n_src != 1) {
      okp_set_error("okp_fire_chain_forward: module %d: bf16 / fp16 single-tap 1x1 plans of one type expected", m); return OKP_EINVAL;
    }
    if (m == 0) cin = sq->cin[0];
    if (sq->cin[0] != cin || sq->cout * 2 != cin || ex->cin[0] != sq->cout || ex->cout != sq->cout) {
      okp_set_error("okp_fire_chain_forward: module %d is not a %d -> %d -> %d fire module", m, cin, cin / 2, cin); return OKP_EINVAL;
    }
    if (int e = okp_ensure_frags(sq, (hipStream_t)stream)) return e;
    if (int e = okp_ensure_frags(ex, (hipStream_t)stream)) return e;
    p.mod[m].w1 = sq->frag_dev; p.mod[m].w1_cout_pad = sq->cout_pad; p.mod[m].b1 = sq->bias_dev;
    p.mod[m].wa = ex->frag_dev; p.mod[m].wa_cout_pad = ex->cout_pad; p.mod[m].ba = ex->bias_dev;
    p.mod[m].wd = dw_w_dev[m]; p.mod[m].bd = dw_bias_dev[m];
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
  const bool bf = squeeze[0]->dtype == OKP_BF16;
  if (cin == 512) {
    if (bf) hipLaunchKernelGGL((okp_fire_chain_kernel<__bf16, 512, 1>), dim3(n), dim3(ChainCfg<512, 1>::NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((okp_fire_chain_kernel<_Float16, 512, 1>), dim3(n), dim3(ChainCfg<512, 1>::NT), 0, (hipStream_t)stream, p);
  } else if (small) {
    if (bf) hipLaunchKernelGGL((okp_fire_chain_kernel<__bf16, 384, 1>), dim3(n), dim3(ChainCfg<384, 1>::NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((okp_fire_chain_kernel<_Float16, 384, 1>), dim3(n), dim3(ChainCfg<384, 1>::NT), 0, (hipStream_t)stream, p);
  } else {
    if (bf) hipLaunchKernelGGL((okp_fire_chain_kernel<__bf16, 384, 4>), dim3(n), dim3(ChainCfg<384, 4>::NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((okp_fire_chain_kernel<_Float16, 384, 4>), dim3(n), dim3(ChainCfg<384, 4>::NT), 0, (hipStream_t)stream, p);
  }
  return okp_check_hip(hipGetLastError(), "okp_fire_chain launch");
}

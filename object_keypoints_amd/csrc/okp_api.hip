// C-ABI glue of libokp_hip.so: error reporting, convolution plans (weight packing + upload),
// argument validation and launch of the implicit-GEMM kernel.  See include/okp.h.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "okp_internal.h"

namespace {
thread_local char g_err[512] = "";

inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x0040u);   // keep NaN a NaN
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
// IEEE half, round to nearest even (overflow -> inf, subnormals kept): the host compiler's own conversion
inline uint16_t f32_to_f16_rne(float f) {
  const _Float16 h = (_Float16)f;
  uint16_t u;
  std::memcpy(&u, &h, 2);
  return u;
}
}  // namespace

uint16_t okp_f32_to_16(int dtype, float f) { return dtype == OKP_BF16 ? f32_to_bf16_rne(f) : f32_to_f16_rne(f); }

void okp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int okp_check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return OKP_OK;
  okp_set_error("%s: %s", what, hipGetErrorString(e));
  return OKP_EHIP;
}

extern "C" const char* okp_last_error(void) { return g_err; }
extern "C" int okp_abi_version(void) { return OKP_ABI_VERSION; }

extern "C" int okp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int okp_device_arch(int dev, char* buf, int buflen) {
  hipDeviceProp_t prop;
  if (int e = okp_check_hip(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties")) return e;
  snprintf(buf, buflen, "%s", prop.gcnArchName);
  return OKP_OK;
}

static okp_conv* conv_create(int dtype, int n_src, const int32_t* cin, const int32_t* conv_stride, int32_t cout,
                             int32_t n_taps, const okp_tap* taps, const uint8_t* tap_terms, const float* bias, int act);

extern "C" okp_conv* okp_conv_create(int dtype, int n_src, const int32_t* cin, const int32_t* conv_stride, int32_t cout,
                                     int32_t n_taps, const okp_tap* taps, const float* bias, int act) {
  return conv_create(dtype, n_src, cin, conv_stride, cout, n_taps, taps, nullptr, bias, act);
}

extern "C" okp_conv* okp_conv_create_x3(int n_src, const int32_t* cin, const int32_t* conv_stride, int32_t cout, int32_t n_taps,
                                        const okp_tap* taps, const uint8_t* tap_terms, const float* bias, int act) {
  if (tap_terms)
    for (int t = 0; t < n_taps && t < OKP_MAX_TAPS; ++t)
      if (tap_terms[t] != 1 && tap_terms[t] != 3) { okp_set_error("okp_conv_create_x3: tap_terms[%d] = %d, must be 1 or 3", t, (int)tap_terms[t]); return nullptr; }
  return conv_create(OKP_F32X3, n_src, cin, conv_stride, cout, n_taps, taps, tap_terms, bias, act);
}

static okp_conv* conv_create(int dtype, int n_src, const int32_t* cin, const int32_t* conv_stride, int32_t cout,
                             int32_t n_taps, const okp_tap* taps, const uint8_t* tap_terms, const float* bias, int act) {
  if (dtype != OKP_F32 && dtype != OKP_F32X3 && !okp_is16(dtype)) { okp_set_error("okp_conv_create: bad dtype %d", dtype); return nullptr; }
  if (n_src < 1 || n_src > 2 || !cin || !conv_stride || !taps) { okp_set_error("okp_conv_create: bad sources"); return nullptr; }
  if (n_taps < 1 || n_taps > OKP_MAX_TAPS) { okp_set_error("okp_conv_create: n_taps %d not in [1,%d]", n_taps, OKP_MAX_TAPS); return nullptr; }
  if (cout < 8 || cout % 8) { okp_set_error("okp_conv_create: cout %d must be a positive multiple of 8", cout); return nullptr; }
  if (act != OKP_ACT_NONE && act != OKP_ACT_RELU) { okp_set_error("okp_conv_create: activation %d unsupported here", act); return nullptr; }
  const int esz = okp_esz(dtype);
  const int KE = 128 / esz;            // elements per K-slice
  for (int s = 0; s < n_src; ++s) {
    if (cin[s] < 1 || (cin[s] * esz) % 16) { okp_set_error("okp_conv_create: cin[%d]=%d is not a multiple of 16 bytes", s, cin[s]); return nullptr; }
    if (conv_stride[s] < 1) { okp_set_error("okp_conv_create: conv_stride[%d]=%d", s, conv_stride[s]); return nullptr; }
  }
  for (int t = 0; t < n_taps; ++t)
    if (taps[t].src < 0 || taps[t].src >= n_src || !taps[t].w) { okp_set_error("okp_conv_create: tap %d invalid", t); return nullptr; }

  okp_conv* plan = new okp_conv();
  std::memset(plan, 0, sizeof(*plan));
  plan->dtype = dtype; plan->n_src = n_src; plan->cout = cout; plan->act = act; plan->n_taps = n_taps;
  plan->cout_pad = (cout + 63) / 64 * 64;
  for (int s = 0; s < 2; ++s) { plan->cin[s] = s < n_src ? cin[s] : 0; plan->conv_stride[s] = s < n_src ? conv_stride[s] : 1; }
  for (int t = 0; t < n_taps; ++t) { plan->taps[t].dy = taps[t].dy; plan->taps[t].dx = taps[t].dx; plan->taps[t].src = taps[t].src; plan->taps[t].pad = 0; }

  // ---- slice table -------------------------------------------------------------------------
  std::vector<OkpSlice> slices;
  for (int t = 0; t < n_taps;) {
    const int s = taps[t].src, c = cin[s];
    if (c * esz == 64 && t + 1 < n_taps && taps[t + 1].src == s) {
      // two half-slice taps share one 128-byte slice (the bf16 stem: 8 px x 4 ch per kernel row)
      if (tap_terms && (tap_terms[t] == 1 || tap_terms[t + 1] == 1)) {
        okp_set_error("okp_conv_create_x3: taps %d / %d share one K-slice (64-byte sources): single-term products are not supported there", t, t + 1);
        delete plan; return nullptr;
      }
      OkpSlice sl{(uint8_t)t, (uint8_t)(t + 1), (uint8_t)s, 8, 0, 0, 0};
      slices.push_back(sl);
      t += 2;
      continue;
    }
    for (int c0 = 0; c0 < c; c0 += KE) {
      const int chunks = ((c - c0) * esz + 15) / 16;
      OkpSlice sl{(uint8_t)t, (uint8_t)t, (uint8_t)s, (uint8_t)(chunks > 8 ? 8 : chunks), c0, c0 + KE / 2, (tap_terms && tap_terms[t] == 1) ? 1 : 0};
      slices.push_back(sl);
    }
    ++t;
  }
  // ---- patch geometries (okp_igemm_patch.hip): which window of its source every tap reads ----------------------------
  // One geometry per source whose taps sit on one lattice (stride 1: the 3x3 window, 18x18 patch pixels for a 16x16 block;
  // a single tap: 16x16 with the conv stride as pixel step); a strided multi-tap source is split into the residue classes
  // of (dy, dx) modulo the stride - a stride-2 3x3 reads four interleaved sub-lattices of 17x17, 17x16, 16x17 and 16x16.
  int tap_geom[OKP_MAX_TAPS] = {0}, tap_ty[OKP_MAX_TAPS] = {0}, tap_tx[OKP_MAX_TAPS] = {0};
  // (16-bit plans: chunks of 64 channels; split-product plans, okp_igemm_patch_x3.hip: 32 fp32 channels - KE either way - and no
  //  single-term taps, which would split the K order in two ranges)
  bool patch_ok = (okp_is16(dtype) || (dtype == OKP_F32X3 && !tap_terms)) && plan->cout_pad % 256 == 0;
  plan->patch_n_geom = 0;
  for (int sidx = 0; sidx < n_src && patch_ok; ++sidx) {
    const int cs = conv_stride[sidx];
    if (cin[sidx] % KE) { patch_ok = false; break; }
    bool single = true; int first = -1;
    for (int t = 0; t < n_taps; ++t) if (taps[t].src == sidx) {
      if (first < 0) first = t;
      single = single && taps[t].dy == taps[first].dy && taps[t].dx == taps[first].dx;
    }
    if (first < 0) { patch_ok = false; break; }
    const int m = (cs == 1 || single) ? 1 : cs;                 // lattice classes per axis
    auto cls = [&](int d) { return ((d % m) + m) % m; };
    for (int cy = 0; cy < m && patch_ok; ++cy)
      for (int cx = 0; cx < m && patch_ok; ++cx) {
        int lo_y = 1 << 20, lo_x = 1 << 20, hi_y = -(1 << 20), hi_x = -(1 << 20), cnt = 0;
        for (int t = 0; t < n_taps; ++t) if (taps[t].src == sidx && cls(taps[t].dy) == cy && cls(taps[t].dx) == cx) {
          lo_y = std::min(lo_y, (int)taps[t].dy); hi_y = std::max(hi_y, (int)taps[t].dy);
          lo_x = std::min(lo_x, (int)taps[t].dx); hi_x = std::max(hi_x, (int)taps[t].dx);
          ++cnt;
        }
        if (!cnt) continue;
        const int g = plan->patch_n_geom;
        if (g >= OKP_PATCH_MAX_GEOM) { patch_ok = false; break; }
        plan->patch_src[g] = sidx; plan->patch_step[g] = single ? cs : m;
        plan->patch_oy[g] = lo_y; plan->patch_ox[g] = lo_x;
        plan->patch_PH[g] = 16 + (hi_y - lo_y) / m; plan->patch_PW[g] = 16 + (hi_x - lo_x) / m;
        if (plan->patch_PH[g] > 18 || plan->patch_PW[g] > 18) { patch_ok = false; break; }
        for (int t = 0; t < n_taps; ++t) if (taps[t].src == sidx && cls(taps[t].dy) == cy && cls(taps[t].dx) == cx) {
          tap_geom[t] = g; tap_ty[t] = (taps[t].dy - lo_y) / m; tap_tx[t] = (taps[t].dx - lo_x) / m;
        }
        ++plan->patch_n_geom;
      }
  }
  if (!patch_ok) for (int t = 0; t < n_taps; ++t) tap_geom[t] = 0;

  // K order: channel chunk outermost, taps inside (any order of the K-slices computes the same sum).  The taps of one
  // 128-byte channel chunk then follow each other, so the lines one tap gathers are re-read by its neighbours within a
  // few slices - L2 hits - instead of a whole tap (4-8 slices of every CU of the XCD) later.  Measured with per-dispatch
  // FETCH_SIZE (scripts/pmc_per_dispatch.sh): pre[1].conv2+skip 3.25 -> 1.00 GB, the 3x3 convolutions at 64x64
  // 570-790 -> 170-310 MB per launch (134 MB of input), +3 % on those launches.  Plans whose tap count is a multiple of
  // four keep four equal tap groups contiguous: the sub-pixel classes of the transposed convolution (n_classes = 4).
  {
    const int group_taps = (n_taps % 4 == 0 && n_taps >= 8) ? n_taps / 4 : n_taps;
    std::stable_sort(slices.begin(), slices.end(), [&](const OkpSlice& a, const OkpSlice& b) {
      const int ga = a.tap_lo / group_taps, gb = b.tap_lo / group_taps;
      if (ga != gb) return ga < gb;
      if (a.src != b.src) return a.src < b.src;
      if (a.c0_lo != b.c0_lo) return a.c0_lo < b.c0_lo;
      return tap_geom[a.tap_lo] < tap_geom[b.tap_lo];          // the taps of one patch geometry follow each other
    });
  }
  // split-product plans with per-tap term counts: the single-term slices go first (the kernel runs them as a loop of their own)
  plan->n_single_slices = 0;
  if (tap_terms) {
    for (const OkpSlice& sl : slices) plan->n_single_slices += sl.pad == 1 ? 1 : 0;
    if (plan->n_single_slices && (n_taps % 4 == 0 && n_taps >= 8) && plan->n_single_slices != (int)slices.size()) {
      // (four equal tap groups are the sub-pixel classes of a transposed convolution: each class is a K range of its own)
      okp_set_error("okp_conv_create_x3: plans with four tap groups (sub-pixel classes) take one term count for all taps"); delete plan; return nullptr;
    }
    std::stable_partition(slices.begin(), slices.end(), [](const OkpSlice& sl) { return sl.pad == 1; });
  }
  plan->n_slices = (int)slices.size();
  if (plan->n_slices > 256) { okp_set_error("okp_conv_create: %d K-slices exceed the 256-entry in-LDS slice table", plan->n_slices); delete plan; return nullptr; }

  // ---- packed weights [slice][cout_pad][KE] ---------------------------------------------------
  const size_t n_el = (size_t)plan->n_slices * plan->cout_pad * KE;
  const size_t w_bytes = n_el * esz;
  if (w_bytes >= 0x7FFF0000ull) { okp_set_error("okp_conv_create: packed weights %zu bytes exceed 2 GiB", w_bytes); delete plan; return nullptr; }
  std::vector<float> packed(n_el, 0.f);
  for (int si = 0; si < plan->n_slices; ++si) {
    const OkpSlice& sl = slices[si];
    for (int half = 0; half < 2; ++half) {
      const int t = half ? sl.tap_hi : sl.tap_lo;
      const int c0 = half ? sl.c0_hi : sl.c0_lo;
      const int c = cin[taps[t].src];
      const float* w = taps[t].w;
      for (int co = 0; co < cout; ++co) {
        float* dst = &packed[((size_t)si * plan->cout_pad + co) * KE + half * (KE / 2)];
        for (int e = 0; e < KE / 2; ++e) {
          const int ci = c0 + e;
          if (ci < c) dst[e] = w[(size_t)co * c + ci];
        }
      }
    }
  }
  // ---- step table of the patch-resident kernels (256-channel tiles; okp_igemm_patch.hip, okp_igemm_patch_x3.hip) -----------------
  // Eligible when every slice is one whole chunk (64 16-bit / 32 fp32 channels) of ONE tap: the 16x16-pixel tile's input patch of a
  // chunk and geometry is loaded once and serves all its taps (consecutive K-steps).
  std::vector<OkpPatchStep> psteps;
  {
    const bool x3 = dtype == OKP_F32X3;
    bool ok = patch_ok && plan->n_slices <= 256 && cin[0] <= 255 * KE && (n_src < 2 || cin[1] <= 255 * KE);
    struct Group { int geom, c0, first, n; };
    std::vector<Group> groups;
    for (int si = 0; si < plan->n_slices && ok; ++si) {
      const OkpSlice& sl = slices[si];
      if (sl.tap_lo != sl.tap_hi || sl.nvalid != 8 || sl.c0_hi != sl.c0_lo + KE / 2 || sl.c0_lo % KE || sl.pad) { ok = false; break; }
      const int g = tap_geom[sl.tap_lo];
      if (groups.empty() || groups.back().geom != g || groups.back().c0 != sl.c0_lo) groups.push_back({g, sl.c0_lo, si, 0});
      ++groups.back().n;
    }
    if (ok && !groups.empty()) {
      auto passes = [&](int g) { return (plan->patch_PH[g] * 18 + 63) / 64; };   // rows of 18 pixels (the kernel's pitch); 64 px = 8 KiB per pass of 512 lanes
      psteps.resize(plan->n_slices);
      for (size_t gi = 0; gi < groups.size(); ++gi) {
        const Group& G = groups[gi];
        const bool has_next = gi + 1 < groups.size();
        const int np = has_next ? passes(groups[gi + 1].geom) : 0;
        // split-product plans: the next patch lands as fp32 and is split in place during this group's LAST step, so its requests go
        // into steps 0 .. n-2; behind a one-step group (and at the first step of a tile) the patch is split at the start of its own
        // first step instead (OKP_PSTEP_CVT_*, okp_igemm_patch_x3.hip)
        const int req_steps = (x3 && G.n >= 2) ? G.n - 1 : G.n;
        const int per = (np + req_steps - 1) / req_steps;
        const bool self = x3 && (gi == 0 || groups[gi - 1].n < 2);
        for (int i = 0; i < G.n; ++i) {
          const OkpSlice& sl = slices[G.first + i];
          OkpPatchStep st{};
          st.tap_bytes = (uint32_t)((tap_ty[sl.tap_lo] * 18 + tap_tx[sl.tap_lo]) * 128);
          st.tx = (uint8_t)tap_tx[sl.tap_lo];
          st.c0q = (uint8_t)(G.c0 / KE);
          st.grp_last = (uint8_t)(G.first + G.n - 1);
          st.pbuf = (uint8_t)(gi & 1);
          if (self && i == 0) st.pbuf |= OKP_PSTEP_CVT_SELF;
          if (x3 && has_next && G.n >= 2 && i == G.n - 1) st.pbuf |= OKP_PSTEP_CVT_NEXT;
          st.geom = (uint8_t)G.geom;
          st.nx_k0 = (uint8_t)std::min(np, i * per); st.nx_k1 = (uint8_t)std::min(np, (i + 1) * per);
          if (i >= req_steps) st.nx_k0 = st.nx_k1 = 0;
          st.nx_geom = (uint8_t)(has_next ? groups[gi + 1].geom : 0);
          st.nx_c0b = (uint32_t)(has_next ? groups[gi + 1].c0 * esz : 0);
          psteps[G.first + i] = st;
        }
      }
    } else {
      psteps.clear();
    }
  }

  // bias is read as float4 per 32-row block of a 256-row tile: pad to a multiple of 256
  const int bias_pad = (cout + 255) / 256 * 256;
  std::vector<float> bias_h(bias_pad, 0.f);
  if (bias) std::memcpy(bias_h.data(), bias, sizeof(float) * cout);

  bool ok = true;
  ok = ok && !okp_check_hip(hipMalloc(&plan->weights_dev, w_bytes), "hipMalloc(weights)");
  ok = ok && !okp_check_hip(hipMalloc((void**)&plan->bias_dev, sizeof(float) * bias_pad), "hipMalloc(bias)");
  ok = ok && !okp_check_hip(hipMalloc((void**)&plan->slices_dev, sizeof(OkpSlice) * slices.size()), "hipMalloc(slices)");
  if (ok) {
    if (okp_is16(dtype)) {
      std::vector<uint16_t> h(n_el);
      if (dtype == OKP_BF16) for (size_t i = 0; i < n_el; ++i) h[i] = f32_to_bf16_rne(packed[i]);
      else for (size_t i = 0; i < n_el; ++i) h[i] = f32_to_f16_rne(packed[i]);
      ok = !okp_check_hip(hipMemcpy(plan->weights_dev, h.data(), w_bytes, hipMemcpyHostToDevice), "hipMemcpy(weights)");
      // 1x1 plans consumed by the resident kernels (okp_fire2, okp_fire_chain, okp_heads) also get their weights in MFMA-fragment
      // order: lane (j = l & 15, q = l >> 4) of wave w, block b, k-step ks holds the 16 bytes of channel 32 w + 2 j + b at K offset
      // 32 ks + 8 q, so that one wave load is 1 KiB contiguous.  Built here, synchronously, so that a plan is immutable after
      // creation (no lazy allocation inside a launch: safe under hipGraph capture and with several host threads / streams).
      if (ok && n_taps == 1 && n_src == 1 && cin[0] % 32 == 0 && cout % 32 == 0) {
        const int ksteps = cin[0] / 32, n_waves = cout / 32;
        const size_t n16 = (size_t)n_waves * 2 * ksteps * 64;               // 16-byte units
        std::vector<uint16_t> fr(n16 * 8);
        for (size_t i = 0; i < n16; ++i) {
          const int lane = (int)(i & 63), ks = (int)((i >> 6) % ksteps), b = (int)(((i >> 6) / ksteps) & 1), w = (int)((i >> 6) / ksteps / 2);
          const int ch = 32 * w + 2 * (lane & 15) + b, q = lane >> 4;
          std::memcpy(&fr[i * 8], &h[(((size_t)(ks >> 1) * plan->cout_pad + ch) * 8 + (ks & 1) * 4 + q) * 8], 16);
        }
        ok = !okp_check_hip(hipMalloc(&plan->frag_dev, n16 * 16), "hipMalloc(fragment-order weights)");
        ok = ok && !okp_check_hip(hipMemcpy(plan->frag_dev, fr.data(), n16 * 16, hipMemcpyHostToDevice), "hipMemcpy(fragment-order weights)");
        // okp_fire2 multiplies with the CHANNELS as MFMA rows (a lane then holds eight adjacent channels of one pixel: 16-byte
        // loads / stores): row i = lane & 15 of block b is channel 32 w + 8 (i >> 2) + 4 b + (i & 3)
        for (size_t i = 0; i < n16; ++i) {
          const int lane = (int)(i & 63), ks = (int)((i >> 6) % ksteps), b = (int)(((i >> 6) / ksteps) & 1), w = (int)((i >> 6) / ksteps / 2);
          const int r = lane & 15, ch = 32 * w + 8 * (r >> 2) + 4 * b + (r & 3), q = lane >> 4;
          std::memcpy(&fr[i * 8], &h[(((size_t)(ks >> 1) * plan->cout_pad + ch) * 8 + (ks & 1) * 4 + q) * 8], 16);
        }
        ok = ok && !okp_check_hip(hipMalloc(&plan->fragT_dev, n16 * 16), "hipMalloc(fragment-order weights, channel rows)");
        ok = ok && !okp_check_hip(hipMemcpy(plan->fragT_dev, fr.data(), n16 * 16, hipMemcpyHostToDevice), "hipMemcpy(fragment-order weights, channel rows)");
      }
    } else if (dtype == OKP_F32X3) {
      // split-product plans: every 128-byte row (32 fp32 of K) becomes [hi k0-15 | lo k0-15 | hi k16-31 | lo k16-31] in fp16,
      // hi = fp16(w), lo = fp16(w - hi): the weight fragments of okp_igemm_kernel<F32S> are read ready-made.
      // A channel's weights are first multiplied by a power of two that puts the largest of them in [256, 512): the low halves
      // of all weights down to 2^-12 of it are then normal fp16 numbers whatever the scale BatchNorm folding left them at (a row of
      // weights around 1e-5 would otherwise have no low halves at all).  The kernel multiplies the accumulator by 1 / scale (exact).
      std::vector<float> osc(bias_pad, 1.f);
      for (int co = 0; co < cout; ++co) {
        float mx = 0.f;
        for (int si = 0; si < plan->n_slices; ++si)
          for (int k = 0; k < 32; ++k) mx = std::max(mx, std::fabs(packed[((size_t)si * plan->cout_pad + co) * 32 + k]));
        if (mx > 0.f && std::isfinite(mx)) {
          int e; std::frexp(mx, &e);                       // mx = f * 2^e, f in [0.5, 1)
          const float sc = std::ldexp(1.f, 9 - e);         // mx * sc in [256, 512)
          osc[co] = 1.f / sc;
          for (int si = 0; si < plan->n_slices; ++si)
            for (int k = 0; k < 32; ++k) packed[((size_t)si * plan->cout_pad + co) * 32 + k] *= sc;
        }
      }
      ok = !okp_check_hip(hipMalloc((void**)&plan->oscale_dev, sizeof(float) * bias_pad), "hipMalloc(oscale)");
      ok = ok && !okp_check_hip(hipMemcpy(plan->oscale_dev, osc.data(), sizeof(float) * bias_pad, hipMemcpyHostToDevice), "hipMemcpy(oscale)");
      std::vector<uint16_t> h(n_el * 2);
      for (size_t r = 0; r < n_el / 32; ++r)
        for (int k = 0; k < 32; ++k) {
          const float w = packed[r * 32 + k];
          const _Float16 hi = (_Float16)w;
          const _Float16 lo = (_Float16)(w - (float)hi);
          uint16_t uh, ul;
          std::memcpy(&uh, &hi, 2); std::memcpy(&ul, &lo, 2);
          const size_t base = r * 64 + (size_t)(k / 16) * 32 + (k % 16);
          h[base] = uh; h[base + 16] = ul;
        }
      ok = ok && !okp_check_hip(hipMemcpy(plan->weights_dev, h.data(), w_bytes, hipMemcpyHostToDevice), "hipMemcpy(weights)");
      // 1x1 plans consumed by the one-launch fire module of this configuration (okp_fire_x3.hip) also get their (scaled) weights in
      // MFMA-fragment order, channels as rows: lane (i = l & 15, q = l >> 4) of wave w, k-step ks holds the hi (then the lo) halves of
      // channel 16 w + i at K offset 32 ks + 8 q .. + 7 - one wave load = 1 KiB contiguous.  Built here: a plan is immutable afterwards.
      if (ok && n_taps == 1 && n_src == 1 && cin[0] % 32 == 0 && cout % 16 == 0 && !tap_terms) {
        const int ksteps = cin[0] / 32, n_waves = cout / 16;
        std::vector<uint16_t> fr((size_t)n_waves * 2 * ksteps * 64 * 8);
        for (int w = 0; w < n_waves; ++w)
          for (int ks = 0; ks < ksteps; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
              const int ch = 16 * w + (lane & 15), q = lane >> 4;
              for (int e = 0; e < 8; ++e) {
                const float v = packed[((size_t)ks * plan->cout_pad + ch) * 32 + 8 * q + e];      // (slice ks = channels 32 ks .. 32 ks + 31 of the one tap)
                const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                std::memcpy(&fr[((((size_t)w * 2 + 0) * ksteps + ks) * 64 + lane) * 8 + e], &hi, 2);
                std::memcpy(&fr[((((size_t)w * 2 + 1) * ksteps + ks) * 64 + lane) * 8 + e], &lo, 2);
              }
            }
        ok = !okp_check_hip(hipMalloc(&plan->fragT_dev, fr.size() * 2), "hipMalloc(fragment-order split weights)");
        ok = ok && !okp_check_hip(hipMemcpy(plan->fragT_dev, fr.data(), fr.size() * 2, hipMemcpyHostToDevice), "hipMemcpy(fragment-order split weights)");
      }
    } else {
      ok = !okp_check_hip(hipMemcpy(plan->weights_dev, packed.data(), w_bytes, hipMemcpyHostToDevice), "hipMemcpy(weights)");
    }
    ok = ok && !okp_check_hip(hipMemcpy(plan->bias_dev, bias_h.data(), sizeof(float) * bias_pad, hipMemcpyHostToDevice), "hipMemcpy(bias)");
    ok = ok && !okp_check_hip(hipMemcpy(plan->slices_dev, slices.data(), sizeof(OkpSlice) * slices.size(), hipMemcpyHostToDevice), "hipMemcpy(slices)");
    if (ok && !psteps.empty()) {
      ok = !okp_check_hip(hipMalloc((void**)&plan->patch_steps_dev, sizeof(OkpPatchStep) * psteps.size()), "hipMalloc(patch steps)");
      ok = ok && !okp_check_hip(hipMemcpy(plan->patch_steps_dev, psteps.data(), sizeof(OkpPatchStep) * psteps.size(), hipMemcpyHostToDevice), "hipMemcpy(patch steps)");
    }
  }
  if (!ok) { okp_conv_destroy(plan); return nullptr; }
  plan->w_bytes = (uint32_t)w_bytes;
  return plan;
}

extern "C" int okp_conv_set_range_flag(okp_conv* plan, int32_t* flag_dev) {
  if (!plan || plan->dtype != OKP_F32X3) { okp_set_error("okp_conv_set_range_flag: a split-product (OKP_F32X3) plan"); return OKP_EINVAL; }
  plan->range_flag = flag_dev;
  return OKP_OK;
}

extern "C" void okp_conv_destroy(okp_conv* plan) {
  if (!plan) return;
  if (plan->weights_dev) (void)hipFree(plan->weights_dev);
  if (plan->bias_dev) (void)hipFree(plan->bias_dev);
  if (plan->oscale_dev) (void)hipFree(plan->oscale_dev);
  if (plan->slices_dev) (void)hipFree(plan->slices_dev);
  if (plan->frag_dev) (void)hipFree(plan->frag_dev);
  if (plan->fragT_dev) (void)hipFree(plan->fragT_dev);
  if (plan->patch_steps_dev) (void)hipFree(plan->patch_steps_dev);
  delete plan;
}

namespace {
int check_view(const char* name, const okp_tensor& t, int esz, bool required, int align = 16, int64_t max_bytes = 0x7FFF0000ll) {
  if (!t.data) {
    if (required) { okp_set_error("okp_conv_forward: %s is null", name); return OKP_EINVAL; }
    return OKP_OK;
  }
  if ((t.pix_stride * esz) % align || ((uintptr_t)t.data) % align) { okp_set_error("okp_conv_forward: %s is not %d-byte aligned (pix_stride %d)", name, align, t.pix_stride); return OKP_EINVAL; }
  if (t.bytes <= 0 || t.bytes >= max_bytes) { okp_set_error("okp_conv_forward: %s spans %lld bytes; views must be < 2 GiB (sub-batch the frames)", name, (long long)t.bytes); return OKP_EINVAL; }
  if (t.h < 1 || t.w < 1) { okp_set_error("okp_conv_forward: %s has empty spatial size", name); return OKP_EINVAL; }
  return OKP_OK;
}
}  // namespace

static int select_tile(const okp_conv* plan, const okp_conv_args* a);

extern "C" int okp_conv_forward(const okp_conv* plan, const okp_conv_args* a, void* stream) {
  if (!plan || !a) { okp_set_error("okp_conv_forward: null plan/args"); return OKP_EINVAL; }
  const int esz = okp_esz(plan->dtype);
  if (a->n < 1 || a->ho < 1 || a->wo < 1) { okp_set_error("okp_conv_forward: empty problem n=%d ho=%d wo=%d", a->n, a->ho, a->wo); return OKP_EINVAL; }
  if ((long)a->n * a->ho * a->wo >= 0x7FFFFFFFl) { okp_set_error("okp_conv_forward: too many output pixels"); return OKP_EINVAL; }
  for (int s = 0; s < plan->n_src; ++s) {
    // sources are read with dword-aligned 16-byte buffer loads (the packed stem frame has 8-byte bf16 pixels);
    // out/res rows are written/read as aligned 16-byte vectors
    // (a SOURCE of a split-product plan may pass 2 GiB where the launch runs on the patch-resident kernel, which addresses frame by frame:
    //  checked again below, once the tile is known)
    if (int e = check_view(s ? "src[1]" : "src[0]", a->src[s], esz, true, 8, plan->dtype == OKP_F32X3 ? (1ll << 40) : 0x7FFF0000ll)) return e;
    // cin may span several consecutive pixels of a row (the stem reads 8 px x 4 ch per tap)
    if (a->src[s].pix_stride < plan->cin[s] && plan->cin[s] % a->src[s].pix_stride != 0) {
      okp_set_error("okp_conv_forward: src[%d] pix_stride %d incompatible with cin %d", s, a->src[s].pix_stride, plan->cin[s]); return OKP_EINVAL;
    }
  }
  const bool x3 = plan->dtype == OKP_F32X3;
  if ((a->out16.data || a->res_is_f16) && !x3) { okp_set_error("okp_conv_forward: out16 / res_is_f16 belong to OKP_F32X3 plans"); return OKP_EINVAL; }
  if ((a->out16.data || a->res_is_f16) && (a->dw_w_dev)) { okp_set_error("okp_conv_forward: out16 / res_is_f16 cannot be combined with the depth-wise branch"); return OKP_EINVAL; }
  const bool sub2 = a->out_subsample == 2;
  if (a->out_subsample != 0 && a->out_subsample != 1 && !sub2) { okp_set_error("okp_conv_forward: out_subsample %d (0, 1 or 2)", a->out_subsample); return OKP_EINVAL; }
  if (sub2 && (!x3 || !a->out16.data || a->out_step != 1 || a->out_oy || a->out_ox || a->n_classes > 1 || a->dw_w_dev ||
               a->out16.h != a->ho || a->out16.w != a->wo || a->out.h != (a->ho + 1) / 2 || a->out.w != (a->wo + 1) / 2)) {
    okp_set_error("okp_conv_forward: out_subsample 2 needs an OKP_F32X3 plan, a full-grid out16, out_step 1 and out of ceil(ho/2) x ceil(wo/2) pixels"); return OKP_EINVAL;
  }
  if (a->src_pairs || a->out_pairs) {
    bool ok = x3 && !a->out16.data && !a->res_is_f16 && !sub2 && !a->dw_w_dev && a->out.data && (a->src_pairs >> plan->n_src) == 0 && a->src_pairs >= 0 && plan->cout % 8 == 0;
    for (int s = 0; s < plan->n_src; ++s)
      if ((a->src_pairs >> s) & 1) ok = ok && plan->cin[s] % 32 == 0 && a->src[s].pix_stride % 8 == 0 && ((uintptr_t)a->src[s].data) % 32 == 0;
    // (a pair group is 8 channels = 32 bytes of the tensor: a window starts on a group, pixels are whole groups apart)
    if (a->out_pairs) ok = ok && a->out.pix_stride % 8 == 0 && ((uintptr_t)a->out.data) % 32 == 0;
    if (!ok) {
      okp_set_error("okp_conv_forward: src_pairs / out_pairs (pair-format tensors) belong to OKP_F32X3 plans with an fp32-geometry out, whole 32-channel chunks per pair source, views that start on a 32-byte pair group with pix_stride % 8 == 0, and no out16 / res_is_f16 / out_subsample / depth-wise branch");
      return OKP_EINVAL;
    }
  }
  if (a->out16.data) {
    if (int e = check_view("out16", a->out16, 2, true)) return e;
    if ((!sub2 && (a->out16.h != a->out.h || a->out16.w != a->out.w)) || a->out16.pix_stride < plan->cout) { okp_set_error("okp_conv_forward: out16 does not match out"); return OKP_EINVAL; }
    if (!a->out.data && (a->out.h < 1 || a->out.w < 1 || a->out.pix_stride < plan->cout)) { okp_set_error("okp_conv_forward: out (data NULL) must still describe the output grid"); return OKP_EINVAL; }
  }
  if (a->out.data || !a->out16.data) { if (int e = check_view("out", a->out, esz, true)) return e; }
  if (int e = check_view("res", a->res, a->res_is_f16 ? 2 : esz, false)) return e;
  if (a->out_step < 1 || a->out_oy < 0 || a->out_ox < 0 ||
      (!sub2 && ((a->ho - 1) * a->out_step + a->out_oy >= a->out.h || (a->wo - 1) * a->out_step + a->out_ox >= a->out.w))) {
    okp_set_error("okp_conv_forward: output grid %dx%d (step %d, offset %d,%d) does not fit out %dx%d", a->ho, a->wo, a->out_step, a->out_oy, a->out_ox, a->out.h, a->out.w);
    return OKP_EINVAL;
  }
  if (a->out.pix_stride < plan->cout || (a->res.data && a->res.pix_stride < plan->cout)) { okp_set_error("okp_conv_forward: out/res pix_stride < cout %d", plan->cout); return OKP_EINVAL; }
  if (a->res.data && (a->res.h != (sub2 ? a->out16.h : a->out.h) || a->res.w != (sub2 ? a->out16.w : a->out.w))) { okp_set_error("okp_conv_forward: residual spatial size differs from out"); return OKP_EINVAL; }
  if (a->tile < 0 || a->tile > 14 || !((1u << a->tile) & 0x615Fu)) {       // 0 (heuristic), 1, 2, 3, 4, 6, 8, 13, 14
    okp_set_error("okp_conv_forward: tile %d is not one of 0 (heuristic), 1, 2, 3, 4, 6, 8, 13, 14", a->tile); return OKP_EINVAL;
  }

  OkpIgemmParams p;
  std::memset(&p, 0, sizeof(p));
  for (int s = 0; s < 2; ++s) {
    const int ss = s < plan->n_src ? s : 0;
    p.src[s] = a->src[ss].data; p.src_bytes[s] = (uint32_t)(a->src[ss].bytes < 0x7FFF0000ll ? a->src[ss].bytes : 0x7FFF0000ll); p.src_bytes64[s] = a->src[ss].bytes;
    p.srcH[s] = a->src[ss].h; p.srcW[s] = a->src[ss].w; p.src_pix_stride[s] = a->src[ss].pix_stride;
    p.conv_stride[s] = plan->conv_stride[ss];
  }
  p.weights = plan->weights_dev; p.w_bytes = plan->w_bytes; p.cout_pad = plan->cout_pad; p.bias = plan->bias_dev; p.oscale = plan->oscale_dev;
  p.slices = plan->slices_dev; p.n_slices = plan->n_slices; p.n_taps = plan->n_taps;
  p.N = a->n; p.Ho = a->ho; p.Wo = a->wo;
  p.div_howo = okp_fastdiv((uint32_t)(a->ho * a->wo)); p.div_wo = okp_fastdiv((uint32_t)a->wo);
  p.out = a->out.data; p.OH = a->out.h; p.OW = a->out.w; p.out_step = a->out_step; p.out_oy = a->out_oy; p.out_ox = a->out_ox;
  p.out_pix_stride = a->out.pix_stride; p.cout = plan->cout;
  p.out_bytes = (uint32_t)a->out.bytes; p.res_bytes = a->res.data ? (uint32_t)a->res.bytes : 0u;
  p.res = a->res.data; p.res_pix_stride = a->res.pix_stride; p.act = plan->act;
  for (int t = 0; t < plan->n_taps; ++t) p.taps[t] = plan->taps[t];
  p.n_classes = a->n_classes > 1 ? a->n_classes : 1;
  if (p.n_classes > 1) {
    if (p.n_classes != 4 || a->out_step != 2 || plan->n_slices % 4 || plan->n_taps % 4 || a->dw_w_dev) {
      okp_set_error("okp_conv_forward: n_classes must be 4 with out_step 2 and four equal tap groups"); return OKP_EINVAL;
    }
    if ((a->ho - 1) * 2 + a->out_oy + 1 >= a->out.h || (a->wo - 1) * 2 + a->out_ox + 1 >= a->out.w) { okp_set_error("okp_conv_forward: sub-pixel classes do not fit out"); return OKP_EINVAL; }
  }
  if (p.n_classes > 1 && plan->n_single_slices != 0 && plan->n_single_slices != plan->n_slices) {
    // (single-term slices are sorted in front of the whole K range: with sub-pixel classes, each a K range of its own, that order
    //  would move slices across class boundaries - a 4-tap plan passes the creation-time check of the 8+-tap case)
    okp_set_error("okp_conv_forward: sub-pixel classes need one term count for all taps of the plan (%d of %d K-slices are single-term)",
                  plan->n_single_slices, plan->n_slices);
    return OKP_EINVAL;
  }
  p.slices_per_class = plan->n_slices / p.n_classes;
  p.n_single_slices = plan->n_single_slices / p.n_classes;
  p.out16 = a->out16.data; p.out16_pix_stride = a->out16.pix_stride; p.res16 = a->res_is_f16 ? 1 : 0;
  if (sub2) { p.out_sub2 = 1; p.OH2 = a->out.h; p.OW2 = a->out.w; p.OH = a->out16.h; p.OW = a->out16.w; }
  p.src_pairs = a->src_pairs; p.out_pairs = a->out_pairs ? 1 : 0;
  p.range_flag = plan->range_flag;
  if (a->dw_w_dev) {
    if (!a->dw_bias_dev) { okp_set_error("okp_conv_forward: dw_w_dev without dw_bias_dev"); return OKP_EINVAL; }
    if (plan->cin[0] != plan->cout || a->out_step != 1) { okp_set_error("okp_conv_forward: the fused depth-wise branch needs cin[0] == cout and out_step 1"); return OKP_EINVAL; }
    if (int e = check_view("dw_out", a->dw_out, esz, true)) return e;
    if (int e = check_view("dw_res", a->dw_res, esz, false)) return e;
    if (a->dw_out.h != a->out.h || a->dw_out.w != a->out.w || a->dw_out.pix_stride < plan->cout) { okp_set_error("okp_conv_forward: dw_out does not match out"); return OKP_EINVAL; }
    p.dw_w = a->dw_w_dev; p.dw_bias = a->dw_bias_dev; p.dw_out = a->dw_out.data; p.dw_res = a->dw_res.data;
    p.dw_out_pix_stride = a->dw_out.pix_stride; p.dw_res_pix_stride = a->dw_res.pix_stride;
  }
  const int tile = a->tile ? a->tile : select_tile(plan, a);
  for (int s = 0; s < plan->n_src; ++s)
    if (a->src[s].bytes >= 0x7FFF0000ll && tile != 13) {
      okp_set_error("okp_conv_forward: src[%d] spans %lld bytes; only the patch-resident split-product kernel (tile 13; this launch: tile %d) reads views of 2 GiB and more", s, (long long)a->src[s].bytes, tile);
      return OKP_EINVAL;
    }
  if ((p.src_pairs || p.out_pairs) && tile != 13) {
    okp_set_error("okp_conv_forward: pair-format tensors are read / written by the patch-resident split-product kernel only (tile 13; this launch: tile %d)", tile);
    return OKP_EINVAL;
  }
  return okp_launch_igemm(plan, p, tile, (hipStream_t)stream);
}

// Tile heuristic of a launch.  The patch-resident kernel (13) takes over from the 256x256 gather tile where a source has
// several taps (3x3 convolutions: the patch is read once instead of once per tap) and the problem is made of whole
// 16x16-pixel blocks written densely.
static int select_tile(const okp_conv* plan, const okp_conv_args* a) {
  // (the sub-pixel classes of a transposed convolution are separate tiles of the same launch: they count towards filling the CUs -
  //  8x8 -> 16x16 at N=64, 384 channels: 128x128 tiles 34 us, the 64x64 ones the per-class count chose 45 us; 4x4 -> 8x8 stays at 64x64: 17 vs 23 us)
  const int ncls0 = a->n_classes > 1 ? a->n_classes : 1;
  const long px = (long)a->n * a->ho * a->wo;
  int tile = okp_select_tile(plan->dtype, plan->cout_pad, px);
  if (ncls0 > 1 && tile != 6 && tile != 3) {
    const int t2 = okp_select_tile(plan->dtype, plan->cout_pad, px * ncls0);
    if (t2 == 2 || tile == 2) tile = 2;
  }
  if (plan->dtype == OKP_F32X3 && tile == 3) {
    // split-product plans with a short reduction (1x1 convolutions: K <= 512) are bound by their fp32 tensors, not by the matrix pipe:
    // the 128 x 256 tile runs two workgroups per CU, so one streams its rows in while the other writes its tile out (N=64, 64x64:
    // heads layer 1 342 -> 230 us, pre[2] skip 252 -> 199 us, pre[1] skip 704 -> 562 us, two-source merge 373 -> 359 us)
    int k = 0;
    for (int t = 0; t < plan->n_taps; ++t) k += plan->cin[plan->taps[t].src];
    if (k <= 512) tile = 4;
  }
  static const bool patch_on = [] { const char* e = getenv("OKP_PATCH"); return !(e && e[0] == '0'); }();   // OKP_PATCH=0: A/B against the gather tile
  const int ncls = a->n_classes > 1 ? a->n_classes : 1;
  const bool dense1 = ncls == 1 && a->out_step == 1 && a->out_oy == 0 && a->out_ox == 0 && a->out.h == a->ho && a->out.w == a->wo;
  const long patch_tiles = (long)ncls * a->n * (a->ho / 16) * (a->wo / 16) * (plan->cout_pad / 256);
  if (patch_on && plan->patch_steps_dev && plan->n_taps > plan->patch_n_geom && !a->dw_w_dev && a->ho % 16 == 0 && a->wo % 16 == 0 &&
      (((tile == 6 || (tile == 3 && plan->dtype == OKP_F32X3)) && dense1) || (ncls == 4 && patch_tiles >= 256))) {
    bool ok = true;
    for (int s = 0; s < plan->n_src; ++s) ok = ok && a->src[s].pix_stride >= plan->cin[s] && (a->src[s].pix_stride * okp_esz(plan->dtype)) % 8 == 0;
    // (split-product plans: the fp16 side output / fp16 residual / subsampled output of the mixed configuration stay with the gather tiles)
    if (plan->dtype == OKP_F32X3 && (a->out16.data || a->res_is_f16 || a->out_subsample == 2 || !a->out.data || plan->n_single_slices)) ok = false;
    if (ok) return 13;
  }
  return tile;
}

extern "C" int okp_conv_select_tile(const okp_conv* plan, const okp_conv_args* a) {
  if (!plan || !a) return 0;
  if (a->tile) return a->tile;
  return select_tile(plan, a);
}

// okp_patch_supported on the fields of the args it reads: what a caller that forces tile 13 (or wants pair-format tensors) asks first
extern "C" int okp_conv_patch_applies(const okp_conv* plan, const okp_conv_args* a) {
  if (!plan || !a) return 0;
  OkpIgemmParams p;
  std::memset(&p, 0, sizeof(p));
  p.n_classes = a->n_classes > 1 ? a->n_classes : 1;
  p.dw_w = a->dw_w_dev; p.Ho = a->ho; p.Wo = a->wo;
  p.out16 = a->out16.data; p.res16 = a->res_is_f16 ? 1 : 0; p.out_sub2 = a->out_subsample == 2 ? 1 : 0;
  p.out = a->out.data ? a->out.data : (a->out16.data ? nullptr : (void*)1);      // (shape-only queries carry no pointers: "out is written")
  for (int s = 0; s < plan->n_src; ++s) p.src_pix_stride[s] = a->src[s].pix_stride;
  return okp_patch_supported(plan, p) ? 1 : 0;
}

extern "C" int64_t okp_conv_macs(const okp_conv* plan, const okp_conv_args* a) {
  if (!plan || !a) return 0;
  int64_t k = 0;
  for (int t = 0; t < plan->n_taps; ++t) k += plan->cin[plan->taps[t].src];
  return (int64_t)a->n * a->ho * a->wo * plan->cout * k;
}

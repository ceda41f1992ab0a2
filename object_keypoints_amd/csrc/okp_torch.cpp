// torch.ops.okp.* : the launch entry points of the C ABI (include/okp.h) registered with the PyTorch dispatcher.
//
// north_star asks for the HIP kernels to be "surfaced through PyTorch-ROCm custom ops" ("What the native
// replacement must export").  This translation unit is a thin shim over the SAME extern "C" symbols of libokp_hip.so that the
// ctypes binding uses: it takes torch tensors (NHWC activations as (tensor, first channel) pairs: a channel window of a wider
// tensor is how concat / split are free), checks device / dtype / layout with TORCH_CHECK as SURVEY §8(b) "Error conventions"
// prescribes, fills the C structs and calls the C entry point on the stream given by the caller.  No kernel lives here and
// no HIP header is needed.  Round 6: EVERY entry point of include/okp.h is registered - plan creation from host weight tensors
// (conv_create, conv_bn_create = fold BatchNorm + build the tap list + pack, stem_create), the ABI-5/6 arguments of
// okp_conv_forward (fp16 side output, fp16 residual, subsampled output, pair-format sources / outputs), the stems, the heads
// (16-bit and split-product), the boundary kernels (pack / preprocess / cast / add) and the geometry (undistort, unproject,
// triangulate) - so a torch-only caller builds and runs the whole path; plans are opaque int handles owned by the caller
// (conv_destroy / stem_destroy).
//
// Cost per launch from Python: one dispatcher call (~4 us) instead of ~20 us of ctypes struct filling.
#include <ATen/ATen.h>
#include <cstring>
#include <torch/library.h>

#include <tuple>
#include <vector>

#include "okp.h"

namespace {

using at::Tensor;
using c10::optional;

inline void check_rc(int rc, const char* what) {
  TORCH_CHECK(rc == OKP_OK, what, " failed (", rc, "): ", okp_last_error());
}

inline int esz_of(const Tensor& t) {
  switch (t.scalar_type()) {
    case at::kFloat: return 4;
    case at::kBFloat16: case at::kHalf: return 2;
    default: TORCH_CHECK(false, "okp: unsupported activation dtype ", t.scalar_type()); return 0;
  }
}

inline int dtype_of(const Tensor& t) {
  switch (t.scalar_type()) {
    case at::kFloat: return OKP_F32;
    case at::kBFloat16: return OKP_BF16;
    case at::kHalf: return OKP_F16;
    default: TORCH_CHECK(false, "okp: unsupported activation dtype ", t.scalar_type()); return 0;
  }
}

// NHWC activation window: tensor [N,H,W,C] (contiguous, on the device), channels [c0, C)
inline okp_tensor view(const Tensor& t, int64_t c0, const char* name) {
  TORCH_CHECK(t.is_cuda(), "okp: ", name, " must be a device tensor: the HIP path has no CPU fallback");
  TORCH_CHECK(t.dim() == 4 && t.is_contiguous(), "okp: ", name, " must be a contiguous [N,H,W,C] tensor");
  TORCH_CHECK(c0 >= 0 && c0 < t.size(3), "okp: ", name, ": channel window outside the tensor");
  const int e = esz_of(t);
  okp_tensor v;
  v.data = static_cast<char*>(t.data_ptr()) + c0 * e;
  v.h = (int32_t)t.size(1); v.w = (int32_t)t.size(2); v.pix_stride = (int32_t)t.size(3);
  v.bytes = t.numel() * e - c0 * e;
  return v;
}

const okp_tensor kNull = {nullptr, 0, 0, 0, 0};

void conv_forward(int64_t plan, const Tensor& src0, int64_t src0_c0, const optional<Tensor>& src1, int64_t src1_c0,
                  const Tensor& out, int64_t out_c0, int64_t ho, int64_t wo,
                  const optional<Tensor>& res, int64_t res_c0, int64_t out_step, int64_t oy, int64_t ox, int64_t tile, int64_t n_classes,
                  const optional<Tensor>& dw_w, const optional<Tensor>& dw_b, const optional<Tensor>& dw_out, int64_t dw_out_c0,
                  const optional<Tensor>& dw_res, int64_t dw_res_c0, int64_t stream,
                  const optional<Tensor>& out16, int64_t out16_c0, bool write_out, int64_t out_subsample, int64_t src_pairs, bool out_pairs) {
  okp_conv_args a;
  std::memset(&a, 0, sizeof(a));
  a.n = (int32_t)out.size(0); a.ho = (int32_t)ho; a.wo = (int32_t)wo;
  a.src[0] = view(src0, src0_c0, "src[0]");
  a.src[1] = src1.has_value() ? view(*src1, src1_c0, "src[1]") : kNull;
  TORCH_CHECK(!src1.has_value() || src1->scalar_type() == src0.scalar_type(), "okp: sources of different dtypes");
  TORCH_CHECK(out.scalar_type() == src0.scalar_type(), "okp: out dtype differs from the sources'");
  a.out = view(out, out_c0, "out");
  a.out_step = (int32_t)out_step; a.out_oy = (int32_t)oy; a.out_ox = (int32_t)ox;
  a.res = res.has_value() ? view(*res, res_c0, "res") : kNull;
  // ABI 5 / 6 (split-product plans): fp16 side output, fp16 residual (told by the residual's dtype), subsampled fp32 output, pair format
  if (res.has_value() && res->scalar_type() != out.scalar_type()) {
    TORCH_CHECK(res->scalar_type() == at::kHalf && out.scalar_type() == at::kFloat, "okp: the residual has the plan's element type (split-product plans also take a float16 one)");
    a.res_is_f16 = 1;
  }
  if (out16.has_value()) {
    TORCH_CHECK(out16->scalar_type() == at::kHalf, "okp: out16 is a float16 tensor");
    a.out16 = view(*out16, out16_c0, "out16");
  }
  TORCH_CHECK(write_out || out16.has_value(), "okp: write_out = False needs out16");
  if (!write_out) a.out.data = nullptr;
  a.out_subsample = (int32_t)out_subsample;
  a.src_pairs = (int32_t)src_pairs; a.out_pairs = out_pairs ? 1 : 0;
  a.tile = (int32_t)tile;
  a.n_classes = (int32_t)n_classes;
  a.dw_w_dev = nullptr; a.dw_bias_dev = nullptr; a.dw_out = kNull; a.dw_res = kNull;
  if (dw_w.has_value()) {
    TORCH_CHECK(dw_b.has_value() && dw_out.has_value(), "okp: dw_w without dw_b / dw_out");
    TORCH_CHECK(dw_w->is_cuda() && dw_w->scalar_type() == at::kFloat && dw_b->is_cuda() && dw_b->scalar_type() == at::kFloat, "okp: depth-wise weights must be fp32 device tensors");
    a.dw_w_dev = dw_w->data_ptr<float>(); a.dw_bias_dev = dw_b->data_ptr<float>();
    a.dw_out = view(*dw_out, dw_out_c0, "dw_out");
    if (dw_res.has_value()) a.dw_res = view(*dw_res, dw_res_c0, "dw_res");
  }
  check_rc(okp_conv_forward(reinterpret_cast<const okp_conv*>(plan), &a, reinterpret_cast<void*>(stream)), "okp_conv_forward");
}

int64_t conv_select_tile(int64_t plan, const Tensor& src0, int64_t src0_c0, const optional<Tensor>& src1, int64_t src1_c0,
                         const Tensor& out, int64_t out_c0, int64_t ho, int64_t wo, int64_t out_step, int64_t oy, int64_t ox,
                         int64_t tile, int64_t n_classes, bool has_dw, bool extended) {
  okp_conv_args a;
  std::memset(&a, 0, sizeof(a));
  a.n = (int32_t)out.size(0); a.ho = (int32_t)ho; a.wo = (int32_t)wo;
  a.src[0] = view(src0, src0_c0, "src[0]");
  a.src[1] = src1.has_value() ? view(*src1, src1_c0, "src[1]") : kNull;
  a.out = view(out, out_c0, "out");
  a.out_step = (int32_t)out_step; a.out_oy = (int32_t)oy; a.out_ox = (int32_t)ox;
  a.res = kNull; a.tile = (int32_t)tile; a.n_classes = (int32_t)n_classes;
  static const float dummy = 0.f;
  a.dw_w_dev = has_dw ? &dummy : nullptr; a.dw_bias_dev = nullptr; a.dw_out = kNull; a.dw_res = kNull;
  if (extended) a.res_is_f16 = 1;      // (out16 / fp16 residual / subsampled output: the heuristic only asks whether any of them is present)
  return okp_conv_select_tile(reinterpret_cast<const okp_conv*>(plan), &a);
}

void fire_forward(int64_t squeeze, int64_t expand, const Tensor& wd, const Tensor& bd, const Tensor& x, int64_t x_c0,
                  const Tensor& out, int64_t out_c0, int64_t stride, bool skip, int64_t stream) {
  TORCH_CHECK(wd.is_cuda() && bd.is_cuda() && wd.scalar_type() == at::kFloat && bd.scalar_type() == at::kFloat, "okp: depth-wise weights must be fp32 device tensors");
  okp_fire_args a;
  a.n = (int32_t)x.size(0);
  a.x = view(x, x_c0, "x"); a.out = view(out, out_c0, "out");
  a.stride = (int32_t)stride; a.skip = skip ? 1 : 0;
  check_rc(okp_fire_forward(reinterpret_cast<const okp_conv*>(squeeze), reinterpret_cast<const okp_conv*>(expand), wd.data_ptr<float>(),
                            bd.data_ptr<float>(), &a, reinterpret_cast<void*>(stream)), "okp_fire_forward");
}

void fire_chain_forward(at::IntArrayRef squeeze, at::IntArrayRef expand, at::TensorList wd, at::TensorList bd,
                        const Tensor& x, int64_t x_c0, const Tensor& out, int64_t out_c0, int64_t stream) {
  const size_t n = squeeze.size();
  TORCH_CHECK(n >= 1 && n <= OKP_FIRE_CHAIN_MAX && expand.size() == n && wd.size() == n && bd.size() == n, "okp: fire chain of 1..", OKP_FIRE_CHAIN_MAX, " modules");
  okp_conv* sq[OKP_FIRE_CHAIN_MAX]; okp_conv* ex[OKP_FIRE_CHAIN_MAX];
  const float* w[OKP_FIRE_CHAIN_MAX]; const float* b[OKP_FIRE_CHAIN_MAX];
  for (size_t i = 0; i < n; ++i) {
    TORCH_CHECK(wd[i].is_cuda() && wd[i].scalar_type() == at::kFloat && bd[i].is_cuda() && bd[i].scalar_type() == at::kFloat, "okp: depth-wise weights must be fp32 device tensors");
    sq[i] = reinterpret_cast<okp_conv*>(squeeze[i]); ex[i] = reinterpret_cast<okp_conv*>(expand[i]);
    w[i] = wd[i].data_ptr<float>(); b[i] = bd[i].data_ptr<float>();
  }
  const okp_tensor xv = view(x, x_c0, "x"), ov = view(out, out_c0, "out");
  check_rc(okp_fire_chain_forward((int32_t)n, sq, ex, w, b, (int32_t)x.size(0), &xv, &ov, reinterpret_cast<void*>(stream)), "okp_fire_chain_forward");
}

void fill_head_args(okp_head_out_args& a, int64_t n, int64_t h, int64_t w, at::IntArrayRef in_c_off, at::IntArrayRef act,
                    at::TensorList outs, at::IntArrayRef out_ch, const Tensor& w3, const Tensor& b3) {
  const size_t no = in_c_off.size();
  TORCH_CHECK(no >= 1 && no <= OKP_HEAD_MAX_OUT && act.size() == no && outs.size() == no && out_ch.size() == no, "okp: 1..", OKP_HEAD_MAX_OUT, " head outputs");
  TORCH_CHECK(w3.is_cuda() && b3.is_cuda() && w3.scalar_type() == at::kFloat && b3.scalar_type() == at::kFloat, "okp: head output weights must be fp32 device tensors");
  a.n = (int32_t)n; a.h = (int32_t)h; a.w = (int32_t)w;
  a.src = kNull;
  a.n_out = (int32_t)no;
  for (size_t o = 0; o < no; ++o) {
    const Tensor& t = outs[o];
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.dim() == 4 && t.is_contiguous(), "okp: head outputs are contiguous fp32 [N,C,H,W] device tensors");
    TORCH_CHECK(out_ch[o] >= 0 && out_ch[o] < t.size(1) && t.size(2) == h && t.size(3) == w && t.size(0) == n, "okp: head output ", o, " does not match the maps");
    a.in_c_off[o] = (int32_t)in_c_off[o]; a.act[o] = (int32_t)act[o];
    a.out_ptr[o] = t.data_ptr<float>() + out_ch[o] * h * w;
    a.out_n_stride[o] = t.size(1) * h * w;
  }
  a.w_dev = w3.data_ptr<float>(); a.bias_dev = b3.data_ptr<float>();
}

void heads_forward(int64_t l1, int64_t l2, const Tensor& x, int64_t x_c0, at::IntArrayRef in_c_off, at::IntArrayRef act,
                   at::TensorList outs, at::IntArrayRef out_ch, const Tensor& w3, const Tensor& b3, int64_t stream) {
  okp_head_out_args a;
  fill_head_args(a, x.size(0), x.size(1), x.size(2), in_c_off, act, outs, out_ch, w3, b3);
  const okp_tensor xv = view(x, x_c0, "x");
  check_rc(okp_heads_forward(reinterpret_cast<const okp_conv*>(l1), reinterpret_cast<const okp_conv*>(l2), &a, &xv, reinterpret_cast<void*>(stream)), "okp_heads_forward");
}

void head_out_forward(const Tensor& src, int64_t src_c0, at::IntArrayRef in_c_off, at::IntArrayRef act, at::TensorList outs,
                      at::IntArrayRef out_ch, const Tensor& w3, const Tensor& b3, int64_t stream) {
  okp_head_out_args a;
  fill_head_args(a, src.size(0), src.size(1), src.size(2), in_c_off, act, outs, out_ch, w3, b3);
  a.src = view(src, src_c0, "src");
  check_rc(okp_head_out_forward(dtype_of(src), &a, reinterpret_cast<void*>(stream)), "okp_head_out_forward");
}

void stem_forward_nchw(int64_t stem, const Tensor& frames, const Tensor& out, int64_t out_c0, int64_t stream) {
  TORCH_CHECK(frames.is_cuda() && frames.scalar_type() == at::kFloat && frames.dim() == 4 && frames.size(1) == 3 && frames.is_contiguous(),
              "okp: frames must be a contiguous float32 [N,3,H,W] device tensor");
  const okp_tensor ov = view(out, out_c0, "out");
  check_rc(okp_stem_forward_nchw(reinterpret_cast<const okp_stem*>(stem), (int32_t)frames.size(0), (int32_t)frames.size(2), (int32_t)frames.size(3),
                                 frames.data_ptr<float>(), &ov, reinterpret_cast<void*>(stream)), "okp_stem_forward_nchw");
}

void peak_nms(const Tensor& heat, int64_t cap, const Tensor& count, const Tensor& yx, const Tensor& xyc, int64_t stream) {
  TORCH_CHECK(heat.is_cuda(), "okp: heat must be a device tensor: the HIP path has no CPU fallback");
  TORCH_CHECK(heat.scalar_type() == at::kFloat && heat.dim() == 4 && heat.is_contiguous(), "okp: heat must be a contiguous float32 [N,K,H,W] tensor");
  const int64_t maps = heat.size(0) * heat.size(1);
  TORCH_CHECK(count.is_cuda() && count.scalar_type() == at::kInt && count.numel() == maps && yx.scalar_type() == at::kInt && yx.numel() == maps * cap * 2 &&
              xyc.scalar_type() == at::kFloat && xyc.numel() == maps * cap * 3 && yx.is_cuda() && xyc.is_cuda(), "okp: peak buffers do not match the maps / capacity");
  check_rc(okp_peak_nms(heat.data_ptr<float>(), (int32_t)maps, (int32_t)heat.size(2), (int32_t)heat.size(3), (int32_t)cap, count.data_ptr<int32_t>(),
                        yx.data_ptr<int32_t>(), xyc.data_ptr<float>(), reinterpret_cast<void*>(stream)), "okp_peak_nms");
}

// ---- plan creation ("a prepare_weights(state_dict) that folds BN and re-lays out weights") ---------------------------

inline const float* host_f32(const Tensor& t, const char* name, std::vector<Tensor>& keep) {
  TORCH_CHECK(!t.is_cuda(), "okp: ", name, " is a HOST tensor (weights are packed and uploaded once, at plan creation)");
  Tensor c = t.to(at::kFloat).contiguous();
  keep.push_back(c);
  return c.data_ptr<float>();
}

// taps as three int lists + one [cout, cin[src]] host weight per tap; tap_terms: per tap 3 / 1 (split-product plans), empty = default
int64_t conv_create(int64_t dtype, at::IntArrayRef cin, at::IntArrayRef conv_stride, int64_t cout, at::IntArrayRef tap_src, at::IntArrayRef tap_dy,
                    at::IntArrayRef tap_dx, at::TensorList tap_w, const optional<Tensor>& bias, int64_t act, at::IntArrayRef tap_terms) {
  const size_t ns = cin.size(), nt = tap_src.size();
  TORCH_CHECK(ns >= 1 && ns <= 2 && conv_stride.size() == ns, "okp: one or two sources, a stride per source");
  TORCH_CHECK(nt >= 1 && nt <= 32 && tap_dy.size() == nt && tap_dx.size() == nt && tap_w.size() == nt, "okp: 1..32 taps, (src, dy, dx, weight) per tap");
  TORCH_CHECK(tap_terms.empty() || (tap_terms.size() == nt && dtype == OKP_F32X3), "okp: tap_terms has one entry per tap and belongs to OKP_F32X3 plans");
  int32_t ci[2] = {0, 0}, st[2] = {1, 1};
  for (size_t s = 0; s < ns; ++s) { ci[s] = (int32_t)cin[s]; st[s] = (int32_t)conv_stride[s]; }
  std::vector<Tensor> keep;
  std::vector<okp_tap> taps(nt);
  for (size_t t = 0; t < nt; ++t) {
    TORCH_CHECK(tap_src[t] >= 0 && tap_src[t] < (int64_t)ns, "okp: tap ", t, " reads source ", tap_src[t]);
    TORCH_CHECK(tap_w[t].dim() == 2 && tap_w[t].size(0) == cout && tap_w[t].size(1) == ci[tap_src[t]], "okp: tap ", t, ": weight is [cout, cin[src]]");
    taps[t].src = (int32_t)tap_src[t]; taps[t].dy = (int32_t)tap_dy[t]; taps[t].dx = (int32_t)tap_dx[t];
    taps[t].w = host_f32(tap_w[t], "a tap weight", keep);
  }
  const float* b = nullptr;
  if (bias.has_value()) { TORCH_CHECK(bias->numel() == cout, "okp: bias has cout entries"); b = host_f32(*bias, "bias", keep); }
  okp_conv* plan;
  if (!tap_terms.empty()) {
    std::vector<uint8_t> terms(nt);
    for (size_t t = 0; t < nt; ++t) terms[t] = (uint8_t)tap_terms[t];
    plan = okp_conv_create_x3((int)ns, ci, st, (int32_t)cout, (int32_t)nt, taps.data(), terms.data(), b, (int)act);
  } else {
    plan = okp_conv_create((int)dtype, (int)ns, ci, st, (int32_t)cout, (int32_t)nt, taps.data(), b, (int)act);
  }
  TORCH_CHECK(plan != nullptr, "okp_conv_create failed: ", okp_last_error());
  return reinterpret_cast<int64_t>(plan);
}

// eval-mode BatchNorm folded into an OIHW convolution weight in fp64 (scale = gamma / sqrt(var + eps), shift = beta - mean * scale
// [+ conv_bias * scale]) - the arithmetic of backbone.fold_bn, bit for bit - returned as host fp32 (weight, bias)
std::tuple<Tensor, Tensor> fold_bn(const Tensor& weight, const optional<Tensor>& bn_weight, const optional<Tensor>& bn_bias, const optional<Tensor>& bn_mean,
                                   const optional<Tensor>& bn_var, double eps, const optional<Tensor>& conv_bias) {
  Tensor w = weight.detach().to(at::kCPU, at::kDouble);
  const int64_t cout = w.size(0);
  Tensor scale, shift;
  if (bn_weight.has_value()) {
    TORCH_CHECK(bn_bias.has_value() && bn_mean.has_value() && bn_var.has_value(), "okp: BatchNorm needs weight, bias, running_mean and running_var");
    auto d = [](const Tensor& t) { return t.detach().to(at::kCPU, at::kFloat).to(at::kDouble); };
    scale = d(*bn_weight) / at::sqrt(d(*bn_var) + eps);
    shift = d(*bn_bias) - d(*bn_mean) * scale;
  } else {
    scale = at::ones({cout}, at::kDouble); shift = at::zeros({cout}, at::kDouble);
  }
  if (conv_bias.has_value()) shift = shift + conv_bias->detach().to(at::kCPU, at::kFloat).to(at::kDouble) * scale;
  std::vector<int64_t> shp(w.dim(), 1); shp[0] = cout;
  w = w.to(at::kFloat).to(at::kDouble) * scale.reshape(shp);
  return {w.to(at::kFloat).contiguous(), shift.to(at::kFloat).contiguous()};
}

// convolution(k, cin, cout, stride) [+ BatchNorm] [+ ReLU] of the reference (py_utils/utils.py:143-156) as ONE plan straight from the module's
// tensors: folds BN, builds the row-major tap list with pad (k - 1) / 2, packs and uploads.  dtype: OKP_F32 / OKP_BF16 / OKP_F16 / OKP_F32X3.
int64_t conv_bn_create(int64_t dtype, const Tensor& weight, const optional<Tensor>& bn_weight, const optional<Tensor>& bn_bias, const optional<Tensor>& bn_mean,
                       const optional<Tensor>& bn_var, double eps, const optional<Tensor>& conv_bias, int64_t stride, bool relu) {
  TORCH_CHECK(weight.dim() == 4, "okp: convolution weight is [cout, cin, kh, kw]");
  auto [w, b] = fold_bn(weight, bn_weight, bn_bias, bn_mean, bn_var, eps, conv_bias);
  const int64_t cout = w.size(0), cin = w.size(1), kh = w.size(2), kw = w.size(3);
  std::vector<int64_t> src, dy, dx; std::vector<Tensor> tw;
  for (int64_t r = 0; r < kh; ++r)
    for (int64_t c = 0; c < kw; ++c) {
      src.push_back(0); dy.push_back(r - (kh - 1) / 2); dx.push_back(c - (kw - 1) / 2);
      tw.push_back(w.select(3, c).select(2, r).contiguous());
    }
  return conv_create(dtype, {cin}, {stride}, cout, src, dy, dx, tw, b, relu ? OKP_ACT_RELU : OKP_ACT_NONE, {});
}

// shape-only question (no tensors): does a dense launch of this plan on n x (ho x wo) output blocks run on the patch-resident kernel - the one
// kernel that reads and writes pair-format tensors?  tile = 0: the heuristic's answer; a forced tile 13: whether it applies at all.
bool conv_picks_patch(int64_t plan, int64_t n, int64_t ho, int64_t wo, at::IntArrayRef src_pix_strides, int64_t cout, int64_t out_step, int64_t n_classes, int64_t tile) {
  TORCH_CHECK(src_pix_strides.size() >= 1 && src_pix_strides.size() <= 2, "okp: one or two sources");
  okp_conv_args a;
  std::memset(&a, 0, sizeof(a));
  a.n = (int32_t)n; a.ho = (int32_t)ho; a.wo = (int32_t)wo;
  for (size_t i = 0; i < src_pix_strides.size(); ++i) { a.src[i].h = a.src[i].w = 1; a.src[i].pix_stride = (int32_t)src_pix_strides[i]; a.src[i].bytes = 1; }
  a.out.h = (int32_t)(ho * out_step); a.out.w = (int32_t)(wo * out_step); a.out.pix_stride = (int32_t)cout; a.out.bytes = 1;
  a.out_step = (int32_t)out_step; a.n_classes = (int32_t)n_classes;
  const okp_conv* pl = reinterpret_cast<const okp_conv*>(plan);
  if (tile) return tile == 13 && okp_conv_patch_applies(pl, &a) != 0;
  static int dummy_target;
  a.src[0].data = &dummy_target; a.out.data = &dummy_target;       // (never dereferenced: the heuristic reads strides and the grid)
  return okp_conv_select_tile(pl, &a) == 13;
}

void conv_destroy(int64_t plan) { okp_conv_destroy(reinterpret_cast<okp_conv*>(plan)); }

int64_t stem_create(int64_t dtype, const Tensor& w, const Tensor& bias) {
  TORCH_CHECK(w.dim() == 4 && w.size(0) == 128 && w.size(1) == 3 && w.size(2) == 7 && w.size(3) == 7 && bias.numel() == 128, "okp: the stem kernel is 7x7, 3 -> 128 channels");
  std::vector<Tensor> keep;
  okp_stem* st = okp_stem_create_dtype((int)dtype, host_f32(w, "the stem weight", keep), host_f32(bias, "the stem bias", keep));
  TORCH_CHECK(st != nullptr, "okp_stem_create failed: ", okp_last_error());
  return reinterpret_cast<int64_t>(st);
}

void stem_destroy(int64_t stem) { okp_stem_destroy(reinterpret_cast<okp_stem*>(stem)); }

int64_t conv_macs(int64_t plan, int64_t n, int64_t ho, int64_t wo) {
  okp_conv_args a;
  std::memset(&a, 0, sizeof(a));
  a.n = (int32_t)n; a.ho = (int32_t)ho; a.wo = (int32_t)wo;
  return okp_conv_macs(reinterpret_cast<const okp_conv*>(plan), &a);
}

// ---- the remaining launch entry points ---------------------------------------------------------------------------------------------

inline void* sp(int64_t stream) { return reinterpret_cast<void*>(stream); }

inline void check_dev(const Tensor& t, at::ScalarType ty, const char* name) {
  TORCH_CHECK(t.is_cuda(), "okp: ", name, " must be a device tensor: the HIP path has no CPU fallback");
  TORCH_CHECK(t.scalar_type() == ty && t.is_contiguous(), "okp: ", name, " must be a contiguous ", ty, " tensor");
}

void stem_forward_nchw_pairs(int64_t stem, const Tensor& frames, const Tensor& out, int64_t out_c0, int64_t stream) {
  TORCH_CHECK(frames.is_cuda() && frames.scalar_type() == at::kFloat && frames.dim() == 4 && frames.size(1) == 3 && frames.is_contiguous(),
              "okp: frames must be a contiguous float32 [N,3,H,W] device tensor");
  const okp_tensor ov = view(out, out_c0, "out");
  check_rc(okp_stem_forward_nchw_pairs(reinterpret_cast<const okp_stem*>(stem), (int32_t)frames.size(0), (int32_t)frames.size(2), (int32_t)frames.size(3),
                                       frames.data_ptr<float>(), &ov, sp(stream)), "okp_stem_forward_nchw_pairs");
}

// packed: the [N, h + 6, wp, 4] output of pack_frames / pack_frames_u8 / preprocess_u8 for h x w frames
void stem_forward(int64_t stem, const Tensor& packed, int64_t h, int64_t w, const Tensor& out, int64_t out_c0, int64_t stream) {
  const okp_tensor pv = view(packed, 0, "packed"), ov = view(out, out_c0, "out");
  TORCH_CHECK(packed.size(3) == 4 && packed.size(1) == h + 6, "okp: packed frames are [N, h + 6, wp, 4]");
  check_rc(okp_stem_forward(reinterpret_cast<const okp_stem*>(stem), (int32_t)packed.size(0), (int32_t)h, (int32_t)w, &pv, &ov, sp(stream)), "okp_stem_forward");
}

void pack_frames(const Tensor& frames, const Tensor& out, int64_t stream) {
  check_dev(frames, at::kFloat, "frames");
  TORCH_CHECK(frames.dim() == 4 && frames.size(1) == 3 && out.is_cuda() && out.dim() == 4 && out.is_contiguous() && out.size(0) == frames.size(0) &&
              out.size(1) == frames.size(2) + 6 && out.size(3) == 4, "okp: frames [N,3,H,W] float32 -> out [N,H+6,wp,4]");
  check_rc(okp_pack_frames(dtype_of(out), frames.data_ptr<float>(), (int32_t)frames.size(0), (int32_t)frames.size(2), (int32_t)frames.size(3), out.data_ptr(),
                           (int32_t)out.size(2), sp(stream)), "okp_pack_frames");
}

inline void three(at::ArrayRef<double> v, float* o, const char* name) {
  TORCH_CHECK(v.size() == 3, "okp: ", name, " has three entries (R, G, B)");
  for (int i = 0; i < 3; ++i) o[i] = (float)v[i];
}

void pack_frames_u8(const Tensor& frames, at::ArrayRef<double> mean, at::ArrayRef<double> std_, const Tensor& out, int64_t stream) {
  check_dev(frames, at::kByte, "frames");
  TORCH_CHECK(frames.dim() == 4 && frames.size(3) == 3 && out.is_cuda() && out.dim() == 4 && out.is_contiguous() && out.size(0) == frames.size(0) &&
              out.size(1) == frames.size(1) + 6 && out.size(3) == 4, "okp: frames [N,H,W,3] uint8 -> out [N,H+6,wp,4]");
  float m[3], sd[3]; three(mean, m, "mean"); three(std_, sd, "std");
  check_rc(okp_pack_frames_u8(dtype_of(out), frames.data_ptr<uint8_t>(), (int32_t)frames.size(0), (int32_t)frames.size(1), (int32_t)frames.size(2), m, sd,
                              out.data_ptr(), (int32_t)out.size(2), sp(stream)), "okp_pack_frames_u8");
}

void preprocess_u8(const Tensor& frames, int64_t resized_h, int64_t resized_w, int64_t crop_y, int64_t crop_x, int64_t h, int64_t w, at::ArrayRef<double> mean,
                   at::ArrayRef<double> std_, const Tensor& out, int64_t stream) {
  check_dev(frames, at::kByte, "frames");
  TORCH_CHECK(frames.dim() == 4 && frames.size(3) == 3 && out.is_cuda() && out.dim() == 4 && out.is_contiguous() && out.size(0) == frames.size(0) &&
              out.size(1) == h + 6 && out.size(3) == 4, "okp: frames [N,H,W,3] uint8 -> out [N,h+6,wp,4]");
  float m[3], sd[3]; three(mean, m, "mean"); three(std_, sd, "std");
  check_rc(okp_preprocess_u8(dtype_of(out), frames.data_ptr<uint8_t>(), (int32_t)frames.size(0), (int32_t)frames.size(1), (int32_t)frames.size(2), (int32_t)resized_h,
                             (int32_t)resized_w, (int32_t)crop_y, (int32_t)crop_x, (int32_t)h, (int32_t)w, m, sd, out.data_ptr(), (int32_t)out.size(2), sp(stream)),
           "okp_preprocess_u8");
}

inline int32_t* flag_ptr(const optional<Tensor>& f) {
  if (!f.has_value()) return nullptr;
  TORCH_CHECK(f->is_cuda() && f->scalar_type() == at::kInt && f->numel() >= 1, "okp: range_flag is an int32 device tensor");
  return f->data_ptr<int32_t>();
}

void cast(const Tensor& src, const Tensor& dst, int64_t stream, const optional<Tensor>& range_flag) {
  TORCH_CHECK(src.is_cuda() && dst.is_cuda() && src.is_contiguous() && dst.is_contiguous() && src.numel() == dst.numel(), "okp: cast takes contiguous device tensors of one size");
  check_rc(okp_cast(dtype_of(src), src.data_ptr(), dtype_of(dst), dst.data_ptr(), src.numel(), flag_ptr(range_flag), sp(stream)), "okp_cast");
}

void add_f16_f32(const Tensor& a, const Tensor& b, const Tensor& out, int64_t act, int64_t stream, const optional<Tensor>& range_flag) {
  check_dev(a, at::kHalf, "a"); check_dev(b, at::kFloat, "b"); check_dev(out, at::kFloat, "out");
  TORCH_CHECK(a.numel() == b.numel() && b.numel() == out.numel(), "okp: add_f16_f32 takes tensors of one size");
  check_rc(okp_add_f16_f32(a.data_ptr(), b.data_ptr<float>(), out.data_ptr<float>(), out.numel(), (int)act, flag_ptr(range_flag), sp(stream)), "okp_add_f16_f32");
}

void dwconv3x3_forward(const Tensor& src, int64_t src_c0, int64_t c, int64_t conv_stride, const Tensor& w, const Tensor& bias, const optional<Tensor>& res, int64_t res_c0,
                       const Tensor& out, int64_t out_c0, int64_t act, int64_t stream) {
  check_dev(w, at::kFloat, "depth-wise weights"); check_dev(bias, at::kFloat, "depth-wise bias");
  TORCH_CHECK(out.scalar_type() == src.scalar_type() && (!res.has_value() || res->scalar_type() == src.scalar_type()), "okp: dwconv3x3: one element type");
  const okp_tensor sv = view(src, src_c0, "src"), ov = view(out, out_c0, "out");
  okp_tensor rv = kNull;
  if (res.has_value()) rv = view(*res, res_c0, "res");
  check_rc(okp_dwconv3x3_forward(dtype_of(src), (int32_t)src.size(0), (int32_t)c, (int32_t)conv_stride, &sv, w.data_ptr<float>(), bias.data_ptr<float>(),
                                 res.has_value() ? &rv : nullptr, &ov, (int)act, sp(stream)), "okp_dwconv3x3_forward");
}

void nms_maxpool(const Tensor& x, int64_t size, const Tensor& out, int64_t stream) {
  check_dev(x, at::kFloat, "x"); check_dev(out, at::kFloat, "out");
  TORCH_CHECK(x.dim() == 4 && out.sizes() == x.sizes(), "okp: nms takes float32 [N,C,H,W] maps");
  check_rc(okp_nms_maxpool(x.data_ptr<float>(), (int32_t)(x.size(0) * x.size(1)), (int32_t)x.size(2), (int32_t)x.size(3), (int32_t)size, out.data_ptr<float>(), sp(stream)),
           "okp_nms_maxpool");
}

void capacity_overflow(const Tensor& count, int64_t K, int64_t cap, int64_t max_obj, const Tensor& flag, int64_t stream, const optional<Tensor>& range_flag) {
  check_dev(count, at::kInt, "count"); check_dev(flag, at::kInt, "flag");
  if (range_flag.has_value()) check_dev(*range_flag, at::kInt, "range_flag");
  check_rc(okp_capacity_overflow(count.data_ptr<int32_t>(), (int32_t)count.numel(), (int32_t)K, (int32_t)cap, (int32_t)max_obj,
                                 range_flag.has_value() ? range_flag->data_ptr<int32_t>() : nullptr, flag.data_ptr<int32_t>(), sp(stream)), "okp_capacity_overflow");
}

// fp16-range guard of split-product plans: the plan's launches OR 1 into flag[0] when a result leaves the fp16 range (include/okp.h).  The
// caller keeps `flag` alive as long as the plan is used.
void conv_set_range_flag(int64_t plan, const optional<Tensor>& flag) {
  if (flag.has_value()) check_dev(*flag, at::kInt, "flag");
  check_rc(okp_conv_set_range_flag(reinterpret_cast<okp_conv*>(plan), flag.has_value() ? flag->data_ptr<int32_t>() : nullptr), "okp_conv_set_range_flag");
}
void stem_set_range_flag(int64_t stem, const optional<Tensor>& flag) {
  if (flag.has_value()) check_dev(*flag, at::kInt, "flag");
  check_rc(okp_stem_set_range_flag(reinterpret_cast<okp_stem*>(stem), flag.has_value() ? flag->data_ptr<int32_t>() : nullptr), "okp_stem_set_range_flag");
}

void stream_wait_stream(int64_t waiter, int64_t signaller) { check_rc(okp_stream_wait_stream(sp(waiter), sp(signaller)), "okp_stream_wait_stream"); }

inline okp_camera camera_of(at::ArrayRef<double> c) {
  TORCH_CHECK(c.size() == 9, "okp: camera = [fx, fy, cx, cy, d0, d1, d2, d3, model]");
  okp_camera cam{c[0], c[1], c[2], c[3], {c[4], c[5], c[6], c[7]}, (int32_t)c[8], 0};
  return cam;
}

void camera_undistort(at::ArrayRef<double> camera, const Tensor& xy, const Tensor& out, int64_t stream) {
  check_dev(xy, at::kFloat, "xy"); check_dev(out, at::kDouble, "out");
  TORCH_CHECK(xy.dim() == 2 && xy.size(1) == 2 && out.numel() == xy.numel(), "okp: xy [M,2] float32 -> out [M,2] float64");
  const okp_camera cam = camera_of(camera);
  check_rc(okp_camera_undistort(&cam, xy.data_ptr<float>(), (int32_t)xy.size(0), out.data_ptr<double>(), sp(stream)), "okp_camera_undistort");
}

void unproject_depth(at::ArrayRef<double> camera, const Tensor& xy, const Tensor& map_id, const Tensor& depth, int64_t max_x, int64_t max_y, const Tensor& out, int64_t stream) {
  check_dev(xy, at::kFloat, "xy"); check_dev(map_id, at::kInt, "map_id"); check_dev(depth, at::kFloat, "depth"); check_dev(out, at::kDouble, "out");
  TORCH_CHECK(xy.dim() == 2 && xy.size(1) == 2 && map_id.numel() == xy.size(0) && depth.dim() == 3 && out.numel() == xy.size(0) * 3, "okp: unproject_depth buffers do not match");
  const okp_camera cam = camera_of(camera);
  check_rc(okp_unproject_depth(&cam, xy.data_ptr<float>(), map_id.data_ptr<int32_t>(), (int32_t)xy.size(0), depth.data_ptr<float>(), (int32_t)depth.size(1), (int32_t)depth.size(2),
                               (int32_t)max_x, (int32_t)max_y, out.data_ptr<double>(), sp(stream)), "okp_unproject_depth");
}

// T_RL: 12 doubles (row-major 3 x 4); F: 9 doubles (row-major 3 x 3) = Hartley-Sturm correction on, empty = off
void triangulate_dlt(at::ArrayRef<double> left, at::ArrayRef<double> right, at::ArrayRef<double> T_RL, at::ArrayRef<double> F, const Tensor& left_xy, const Tensor& right_xy,
                     const Tensor& out, int64_t stream) {
  check_dev(left_xy, at::kFloat, "left_xy"); check_dev(right_xy, at::kFloat, "right_xy"); check_dev(out, at::kDouble, "out");
  TORCH_CHECK(left_xy.dim() == 2 && left_xy.size(1) == 2 && right_xy.sizes() == left_xy.sizes() && out.numel() == left_xy.size(0) * 3, "okp: left / right [M,2] float32 -> out [M,3] float64");
  TORCH_CHECK(T_RL.size() == 12 && (F.empty() || F.size() == 9), "okp: T_RL is row-major 3 x 4, F row-major 3 x 3 (or empty)");
  const okp_camera cl = camera_of(left), cr = camera_of(right);
  check_rc(okp_triangulate_dlt(&cl, &cr, T_RL.data(), F.empty() ? nullptr : F.data(), F.empty() ? 0 : 1, left_xy.data_ptr<float>(), right_xy.data_ptr<float>(),
                               (int32_t)left_xy.size(0), out.data_ptr<double>(), sp(stream)), "okp_triangulate_dlt");
}

void lift_peaks(at::ArrayRef<double> camera, const Tensor& count, const Tensor& xyc, const Tensor& depth, int64_t max_x, int64_t max_y, const Tensor& out, int64_t stream) {
  TORCH_CHECK(xyc.is_cuda() && xyc.dim() == 4 && xyc.scalar_type() == at::kFloat && depth.is_cuda() && depth.scalar_type() == at::kFloat && depth.dim() == 4 && depth.is_contiguous() &&
              depth.size(0) == xyc.size(0) && depth.size(1) == xyc.size(1) && count.scalar_type() == at::kInt && out.scalar_type() == at::kDouble &&
              out.numel() == xyc.size(0) * xyc.size(1) * xyc.size(2) * 4, "okp: lift_peaks buffers do not match");
  const okp_camera cam = camera_of(camera);
  check_rc(okp_lift_peaks(&cam, count.data_ptr<int32_t>(), xyc.data_ptr<float>(), (int32_t)(xyc.size(0) * xyc.size(1)), (int32_t)xyc.size(2), depth.data_ptr<float>(),
                          (int32_t)depth.size(2), (int32_t)depth.size(3), (int32_t)max_x, (int32_t)max_y, out.data_ptr<double>(), reinterpret_cast<void*>(stream)), "okp_lift_peaks");
}

void group_objects(const Tensor& count, const Tensor& xyc, const Tensor& centers, at::IntArrayRef type_count, double max_dist, int64_t max_obj, int64_t max_sel,
                   const Tensor& n_obj, const Tensor& sel, const Tensor& votes, const Tensor& assign, const Tensor& pred, int64_t stream, const optional<Tensor>& reduced) {
  TORCH_CHECK(xyc.is_cuda() && xyc.dim() == 4 && centers.is_cuda() && centers.dim() == 5 && centers.is_contiguous() && centers.scalar_type() == at::kFloat, "okp: group_objects inputs");
  const int64_t n = xyc.size(0), K = xyc.size(1), cap = xyc.size(2);
  TORCH_CHECK((int64_t)type_count.size() == K - 1 && K <= 8, "okp: type_count has K-1 entries, K <= 8");
  int32_t tc[8];
  for (int64_t i = 0; i < K - 1; ++i) tc[i] = (int32_t)type_count[i];
  float* red = nullptr;
  if (reduced.has_value()) {
    TORCH_CHECK(reduced->is_cuda() && reduced->scalar_type() == at::kFloat && reduced->is_contiguous() && reduced->numel() == n * max_obj * (K - 1) * max_sel * 2,
                "okp: reduced is a contiguous float32 [N, max_obj, K-1, max_sel, 2] device tensor");
    red = reduced->data_ptr<float>();
  }
  check_rc(okp_group_objects(count.data_ptr<int32_t>(), xyc.data_ptr<float>(), centers.data_ptr<float>(), (int32_t)n, (int32_t)K, (int32_t)cap,
                             (int32_t)centers.size(3), (int32_t)centers.size(4), tc, (float)max_dist, (int32_t)max_obj, (int32_t)max_sel,
                             n_obj.data_ptr<int32_t>(), sel.data_ptr<int32_t>(), votes.data_ptr<int32_t>(), assign.data_ptr<int32_t>(), pred.data_ptr<double>(),
                             red, reinterpret_cast<void*>(stream)), "okp_group_objects");
}

}  // namespace

TORCH_LIBRARY(okp, m) {
  m.def("conv_forward(int plan, Tensor src0, int src0_c0, Tensor? src1, int src1_c0, Tensor(a!) out, int out_c0, int ho, int wo, Tensor? res, int res_c0, "
        "int out_step, int oy, int ox, int tile, int n_classes, Tensor? dw_w, Tensor? dw_b, Tensor(b!)? dw_out, int dw_out_c0, Tensor? dw_res, int dw_res_c0, int stream, "
        "Tensor(c!)? out16=None, int out16_c0=0, bool write_out=True, int out_subsample=1, int src_pairs=0, bool out_pairs=False) -> ()", conv_forward);
  m.def("conv_select_tile(int plan, Tensor src0, int src0_c0, Tensor? src1, int src1_c0, Tensor out, int out_c0, int ho, int wo, int out_step, int oy, int ox, "
        "int tile, int n_classes, bool has_dw, bool extended=False) -> int", conv_select_tile);
  m.def("fire_forward(int squeeze, int expand, Tensor wd, Tensor bd, Tensor x, int x_c0, Tensor(a!) out, int out_c0, int stride, bool skip, int stream) -> ()", fire_forward);
  m.def("fire_chain_forward(int[] squeeze, int[] expand, Tensor[] wd, Tensor[] bd, Tensor x, int x_c0, Tensor(a!) out, int out_c0, int stream) -> ()", fire_chain_forward);
  m.def("heads_forward(int l1, int l2, Tensor x, int x_c0, int[] in_c_off, int[] act, Tensor(a!)[] outs, int[] out_ch, Tensor w3, Tensor b3, int stream) -> ()", heads_forward);
  m.def("head_out_forward(Tensor src, int src_c0, int[] in_c_off, int[] act, Tensor(a!)[] outs, int[] out_ch, Tensor w3, Tensor b3, int stream) -> ()", head_out_forward);
  m.def("stem_forward_nchw(int stem, Tensor frames, Tensor(a!) out, int out_c0, int stream) -> ()", stem_forward_nchw);
  m.def("peak_nms(Tensor heat, int cap, Tensor(a!) count, Tensor(b!) yx, Tensor(c!) xyc, int stream) -> ()", peak_nms);
  m.def("lift_peaks(float[] camera, Tensor count, Tensor xyc, Tensor depth, int max_x, int max_y, Tensor(a!) out, int stream) -> ()", lift_peaks);
  m.def("group_objects(Tensor count, Tensor xyc, Tensor centers, int[] type_count, float max_dist, int max_obj, int max_sel, Tensor(a!) n_obj, Tensor(b!) sel, Tensor(c!) votes, "
        "Tensor(d!) assign, Tensor(e!) pred, int stream, Tensor(f!)? reduced=None) -> ()", group_objects);
  // plan creation (host tensors in, opaque handle out)
  m.def("conv_create(int dtype, int[] cin, int[] conv_stride, int cout, int[] tap_src, int[] tap_dy, int[] tap_dx, Tensor[] tap_w, Tensor? bias, int act, int[] tap_terms) -> int", conv_create);
  m.def("fold_bn(Tensor weight, Tensor? bn_weight, Tensor? bn_bias, Tensor? bn_mean, Tensor? bn_var, float eps, Tensor? conv_bias) -> (Tensor, Tensor)", fold_bn);
  m.def("conv_bn_create(int dtype, Tensor weight, Tensor? bn_weight, Tensor? bn_bias, Tensor? bn_mean, Tensor? bn_var, float eps, Tensor? conv_bias, int stride, bool relu) -> int", conv_bn_create);
  m.def("conv_destroy(int plan) -> ()", conv_destroy);
  m.def("conv_picks_patch(int plan, int n, int ho, int wo, int[] src_pix_strides, int cout, int out_step, int n_classes, int tile) -> bool", conv_picks_patch);
  m.def("conv_macs(int plan, int n, int ho, int wo) -> int", conv_macs);
  m.def("stem_create(int dtype, Tensor w, Tensor bias) -> int", stem_create);
  m.def("stem_destroy(int stem) -> ()", stem_destroy);
  // the rest of include/okp.h
  m.def("stem_forward_nchw_pairs(int stem, Tensor frames, Tensor(a!) out, int out_c0, int stream) -> ()", stem_forward_nchw_pairs);
  m.def("stem_forward(int stem, Tensor packed, int h, int w, Tensor(a!) out, int out_c0, int stream) -> ()", stem_forward);
  m.def("pack_frames(Tensor frames, Tensor(a!) out, int stream) -> ()", pack_frames);
  m.def("pack_frames_u8(Tensor frames, float[] mean, float[] std, Tensor(a!) out, int stream) -> ()", pack_frames_u8);
  m.def("preprocess_u8(Tensor frames, int resized_h, int resized_w, int crop_y, int crop_x, int h, int w, float[] mean, float[] std, Tensor(a!) out, int stream) -> ()", preprocess_u8);
  m.def("cast(Tensor src, Tensor(a!) dst, int stream, Tensor(b!)? range_flag=None) -> ()", cast);
  m.def("add_f16_f32(Tensor a, Tensor b, Tensor(a!) out, int act, int stream, Tensor(b!)? range_flag=None) -> ()", add_f16_f32);
  m.def("dwconv3x3_forward(Tensor src, int src_c0, int c, int conv_stride, Tensor w, Tensor bias, Tensor? res, int res_c0, Tensor(a!) out, int out_c0, int act, int stream) -> ()", dwconv3x3_forward);
  m.def("nms_maxpool(Tensor x, int size, Tensor(a!) out, int stream) -> ()", nms_maxpool);
  m.def("capacity_overflow(Tensor count, int K, int cap, int max_obj, Tensor(a!) flag, int stream, Tensor? range_flag=None) -> ()", capacity_overflow);
  m.def("conv_set_range_flag(int plan, Tensor? flag) -> ()", conv_set_range_flag);
  m.def("stem_set_range_flag(int stem, Tensor? flag) -> ()", stem_set_range_flag);
  m.def("stream_wait_stream(int waiter, int signaller) -> ()", stream_wait_stream);
  m.def("camera_undistort(float[] camera, Tensor xy, Tensor(a!) out, int stream) -> ()", camera_undistort);
  m.def("unproject_depth(float[] camera, Tensor xy, Tensor map_id, Tensor depth, int max_x, int max_y, Tensor(a!) out, int stream) -> ()", unproject_depth);
  m.def("triangulate_dlt(float[] left, float[] right, float[] T_RL, float[] F, Tensor left_xy, Tensor right_xy, Tensor(a!) out, int stream) -> ()", triangulate_dlt);
}

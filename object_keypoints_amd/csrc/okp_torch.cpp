// torch.ops.okp.* : the launch entry points of the C ABI (include/okp.h) registered with the PyTorch dispatcher.
//
// north_star asks for the HIP kernels to be "surfaced through PyTorch-ROCm custom ops" ("What the native
// replacement must export").  This translation unit is a thin shim over the SAME extern "C" symbols of libokp_hip.so that the
// ctypes binding uses: it takes torch tensors (NHWC activations as (tensor, first channel) pairs: a channel window of a wider
// tensor is how concat / split are free), checks device / dtype / layout with TORCH_CHECK as SURVEY §8(b) "Error conventions"
// prescribes, fills the C structs and calls the C entry point on the stream given by the caller.  No kernel lives here and
// no HIP header is needed: plans stay opaque handles (int64) created through the C ABI.
//
// Cost per launch from Python: one dispatcher call (~4 us) instead of ~20 us of ctypes struct filling.
#include <ATen/ATen.h>
#include <cstring>
#include <torch/library.h>

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
                  const optional<Tensor>& dw_res, int64_t dw_res_c0, int64_t stream) {
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
                         int64_t tile, int64_t n_classes, bool has_dw) {
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

inline okp_camera camera_of(at::ArrayRef<double> c) {
  TORCH_CHECK(c.size() == 9, "okp: camera = [fx, fy, cx, cy, d0, d1, d2, d3, model]");
  okp_camera cam{c[0], c[1], c[2], c[3], {c[4], c[5], c[6], c[7]}, (int32_t)c[8], 0};
  return cam;
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
                   const Tensor& n_obj, const Tensor& sel, const Tensor& votes, const Tensor& assign, const Tensor& pred, int64_t stream) {
  TORCH_CHECK(xyc.is_cuda() && xyc.dim() == 4 && centers.is_cuda() && centers.dim() == 5 && centers.is_contiguous() && centers.scalar_type() == at::kFloat, "okp: group_objects inputs");
  const int64_t n = xyc.size(0), K = xyc.size(1), cap = xyc.size(2);
  TORCH_CHECK((int64_t)type_count.size() == K - 1 && K <= 8, "okp: type_count has K-1 entries, K <= 8");
  int32_t tc[8];
  for (int64_t i = 0; i < K - 1; ++i) tc[i] = (int32_t)type_count[i];
  check_rc(okp_group_objects(count.data_ptr<int32_t>(), xyc.data_ptr<float>(), centers.data_ptr<float>(), (int32_t)n, (int32_t)K, (int32_t)cap,
                             (int32_t)centers.size(3), (int32_t)centers.size(4), tc, (float)max_dist, (int32_t)max_obj, (int32_t)max_sel,
                             n_obj.data_ptr<int32_t>(), sel.data_ptr<int32_t>(), votes.data_ptr<int32_t>(), assign.data_ptr<int32_t>(), pred.data_ptr<double>(),
                             reinterpret_cast<void*>(stream)), "okp_group_objects");
}

}  // namespace

TORCH_LIBRARY(okp, m) {
  m.def("conv_forward(int plan, Tensor src0, int src0_c0, Tensor? src1, int src1_c0, Tensor(a!) out, int out_c0, int ho, int wo, Tensor? res, int res_c0, "
        "int out_step, int oy, int ox, int tile, int n_classes, Tensor? dw_w, Tensor? dw_b, Tensor(b!)? dw_out, int dw_out_c0, Tensor? dw_res, int dw_res_c0, int stream) -> ()", conv_forward);
  m.def("conv_select_tile(int plan, Tensor src0, int src0_c0, Tensor? src1, int src1_c0, Tensor out, int out_c0, int ho, int wo, int out_step, int oy, int ox, "
        "int tile, int n_classes, bool has_dw) -> int", conv_select_tile);
  m.def("fire_forward(int squeeze, int expand, Tensor wd, Tensor bd, Tensor x, int x_c0, Tensor(a!) out, int out_c0, int stride, bool skip, int stream) -> ()", fire_forward);
  m.def("fire_chain_forward(int[] squeeze, int[] expand, Tensor[] wd, Tensor[] bd, Tensor x, int x_c0, Tensor(a!) out, int out_c0, int stream) -> ()", fire_chain_forward);
  m.def("heads_forward(int l1, int l2, Tensor x, int x_c0, int[] in_c_off, int[] act, Tensor(a!)[] outs, int[] out_ch, Tensor w3, Tensor b3, int stream) -> ()", heads_forward);
  m.def("head_out_forward(Tensor src, int src_c0, int[] in_c_off, int[] act, Tensor(a!)[] outs, int[] out_ch, Tensor w3, Tensor b3, int stream) -> ()", head_out_forward);
  m.def("stem_forward_nchw(int stem, Tensor frames, Tensor(a!) out, int out_c0, int stream) -> ()", stem_forward_nchw);
  m.def("peak_nms(Tensor heat, int cap, Tensor(a!) count, Tensor(b!) yx, Tensor(c!) xyc, int stream) -> ()", peak_nms);
  m.def("lift_peaks(float[] camera, Tensor count, Tensor xyc, Tensor depth, int max_x, int max_y, Tensor(a!) out, int stream) -> ()", lift_peaks);
  m.def("group_objects(Tensor count, Tensor xyc, Tensor centers, int[] type_count, float max_dist, int max_obj, int max_sel, Tensor(a!) n_obj, Tensor(b!) sel, Tensor(c!) votes, "
        "Tensor(d!) assign, Tensor(e!) pred, int stream) -> ()", group_objects);
}

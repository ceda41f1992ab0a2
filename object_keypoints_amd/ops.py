"""Thin, torch-aware wrappers over the C ABI (include/okp.h).

torch is used for device memory and streams only: every computation below is a call into
libokp_hip.so on `torch.cuda.current_stream()`.  Nothing here has a CPU path.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import OKP_BF16, OKP_F16, OKP_F32, OKP_F32X3, ACT_NONE, ACT_RELU, ACT_SIGMOID, OkpError

_DTYPES = {torch.float32: OKP_F32, torch.bfloat16: OKP_BF16, torch.float16: OKP_F16}
HALF_DTYPES = (torch.bfloat16, torch.float16)     # 16-bit activations / weights, fp32 accumulate: the MFMA throughput precisions


# "float32x3": fp32 tensors everywhere, but the convolution plans multiply on the fp16 matrix pipe as a three-term split of the
# fp32 operands (OKP_F32X3, include/okp.h): fp32-grade results at a multiple of the exact-fp32 kernels' speed.  The mode is a property
# of a PLAN (chosen when the plan is built); activations are plain torch.float32 tensors in both fp32 modes.
F32X3 = "float32x3"
# "float32mix": the sensitivity-guided mixed configuration on top of float32x3.  The skip stream (every tensor a residual / fire
# module / merge hands on) stays fp32 and is multiplied with three terms; what reaches the output attenuated is computed cheaper:
#   * the 3x3 convolutions INSIDE the residual blocks of the trunk (pre.1, pre.2, inters.0: conv1, conv2 - 70 % of the network's MACs,
#     ~3 % of the fp16 error variance) multiply with the single term x_hi * w_hi (okp_conv_create_x3, tap_terms = 1);
#   * the innermost MIX_FP16_LEVELS levels of both hourglasses (16 x 16 pixels and below: 1 % of the variance) run entirely in fp16 on
#     the fused fp16 kernels, between two okp_cast launches.
# tests/precision/attribute.py measures those shares on given weights (tests/golden/precision_attribution.json for the test network) and
# tests/precision/emulate.py: mixed_policy() predicts the result on the CPU: heat error 3.9e-4 max on the test networks (bar 1e-3).
F32MIX = "float32mix"
# ops.F32_SPLIT: plans built for torch.float32 while it is set are OKP_F32X3 plans; ops.F32_MIX: ... with single-term residual branches and
# fp16 inner hourglass levels.  Both are PER-THREAD state set by the f32_split() context manager (module __getattr__ below): two networks of
# different configurations may run on two host threads.
import threading
_MODE = threading.local()


def __getattr__(name):
    if name == "F32_SPLIT":
        return getattr(_MODE, "split", False)
    if name == "F32_MIX":
        return getattr(_MODE, "mix", False)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")

MIX_FP16_LEVELS = int(os.environ.get("OKP_MIX_FP16_LEVELS", "2"))    # hg_module levels n <= this run in fp16 (n = 2: the 16 x 16 level)
MIX_BRANCH_SINGLE = True        # (module attributes; the measured trade-off of each is in DESIGN.md 2.2)
# ... and those single-term branches run on the fp16 kernels proper (the patch-resident 3x3 kernel: half the bytes, no in-loop conversion):
# the producer of a stream tensor writes an fp16 copy next to it (okp_conv_args.out16), conv1 / conv2 are fp16 plans, and the block
# closes with the three-term projected skip taking the fp16 branch as its residual (or okp_add_f16_f32).  0 = single-term taps of the
# split-product kernel on the fp32 tensors (same products, slower).
MIX_BRANCH_FP16 = True
# ... and a stream tensor whose only fp32 reader is the next block's stride-2 skip (stem output, pre[1] output) is kept in fp32 at even
# rows / columns only (okp_conv_args.out_subsample): three quarters of its bytes were written and never read
MIX_COMPACT = True
# ... and the stem then runs as two launches: the fp16 stem kernel writes the full-grid fp16 tensor (read by pre[1].conv1 only: a branch
# input), a stride-4 three-term launch the fp32 values at even pixels (read by pre[1]'s skip: the stream) - a quarter of the three-term work
MIX_STEM_FP16 = True


class f32_split:
    """Context manager: convolution plans built inside are split-product plans (KeypointNet(compute_dtype=ops.F32X3 / ops.F32MIX) wraps
    its passes in it).  range_flag (int32 device tensor of one element, optional): every split-product plan and stem BUILT inside gets it as
    its fp16-range flag (okp_conv_set_range_flag: a launch ORs 1 into it when one of its results would not survive the next layer's split
    into fp16 halves - |x| > 65504 or not finite)."""

    def __init__(self, enabled=True, mixed=False, range_flag=None):
        self.enabled, self.mixed = bool(enabled), bool(enabled and mixed)
        self.range_flag = range_flag if self.enabled else None

    def __enter__(self):
        self.prev = (getattr(_MODE, "split", False), getattr(_MODE, "mix", False), getattr(_MODE, "range_flag", None))
        _MODE.split, _MODE.mix, _MODE.range_flag = self.enabled, self.mixed, self.range_flag

    def __exit__(self, *exc):
        _MODE.split, _MODE.mix, _MODE.range_flag = self.prev


def _attach_range_flag(plan, setter_name):
    """Give a freshly built split-product plan the range flag of the enclosing f32_split context, if there is one."""
    flag = getattr(_MODE, "range_flag", None)
    plan._range_flag = None
    if flag is None or not plan.split:
        return
    require_cuda(flag, "range_flag")
    if flag.dtype != torch.int32 or flag.numel() != 1:
        raise OkpError("range_flag is an int32 device tensor of one element")
    T = _ops()
    if T is not None:
        _dispatch(getattr(T, setter_name), plan._h, flag)
    else:
        _lib.check(getattr(_lib.lib(), "okp_" + setter_name)(plan._h, flag.data_ptr()), "okp_" + setter_name)
    plan._range_flag = flag              # (the plan's launches write through this pointer: keep the tensor alive with the plan)


def parse_compute_dtype(compute_dtype):
    """-> (torch dtype of the stream tensors, split-product flag, mixed flag) for torch.float32 / bfloat16 / float16, ops.F32X3 or ops.F32MIX."""
    if isinstance(compute_dtype, str):
        if compute_dtype.lower() in (F32X3, "f32x3"):
            return torch.float32, True, False
        if compute_dtype.lower() in (F32MIX, "f32mix"):
            return torch.float32, True, True
        compute_dtype = getattr(torch, compute_dtype, compute_dtype)
    if compute_dtype not in _DTYPES:
        raise OkpError(f"unsupported compute dtype {compute_dtype}; use torch.float32, torch.bfloat16, torch.float16, '{F32X3}' or '{F32MIX}'")
    return compute_dtype, False, False


def okp_dtype(torch_dtype):
    try:
        return _DTYPES[torch_dtype]
    except KeyError:
        raise OkpError(f"unsupported activation dtype {torch_dtype}; use torch.float32, torch.bfloat16 or torch.float16")


# torch.cuda.current_stream() builds a Stream object through four Python layers (8 us under a profiler, ~75 calls per network pass);
# the raw handle of the current stream of the current device is one C call
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_int():
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch._C._cuda_getDevice())
    return torch.cuda.current_stream().cuda_stream


def stream_handle():
    return ctypes.c_void_p(stream_int())


def _dispatch(fn, *args):
    """Call a torch.ops.okp entry point; its TORCH_CHECK failures surface as OkpError like the ctypes path's error codes."""
    try:
        return fn(*args)
    except RuntimeError as e:
        raise OkpError(str(e).split("\n")[0]) from None


def _ops():
    """torch.ops.okp, or None with the ctypes call that follows counted (COUNTERS["ctypes_launches"])."""
    T = _lib.torch_ops()
    if T is None:
        COUNTERS["ctypes_launches"] += 1
    return T


def _cam_list(cam):
    return [cam.fx, cam.fy, cam.cx, cam.cy, cam.d[0], cam.d[1], cam.d[2], cam.d[3], float(cam.model)]


def require_cuda(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise OkpError(f"{what} must be a device tensor: the HIP path has no CPU fallback")


class Act:
    """NHWC activation view: a contiguous device tensor [N,H,W,Ctot] and a channel window."""

    __slots__ = ("t", "c0", "c", "orig_hw", "shadow", "compact", "pairs")

    def __init__(self, t, c0=0, c=None):
        self.orig_hw = None          # set by pack_frames: (H, W) of the un-padded frame
        self.shadow = None           # mixed configuration: fp16 copy of an fp32 stream tensor, written by its producer (ConvPlan(out16=...))
        self.compact = False         # ... and this fp32 tensor holds the even rows / columns of the grid only (out_subsample=2); shadow: full grid
        self.pairs = False           # split-product plans: the float32 storage holds [8 x fp16 hi | 8 x fp16 lo] per 8 channels (okp_conv_args.out_pairs):
                                     # written by ConvPlan(out_pairs=True), read by the patch-resident 3x3 kernel only - every other op refuses it
        require_cuda(t, "activation")
        if t.dim() != 4 or not t.is_contiguous():
            raise OkpError("activation must be a contiguous [N,H,W,C] tensor")
        self.t = t
        self.c0 = c0
        self.c = t.shape[3] - c0 if c is None else c
        if self.c0 < 0 or self.c0 + self.c > t.shape[3]:
            raise OkpError("channel window outside tensor")

    n = property(lambda s: s.t.shape[0])
    h = property(lambda s: s.t.shape[1])
    w = property(lambda s: s.t.shape[2])
    dtype = property(lambda s: s.t.dtype)

    def slice(self, c0, c):
        a = Act(self.t, self.c0 + c0, c)
        if self.pairs:
            if c0 % 8 or c % 8:
                raise OkpError("a pair-format activation is sliced in whole 8-channel groups")
            a.pairs = True
        return a

    def view(self, pairs_ok=False):
        if self.pairs and not pairs_ok:
            raise OkpError("this activation is in pair format ([hi | lo] fp16 per 8 channels): only a split-product 3x3 ConvPlan reads it")
        esz = self.t.element_size()
        return _lib.okp_tensor(self.t.data_ptr() + self.c0 * esz, self.h, self.w, self.t.shape[3],
                               self.t.numel() * esz - self.c0 * esz)

    @staticmethod
    def empty(n, h, w, c, dtype, device):
        return Act(torch.empty((n, h, w, c), dtype=dtype, device=device))

    @staticmethod
    def from_nchw(x, dtype):
        """Layout change at the public boundary (torch copy kernels; not part of the hot path)."""
        require_cuda(x, "input")
        return Act(x.permute(0, 2, 3, 1).contiguous().to(dtype))

    def to_nchw(self):
        if self.pairs:
            return self.pairs_to_float().to_nchw()
        return self.t[..., self.c0:self.c0 + self.c].permute(0, 3, 1, 2).float().contiguous()

    def pairs_to_float(self):
        """hi + lo of a pair-format activation as an ordinary float32 one (torch kernels: tests and debugging, not the hot path)."""
        if not self.pairs or self.c0 % 8 or self.c % 8:
            raise OkpError("pairs_to_float: a pair-format activation with whole 8-channel groups")
        n, h, w, ct = self.t.shape
        hl = self.t.view(torch.float16).view(n, h, w, ct // 8, 2, 8)[:, :, :, self.c0 // 8:(self.c0 + self.c) // 8]
        return Act((hl[..., 0, :].float() + hl[..., 1, :].float()).reshape(n, h, w, self.c).contiguous())

    @staticmethod
    def float_to_pairs(t):
        """A float32 [N,H,W,C] tensor in pair format (C % 8 == 0): hi = fp16(x), lo = fp16(x - hi), as the kernels split (tests)."""
        require_cuda(t, "activation")
        n, h, w, c = t.shape
        x = t.float().reshape(n, h, w, c // 8, 8)
        hi = x.half()
        lo = (x - hi.float()).half()
        a = Act(torch.stack((hi, lo), dim=4).reshape(n, h, w, c * 2).contiguous().view(torch.float32))
        a.pairs = True
        return a


_NULL_TENSOR = _lib.okp_tensor(None, 0, 0, 0, 0)


def _refuse_pairs(*acts):
    for a in acts:
        if a is not None and getattr(a, "pairs", False):
            raise OkpError("this activation is in pair format ([hi | lo] fp16 per 8 channels): only a split-product 3x3 ConvPlan reads it")


class ConvPlan:
    """One okp_conv plan.  taps: [(src, dy, dx, weight[cout, cin_src] float32)]."""

    def __init__(self, dtype, cins, strides, cout, taps, bias=None, relu=False, alg_k=None, tap_terms=None):
        """tap_terms (split-product plans only): per tap 3 (three-term product, the default) or 1 (single fp16 term)."""
        L = _lib.lib()
        self.dtype = dtype
        self.split = bool(getattr(_MODE, "split", False) and dtype == torch.float32)      # OKP_F32X3: fp32 tensors, products on the fp16 matrix pipe
        self.allow_compact = False       # set by the one consumer that knows how to read an even-pixels-only source (residual's skip plan)
        self.tap_terms = list(tap_terms) if (tap_terms is not None and self.split) else None
        self.cout = cout
        self.n_src = len(cins)
        self.cins = list(cins)
        self._strides = (list(strides) + [1])[:2]
        self._tap_srcs = [t[0] for t in taps]
        n_taps = len(taps)
        keep = []
        k = 0
        for i, (src, dy, dx, w) in enumerate(taps):
            w = np.ascontiguousarray(w, dtype=np.float32)
            if w.shape != (cout, cins[src]):
                raise OkpError(f"tap {i}: weight shape {w.shape} != {(cout, cins[src])}")
            keep.append(w)
            k += cins[src]
        self.alg_k = k if alg_k is None else alg_k      # algorithmic reduction length per output element
        b = None
        if bias is not None:
            b = np.ascontiguousarray(bias, dtype=np.float32)
            if b.shape != (cout,):
                raise OkpError("bias shape")
        if self.tap_terms is not None and len(self.tap_terms) != n_taps:
            raise OkpError("tap_terms: one entry per tap")
        code = OKP_F32X3 if self.split else okp_dtype(dtype)
        act = ACT_RELU if relu else ACT_NONE
        T = _ops()
        if T is not None:       # torch.ops.okp.conv_create: host tensors in, opaque handle out
            self._h = _dispatch(T.conv_create, code, list(cins), (list(strides) + [1] * len(cins))[:len(cins)], cout, [t[0] for t in taps], [t[1] for t in taps], [t[2] for t in taps],
                                [torch.from_numpy(w) for w in keep], torch.from_numpy(b) if b is not None else None, act, self.tap_terms or [])
            _attach_range_flag(self, "conv_set_range_flag")
            return
        arr_t = (_lib.okp_tap * n_taps)()
        for i, (src, dy, dx, _) in enumerate(taps):
            arr_t[i] = _lib.okp_tap(src, dy, dx, keep[i].ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        cin_arr = (ctypes.c_int32 * 2)(*(list(cins) + [0])[:2])
        st_arr = (ctypes.c_int32 * 2)(*(list(strides) + [1])[:2])
        bp = b.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if b is not None else None
        if self.tap_terms is not None:
            terms = (ctypes.c_uint8 * n_taps)(*self.tap_terms)
            self._h = L.okp_conv_create_x3(self.n_src, cin_arr, st_arr, cout, n_taps, arr_t, terms, bp, act)
        else:
            self._h = L.okp_conv_create(code, self.n_src, cin_arr, st_arr, cout, n_taps, arr_t, bp, act)
        if not self._h:
            raise OkpError("okp_conv_create: " + L.okp_last_error().decode())
        _attach_range_flag(self, "conv_set_range_flag")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        try:
            if h and _lib is not None and _lib._lib is not None:
                _lib._lib.okp_conv_destroy(h)       # (the same C symbol torch.ops.okp.conv_destroy calls; safe during interpreter shutdown)
        except Exception:       # interpreter shutdown: the process is going away with its HBM
            pass

    def picks_patch(self, n, ho, wo, src_pix_strides, out_step=1, n_classes=1, tile=0):
        """True if a launch of this shape (dense output of ho*out_step x wo*out_step pixels, no depth-wise branch / fp16 side output) runs
        on the patch-resident kernel (tile 13 of okp_conv_forward's heuristic, okp_conv_select_tile) - the one kernel that reads and
        writes pair-format activations."""
        tile = tile or FORCE_TILE
        T = _ops()
        if T is not None:
            return bool(_dispatch(T.conv_picks_patch, self._h, n, ho, wo, list(src_pix_strides), self.cout, out_step, n_classes, tile))
        a = _lib.okp_conv_args()
        a.n, a.ho, a.wo = n, ho, wo
        for i, ps in enumerate(src_pix_strides):
            a.src[i] = _lib.okp_tensor(1, 1, 1, ps, 1)          # (the heuristic reads pixel strides and the output grid only)
        a.out = _lib.okp_tensor(1, ho * out_step, wo * out_step, self.cout, 1)
        a.out_step, a.n_classes = out_step, n_classes
        if tile:                                                # a forced tile 13 on a shape the kernel refuses is not a "yes" (the launch would end in EINVAL)
            return tile == 13 and bool(_lib.lib().okp_conv_patch_applies(self._h, ctypes.byref(a)))
        return _lib.lib().okp_conv_select_tile(self._h, ctypes.byref(a)) == 13

    def _algorithmic_bytes(self, srcs, out, ho, wo, res, out_step, n_classes, dw, out16, write_out, out_subsample):
        """Compulsory HBM bytes of one launch (bench.py's roofline of HBM-bound launches): every source pixel the taps sample once,
        every output element once (+ the fp16 copy), the residual once; weights not counted (L2-resident, shared by all tiles)."""
        esz = lambda a: a.t.element_size()
        b = 0
        for i, s_ in enumerate(srcs):
            n_taps_src = sum(1 for t in self._tap_srcs if t == i)
            stride = self._strides[i]
            px = s_.h * s_.w if (n_taps_src > 1 or stride == 1) else ho * wo       # a strided 1x1 samples one pixel per output pixel
            b += s_.n * px * self.cins[i] * esz(s_)
        opx = out.n * ho * wo * (n_classes if n_classes > 1 else 1)
        if write_out:
            b += (opx // (out_subsample * out_subsample)) * self.cout * esz(out)
        if out16 is not None:
            b += opx * self.cout * 2
        if res is not None:
            b += opx * self.cout * esz(res)
        if dw is not None:
            b += opx * self.cout * esz(dw[2]) + (opx * self.cout * esz(dw[3]) if dw[3] is not None else 0)
        return b

    def __call__(self, srcs, out, ho, wo, res=None, out_step=1, oy=0, ox=0, tile=0, dw=None, n_classes=1, out16=None, write_out=True, out_subsample=1,
                 out_pairs=False):
        """dw = (w_dev [9,cout] fp32, bias_dev [cout] fp32, dw_out Act, dw_res Act|None): fused depth-wise 3x3 branch.
        out_pairs (split-product 3x3 plans on the patch-resident kernel): `out` is written in pair format (Act.pairs); sources that are
        in pair format are passed as such (okp_conv_args.src_pairs / out_pairs).
        Split-product plans: out16 = fp16 Act that receives the result as well (write_out=False: only that copy is written; `out` then
        only describes the grid), and `res` may be an fp16 tensor (okp_conv_args.out16 / res_is_f16)."""
        for s in srcs:
            if s.dtype != self.dtype:
                raise OkpError("source dtype differs from plan dtype")
            if s.compact and not self.allow_compact:
                raise OkpError("source keeps the even rows / columns of its grid only (out_subsample): this plan does not expect that")
        if out.dtype != self.dtype or out.c != self.cout:
            raise OkpError(f"out has {out.c} channels / {out.dtype}, plan has {self.cout} / {self.dtype}")
        res16 = res is not None and self.split and res.dtype == torch.float16
        if res is not None and res.dtype != self.dtype and not res16:
            raise OkpError(f"residual is {res.dtype}, plan is {self.dtype}" + (" (split-product plans also take a float16 residual)" if self.split else ""))
        if dw is not None and (dw[2].dtype != self.dtype or (dw[3] is not None and dw[3].dtype != self.dtype)):
            raise OkpError("depth-wise branch: dw_out / dw_res dtype differs from plan dtype")
        if (out16 is not None or res16 or not write_out) and not self.split:
            raise OkpError("out16 / fp16 residual / write_out=False belong to split-product (ops.F32X3) plans")
        if out16 is not None and (out16.dtype != torch.float16 or out16.c != self.cout):
            raise OkpError("out16 must be a float16 activation with the plan's output channels")
        if not write_out and out16 is None:
            raise OkpError("write_out=False needs out16")
        if out_subsample not in (1, 2) or (out_subsample == 2 and out16 is None):
            raise OkpError("out_subsample is 1 or 2, and 2 needs out16")
        src_pairs = sum(1 << i for i, s in enumerate(srcs) if s.pairs)
        if (src_pairs or out_pairs) and not self.split:
            raise OkpError("pair-format activations belong to split-product (ops.F32X3) plans")
        if (res is not None and res.pairs) or (dw is not None and (src_pairs or out_pairs)):
            raise OkpError("pair format: the residual stays float32, and the depth-wise branch does not take it")
        if out_pairs and (out.c0 % 8 or out.t.shape[3] % 8):
            raise OkpError("pair format: the output window starts on a whole 8-channel group of its tensor")
        extended = out16 is not None or res16 or out_subsample == 2 or not write_out      # (what keeps a split-product launch off the patch-resident kernel)
        macs = out.n * ho * wo * self.cout * self.alg_k
        macs_dw = out.n * ho * wo * self.cout * 9 if dw is not None else 0
        tile = tile or FORCE_TILE
        out.pairs = False                # whatever this Act held before, the launch overwrites it with ordinary values - or pairs, set below
        T = _ops()
        hook = LAUNCH_HOOK
        if T is not None:
            s1 = srcs[1] if len(srcs) > 1 else None
            if hook is not None:
                self.last_launch = (out.n, ho, wo, dw is not None, res is not None, n_classes)
                self.last_bytes = self._algorithmic_bytes(srcs, out, ho, wo, res, out_step, n_classes, dw, out16, write_out, out_subsample)
                tile = _dispatch(T.conv_select_tile, self._h, srcs[0].t, srcs[0].c0, s1.t if s1 else None, s1.c0 if s1 else 0, out.t, out.c0, ho, wo,
                                 out_step, oy, ox, tile, n_classes, dw is not None, extended)
                token = hook.before(self, tile, macs)
            if dw is not None:
                dw_w, dw_b, dw_out, dw_res = dw
                dw_out.pairs = False
                dwa = (dw_w, dw_b, dw_out.t, dw_out.c0, dw_res.t if dw_res is not None else None, dw_res.c0 if dw_res is not None else 0)
            else:
                dwa = (None, None, None, 0, None, 0)
            _dispatch(T.conv_forward, self._h, srcs[0].t, srcs[0].c0, s1.t if s1 else None, s1.c0 if s1 else 0, out.t, out.c0, ho, wo,
                      res.t if res is not None else None, res.c0 if res is not None else 0, out_step, oy, ox, tile, n_classes, *dwa, stream_int(),
                      out16.t if out16 is not None else None, out16.c0 if out16 is not None else 0, bool(write_out), out_subsample, src_pairs, bool(out_pairs))
            if hook is not None:
                hook.after(token)
        else:
            a = _lib.okp_conv_args()
            a.n, a.ho, a.wo = out.n, ho, wo
            for i, s in enumerate(srcs):
                a.src[i] = s.view(pairs_ok=True)
            a.src_pairs, a.out_pairs = src_pairs, 1 if out_pairs else 0
            a.out = out.view()
            if not write_out:
                a.out.data = None
            if out16 is not None:
                a.out16 = out16.view()
            a.res_is_f16 = 1 if res16 else 0
            a.out_subsample = out_subsample
            a.out_step, a.out_oy, a.out_ox = out_step, oy, ox
            a.res = res.view() if res is not None else _NULL_TENSOR
            a.tile = tile
            a.n_classes = n_classes
            if dw is not None:
                dw_w, dw_b, dw_out, dw_res = dw
                dw_out.pairs = False
                a.dw_w_dev, a.dw_bias_dev = dw_w.data_ptr(), dw_b.data_ptr()
                a.dw_out = dw_out.view()
                a.dw_res = dw_res.view() if dw_res is not None else _NULL_TENSOR
            if hook is not None:
                self.last_launch = (out.n, ho, wo, dw is not None, res is not None, n_classes)
                self.last_bytes = self._algorithmic_bytes(srcs, out, ho, wo, res, out_step, n_classes, dw, out16, write_out, out_subsample)
                a.tile = _lib.lib().okp_conv_select_tile(self._h, ctypes.byref(a))
                token = hook.before(self, a.tile, macs)
            _lib.check(_lib.lib().okp_conv_forward(self._h, ctypes.byref(a), stream_handle()), "okp_conv_forward")
            if hook is not None:
                hook.after(token)
        out.pairs = bool(out_pairs)
        COUNTERS["pair_outputs"] += 1 if out_pairs else 0
        COUNTERS["macs"] += macs + macs_dw
        COUNTERS["launches"] += 1


# "ctypes_launches": C-ABI calls (launches and plan creations) that went through the ctypes binding instead of torch.ops.okp - stays 0 while the
# dispatcher shim is loaded (tests/test_gpu_torch_ops.py); the ctypes binding serves non-torch hosts and OKP_TORCH_OPS=0
COUNTERS = {"macs": 0, "launches": 0, "pair_outputs": 0, "ctypes_launches": 0}
LAUNCH_HOOK = None      # bench.py: object with before(plan, tile, macs) / after(token) bracketing conv launches
FORCE_TILE = 0          # tests: 1/2/3 pins the implicit-GEMM tile (64/128/256), 0 = heuristic


def fire_fused(squeeze, expand, wd_dev, bd_dev, x, out, stride, skip):
    """One-launch fire module (16-bit plans: okp_fire2.hip; split-product plans: okp_fire_x3.hip): see okp_fire_forward in include/okp.h."""
    _refuse_pairs(x)
    out.pairs = False
    T = _ops()
    if T is not None:
        _dispatch(T.fire_forward, squeeze._h, expand._h, wd_dev, bd_dev, x.t, x.c0, out.t, out.c0, stride, bool(skip), stream_int())
    else:
        a = _lib.okp_fire_args()
        a.n = x.n
        a.x, a.out = x.view(), out.view()
        a.stride, a.skip = stride, 1 if skip else 0
        _lib.check(_lib.lib().okp_fire_forward(squeeze._h, expand._h, wd_dev.data_ptr(), bd_dev.data_ptr(), ctypes.byref(a), stream_handle()),
                   "okp_fire_forward")
    half = expand.cout
    COUNTERS["macs"] += x.n * x.h * x.w * squeeze.cout * squeeze.cins[0] + out.n * out.h * out.w * half * (expand.cins[0] + 9)
    COUNTERS["launches"] += 1


FIRE_CHAIN_MAX = 8
# shortest run of chainable modules that takes the resident launch: 1 - a single fire(384, 384) on an 8 x 8 map (low1[1] of the 16 x 16 level)
# is one latency-bound tile per workgroup either way, and the resident kernel has the shorter chain of waits (step -0.6 %, ABBA)
# (the FUSE_* / *_MIN values below are module attributes that tests flip to compare paths; they are not environment switches)
FIRE_CHAIN_MIN = 1
FUSE_FIRE_CHAIN_FRAMED = True   # entry (stride 2) + chain + exit modules of the innermost level in one launch
FUSE_FIRE_CHAIN = True          # consecutive 512-channel fire modules on <= 4x4 maps: one resident launch


def fire_chain(modules, x, out):
    """modules: list of (squeeze plan, expand plan, dw weights, dw bias) of consecutive fire(512, 512) modules; x, out: Acts."""
    _refuse_pairs(x)
    n = len(modules)
    out.pairs = False
    T = _ops()
    if T is not None:
        _dispatch(T.fire_chain_forward, [m[0]._h for m in modules], [m[1]._h for m in modules], [m[2] for m in modules], [m[3] for m in modules],
                  x.t, x.c0, out.t, out.c0, stream_int())
    else:
        sq = (ctypes.c_void_p * n)(*[m[0]._h for m in modules])
        ex = (ctypes.c_void_p * n)(*[m[1]._h for m in modules])
        wd = (ctypes.c_void_p * n)(*[m[2].data_ptr() for m in modules])
        bd = (ctypes.c_void_p * n)(*[m[3].data_ptr() for m in modules])
        xv, ov = x.view(), out.view()
        _lib.check(_lib.lib().okp_fire_chain_forward(n, sq, ex, wd, bd, x.n, ctypes.byref(xv), ctypes.byref(ov), stream_handle()), "okp_fire_chain_forward")
    for i, m in enumerate(modules):
        half = m[1].cout
        sq_px = x.h * x.w if i == 0 else out.h * out.w          # (entry / exit form: the entry module squeezes at the input resolution)
        COUNTERS["macs"] += x.n * (sq_px * m[0].cout * m[0].cins[0] + out.h * out.w * half * (m[1].cins[0] + 9))
    COUNTERS["launches"] += 1


SIDE_STREAMS = True      # hourglass up1 branches run on side streams, concurrently with the low path
SIDE_MIN_BATCH = int(os.environ.get("OKP_SIDE_MIN_BATCH", "8"))    # eager passes of fewer frames are host-bound and a fork / join costs the host ~40 us:
                                                                   # below this batch the branches run on the main stream (unless a hipGraph is being captured)
FUSE_FIRE = True        # one-launch streaming fire module (okp_fire2.hip) where it exists: 256 -> 128 -> 256, stride 1
                        # (the two high-resolution hourglass levels): 98-104 us vs 140 us per module at 64x64, N=64
FUSE_FIRE_MIN_HW = 8    # 4x4 maps: two launches are faster (25 vs 20 us)


_FIRE2_CONFIGS = {1: {(256, 128), (384, 192), (512, 256), (384, 128), (512, 192)},      # stride -> (cin, mid) instances of okp_fire2.hip
                  2: {(256, 128), (384, 192), (384, 256), (256, 192)}}
if os.environ.get("OKP_FIRE2_S2_256_192", "1") == "0":      # A/B switch (scripts/README.md): the 32 x 32 -> 16 x 16 module as squeeze + fused tail, as until round 6
    _FIRE2_CONFIGS[2].discard((256, 192))
FUSE_FIRE_S2 = True


def fire_fusable(inp_dim, mid, stride, h, w):
    if stride == 2 and not FUSE_FIRE_S2:
        return False
    return FUSE_FIRE and (inp_dim, mid) in _FIRE2_CONFIGS.get(stride, ()) and min(h, w) // stride >= FUSE_FIRE_MIN_HW


FUSE_FIRE_X3 = True     # split-product plans: the one-launch fire module okp_fire_x3.hip (256 -> 128 -> 256 with skip, stride 1)
FUSE_FIRE_X3_MIN_HW = 16


FUSE_FIRE_X3_S2 = True  # ... and its stride-2 form without skip (64 x 64 -> 32 x 32: low1[0] of the outermost hourglass level)


def fire_fusable_x3(inp_dim, mid, stride, skip, h, w):
    if not (FUSE_FIRE_X3 and (inp_dim, mid) == (256, 128)):
        return False
    if stride == 1:
        return bool(skip) and min(h, w) >= FUSE_FIRE_X3_MIN_HW
    return FUSE_FIRE_X3_S2 and stride == 2 and not skip and min(h, w) // 2 >= FUSE_FIRE_X3_MIN_HW


LIGHT_EVENTS = True     # forks / joins through okp_stream_wait_stream (events without a system-scope fence); False: torch's wait_stream


def stream_wait(waiter, signaller):
    """`waiter` (torch.cuda.Stream) waits for the work enqueued on `signaller` so far."""
    if LIGHT_EVENTS:
        T = _ops()
        if T is not None:
            _dispatch(T.stream_wait_stream, waiter.cuda_stream, signaller.cuda_stream)
        else:
            _lib.check(_lib.lib().okp_stream_wait_stream(ctypes.c_void_p(waiter.cuda_stream), ctypes.c_void_p(signaller.cuda_stream)), "okp_stream_wait_stream")
    else:
        waiter.wait_stream(signaller)


def cast(src, dtype):
    """Act -> Act of another element type (okp_cast: fp32 <-> fp16 / bf16): the boundary of an fp16 sub-network inside the fp32 stream."""
    _refuse_pairs(src)
    if src.c0 != 0 or src.c != src.t.shape[3]:
        raise OkpError("cast takes a whole tensor, not a channel window")
    out = Act(torch.empty(src.t.shape, dtype=dtype, device=src.t.device))
    okp_dtype(dtype)
    # to fp32 inside a split-product pass: what the fp16 sub-network made is about to be halved again - its range is checked here, exactly
    flag = getattr(_MODE, "range_flag", None) if dtype == torch.float32 else None
    T = _ops()
    if T is not None:
        _dispatch(T.cast, src.t, out.t, stream_int(), flag)
    else:
        _lib.check(_lib.lib().okp_cast(okp_dtype(src.dtype), src.t.data_ptr(), okp_dtype(dtype), out.t.data_ptr(), src.t.numel(),
                                       flag.data_ptr() if flag is not None else None, stream_handle()), "okp_cast")
    COUNTERS["launches"] += 1
    return out


def add_f16_f32(a16, b32, relu=True):
    """relu(a + b): a an fp16 Act, b an fp32 Act of the same shape -> fp32 Act (okp_add_f16_f32)."""
    _refuse_pairs(a16, b32)
    if a16.dtype != torch.float16 or b32.dtype != torch.float32 or a16.t.shape != b32.t.shape or a16.c0 or b32.c0 or a16.c != a16.t.shape[3] or b32.c != b32.t.shape[3]:
        raise OkpError("add_f16_f32 takes whole float16 / float32 tensors of one shape")
    out = Act(torch.empty_like(b32.t))
    flag = getattr(_MODE, "range_flag", None)
    T = _ops()
    if T is not None:
        _dispatch(T.add_f16_f32, a16.t, b32.t, out.t, ACT_RELU if relu else ACT_NONE, stream_int(), flag)
    else:
        _lib.check(_lib.lib().okp_add_f16_f32(a16.t.data_ptr(), b32.t.data_ptr(), out.t.data_ptr(), out.t.numel(), ACT_RELU if relu else ACT_NONE,
                                              flag.data_ptr() if flag is not None else None, stream_handle()), "okp_add_f16_f32")
    COUNTERS["launches"] += 1
    return out


def dwconv3x3(src, w_dev, bias_dev, out, stride, res=None, relu=True):
    _refuse_pairs(src, res)
    dt = okp_dtype(src.dtype)
    out.pairs = False
    T = _ops()
    if T is not None:
        _dispatch(T.dwconv3x3_forward, src.t, src.c0, src.c, stride, w_dev, bias_dev, res.t if res is not None else None, res.c0 if res is not None else 0,
                  out.t, out.c0, ACT_RELU if relu else ACT_NONE, stream_int())
        COUNTERS["macs"] += out.n * out.h * out.w * src.c * 9
        COUNTERS["launches"] += 1
        return
    sv, ov = src.view(), out.view()
    rv = res.view() if res is not None else None
    _lib.check(_lib.lib().okp_dwconv3x3_forward(dt, src.n, src.c, stride, ctypes.byref(sv), w_dev.data_ptr(), bias_dev.data_ptr(),
                                                ctypes.byref(rv) if rv is not None else None, ctypes.byref(ov),
                                                ACT_RELU if relu else ACT_NONE, stream_handle()), "okp_dwconv3x3_forward")
    COUNTERS["macs"] += out.n * out.h * out.w * src.c * 9
    COUNTERS["launches"] += 1


def stem_packed_width(w):
    wo = (w + 6 - 7) // 2 + 1
    need = max(w + 6, 2 * (wo - 1) + 8)
    return (need + 3) // 4 * 4


def pack_frames(frames, dtype):
    """NCHW fp32 frames -> NHWC4 with zero halo 3 (input layout of the 7x7/s2 stem)."""
    require_cuda(frames, "frames")
    if frames.dtype != torch.float32 or frames.dim() != 4 or frames.shape[1] != 3:
        raise OkpError("frames must be float32 [N,3,H,W]")
    frames = frames.contiguous()
    n, _, h, w = frames.shape
    wp = stem_packed_width(w)
    out = torch.empty((n, h + 6, wp, 4), dtype=dtype, device=frames.device)
    okp_dtype(dtype)
    T = _ops()
    if T is not None:
        _dispatch(T.pack_frames, frames, out, stream_int())
    else:
        _lib.check(_lib.lib().okp_pack_frames(okp_dtype(dtype), frames.data_ptr(), n, h, w, out.data_ptr(), wp, stream_handle()), "okp_pack_frames")
    COUNTERS["launches"] += 1
    act = Act(out)
    act.orig_hw = (h, w)
    return act


class StemPlan:
    """The 7x7/s2 stem kernels (okp_stem_*): w [128,3,7,7] and bias [128] with BatchNorm folded (host fp32).  dtype bfloat16 / float16:
    the 16-bit kernel; float32 inside ops.f32_split(): the split-product kernel (fp32 NHWC output, three-term products; it reads the
    caller's fp32 NCHW frames: from_nchw only)."""

    def __init__(self, w, bias, dtype=torch.bfloat16):
        w = np.ascontiguousarray(w, dtype=np.float32)
        b = np.ascontiguousarray(bias, dtype=np.float32)
        if w.shape != (128, 3, 7, 7) or b.shape != (128,):
            raise OkpError("the stem kernel is 7x7, 3 -> 128 channels")
        self.split = bool(dtype == torch.float32 and getattr(_MODE, "split", False))
        if dtype not in HALF_DTYPES and not self.split:
            raise OkpError("the stem kernel computes in bfloat16, float16 or (float32 inside ops.f32_split()) split products")
        self.dtype = dtype
        L = _lib.lib()
        T = _ops()
        if T is not None:
            self._h = _dispatch(T.stem_create, OKP_F32X3 if self.split else okp_dtype(dtype), torch.from_numpy(w), torch.from_numpy(b))
            _attach_range_flag(self, "stem_set_range_flag")
            return
        self._h = L.okp_stem_create_dtype(OKP_F32X3 if self.split else okp_dtype(dtype), w.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), b.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        if not self._h:
            raise OkpError("okp_stem_create: " + L.okp_last_error().decode())
        _attach_range_flag(self, "stem_set_range_flag")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        try:
            if h and _lib is not None and _lib._lib is not None:
                _lib._lib.okp_stem_destroy(h)
        except Exception:
            pass

    def from_nchw(self, frames, out, out_pairs=False):
        """frames: fp32 NCHW [N,3,H,W] on the device (the reference's input layout); no packing pass.
        out_pairs (split-product stem): `out` is written in pair format (Act.pairs; okp_stem_forward_nchw_pairs)."""
        require_cuda(frames, "frames")
        if frames.dtype != torch.float32 or frames.dim() != 4 or frames.shape[1] != 3 or out.dtype != self.dtype:
            raise OkpError("frames must be float32 [N,3,H,W] and the output of the plan's element type")
        frames = frames.contiguous()
        n, _, h, w = frames.shape
        T = _ops()
        out.pairs = False
        if out_pairs:
            if not self.split:
                raise OkpError("pair-format output belongs to the split-product stem")
            if T is not None:
                _dispatch(T.stem_forward_nchw_pairs, self._h, frames, out.t, out.c0, stream_int())
            else:
                ov = out.view()
                _lib.check(_lib.lib().okp_stem_forward_nchw_pairs(self._h, n, h, w, frames.data_ptr(), ctypes.byref(ov), stream_handle()), "okp_stem_forward_nchw_pairs")
            out.pairs = True
            COUNTERS["pair_outputs"] += 1
        elif T is not None:
            _dispatch(T.stem_forward_nchw, self._h, frames, out.t, out.c0, stream_int())
        else:
            ov = out.view()
            _lib.check(_lib.lib().okp_stem_forward_nchw(self._h, n, h, w, frames.data_ptr(), ctypes.byref(ov), stream_handle()), "okp_stem_forward_nchw")
        COUNTERS["macs"] += out.n * out.h * out.w * 128 * 147
        COUNTERS["launches"] += 1

    def __call__(self, packed, out):
        if packed.orig_hw is None or packed.dtype != self.dtype or out.dtype != self.dtype or self.split:
            raise OkpError("stem input must be the output of ops.pack_frames in the plan's 16-bit type (the split-product stem reads NCHW frames: from_nchw)")
        h, w = packed.orig_hw
        out.pairs = False
        T = _ops()
        if T is not None:
            _dispatch(T.stem_forward, self._h, packed.t, h, w, out.t, out.c0, stream_int())
        else:
            pv, ov = packed.view(), out.view()
            _lib.check(_lib.lib().okp_stem_forward(self._h, packed.n, h, w, ctypes.byref(pv), ctypes.byref(ov), stream_handle()), "okp_stem_forward")
        COUNTERS["macs"] += out.n * out.h * out.w * 128 * 147
        COUNTERS["launches"] += 1


# normalisation constants of the reference's data loader (perception/datasets/video.py:55-56)
RGB_MEAN = (0.40789654, 0.44719302, 0.47026115)
RGB_STD = (0.28863828, 0.27408164, 0.27809835)


def pack_frames_u8(frames, dtype, mean=RGB_MEAN, std=RGB_STD):
    """uint8 RGB frames [N,H,W,3] (device) -> normalised, packed stem input (see okp_pack_frames_u8)."""
    require_cuda(frames, "frames")
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
        raise OkpError("frames must be uint8 [N,H,W,3]")
    frames = frames.contiguous()
    n, h, w, _ = frames.shape
    wp = stem_packed_width(w)
    out = torch.empty((n, h + 6, wp, 4), dtype=dtype, device=frames.device)
    okp_dtype(dtype)
    T = _ops()
    if T is not None:
        _dispatch(T.pack_frames_u8, frames, [float(np.float32(v)) for v in mean], [float(np.float32(v)) for v in std], out, stream_int())
        COUNTERS["launches"] += 1
        act = Act(out)
        act.orig_hw = (h, w)
        return act
    m = (ctypes.c_float * 3)(*mean)
    sd = (ctypes.c_float * 3)(*std)
    _lib.check(_lib.lib().okp_pack_frames_u8(okp_dtype(dtype), frames.data_ptr(), n, h, w, m, sd, out.data_ptr(), wp, stream_handle()), "okp_pack_frames_u8")
    COUNTERS["launches"] += 1
    act = Act(out)
    act.orig_hw = (h, w)
    return act


def smallest_max_size(h, w, max_size):
    """albumentations.SmallestMaxSize geometry: scale the shortest side to max_size, round half to even like py3round."""
    scale = max_size / min(h, w)
    return int(round(h * scale)), int(round(w * scale))


def preprocess_u8(frames, dtype, size=511, mean=RGB_MEAN, std=RGB_STD):
    """Raw uint8 RGB frames [N,H,W,3] (device) -> resized (shortest side = size), centre-cropped size x size, normalised and
    packed stem input (okp_preprocess_u8; the reference's SmallestMaxSize + CenterCrop + normalisation, video.py:95-96,215)."""
    require_cuda(frames, "frames")
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
        raise OkpError("frames must be uint8 [N,H,W,3]")
    frames = frames.contiguous()
    n, sh, sw, _ = frames.shape
    rh, rw = smallest_max_size(sh, sw, size)
    if rh < size or rw < size:
        raise OkpError("resized frame is smaller than the crop")
    cy, cx = (rh - size) // 2, (rw - size) // 2
    wp = stem_packed_width(size)
    out = torch.empty((n, size + 6, wp, 4), dtype=dtype, device=frames.device)
    okp_dtype(dtype)
    T = _ops()
    if T is not None:
        _dispatch(T.preprocess_u8, frames, rh, rw, cy, cx, size, size, [float(np.float32(v)) for v in mean], [float(np.float32(v)) for v in std], out, stream_int())
        COUNTERS["launches"] += 1
        act = Act(out)
        act.orig_hw = (size, size)
        return act
    m = (ctypes.c_float * 3)(*mean)
    sd = (ctypes.c_float * 3)(*std)
    _lib.check(_lib.lib().okp_preprocess_u8(okp_dtype(dtype), frames.data_ptr(), n, sh, sw, rh, rw, cy, cx, size, size, m, sd,
                                            out.data_ptr(), wp, stream_handle()), "okp_preprocess_u8")
    COUNTERS["launches"] += 1
    act = Act(out)
    act.orig_hw = (size, size)
    return act


def _head_out_args(n, h, w, outputs, w_dev, bias_dev):
    a = _lib.okp_head_out_args()
    a.n, a.h, a.w = n, h, w
    a.n_out = len(outputs)
    for i, (off, act, t, ch) in enumerate(outputs):
        a.in_c_off[i] = off
        a.act[i] = act
        a.out_ptr[i] = t.data_ptr() + ch * t.shape[2] * t.shape[3] * 4
        a.out_n_stride[i] = t.shape[1] * t.shape[2] * t.shape[3]
    a.w_dev, a.bias_dev = w_dev.data_ptr(), bias_dev.data_ptr()
    return a


def head_out(src, outputs, w_dev, bias_dev):
    """outputs: list of (in_c_off, act, out_tensor[N,Cx,H,W] fp32, channel index)."""
    _refuse_pairs(src)
    T = _ops()
    if T is not None:
        _dispatch(T.head_out_forward, src.t, src.c0, [o[0] for o in outputs], [o[1] for o in outputs], [o[2] for o in outputs], [o[3] for o in outputs],
                  w_dev, bias_dev, stream_int())
    else:
        a = _head_out_args(src.n, src.h, src.w, outputs, w_dev, bias_dev)
        a.src = src.view()
        _lib.check(_lib.lib().okp_head_out_forward(okp_dtype(src.dtype), ctypes.byref(a), stream_handle()), "okp_head_out_forward")
    COUNTERS["macs"] += src.n * src.h * src.w * 32 * len(outputs)
    COUNTERS["launches"] += 1


FUSE_HEADS = True    # 16-bit, 128 features: the three heads of a stack in one launch
FUSE_HEADS_X3 = True # split-product configuration: the same with three-term products, on the pair-format output of the last `cnvs` convolution


def heads_fused(l1, l2, x, outputs, w_dev, bias_dev):
    """The three prediction heads in one launch (okp_heads_forward): l1 256 -> 384, l2 block-diagonal 384 -> 96 plans.
    Split-product plans: x is a pair-format activation (Act.pairs; written by the 3x3 `cnvs` convolution in front of the heads)."""
    if l1.split != bool(x.pairs):
        raise OkpError("heads_fused: split-product plans read a pair-format activation, the 16-bit plans an ordinary one")
    T = _ops()
    if T is not None:
        _dispatch(T.heads_forward, l1._h, l2._h, x.t, x.c0, [o[0] for o in outputs], [o[1] for o in outputs], [o[2] for o in outputs], [o[3] for o in outputs],
                  w_dev, bias_dev, stream_int())
    else:
        a = _head_out_args(x.n, x.h, x.w, outputs, w_dev, bias_dev)
        xv = x.view(pairs_ok=True)
        _lib.check(_lib.lib().okp_heads_forward(l1._h, l2._h, ctypes.byref(a), ctypes.byref(xv), stream_handle()), "okp_heads_forward")
    COUNTERS["macs"] += x.n * x.h * x.w * (l1.cout * l1.alg_k + l2.cout * l2.alg_k + 32 * len(outputs))
    COUNTERS["launches"] += 1


def peak_nms(heat, cap=64):
    """heat [N,K,H,W] fp32 (device) -> count [N,K] int32, yx [N,K,cap,2] int32, xyc [N,K,cap,3] fp32 (device)."""
    require_cuda(heat, "heat")
    if heat.dtype != torch.float32 or heat.dim() != 4:
        raise OkpError("heat must be float32 [N,K,H,W]")
    heat = heat.contiguous()
    n, k, h, w = heat.shape
    # no fill kernels: okp_peak_nms writes every count and defines the unused slots of yx / xyc (zeros) itself
    count = torch.empty((n, k), dtype=torch.int32, device=heat.device)
    yx = torch.empty((n, k, cap, 2), dtype=torch.int32, device=heat.device)
    xyc = torch.empty((n, k, cap, 3), dtype=torch.float32, device=heat.device)
    T = _ops()
    if T is not None:
        _dispatch(T.peak_nms, heat, cap, count, yx, xyc, stream_int())
    else:
        _lib.check(_lib.lib().okp_peak_nms(heat.data_ptr(), n * k, h, w, cap, count.data_ptr(), yx.data_ptr(), xyc.data_ptr(), stream_handle()), "okp_peak_nms")
    return count, yx, xyc


def nms_maxpool(x, size=5):
    """x * (x == maxpool_{size}(x)) on [N,C,H,W] fp32 device maps (perception/models.py:55-58)."""
    require_cuda(x, "x")
    if x.dtype != torch.float32 or x.dim() != 4:
        raise OkpError("nms expects float32 [N,C,H,W]")
    x = x.contiguous()
    out = torch.empty_like(x)
    n, c, h, w = x.shape
    T = _ops()
    if T is not None:
        _dispatch(T.nms_maxpool, x, size, out, stream_int())
    else:
        _lib.check(_lib.lib().okp_nms_maxpool(x.data_ptr(), n * c, h, w, size, out.data_ptr(), stream_handle()), "okp_nms_maxpool")
    return out


def make_camera(K, D, model=_lib.CAM_EQUIDISTANT):
    K = np.asarray(K, dtype=np.float64)
    D = np.asarray(D, dtype=np.float64).reshape(-1)
    if abs(K[0, 1]) > 1e-12 or abs(K[1, 0]) > 1e-12:
        raise OkpError("camera matrix with skew is not supported")
    cam = _lib.okp_camera(K[0, 0], K[1, 1], K[0, 2], K[1, 2])
    for i in range(4):
        cam.d[i] = D[i] if i < D.size else 0.0
    cam.model = model
    return cam


def camera_undistort(cam, xy):
    """xy [M,2] pixels (device) -> [M,2] fp64 undistorted pixels (P = K), by the camera's distortion model."""
    require_cuda(xy, "xy")
    xy = xy.to(torch.float32).contiguous()
    out = torch.empty((xy.shape[0], 2), dtype=torch.float64, device=xy.device)
    T = _ops()
    if T is not None:
        _dispatch(T.camera_undistort, _cam_list(cam), xy, out, stream_int())
    else:
        _lib.check(_lib.lib().okp_camera_undistort(ctypes.byref(cam), xy.data_ptr(), xy.shape[0], out.data_ptr(), stream_handle()), "okp_camera_undistort")
    return out


fisheye_undistort = camera_undistort


def unproject_depth(cam, xy, map_id, depth, max_x, max_y):
    """xy [M,2] fp32, map_id [M] int32, depth [NM,H,W] fp32 -> [M,3] fp64 (all device)."""
    require_cuda(xy, "xy")
    xy = xy.to(torch.float32).contiguous()
    map_id = map_id.to(torch.int32).contiguous()
    depth = depth.contiguous()
    if depth.dtype != torch.float32 or depth.dim() != 3:
        raise OkpError("depth must be float32 [maps,H,W]")
    out = torch.empty((xy.shape[0], 3), dtype=torch.float64, device=xy.device)
    T = _ops()
    if T is not None:
        _dispatch(T.unproject_depth, _cam_list(cam), xy, map_id, depth, max_x, max_y, out, stream_int())
    else:
        _lib.check(_lib.lib().okp_unproject_depth(ctypes.byref(cam), xy.data_ptr(), map_id.data_ptr(), xy.shape[0], depth.data_ptr(),
                                                  depth.shape[1], depth.shape[2], max_x, max_y, out.data_ptr(), stream_handle()), "okp_unproject_depth")
    return out


def lift_peaks(cam, count, xyc, depth, max_x, max_y):
    """count [N,K] int32, xyc [N,K,cap,3] fp32, depth [N,K,H,W] fp32 (device) -> [N,K,cap,4] fp64 (X,Y,Z,conf)."""
    require_cuda(xyc, "xyc")
    n, k, cap, _ = xyc.shape
    depth = depth.contiguous()
    if depth.dtype != torch.float32 or depth.shape[:2] != (n, k):
        raise OkpError("depth must be float32 [N,K,H,W] matching the peak tensors")
    out = torch.empty((n, k, cap, 4), dtype=torch.float64, device=xyc.device)
    T = _ops()
    if T is not None:
        _dispatch(T.lift_peaks, _cam_list(cam), count, xyc, depth, max_x, max_y, out, stream_int())
    else:
        _lib.check(_lib.lib().okp_lift_peaks(ctypes.byref(cam), count.data_ptr(), xyc.data_ptr(), n * k, cap, depth.data_ptr(),
                                             depth.shape[2], depth.shape[3], max_x, max_y, out.data_ptr(), stream_handle()), "okp_lift_peaks")
    return out


def group_objects(count, xyc, centers, type_count, max_obj=16, max_sel=4, max_dist=20.0):
    """Device-side object grouping (okp_group_objects).  count [N,K] int32, xyc [N,K,cap,3] fp32, centers
    [N,K-1,2,H,W] fp32 -> dict of device tensors: n_obj [N], sel [N,max_obj,K-1,max_sel], n_votes [N,max_obj,K-1],
    assign [N,K,cap] (int32), pred [N,K,cap,2] (fp64 predicted centres) and reduced [N,max_obj,K-1,max_sel,2] (fp32: the cluster centres of
    a multi-instance type that received more votes than it has instances - the reference's k-means branch, pipeline.py:143-148 - NaN elsewhere)."""
    require_cuda(xyc, "xyc")
    n, k, cap, _ = xyc.shape
    centers = centers.contiguous()
    if centers.dtype != torch.float32 or tuple(centers.shape[:3]) != (n, k - 1, 2):
        raise OkpError("centers must be float32 [N,K-1,2,H,W]")
    dev = xyc.device
    n_obj = torch.empty((n,), dtype=torch.int32, device=dev)
    sel = torch.empty((n, max_obj, k - 1, max_sel), dtype=torch.int32, device=dev)
    votes = torch.empty((n, max_obj, k - 1), dtype=torch.int32, device=dev)
    assign = torch.empty((n, k, cap), dtype=torch.int32, device=dev)
    pred = torch.empty((n, k, cap, 2), dtype=torch.float64, device=dev)      # (the kernel zeroes the unused slots)
    reduced = torch.empty((n, max_obj, k - 1, max_sel, 2), dtype=torch.float32, device=dev)     # (NaN where no reduction took place)
    T = _ops()
    if T is not None:
        _dispatch(T.group_objects, count, xyc, centers, [int(c) for c in type_count], float(max_dist), max_obj, max_sel, n_obj, sel, votes, assign, pred, stream_int(), reduced)
    else:
        tc = (ctypes.c_int32 * (k - 1))(*[int(c) for c in type_count])
        _lib.check(_lib.lib().okp_group_objects(count.data_ptr(), xyc.data_ptr(), centers.data_ptr(), n, k, cap, centers.shape[3], centers.shape[4],
                                                tc, float(max_dist), max_obj, max_sel, n_obj.data_ptr(), sel.data_ptr(), votes.data_ptr(),
                                                assign.data_ptr(), pred.data_ptr(), reduced.data_ptr(), stream_handle()), "okp_group_objects")
    return {"n_obj": n_obj, "sel": sel, "n_votes": votes, "assign": assign, "pred": pred, "reduced": reduced}


RANGE_OVERFLOW = 2       # bit of capacity_overflow()'s word: the split-product network's fp16-range flag was raised


def capacity_overflow(count, cap, max_obj, range_flag=None):
    """count [N,K] int32 (device) -> 0-d int32 tensor (device), non-zero when some map holds more than `cap` peaks or some frame more
    than `max_obj` centre peaks (bit 0), or - with `range_flag`, the fp16-range flag of the split-product network that made the maps -
    when that flag is raised (bit 1, RANGE_OVERFLOW).  `bool(flag)` on the host.  One small launch, no host sync, no torch kernel."""
    require_cuda(count, "count")
    flag = torch.empty((1,), dtype=torch.int32, device=count.device)
    n, k = count.shape
    T = _ops()
    if T is not None:
        _dispatch(T.capacity_overflow, count.contiguous(), k, cap, max_obj, flag, stream_int(), range_flag)
    else:
        _lib.check(_lib.lib().okp_capacity_overflow(count.data_ptr(), n * k, k, cap, max_obj, range_flag.data_ptr() if range_flag is not None else None,
                                                    flag.data_ptr(), stream_handle()), "okp_capacity_overflow")
    return flag[0]


def triangulate_dlt(cam_l, cam_r, T_RL, left_xy, right_xy, F=None):
    require_cuda(left_xy, "left_xy")
    left_xy = left_xy.to(torch.float32).contiguous()
    right_xy = right_xy.to(torch.float32).contiguous()
    if left_xy.shape != right_xy.shape:
        raise OkpError("left/right point counts differ")
    T = np.ascontiguousarray(np.asarray(T_RL, dtype=np.float64)[:3, :4])
    TO = _ops()
    if TO is not None:
        out = torch.empty((left_xy.shape[0], 3), dtype=torch.float64, device=left_xy.device)
        Fl = [] if F is None else [float(v) for v in np.asarray(F, dtype=np.float64).reshape(9)]
        _dispatch(TO.triangulate_dlt, _cam_list(cam_l), _cam_list(cam_r), [float(v) for v in T.reshape(12)], Fl, left_xy, right_xy, out, stream_int())
        return out
    Fp = None
    if F is not None:
        Fa = np.ascontiguousarray(np.asarray(F, dtype=np.float64).reshape(3, 3))
        Fp = Fa.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    out = torch.empty((left_xy.shape[0], 3), dtype=torch.float64, device=left_xy.device)
    _lib.check(_lib.lib().okp_triangulate_dlt(ctypes.byref(cam_l), ctypes.byref(cam_r), T.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), Fp,
                                              1 if F is not None else 0, left_xy.data_ptr(), right_xy.data_ptr(), left_xy.shape[0],
                                              out.data_ptr(), stream_handle()), "okp_triangulate_dlt")
    return out

"""CornerNet-Squeeze stacked hourglass on the HIP path.

Each class keeps the reference module's parameters under the reference's attribute names (the
`state_dict` wire format) and, in forward(), runs hand-written HIP kernels
through the C ABI on NHWC activations (`ops.Act`).  Eval mode only: BatchNorm running
statistics are folded into the convolution weights when a plan is first built.

Reference modules mirrored (file:line under the reference tree):
  convolution   perception/corner_net_lite/core/models/py_utils/utils.py:143-156
  residual      .../py_utils/utils.py:158-185
  fire_module   .../core/models/CornerNet_Squeeze.py:10-30 (+ factories :32-51)
  hg_module     .../py_utils/modules.py:25-66
  hg            .../py_utils/modules.py:68-93
"""
import numpy as np
import torch
from torch import nn

from .. import ops
from ..ops import Act, ConvPlan, OkpError, StemPlan


# ------------------------------------------------------------------------------------------
# folding helpers (host side, once per plan)
# ------------------------------------------------------------------------------------------

import os
# (module attributes, flipped by tests that compare the paths; not environment switches)
STEM_DIRECT = True       # 16-bit: the stem kernel reads fp32 NCHW frames itself (no pack launch)
STEM_KERNEL = True       # 16-bit: dedicated stem kernel (okp_stem.hip); False = generic tap-list kernel
STEM_X3_KERNEL = True    # split-product configuration on raw NCHW frames: the stem kernel's three-term form; False = pack + generic kernel
PAIR_TENSORS = True      # split-product configuration: a tensor whose only readers are 3x3 convolutions on the patch-resident kernel is written
                         # in pair format ([hi | lo] fp16 per 8 channels, okp_conv_args.out_pairs) by its producer, so that its consumers do
                         # not split every landed patch between their K-steps: conv1 -> conv2 of the residual blocks, pre[1] -> pre[2],
                         # hourglass -> cnvs.  Bit-identical results (the same split, done once per element in the producer's epilogue)
STEM_PAIRS = True        # ... and the split-product stem kernel writes pairs for pre[1] (okp_stem_forward_nchw_pairs)
BIG_VIEWS = True         # split-product configuration on raw frames: the stem kernel writes, and the patch-resident kernel reads, the stem's map of the WHOLE
                         # batch (2.1 GB at 64 frames: both address frame by frame, include/okp.h) - no frame chunks in front of pre[2];
                         # False = chunks under the 2 GiB view limit as for the other fp32-storage configurations (A/B)
UNPOOL_TILE = 0          # tile code of the transposed-conv launches (0 = heuristic)
SQUEEZE_TILE = 0         # tile code of the squeeze launch of a fire module that has no one-launch kernel (0 = heuristic)
STEM_TILE = 4            # 7x7/s2 stem on the generic kernel: 128 co x 256 px tile measured fastest (603 vs 728 us)


def _np(t):
    return t.detach().to("cpu", torch.float32).numpy()


def fold_bn(weight, bn, conv_bias=None):
    """(w * scale[co], bias) for eval-mode BatchNorm2d after a conv with out-channel dim 0."""
    w = _np(weight).astype(np.float64)
    cout = w.shape[0]
    if isinstance(bn, nn.BatchNorm2d):
        scale = _np(bn.weight).astype(np.float64) / np.sqrt(_np(bn.running_var).astype(np.float64) + bn.eps)
        shift = _np(bn.bias).astype(np.float64) - _np(bn.running_mean).astype(np.float64) * scale
    else:
        scale, shift = np.ones(cout), np.zeros(cout)
    b = shift if conv_bias is None else shift + _np(conv_bias).astype(np.float64) * scale
    w = w * scale.reshape((-1,) + (1,) * (w.ndim - 1))
    return w.astype(np.float32), b.astype(np.float32)


def conv_taps(w, src=0, pad=None):
    """OIHW folded weight -> tap list [(src, dy, dx, W[co, ci])] in row-major kernel order."""
    kh, kw = w.shape[2], w.shape[3]
    ph = (kh - 1) // 2 if pad is None else pad
    pw = (kw - 1) // 2 if pad is None else pad
    return [(src, r - ph, s - pw, np.ascontiguousarray(w[:, :, r, s])) for r in range(kh) for s in range(kw)]


def conv_out_size(n, k, stride, pad):
    return (n + 2 * pad - k) // stride + 1


class _HipModule(nn.Module):
    """Plan cache shared by the mirrored modules; plans are rebuilt when weights are reloaded."""

    def __init__(self):
        super().__init__()
        self._plans = {}
        self._plan_epoch = 0          # bumped whenever the cached plans are dropped (weights reloaded / moved): CapturedStep.stale()
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._drop_plans())

    def _drop_plans(self):
        self._plans.clear()
        self._plan_epoch += 1

    def _plan(self, key, build):
        if self.training:
            raise OkpError("the HIP path implements eval-mode inference only; call .eval()")
        key = (key, ops.F32_SPLIT, ops.F32_MIX)   # fp32 plans exist in three forms: exact fp32 MFMAs / split products (ops.F32X3) / mixed
        p = self._plans.get(key)
        if p is None:
            p = build()
            self._plans[key] = p
        return p

    def _apply(self, fn, *a, **k):          # .to()/.cuda() moves parameters: device constants must follow
        self._drop_plans()
        return super()._apply(fn, *a, **k)


class convolution(_HipModule):
    def __init__(self, k, inp_dim, out_dim, stride=1, with_bn=True):
        super().__init__()
        pad = (k - 1) // 2
        self.k, self.stride, self.pad, self.inp_dim, self.out_dim = k, stride, pad, inp_dim, out_dim
        self.conv = nn.Conv2d(inp_dim, out_dim, (k, k), padding=(pad, pad), stride=(stride, stride), bias=not with_bn)
        self.bn = nn.BatchNorm2d(out_dim) if with_bn else nn.Sequential()

    def _build(self, dtype, stride=None):
        """stride: conv stride of the generic stem plan (4 = every second output pixel and row of the 7x7/s2 stem)."""
        w, b = fold_bn(self.conv.weight, self.bn, self.conv.bias)
        if self.inp_dim == 3:
            if self.k != 7 or self.stride != 2:
                raise OkpError("3-channel input is supported for the 7x7/s2 stem only")
            if dtype in ops.HALF_DTYPES and self.out_dim == 128 and STEM_KERNEL and stride is None:
                return StemPlan(w, b, dtype)
            if stride == "direct":          # split-product stem kernel on fp32 NCHW frames (forward_frames)
                return StemPlan(w, b, dtype)
            # one tap per kernel row: 8 pixels x 4 channels of the packed frame = 32 contiguous elements
            taps = []
            for r in range(7):
                m = np.zeros((self.out_dim, 8, 4), dtype=np.float32)
                m[:, :7, :3] = np.transpose(w[:, :, r, :], (0, 2, 1))
                taps.append((0, r, 0, m.reshape(self.out_dim, 32)))
            return ConvPlan(dtype, [32], [stride or 2], self.out_dim, taps, b, relu=True, alg_k=147)
        return ConvPlan(dtype, [self.inp_dim], [self.stride], self.out_dim, conv_taps(w), b, relu=True)

    def takes_pairs(self, n, h, w, dtype):
        """This 3x3 convolution reads a pair-format input of n x h x w pixels (PAIR_TENSORS): a pure split-product plan whose launch runs
        on the patch-resident kernel."""
        if not (PAIR_TENSORS and dtype == torch.float32 and ops.F32_SPLIT and self.inp_dim != 3 and self.k > 1):     # (also the mixed configuration's `cnvs`)
            return False
        plan = self._plan(("p", dtype), lambda: self._build(dtype))
        return bool(getattr(plan, "split", False)) and plan.picks_patch(n, conv_out_size(h, self.k, self.stride, self.pad),
                                                                        conv_out_size(w, self.k, self.stride, self.pad), [self.inp_dim])

    def forward_frames(self, frames, dtype, shadow=False, compact=False, out_pairs=False):
        """The stem on raw fp32 NCHW frames.  bf16: one launch of the dedicated kernel reading the frames directly;
        otherwise pack (ops.pack_frames) + the generic path.  shadow (split-product plans): also write the fp16 copy.
        out_pairs (split-product stem kernel only; the caller checks Act.pairs): the result in pair format (PAIR_TENSORS)."""
        if self.inp_dim == 3 and dtype in ops.HALF_DTYPES and self.out_dim == 128 and STEM_KERNEL and STEM_DIRECT:
            plan = self._plan(("p", dtype), lambda: self._build(dtype))
            n, _, h, w = frames.shape
            out = Act.empty(n, conv_out_size(h, 7, 2, 3), conv_out_size(w, 7, 2, 3), self.out_dim, dtype, frames.device)
            plan.from_nchw(frames, out)
            return out
        if (dtype == torch.float32 and ops.F32_SPLIT and not shadow and self.inp_dim == 3 and self.out_dim == 128 and STEM_KERNEL and STEM_DIRECT
                and STEM_X3_KERNEL):
            # split-product configuration: the stem kernel's three-term form, fp32 NHWC out of fp32 NCHW in (no pack launch)
            plan = self._plan(("px3", dtype), lambda: self._build(dtype, stride="direct"))
            n, _, h, w = frames.shape
            out = Act.empty(n, conv_out_size(h, 7, 2, 3), conv_out_size(w, 7, 2, 3), self.out_dim, dtype, frames.device)
            plan.from_nchw(frames, out, out_pairs=bool(out_pairs and PAIR_TENSORS and not ops.F32_MIX))
            return out
        if (shadow and compact and ops.F32_MIX and ops.MIX_STEM_FP16 and dtype == torch.float32 and self.inp_dim == 3 and self.out_dim == 128
                and STEM_KERNEL and STEM_DIRECT):
            # mixed configuration, two launches (ops.MIX_STEM_FP16): full-grid fp16 tensor by the fp16 stem kernel, fp32 at even pixels by a
            # stride-4 three-term launch
            n, _, h, w = frames.shape
            ho, wo = conv_out_size(h, 7, 2, 3), conv_out_size(w, 7, 2, 3)
            p16 = self._plan(("p", torch.float16), lambda: self._build(torch.float16))
            sh = Act.empty(n, ho, wo, self.out_dim, torch.float16, frames.device)
            p16.from_nchw(frames, sh)
            p4 = self._plan(("s4", dtype), lambda: self._build(dtype, stride=4))
            out = Act.empty(n, (ho + 1) // 2, (wo + 1) // 2, self.out_dim, dtype, frames.device)
            p4([ops.pack_frames(frames, dtype)], out, out.h, out.w, tile=STEM_TILE)
            out.compact, out.shadow = True, sh
            return out
        return self.forward(ops.pack_frames(frames, dtype), shadow=shadow, compact=compact)

    def forward(self, x, shadow=False, compact=False, out_pairs=False):
        """shadow / compact (split-product plans): also write the fp16 copy; keep the fp32 result at even rows / columns only.
        out_pairs: the result in pair format if this launch runs on the patch-resident split kernel (the caller looks at Act.pairs)."""
        plan = self._plan(("p", x.dtype), lambda: self._build(x.dtype))
        if self.inp_dim == 3:       # x is the packed frame tensor [N, H+6, Wp, 4]
            if x.orig_hw is None:
                raise OkpError("stem input must come from ops.pack_frames")
            h, w = x.orig_hw
            ho, wo = conv_out_size(h, 7, 2, 3), conv_out_size(w, 7, 2, 3)
        else:
            ho, wo = conv_out_size(x.h, self.k, self.stride, self.pad), conv_out_size(x.w, self.k, self.stride, self.pad)
        out = Act.empty(x.n, ho, wo, self.out_dim, x.dtype, x.t.device) if not (compact and shadow and getattr(plan, "split", False)) else None
        if isinstance(plan, StemPlan):
            plan(x, out)
        else:
            sub = 1
            if shadow and plan.split:
                sh = Act.empty(x.n, ho, wo, self.out_dim, torch.float16, x.t.device)
                if compact:
                    out = Act.empty(x.n, (ho + 1) // 2, (wo + 1) // 2, self.out_dim, x.dtype, x.t.device)
                    out.compact, sub = True, 2
                out.shadow = sh
            pairs = bool(out_pairs and self.takes_pairs(x.n, x.h, x.w, x.dtype) and not (shadow and plan.split))
            plan([x], out, ho, wo, tile=STEM_TILE if self.inp_dim == 3 else 0, out16=out.shadow, out_subsample=sub, out_pairs=pairs)
        return out


class residual(_HipModule):
    def __init__(self, inp_dim, out_dim, k=3, stride=1):
        super().__init__()
        p = (k - 1) // 2
        self.k, self.stride, self.pad, self.inp_dim, self.out_dim = k, stride, p, inp_dim, out_dim
        self.conv1 = nn.Conv2d(inp_dim, out_dim, (k, k), padding=(p, p), stride=(stride, stride), bias=False)
        self.bn1 = nn.BatchNorm2d(out_dim)
        self.conv2 = nn.Conv2d(out_dim, out_dim, (k, k), padding=(p, p), bias=False)
        self.bn2 = nn.BatchNorm2d(out_dim)
        self.projected = stride != 1 or inp_dim != out_dim
        self.skip = nn.Sequential(nn.Conv2d(inp_dim, out_dim, (1, 1), stride=(stride, stride), bias=False),
                                  nn.BatchNorm2d(out_dim)) if self.projected else nn.Sequential()

    def _build(self, dtype):
        # mixed configuration (ops.F32MIX): the two 3x3 convolutions of the branch multiply with ONE fp16 term, the projected skip
        # (the path the stream takes) keeps three - see ops.F32MIX for the error budget behind this
        single = ops.F32_MIX and ops.MIX_BRANCH_SINGLE and dtype == torch.float32
        w1, b1 = fold_bn(self.conv1.weight, self.bn1)
        t1 = conv_taps(w1)
        p1 = ConvPlan(dtype, [self.inp_dim], [self.stride], self.out_dim, t1, b1, relu=True, tap_terms=[1] * len(t1) if single else None)
        w2, b2 = fold_bn(self.conv2.weight, self.bn2)
        taps = conv_taps(w2)
        terms = [1] * len(taps) if single else None
        if self.projected:
            # conv2 + projected skip + add + relu as ONE GEMM: the 1x1 skip is a tenth tap on a second source
            ws, bs = fold_bn(self.skip[0].weight, self.skip[1])
            taps = taps + [(1, 0, 0, np.ascontiguousarray(ws[:, :, 0, 0]))]
            p2 = ConvPlan(dtype, [self.out_dim, self.inp_dim], [1, self.stride], self.out_dim, taps, b2 + bs, relu=True,
                          tap_terms=terms + [3] if single else None)
        else:
            p2 = ConvPlan(dtype, [self.out_dim], [1], self.out_dim, taps, b2, relu=True, tap_terms=terms)
        return p1, p2

    def _build_mixed16(self, compact_in=False):
        """Mixed configuration with the branch on the fp16 kernels (ops.MIX_BRANCH_FP16): conv1 and conv2 as fp16 plans (conv2 without
        the closing ReLU: its rounded output is the residual of what follows), the projected skip as a three-term plan of its own."""
        w1, b1 = fold_bn(self.conv1.weight, self.bn1)
        p1 = ConvPlan(torch.float16, [self.inp_dim], [self.stride], self.out_dim, conv_taps(w1), b1, relu=True)
        w2, b2 = fold_bn(self.conv2.weight, self.bn2)
        p2 = ConvPlan(torch.float16, [self.out_dim], [1], self.out_dim, conv_taps(w2), b2, relu=False)
        ps = None
        if self.projected:
            ws, bs = fold_bn(self.skip[0].weight, self.skip[1])
            # (a compact input already holds exactly the pixels a stride-2 skip samples: read with stride 1)
            ps = ConvPlan(torch.float32, [self.inp_dim], [1 if compact_in else self.stride], self.out_dim, [(0, 0, 0, np.ascontiguousarray(ws[:, :, 0, 0]))], bs, relu=True)
            ps.allow_compact = compact_in
        return p1, p2, ps

    def _forward_mixed16(self, x, shadow, out=None, out_shadow=None, compact=False):
        if x.compact and not (self.projected and self.stride == 2 and x.shadow is not None):
            raise OkpError("residual: a compact input needs its fp16 copy and a stride-2 projected block")
        p1, p2, ps = self._plan(("mix16", x.compact), lambda: self._build_mixed16(x.compact))
        x16 = x.shadow if x.shadow is not None else ops.cast(x, torch.float16)      # (the producer wrote the copy when it knew of this consumer)
        ho, wo = conv_out_size(x16.h, self.k, self.stride, self.pad), conv_out_size(x16.w, self.k, self.stride, self.pad)
        dev = x.t.device
        t16 = Act.empty(x.n, ho, wo, self.out_dim, torch.float16, dev)
        p1([x16], t16, ho, wo)
        b16 = Act.empty(x.n, ho, wo, self.out_dim, torch.float16, dev)
        p2([t16], b16, ho, wo)
        if ps is not None:
            compact = compact and shadow
            if out is None:
                out = Act.empty(x.n, (ho + 1) // 2 if compact else ho, (wo + 1) // 2 if compact else wo, self.out_dim, torch.float32, dev)
            out.compact = compact
            if shadow:
                out.shadow = out_shadow if out_shadow is not None else Act.empty(x.n, ho, wo, self.out_dim, torch.float16, dev)
            # relu(skip(x) + branch): three terms on the stream, fp16 residual
            ps([x], out, ho, wo, res=b16, out16=out.shadow, out_subsample=2 if compact else 1)
        else:
            if out is not None:
                raise OkpError("residual: a caller-provided destination is supported for projected blocks only")
            out = ops.add_f16_f32(b16, x, relu=True)
            if shadow:
                out.shadow = ops.cast(out, torch.float16)
        return out

    def on_patch_kernel(self, n, h, w, dtype, in_pix_stride=None):
        """(conv1, conv2) of an input of n x h x w pixels run on the patch-resident split-product kernel, i.e. may exchange pair-format
        tensors (PAIR_TENSORS): pure split-product plans only - the mixed configuration keeps its fp16 side outputs."""
        if not (PAIR_TENSORS and dtype == torch.float32 and ops.F32_SPLIT and not ops.F32_MIX):
            return False, False
        return self.launches_on_patch(n, h, w, dtype, in_pix_stride)

    def launches_on_patch(self, n, h, w, dtype, in_pix_stride=None):
        """(conv1, conv2) run on the patch-resident split-product kernel for this input (whatever the tensor format)."""
        if not (dtype == torch.float32 and ops.F32_SPLIT and not ops.F32_MIX):
            return False, False
        p1, p2 = self._plan(("p", dtype), lambda: self._build(dtype))
        if not p1.split:
            return False, False
        ho, wo = conv_out_size(h, self.k, self.stride, self.pad), conv_out_size(w, self.k, self.stride, self.pad)
        ps = in_pix_stride or self.inp_dim
        return (p1.picks_patch(n, ho, wo, [ps]),
                p2.picks_patch(n, ho, wo, [self.out_dim, ps] if self.projected else [self.out_dim]))

    def takes_pairs(self, n, h, w, dtype):
        """The block reads a pair-format input: conv1 and the projected skip inside conv2's launch both run on the patch-resident kernel
        (an identity skip adds the input in fp32: such a block takes fp32 tensors)."""
        return self.projected and all(self.on_patch_kernel(n, h, w, dtype))

    def forward(self, x, shadow=False, out=None, out_shadow=None, compact=False, out_pairs=False):
        """out / out_shadow: optional destination (an Act over a frame range of a larger tensor) of the block's result and of its fp16
        copy - hg.forward runs the two high-resolution layers in frame chunks and the rest of the network in one pass.
        compact (mixed configuration, with shadow): the fp32 result is kept at even rows / columns only (its one fp32 reader is the
        next block's stride-2 skip).  out_pairs: write the result in pair format if conv2 runs on the patch-resident kernel (the
        caller has checked that every reader takes it: residual.takes_pairs); the returned Act's `pairs` says whether it was."""
        if ops.F32_MIX and ops.MIX_BRANCH_SINGLE and ops.MIX_BRANCH_FP16 and x.dtype == torch.float32:
            return self._forward_mixed16(x, shadow, out, out_shadow, compact)
        p1, p2 = self._plan(("p", x.dtype), lambda: self._build(x.dtype))
        ho, wo = conv_out_size(x.h, self.k, self.stride, self.pad), conv_out_size(x.w, self.k, self.stride, self.pad)
        on1, on2 = self.on_patch_kernel(x.n, x.h, x.w, x.dtype, x.t.shape[3])
        t = Act.empty(x.n, ho, wo, self.out_dim, x.dtype, x.t.device)
        p1([x], t, ho, wo, out_pairs=on1 and on2)       # conv1 -> conv2: one writer, one reader
        if out is None:
            out = Act.empty(x.n, ho, wo, self.out_dim, x.dtype, x.t.device)
        if self.projected:
            p2([t, x], out, ho, wo, out_pairs=out_pairs and on2)
        else:
            p2([t], out, ho, wo, res=x, out_pairs=out_pairs and on2)
        return out


class fire_module(_HipModule):
    def __init__(self, inp_dim, out_dim, sr=2, stride=1):
        super().__init__()
        self.inp_dim, self.out_dim, self.mid, self.stride = inp_dim, out_dim, out_dim // sr, stride
        self.conv1 = nn.Conv2d(inp_dim, out_dim // sr, kernel_size=1, stride=1, bias=False)
        self.bn1 = nn.BatchNorm2d(out_dim // sr)
        self.conv_1x1 = nn.Conv2d(out_dim // sr, out_dim // 2, kernel_size=1, stride=stride, bias=False)
        self.conv_3x3 = nn.Conv2d(out_dim // sr, out_dim // 2, kernel_size=3, padding=1, stride=stride,
                                  groups=out_dim // sr, bias=False)
        self.bn2 = nn.BatchNorm2d(out_dim)
        self.skip = (stride == 1 and inp_dim == out_dim)
        if self.mid != out_dim // 2:
            raise OkpError("fire_module: only sr=2 (depth multiplier 1) is implemented")

    def _build(self, dtype, device):
        half = self.out_dim // 2
        w1, b1 = fold_bn(self.conv1.weight, self.bn1)
        squeeze = ConvPlan(dtype, [self.inp_dim], [1], self.mid, conv_taps(w1), b1, relu=False)
        # bn2 acts per channel on the concatenation: fold its two halves into the two branches
        scale2 = _np(self.bn2.weight).astype(np.float64) / np.sqrt(_np(self.bn2.running_var).astype(np.float64) + self.bn2.eps)
        shift2 = _np(self.bn2.bias).astype(np.float64) - _np(self.bn2.running_mean).astype(np.float64) * scale2
        wa = (_np(self.conv_1x1.weight).astype(np.float64) * scale2[:half, None, None, None]).astype(np.float32)
        expand = ConvPlan(dtype, [self.mid], [self.stride], half, conv_taps(wa), shift2[:half].astype(np.float32), relu=True)
        wd = _np(self.conv_3x3.weight).astype(np.float64)[:, 0] * scale2[half:, None, None]      # [C,3,3]
        wd = np.ascontiguousarray(np.transpose(wd, (1, 2, 0)).reshape(9, half)).astype(np.float32)   # tap-major [9][C]
        return (squeeze, expand, torch.from_numpy(wd).to(device), torch.from_numpy(shift2[half:].astype(np.float32)).to(device))

    def forward(self, x):
        squeeze, expand, wd, bd = self._plan(("p", x.dtype), lambda: self._build(x.dtype, x.t.device))
        half = self.out_dim // 2
        ho, wo = conv_out_size(x.h, 3, self.stride, 1), conv_out_size(x.w, 3, self.stride, 1)
        if (x.dtype in ops.HALF_DTYPES and ops.fire_fusable(self.inp_dim, self.mid, self.stride, x.h, x.w)) or \
                (squeeze.split and expand.split and ops.fire_fusable_x3(self.inp_dim, self.mid, self.stride, self.skip, x.h, x.w)):
            out = Act.empty(x.n, ho, wo, self.out_dim, x.dtype, x.t.device)
            ops.fire_fused(squeeze, expand, wd, bd, x, out, self.stride, self.skip)
            return out
        s = Act.empty(x.n, x.h, x.w, self.mid, x.dtype, x.t.device)
        squeeze([x], s, x.h, x.w, tile=SQUEEZE_TILE)
        ho, wo = conv_out_size(x.h, 3, self.stride, 1), conv_out_size(x.w, 3, self.stride, 1)
        out = Act.empty(x.n, ho, wo, self.out_dim, x.dtype, x.t.device)
        # concat is free: both branches write their channel window of the same NHWC tensor, and both run in ONE
        # launch (expand GEMM + depth-wise taps of the same squeeze tile)
        expand([s], out.slice(0, half), ho, wo, res=x.slice(0, half) if self.skip else None,
               dw=(wd, bd, out.slice(half, half), x.slice(half, half) if self.skip else None))
        return out


def make_pool_layer(dim):
    return nn.Sequential()


class unpool_merge(_HipModule):
    """ConvTranspose2d(dim, dim, 4, 2, 1) with bias, fused with the hourglass merge add.

    Holds the transposed-conv parameters as `weight` / `bias` so that, assigned to `hg_module.up2`,
    the state_dict keys are the reference's `...up2.weight` / `...up2.bias`
    (CornerNet_Squeeze.py:35-36; merge: py_utils/utils.py:139-141, modules.py:64-65).
    Lowered to four sub-pixel 2x2 convolutions (one per output parity) that run as the four classes of ONE
    launch, each writing its quarter of the output with the `up1` tensor added in the epilogue.
    """

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        ref = nn.ConvTranspose2d(dim, dim, kernel_size=4, stride=2, padding=1)
        self.weight = ref.weight          # (Cin, Cout, 4, 4)
        self.bias = ref.bias

    def _build(self, dtype):
        w = _np(self.weight)              # [ci, co, ky, kx]
        b = _np(self.bias)
        # output row 2i+a gathers input rows i+dy with kernel row ky:  a=0: (0,1), (-1,3);  a=1: (1,0), (0,2)
        sel = {0: [(0, 1), (-1, 3)], 1: [(1, 0), (0, 2)]}
        # ONE plan, 16 taps = four groups of four; group k = 2a + bb is the sub-pixel class written at (a, bb)
        taps = [(0, dy, dx, np.ascontiguousarray(w[:, :, ky, kx].T))
                for a in (0, 1) for bb in (0, 1) for dy, ky in sel[a] for dx, kx in sel[bb]]
        return ConvPlan(dtype, [self.dim], [1], self.dim, taps, b, relu=False)

    def forward(self, low, up1, out_pairs=False):
        """out_pairs (split-product plans, PAIR_TENSORS): the merged map in pair format if this launch runs on the patch-resident kernel."""
        plan = self._plan(("p", low.dtype), lambda: self._build(low.dtype))
        if up1.h != 2 * low.h or up1.w != 2 * low.w:
            raise OkpError("unpool_merge: up1 must be twice the size of low")
        out = Act.empty(low.n, 2 * low.h, 2 * low.w, self.dim, low.dtype, low.t.device)
        out_pairs = bool(out_pairs and PAIR_TENSORS and plan.split
                         and plan.picks_patch(low.n, low.h, low.w, [low.t.shape[3]], out_step=2, n_classes=4, tile=UNPOOL_TILE))
        plan([low], out, low.h, low.w, res=up1, out_step=2, n_classes=4, tile=UNPOOL_TILE, out_pairs=out_pairs)     # four output parities, one launch
        return out


def make_unpool_layer(dim):
    return unpool_merge(dim)


def run_fire_modules(mods, x):
    """Apply consecutive fire modules.  Runs of stride-1 fire(512, 512) modules on maps of at most 4 x 4 pixels (the
    innermost hourglass level: six in a row) and of fire(384, 384) modules on maps of at most 8 x 8 pixels go through
    ONE resident launch (ops.fire_chain / okp_fire_chain_forward)."""
    mods = list(mods)

    def chainable(m, x):
        if not (ops.FUSE_FIRE_CHAIN and x.dtype in ops.HALF_DTYPES and m.stride == 1 and m.skip and m.inp_dim == m.out_dim):
            return False
        return (m.inp_dim == 512 and x.h <= 4 and x.w <= 4) or (m.inp_dim == 384 and x.h <= 8 and x.w <= 8)

    def framed_chain(i, x):
        """[stride-2 fire(384, 512) on a map of at most 8 x 8] + fire(512, 512) x k + [fire(512, 384)]: the whole innermost hourglass
        level (low1, low2, low3 of hg_module n = 1) is ONE launch (the entry / exit form of okp_fire_chain_forward)."""
        m0 = mods[i]
        if not (ops.FUSE_FIRE_CHAIN and ops.FUSE_FIRE_CHAIN_FRAMED and x.dtype in ops.HALF_DTYPES and m0.stride == 2 and m0.inp_dim == 384
                and m0.out_dim == 512 and x.h <= 8 and x.w <= 8):
            return 0
        j = i + 1
        while j < len(mods) and mods[j].stride == 1 and mods[j].skip and mods[j].inp_dim == 512 and mods[j].out_dim == 512:
            j += 1
        if j - i - 1 < 1 or j >= len(mods) or j - i + 1 > ops.FIRE_CHAIN_MAX:
            return 0
        mx = mods[j]
        if not (mx.stride == 1 and mx.inp_dim == 512 and mx.out_dim == 384):
            return 0
        return j - i + 1

    i = 0
    while i < len(mods):
        k = framed_chain(i, x)
        if k:
            plans = [m._plan(("p", x.dtype), lambda m=m: m._build(x.dtype, x.t.device)) for m in mods[i:i + k]]
            out = Act.empty(x.n, (x.h - 1) // 2 + 1, (x.w - 1) // 2 + 1, 384, x.dtype, x.t.device)
            ops.fire_chain(plans, x, out)
            x = out
            i += k
            continue
        j = i
        while j < len(mods) and j - i < ops.FIRE_CHAIN_MAX and chainable(mods[j], x) and mods[j].inp_dim == mods[i].inp_dim:
            j += 1
        if j - i >= ops.FIRE_CHAIN_MIN:
            plans = [m._plan(("p", x.dtype), lambda m=m: m._build(x.dtype, x.t.device)) for m in mods[i:j]]
            out = Act.empty(x.n, x.h, x.w, mods[i].out_dim, x.dtype, x.t.device)
            ops.fire_chain(plans, x, out)
            x = out
            i = j
        else:
            x = mods[i](x)
            i += 1
    return x


class _FireSeq(nn.Sequential):
    def forward(self, x):
        return run_fire_modules(self, x)


def make_layer(inp_dim, out_dim, modules):
    return _FireSeq(*([fire_module(inp_dim, out_dim)] + [fire_module(out_dim, out_dim) for _ in range(1, modules)]))


def make_layer_revr(inp_dim, out_dim, modules):
    return _FireSeq(*([fire_module(inp_dim, inp_dim) for _ in range(modules - 1)] + [fire_module(inp_dim, out_dim)]))


def make_hg_layer(inp_dim, out_dim, modules):
    return _FireSeq(*([fire_module(inp_dim, out_dim, stride=2)] + [fire_module(out_dim, out_dim) for _ in range(1, modules)]))


class hg_module(nn.Module):
    def __init__(self, n, dims, modules):
        super().__init__()
        curr_dim, next_dim = dims[0], dims[1]
        self.n = n
        self.up1 = make_layer(curr_dim, curr_dim, modules[0])
        self.max1 = make_pool_layer(curr_dim)
        self.low1 = make_hg_layer(curr_dim, next_dim, modules[0])
        self.low2 = hg_module(n - 1, dims[1:], modules[1:]) if n > 1 else make_layer(next_dim, next_dim, modules[1])
        self.low3 = make_layer_revr(next_dim, curr_dim, modules[0])
        self.up2 = make_unpool_layer(curr_dim)

    def forward(self, x, out_pairs=False):
        """up1(x) and the whole low path are independent until the merge.  The low path is a long chain of small,
        latency-bound launches that leave most of the 256 CUs idle, so up1 runs on a side HIP stream and fills
        them (ops.SIDE_STREAMS; the streams fork/join with events, which also captures cleanly into a hipGraph).
        out_pairs: the caller's only reader of the result takes pair format (unpool_merge.forward, PAIR_TENSORS)."""
        if ops.F32_MIX and x.dtype == torch.float32 and self.n <= ops.MIX_FP16_LEVELS:
            # mixed configuration: this level and everything below it run in fp16 on the fused fp16 kernels (ops.F32MIX)
            return ops.cast(self.forward(ops.cast(x, torch.float16)), torch.float32)
        if not ops.SIDE_STREAMS or (x.n < ops.SIDE_MIN_BATCH and not torch.cuda.is_current_stream_capturing()):
            for held in _HELD_BRANCH_OUTPUTS.values():      # (no fork in this pass: nothing of an earlier pass needs to stay pinned)
                held.clear()
            up1 = self.up1(x)
            low3 = self._low_path(x)                   # max1 is the identity (CornerNet_Squeeze.py:32-33)
            return self.up2(low3, up1, out_pairs=out_pairs)
        main = torch.cuda.current_stream()
        skey, side = self._side_stream(x.t.device)
        ops.stream_wait(side, main)                     # x is ready on the side stream
        # Tensor lifetimes across the two streams WITHOUT Tensor.record_stream (the caching allocator answers a recorded use with an
        # event record on the using stream when the tensor dies - a marker packet in the MAIN queue right behind every merge, 6-8 us of
        # delay for the kernel after it, eight times per step):
        #  * x (main stream's pool) is read by the branch: it is this call's argument, alive until the call returns, and the join
        #    below - in a finally clause: also when the low path raises - puts every branch kernel before anything the main stream
        #    does afterwards;
        #  * up1 (side stream's pool) is read by the merge on the main stream: it is kept alive until this side stream next waits for
        #    the main stream (the next fork that uses it), from where on every side-stream kernel comes after that merge.
        # (all bookkeeping is keyed by the _SIDE_STREAMS key - device, MAIN stream, level group - not by the side stream's raw handle:
        #  torch hands out pooled handles, so two entries may share one)
        held = _HELD_BRANCH_OUTPUTS.setdefault(skey, [])
        held.clear()                                    # (the wait above orders the side stream behind the merges that read these)
        joined = False
        try:
            with torch.cuda.stream(side):
                up1 = self.up1(x)
            # A join covers every branch enqueued on that side stream before it: levels that share a side stream are nested, the inner
            # level's branch is enqueued later and joined first, so the outer level finds its branch already joined and skips the
            # barrier packet (6 us on the main queue).
            seq = _SIDE_SEQ[skey] = _SIDE_SEQ.get(skey, 0) + 1
            low3 = self._low_path(x)
            if _JOINED_SEQ.get(skey, 0) < seq:
                ops.stream_wait(main, side)             # join before the merge
                _JOINED_SEQ[skey] = _SIDE_SEQ.get(skey, seq)
            joined = True
            out = self.up2(low3, up1, out_pairs=out_pairs)
            held.append(up1.t)
            return out
        finally:
            if not joined:                              # a launch failed between fork and join: x must outlive the branch kernels
                ops.stream_wait(main, side)
                _JOINED_SEQ[skey] = _SIDE_SEQ.get(skey, 0)

    def _low_path(self, x):
        if isinstance(self.low2, _FireSeq):             # innermost level: low1, low2, low3 are one list of fire modules
            return run_fire_modules(list(self.low1) + list(self.low2) + list(self.low3), x)
        return self.low3(self.low2(self.low1(x)))

    def _side_stream(self, device):
        """Three side streams per device, shared by every hourglass: the 64 x 64 level, the 32 x 32 level, and one for the two levels
        below.  A stream of its own per hg_module (eight of them, HIP deals streams round-robin onto its four hardware queues) put
        one of the side branches of the second hourglass on the MAIN stream's queue, i.e. back into the critical chain (step +0.7 %
        with the shared three; GPU_MAX_HW_QUEUES above its default of 4 costs 30 %, below it 1-5 %)."""
        # keyed by the MAIN stream as well: a pass that runs (or is being captured) on another stream - another host thread, a
        # second pipeline's hipGraph capture - gets its own side streams instead of recording into a foreign capture
        group = _SIDE_STREAM_OF_LEVEL[4 - self.n] if 1 <= self.n <= 4 else 0
        main = torch.cuda.current_stream(device).cuda_stream
        key = (device.type, device.index, main, group)
        st = _SIDE_STREAMS.get(key)
        if st is None:
            # main streams come and go (host threads, captures): oldest entries first - but never an entry of the CURRENT main stream (an
            # outer level of this very pass may be between its fork and its join on it), and nothing at all while a capture is active
            # (Stream.synchronize is illegal there, and a pass on another thread may be mid-flight on any entry): the table then simply grows
            if len(_SIDE_STREAMS) >= _SIDE_STREAMS_MAX and not torch.cuda.is_current_stream_capturing():
                for old in [k for k in _SIDE_STREAMS if k[2] != main][:len(_SIDE_STREAMS) - _SIDE_STREAMS_MAX + 1]:
                    gone = _SIDE_STREAMS.pop(old)
                    gone.synchronize()
                    _HELD_BRANCH_OUTPUTS.pop(old, None)
                    _SIDE_SEQ.pop(old, None)
                    _JOINED_SEQ.pop(old, None)
            st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
        return key, st


_SIDE_STREAMS = {}
_SIDE_STREAMS_MAX = 48           # (main stream, level group) entries kept: 16 main streams' worth
_SIDE_STREAM_OF_LEVEL = (0, 1, 2, 2)   # side stream of the 64x64, 32x32, 16x16, 8x8 level (every other assignment swept: within noise, DESIGN App. A)
_SIDE_SEQ, _JOINED_SEQ = {}, {}  # per _SIDE_STREAMS key: branches enqueued on that side stream; newest branch a join of its main stream has covered
_HELD_BRANCH_OUTPUTS = {}        # _SIDE_STREAMS key -> branch outputs the main stream may still be reading (hg_module.forward)


class _MergeMod(nn.Sequential):
    def __init__(self):
        super().__init__(nn.Conv2d(256, 256, (1, 1), bias=False), nn.BatchNorm2d(256))


class hg(_HipModule):
    DIMS = [256, 256, 384, 384, 512]
    MODULES = [2, 2, 2, 2, 4]

    def __init__(self, stacks=2):
        super().__init__()
        self.pre = nn.Sequential(convolution(7, 3, 128, stride=2), residual(128, 256, stride=2), residual(256, 256, stride=2))
        self.hgs = nn.ModuleList([hg_module(4, self.DIMS, self.MODULES) for _ in range(stacks)])
        self.cnvs = nn.ModuleList([convolution(3, 256, 256) for _ in range(stacks)])
        self.inters = nn.ModuleList([residual(256, 256) for _ in range(stacks - 1)])
        self.inters_ = nn.ModuleList([_MergeMod() for _ in range(stacks - 1)])
        self.cnvs_ = nn.ModuleList([_MergeMod() for _ in range(stacks - 1)])

    @staticmethod
    def front_chunk(n, h, w, dtype, compact=False):
        """Frames per launch of the stem and pre[1] such that the stem output stays under the 2 GiB view limit (include/okp.h), in
        equal chunks.  compact: its fp32 form holds a quarter of the pixels, the full-grid fp16 copy is the largest tensor."""
        esz = 2 if (dtype in ops.HALF_DTYPES or compact) else 4
        per_frame = conv_out_size(h, 7, 2, 3) * conv_out_size(w, 7, 2, 3) * 128 * esz
        limit = max(1, (0x7FFF0000 - 1) // per_frame)
        parts = -(-n // limit)
        return -(-n // parts)

    def _build_merge(self, i, dtype):
        # relu(inters_[i](inter) + cnvs_[i](cnv)): two 1x1+BN summed = one GEMM over two sources
        wa, ba = fold_bn(self.inters_[i][0].weight, self.inters_[i][1])
        wb, bb = fold_bn(self.cnvs_[i][0].weight, self.cnvs_[i][1])
        taps = [(0, 0, 0, np.ascontiguousarray(wa[:, :, 0, 0])), (1, 0, 0, np.ascontiguousarray(wb[:, :, 0, 0]))]
        return ConvPlan(dtype, [256, 256], [1, 1], 256, taps, ba + bb, relu=True)

    def forward(self, x, dtype=None, cnv_pairs=False):
        """x: packed frames (ops.pack_frames), or raw fp32 NCHW frames with the compute dtype given (the stem then reads
        them itself where it can).  Returns [cnv0, cnv1] as NHWC activations.  cnv_pairs (split-product configuration): the caller reads
        the LAST stack's cnv in pair format if it comes that way (Act.pairs; models.KeypointNet: the fused heads)."""
        # mixed configuration with fp16 branches: producers write the fp16 copy their consumer's conv1 reads (stem -> pre.1, pre.1 -> pre.2)
        mix16 = ops.F32_MIX and ops.MIX_BRANCH_SINGLE and ops.MIX_BRANCH_FP16
        # ... and keep the fp32 stream at even pixels only where its single fp32 reader is the next block's stride-2 skip
        cp = [mix16 and ops.MIX_COMPACT and r.projected and r.stride == 2 for r in (self.pre[1], self.pre[2])]
        if isinstance(x, torch.Tensor):
            n, (fh, fw), sdtype = x.shape[0], x.shape[2:4], dtype
            stem = lambda c0, c1, sh: self.pre[0].forward_frames(x[c0:c1], dtype, shadow=sh, compact=sh and cp[0], out_pairs=stem_pairs)
        else:                                   # packed frames (ops.pack_frames / pack_frames_u8 / preprocess_u8)
            n, (fh, fw), sdtype = x.n, x.orig_hw, x.dtype

            def stem(c0, c1, sh):
                xi = Act(x.t[c0:c1])
                xi.orig_hw = x.orig_hw
                return self.pre[0](xi, shadow=sh, compact=sh and cp[0])
        want_shadow = mix16 and sdtype == torch.float32
        chunk = self.front_chunk(n, fh, fw, sdtype, compact=want_shadow and cp[0])
        if (chunk < n and BIG_VIEWS and isinstance(x, torch.Tensor) and sdtype == torch.float32 and ops.F32_SPLIT and not ops.F32_MIX and STEM_X3_KERNEL
                and STEM_KERNEL and STEM_DIRECT and self.pre[0].inp_dim == 3 and self.pre[0].out_dim == 128
                and all(self.pre[1].launches_on_patch(n, conv_out_size(fh, 7, 2, 3), conv_out_size(fw, 7, 2, 3), sdtype))):
            chunk = n       # the split-product stem kernel and the patch-resident kernel address frame by frame: the whole batch in one launch each
        # pair format between the stem and pre[1] (both of its launches read the stem's map): every frame chunk's launches on the patch kernel
        stem_pairs = STEM_PAIRS and all(self.pre[1].takes_pairs(min(n, q + chunk) - q, conv_out_size(fh, 7, 2, 3), conv_out_size(fw, 7, 2, 3), sdtype)
                         for q in range(0, n, chunk))
        if chunk < n:
            # fp32 tensors: the stem output (128 channels at half resolution, 33.5 MB per frame) of a whole batch would pass the 2 GiB
            # view limit of the 32-bit buffer offsets.  Only the stem and pre[1] see that resolution: they run per frame chunk and
            # pre[1] writes straight into its range of the full batch's tensor; everything behind it runs in ONE pass (the chains of
            # small launches in the hourglasses cost the same for 32 frames as for 64).
            inter = None
            pairs1 = None
            for c0 in range(0, n, chunk):
                c1 = min(n, c0 + chunk)
                a = stem(c0, c1, want_shadow)
                if inter is None:
                    r = self.pre[1]
                    ah, aw = (a.shadow.h, a.shadow.w) if a.compact else (a.h, a.w)
                    ho, wo = conv_out_size(ah, r.k, r.stride, r.pad), conv_out_size(aw, r.k, r.stride, r.pad)
                    c2 = want_shadow and cp[1]
                    inter = Act.empty(n, (ho + 1) // 2 if c2 else ho, (wo + 1) // 2 if c2 else wo, r.out_dim, sdtype, a.t.device)
                    inter.compact = c2
                    if want_shadow:
                        inter.shadow = Act.empty(n, ho, wo, r.out_dim, torch.float16, a.t.device)
                    # pair format between pre[1] and pre[2]: every chunk's launch and both of pre[2]'s must be on the patch-resident kernel
                    pairs1 = (self.pre[2].takes_pairs(n, ho, wo, sdtype)
                              and all(r.on_patch_kernel(min(n, q + chunk) - q, ah, aw, sdtype, a.t.shape[3])[1] for q in range(0, n, chunk)))
                sh = Act(inter.shadow.t[c0:c1]) if inter.shadow is not None else None
                part = self.pre[1](a, shadow=sh is not None, out=Act(inter.t[c0:c1]), out_shadow=sh, compact=inter.compact, out_pairs=pairs1)
                if part.pairs != bool(pairs1):
                    raise OkpError("hg.forward: a frame chunk of pre[1] did not keep the tensor format of the others")
            inter.pairs = bool(pairs1)
        else:
            a = stem(0, n, want_shadow)
            r = self.pre[1]
            ah, aw = (a.shadow.h, a.shadow.w) if a.compact else (a.h, a.w)
            pairs1 = self.pre[2].takes_pairs(n, conv_out_size(ah, r.k, r.stride, r.pad), conv_out_size(aw, r.k, r.stride, r.pad), sdtype)
            inter = r(a, shadow=mix16, compact=want_shadow and cp[1], out_pairs=pairs1)
        inter = self.pre[2](inter)
        cnvs = []
        last = len(self.hgs) - 1
        for i, (hg_, cnv_) in enumerate(zip(self.hgs, self.cnvs)):
            # (the hourglass's merged map has one reader, cnvs[i]: pair format where that 3x3 takes it)
            cnv = cnv_(hg_(inter, out_pairs=cnv_.takes_pairs(inter.n, inter.h, inter.w, inter.dtype)), out_pairs=cnv_pairs and i == last)
            cnvs.append(cnv)
            if i < last:
                merge = self._plan(("m", i, inter.dtype), lambda: self._build_merge(i, inter.dtype))
                merged = Act.empty(inter.n, inter.h, inter.w, 256, inter.dtype, inter.t.device)
                if mix16 and merge.split:
                    merged.shadow = Act.empty(inter.n, inter.h, inter.w, 256, torch.float16, inter.t.device)
                merge([inter, cnv], merged, inter.h, inter.w, out16=merged.shadow)
                inter = self.inters[i](merged)
        return cnvs

"""Drop-in for the reference's `perception/models.py` on the HIP path.

Public surface (reference perception/models.py:13-85): prediction_module, HeatmapHead, DepthHead,
CenterHead, nms, KeypointNet — same constructor arguments, same `state_dict` keys (906 entries for
features=128), same output nesting `((hm1, hm2), (d1, d2), (c1, c2))` with NCHW fp32 tensors.
The deployed wrapper of scripts/package_model.py:21-28 is `KeypointNet.deployed(frames)`.

Device work is done by libokp_hip.so; inputs must be CUDA(HIP) tensors and the module must be in
eval mode.  There is no CPU path here (the CPU restatement lives in oracle/ and is test-only).
"""
import os

import numpy as np
import torch
from torch import nn

from .. import ops
from ..ops import Act, ConvPlan, OkpError
from .backbone import convolution, fold_bn, hg, _HipModule, _np


class prediction_module(nn.Sequential):
    """1x1 256->F +BN+ReLU, 1x1 F->32 +BN+ReLU, 1x1 32->out +bias  (models.py:13-18)."""

    def __init__(self, int_features, features_out):
        super().__init__(convolution(1, 256, int_features, with_bn=True),
                         convolution(1, int_features, 32, with_bn=True),
                         nn.Conv2d(32, features_out, (1, 1), bias=True))
        self.features_out = features_out

    def forward(self, x, sigmoid=False):
        """x: NHWC activation (ops.Act) -> NCHW fp32 tensor [N, out, H, W]."""
        y = self[1](self[0](x))
        w = self[2].weight.detach().reshape(self.features_out, 32).float().contiguous().to(y.t.device)
        b = self[2].bias.detach().float().contiguous().to(y.t.device)
        out = torch.empty((y.n, self.features_out, y.h, y.w), dtype=torch.float32, device=y.t.device)
        act = ops.ACT_SIGMOID if sigmoid else ops.ACT_NONE
        ops.head_out(y, [(0, act, out, o) for o in range(self.features_out)], w, b)
        return out


class _TwoStackHead(nn.Module):
    def __init__(self, features, out_channels):
        super().__init__()
        self.output_head1 = prediction_module(features, out_channels)
        self.output_head2 = prediction_module(features, out_channels)

    def forward(self, x):
        return self.output_head1(x[0]), self.output_head2(x[1])


class HeatmapHead(_TwoStackHead):
    def __init__(self, features, heatmaps):
        super().__init__(features, heatmaps)
        self.output_head1[-1].bias.data.fill_(0.01 / 0.99)
        self.output_head2[-1].bias.data.fill_(0.01 / 0.99)


class DepthHead(_TwoStackHead):
    pass


class CenterHead(_TwoStackHead):
    def __init__(self, features, heatmaps):
        self.outputs = heatmaps - 1
        super().__init__(features, self.outputs * 2)

    def forward(self, x):
        out1, out2 = super().forward(x)
        n, _, h, w = out2.shape
        return out1.reshape(n, self.outputs, 2, h, w), out2.reshape(n, self.outputs, 2, h, w)


def nms(x, size=5):
    """x * (x == max_pool2d(x, size, stride 1, pad size//2))  (models.py:55-58)."""
    return ops.nms_maxpool(x, size)


class KeypointNet(_HipModule):
    """CornerNet-Squeeze hourglass backbone + heat/depth/centre heads (models.py:60-85).

    `compute_dtype` selects the activation/weight precision of the HIP path: torch.float32
    (parity configuration), torch.bfloat16 or torch.float16 (MFMA throughput configurations, fp32 accumulate;
    float16 keeps three more mantissa bits than bfloat16 at the same rate, with a range of +-65504), or
    `ops.F32X3` ("float32x3"): fp32 tensors like torch.float32, every convolution product as a three-term fp16 split on
    the fp16 matrix pipe - the reference's fp32 tolerances hold (measured heat error 3e-6) at a multiple of the fp32
    configuration's speed; activations must stay below 65504 in magnitude.  `ops.F32MIX` ("float32mix") trades most of that
    margin for speed where the rounding error reaches the output attenuated (single-term products inside the trunk's residual
    branches, fp16 inner hourglass levels): heat error 4e-4 on the test networks, still inside the 1e-3 bar.
    """

    def __init__(self, output_size=None, features=128, heatmaps_out=2, dropout=0.1, compute_dtype=torch.float32):
        super().__init__()
        self.backbone = hg()
        self.heatmap_head = HeatmapHead(features, heatmaps_out)
        self.depth_head = DepthHead(features, heatmaps_out)
        self.center_head = CenterHead(features, heatmaps_out)
        self.dropout = nn.Dropout(p=dropout)       # identity in eval mode; kept for the interface
        self.features = features
        self.heatmaps_out = heatmaps_out
        self.compute_dtype, self.mfma_split, self.mixed = ops.parse_compute_dtype(compute_dtype)
        # uint8 frames of any other size are resized (shortest side) and centre-cropped to this size on the device, as the
        # reference's data path does (perception/datasets/video.py:63-69,95-96); None = uint8 frames are taken as they are
        self.raw_frame_size = 511
        # Split-product configurations halve fp32 values into fp16: a value beyond +-65504 has no halves (inf, -inf -> NaN products, which the
        # next ReLU turns into 0 without a trace), where the reference's fp32 arithmetic returns numbers.  The kernels raise a device flag
        # (okp_conv_set_range_flag) whenever a value they hand on leaves that range; what a pass does about it:
        #   "raise"   (default) deployed() / forward() read the flag behind the pass (one device -> host word) and raise OkpError
        #   "float32" ... and run the pass again with the exact fp32 kernels (slow, always right), with a RuntimeWarning the first time
        #   "defer"   nothing is read: the caller asks range_overflow() or folds range_flag() into okp_capacity_overflow's word
        #             (BatchedKeypointPipeline.forward_device does; captured graphs cannot branch on the host anyway)
        self.on_overflow = "raise"
        self._range_flag = None
        self._warned_overflow = False

    # ---- fused three-head plan for one stack ------------------------------------------------
    def _build_heads(self, stack, dtype, device):
        heads = [getattr(h, f"output_head{stack + 1}") for h in (self.heatmap_head, self.depth_head, self.center_head)]
        F = self.features
        w1, b1, w2, b2 = [], [], np.zeros((96, 3 * F), dtype=np.float32), []
        for i, pm in enumerate(heads):
            w, b = fold_bn(pm[0].conv.weight, pm[0].bn)
            w1.append(w[:, :, 0, 0]); b1.append(b)
            w, b = fold_bn(pm[1].conv.weight, pm[1].bn)
            w2[32 * i:32 * i + 32, F * i:F * i + F] = w[:, :, 0, 0]      # block diagonal: heads do not mix
            b2.append(b)
        l1 = ConvPlan(dtype, [256], [1], 3 * F, [(0, 0, 0, np.concatenate(w1, 0))], np.concatenate(b1), relu=True)
        l2 = ConvPlan(dtype, [3 * F], [1], 96, [(0, 0, 0, w2)], np.concatenate(b2), relu=True, alg_k=F)
        w3 = np.concatenate([_np(pm[2].weight).reshape(-1, 32) for pm in heads], 0)
        b3 = np.concatenate([_np(pm[2].bias) for pm in heads], 0)
        return l1, l2, torch.from_numpy(w3).to(device), torch.from_numpy(b3).to(device)

    def range_flag(self, device=None):
        """The network's fp16-range flag (int32 device tensor of one element; split-product configurations only, else None).  Every
        split-product plan built by this network's passes raises it; the passes zero it when they start."""
        if not self.mfma_split:
            return None
        if self._range_flag is None or (device is not None and self._range_flag.device != torch.device(device)):
            dev = device if device is not None else next(self.parameters()).device
            self._range_flag = torch.zeros(1, dtype=torch.int32, device=dev)
            self._drop_all_plans()               # plans hold the old flag's address
        return self._range_flag

    def _drop_all_plans(self):
        for m in self.modules():
            if hasattr(m, "_drop_plans"):
                m._drop_plans()

    def range_overflow(self):
        """True if a value left the fp16 range in a split-product pass since the last pass started (reads one word from the device)."""
        return bool(self.mfma_split and self._range_flag is not None and int(self._range_flag) != 0)

    def _guarded(self, x, run, check):
        """One pass under the range guard: zero the flag, run, then act on `on_overflow` (see __init__).  check=False: never read the flag."""
        if not self.mfma_split:
            return run()
        if self.on_overflow not in ("raise", "float32", "defer"):
            raise OkpError("on_overflow is 'raise', 'float32' or 'defer'")
        self.range_flag(x.device).zero_()
        out = run()
        if not check or self.on_overflow == "defer" or torch.cuda.is_current_stream_capturing():
            return out
        if int(self._range_flag) == 0:
            return out
        if self.on_overflow == "raise":
            raise OkpError(f"{self.configuration()}: an activation left the fp16 range (|x| > 65504 or not finite); the split products of this pass are "
                           "not fp32-grade.  Run these weights with compute_dtype=torch.float32, or load with on_overflow='float32'")
        if not self._warned_overflow:
            import warnings
            warnings.warn(f"{self.configuration()}: an activation left the fp16 range; passes that do are re-run with the exact float32 kernels", RuntimeWarning, stacklevel=3)
            self._warned_overflow = True
        mine = (self.compute_dtype, self.mfma_split, self.mixed)
        try:
            self.compute_dtype, self.mfma_split, self.mixed = torch.float32, False, False
            return run()
        finally:
            self.compute_dtype, self.mfma_split, self.mixed = mine

    def _run_heads(self, stack, cnv, sigmoid):
        with ops.f32_split(self.mfma_split, self.mixed, self.range_flag(cnv.t.device)):      # (allocated here if no guarded pass has run yet: a plan
            return self._run_heads_(stack, cnv, sigmoid)                                     #  built without the flag would be cached without it)

    def _run_heads_(self, stack, cnv, sigmoid):
        l1, l2, w3, b3 = self._plan(("heads", stack, cnv.dtype), lambda: self._build_heads(stack, cnv.dtype, cnv.t.device))
        K = self.heatmaps_out
        n, h, w = cnv.n, cnv.h, cnv.w
        heat = torch.empty((n, K, h, w), dtype=torch.float32, device=cnv.t.device)
        depth = torch.empty((n, K, h, w), dtype=torch.float32, device=cnv.t.device)
        centers = torch.empty((n, 2 * (K - 1), h, w), dtype=torch.float32, device=cnv.t.device)
        outs = [(0, ops.ACT_SIGMOID if sigmoid else ops.ACT_NONE, heat, k) for k in range(K)]
        outs += [(32, ops.ACT_NONE, depth, k) for k in range(K)]
        outs += [(64, ops.ACT_NONE, centers, k) for k in range(2 * (K - 1))]
        if ops.FUSE_HEADS and cnv.dtype in ops.HALF_DTYPES and self.features == 128:
            ops.heads_fused(l1, l2, cnv, outs, w3, b3)           # one launch, the 384- and 96-channel tensors stay in LDS
        elif cnv.pairs:                                          # (split-product configuration: requested from the backbone by fused_heads_x3)
            ops.heads_fused(l1, l2, cnv, outs, w3, b3)
        else:
            a1 = Act.empty(n, h, w, 3 * self.features, cnv.dtype, cnv.t.device)
            l1([cnv], a1, h, w)
            a2 = Act.empty(n, h, w, 96, cnv.dtype, cnv.t.device)
            l2([a1], a2, h, w)
            ops.head_out(a2, outs, w3, b3)
        return heat, depth, centers.reshape(n, K - 1, 2, h, w)

    def _features(self, x):
        with ops.f32_split(self.mfma_split, self.mixed, self.range_flag(x.device) if isinstance(x, torch.Tensor) and x.is_cuda else None):
            return self._features_(x)

    def _features_(self, x):
        ops.require_cuda(x, "frames")
        if self.training:
            raise OkpError("the HIP path implements eval-mode inference only; call .eval()")
        # split-product configuration: the last stack's `cnv` has one reader, the heads - in pair format where its 3x3 convolution can write
        # that, the three heads are one launch (ops.heads_fused; _run_heads_ looks at Act.pairs)
        cnv_pairs = bool(ops.FUSE_HEADS_X3 and self.mfma_split and self.features == 128)      # (the mixed configuration's heads are three-term plans too)
        if x.dtype == torch.uint8:      # raw RGB [N,H,W,3]: normalisation (and resize + centre crop) fused into the packing kernel
            if self.raw_frame_size is not None and tuple(x.shape[1:3]) != (self.raw_frame_size, self.raw_frame_size):
                return self.backbone(ops.preprocess_u8(x, self.compute_dtype, size=self.raw_frame_size), cnv_pairs=cnv_pairs)
            return self.backbone(ops.pack_frames_u8(x, self.compute_dtype), cnv_pairs=cnv_pairs)
        return self.backbone(x.float(), self.compute_dtype, cnv_pairs=cnv_pairs)

    def max_frames_per_pass(self, h, w):
        """Frames per network pass such that the largest activation behind the two high-resolution layers (pre[1]'s output, 256
        channels at quarter resolution - as many bytes per frame as the 16-bit stem output) stays under the 2 GiB view limit of the
        32-bit buffer offsets (include/okp.h).  The stem and pre[1] themselves run in frame chunks inside a pass where the fp32 stem
        output would pass the limit (hg.forward)."""
        esz = 2 if self.compute_dtype in ops.HALF_DTYPES else 4
        h2, w2 = (h + 1) // 2, (w + 1) // 2
        per_frame = max(h2 * w2 * 128 * 2, ((h2 + 1) // 2) * ((w2 + 1) // 2) * 256 * esz)
        return max(1, (0x7FFF0000 - 1) // per_frame)

    def _chunks(self, x):
        h, w = (x.shape[1], x.shape[2]) if x.dtype == torch.uint8 else (x.shape[2], x.shape[3])
        if x.dtype == torch.uint8 and self.raw_frame_size is not None:
            h = w = self.raw_frame_size
        step = self.max_frames_per_pass(h, w)
        return [x[i:i + step] for i in range(0, x.shape[0], step)]

    def forward(self, x, check_range=True):
        """frames [N,3,H,W] fp32 -> ((hm1,hm2),(d1,d2),(c1,c2)), raw logits for the heat maps."""
        return self._guarded(x, lambda: self._forward(x), check_range)

    def _forward(self, x):
        outs = []
        for xc in self._chunks(x):
            feats = self._features(xc)
            outs.append(self._run_heads(0, feats[0], sigmoid=False) + self._run_heads(1, feats[1], sigmoid=False))
        h1, d1, c1, h2, d2, c2 = [torch.cat(t) if len(outs) > 1 else t[0] for t in zip(*outs)]
        return (h1, h2), (d1, d2), (c1, c2)

    def configuration(self):
        """Name of the compute configuration: 'float32mix', 'float32x3', 'float32', 'bfloat16' or 'float16'."""
        if self.mixed:
            return ops.F32MIX
        if self.mfma_split:
            return ops.F32X3
        return str(self.compute_dtype).replace("torch.", "")

    def set_compute_dtype(self, compute_dtype):
        """Switch the configuration of the SAME weights (plans are cached per configuration; nothing is rebuilt that exists)."""
        self.compute_dtype, self.mfma_split, self.mixed = ops.parse_compute_dtype(compute_dtype)
        return self

    def _apply(self, fn, *a, **k):          # .to() / .cuda(): the range flag lives on the old device
        self._range_flag = None
        return super()._apply(fn, *a, **k)

    def precision_audit(self, x, against=ops.F32X3):
        """Run the deployed outputs of frames `x` in this network's compute precision AND in `against` (default: the split-product
        configuration, fp32-grade: 2e-6 on the test networks) on the same weights, both on the HIP path, and return the absolute
        differences {"heat" | "depth" | "centers": {"max", "mean", "p99", "finite"}}.  The error of a 16-bit or mixed configuration depends on
        the weights: this is the on-device check that a plan chosen on one network still holds on another - a few
        frames, two passes, no CPU reference involved."""
        mine = (self.compute_dtype, self.mfma_split, self.mixed)
        with torch.no_grad():
            got = [t.float().clone() for t in self.deployed(x, check_range=False)]
            mine_over = self.range_overflow()
            try:
                self.compute_dtype, self.mfma_split, self.mixed = ops.parse_compute_dtype(against)
                ref = [t.float() for t in self.deployed(x, check_range=False)]
                ref_over = self.range_overflow()
            finally:
                self.compute_dtype, self.mfma_split, self.mixed = mine
        report = {}
        for name, g, r in zip(("heat", "depth", "centers"), got, ref):
            d = (g - r).abs().flatten()
            k = max(1, int(round(0.99 * d.numel())))
            report[name] = {"max": float(d.max()), "mean": float(d.mean()), "p99": float(d.kthvalue(k).values),
                            # (fp16 operands overflow above 65504: the outputs may still look finite - a ReLU swallows the NaN - so the range flags count)
                            "finite": bool(torch.isfinite(g).all()) and bool(torch.isfinite(r).all()) and not mine_over and not ref_over}
        return report

    def deployed(self, x, check_range=True):
        """What the packaged model returns (scripts/package_model.py:26-28):
        sigmoid(heat[-1]), depth[-1], centers[-1]; the dead stack-1 heads are not executed.
        Split-product configurations: the pass runs under the fp16-range guard (on_overflow, see __init__); check_range=False leaves the
        flag unread (the caller folds range_flag() into its own overflow word)."""
        return self._guarded(x, lambda: self._deployed(x), check_range)

    def _deployed(self, x):
        chunks = self._chunks(x)
        outs = []
        for xc in chunks:
            feats = self._features(xc)
            outs.append(self._run_heads(1, feats[1], sigmoid=True))
        return tuple(torch.cat(t) if len(outs) > 1 else t[0] for t in zip(*outs))

"""ORACLE (test infrastructure, not product code): torch-CPU fp32 restatement of the
reference keypoint network.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product path (object_keypoints_amd) never does.

Parity status: PINNED.  tests/golden/make_goldens.py imports the real reference
(/root/reference, build container only) and stores its outputs for per-block and
whole-network cases on procedurally generated weights; tests/test_oracle_net.py
checks this restatement against those fixtures.

What is restated (reference file:line):
  convolution        perception/corner_net_lite/core/models/py_utils/utils.py:143-156
  residual           .../py_utils/utils.py:158-185
  fire_module        .../core/models/CornerNet_Squeeze.py:10-30
  layer factories    .../CornerNet_Squeeze.py:32-51
  hg_module          .../py_utils/modules.py:25-66
  hg                 .../py_utils/modules.py:68-93
  squeeze backbone   .../CornerNet_Squeeze.py:66-89 (only `.hg` is kept, perception/models.py:78)
  heads, KeypointNet perception/models.py:13-53,60-85
  deployed forward   scripts/package_model.py:26-28

Module attribute names are the state_dict wire format and are
therefore identical to the reference's; the bodies are written independently.
"""
import torch
from torch import nn
import torch.nn.functional as F


class convolution(nn.Module):
    """k x k conv, pad (k-1)//2, bias only without BN, then BN, then ReLU."""

    def __init__(self, k, inp_dim, out_dim, stride=1, with_bn=True):
        super().__init__()
        self.conv = nn.Conv2d(inp_dim, out_dim, k, stride=stride, padding=(k - 1) // 2, bias=not with_bn)
        self.bn = nn.BatchNorm2d(out_dim) if with_bn else nn.Sequential()

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


class residual(nn.Module):
    """relu(bn2(conv2(relu(bn1(conv1 x)))) + skip(x)); skip is 1x1 conv + BN when shape changes."""

    def __init__(self, inp_dim, out_dim, k=3, stride=1):
        super().__init__()
        p = (k - 1) // 2
        self.conv1 = nn.Conv2d(inp_dim, out_dim, k, stride=stride, padding=p, bias=False)
        self.bn1 = nn.BatchNorm2d(out_dim)
        self.conv2 = nn.Conv2d(out_dim, out_dim, k, padding=p, bias=False)
        self.bn2 = nn.BatchNorm2d(out_dim)
        projected = stride != 1 or inp_dim != out_dim
        self.skip = nn.Sequential(nn.Conv2d(inp_dim, out_dim, 1, stride=stride, bias=False),
                                  nn.BatchNorm2d(out_dim)) if projected else nn.Sequential()

    def forward(self, x):
        y = self.bn2(self.conv2(F.relu(self.bn1(self.conv1(x)))))
        return F.relu(y + self.skip(x))


class fire_module(nn.Module):
    """Squeeze 1x1 + BN (no ReLU) -> [1x1 || depth-wise 3x3] -> concat -> BN -> (+x) -> ReLU."""

    def __init__(self, inp_dim, out_dim, sr=2, stride=1):
        super().__init__()
        mid = out_dim // sr
        self.conv1 = nn.Conv2d(inp_dim, mid, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(mid)
        self.conv_1x1 = nn.Conv2d(mid, out_dim // 2, 1, stride=stride, bias=False)
        self.conv_3x3 = nn.Conv2d(mid, out_dim // 2, 3, stride=stride, padding=1, groups=mid, bias=False)
        self.bn2 = nn.BatchNorm2d(out_dim)
        self.skip = stride == 1 and inp_dim == out_dim

    def forward(self, x):
        s = self.bn1(self.conv1(x))
        y = self.bn2(torch.cat([self.conv_1x1(s), self.conv_3x3(s)], dim=1))
        return F.relu(y + x if self.skip else y)


def _fires(dims, first_stride=1):
    mods = [fire_module(a, b, stride=first_stride if i == 0 else 1) for i, (a, b) in enumerate(dims)]
    return nn.Sequential(*mods)


def make_layer(inp, out, n):        # fire(inp,out), then n-1 x fire(out,out)
    return _fires([(inp, out)] + [(out, out)] * (n - 1))


def make_layer_revr(inp, out, n):   # n-1 x fire(inp,inp), then fire(inp,out)
    return _fires([(inp, inp)] * (n - 1) + [(inp, out)])


def make_hg_layer(inp, out, n):     # like make_layer but the first fire has stride 2
    return _fires([(inp, out)] + [(out, out)] * (n - 1), first_stride=2)


class hg_module(nn.Module):
    """One hourglass level: up1(x) + unpool(low3(low2(low1(x)))); max1 is the identity here."""

    def __init__(self, n, dims, modules):
        super().__init__()
        cur, nxt = dims[0], dims[1]
        self.n = n
        self.up1 = make_layer(cur, cur, modules[0])
        self.max1 = nn.Sequential()
        self.low1 = make_hg_layer(cur, nxt, modules[0])
        self.low2 = hg_module(n - 1, dims[1:], modules[1:]) if n > 1 else make_layer(nxt, nxt, modules[1])
        self.low3 = make_layer_revr(nxt, cur, modules[0])
        self.up2 = nn.ConvTranspose2d(cur, cur, kernel_size=4, stride=2, padding=1)

    def forward(self, x):
        return self.up1(x) + self.up2(self.low3(self.low2(self.low1(self.max1(x)))))


def _merge_mod():
    return nn.Sequential(nn.Conv2d(256, 256, 1, bias=False), nn.BatchNorm2d(256))


class hg(nn.Module):
    """Two stacked hourglasses with the inter-stack merge; returns [cnv0, cnv1]."""

    DIMS = [256, 256, 384, 384, 512]
    MODULES = [2, 2, 2, 2, 4]

    def __init__(self, stacks=2):
        super().__init__()
        self.pre = nn.Sequential(convolution(7, 3, 128, stride=2),
                                 residual(128, 256, stride=2),
                                 residual(256, 256, stride=2))
        self.hgs = nn.ModuleList([hg_module(4, self.DIMS, self.MODULES) for _ in range(stacks)])
        self.cnvs = nn.ModuleList([convolution(3, 256, 256) for _ in range(stacks)])
        self.inters = nn.ModuleList([residual(256, 256) for _ in range(stacks - 1)])
        self.inters_ = nn.ModuleList([_merge_mod() for _ in range(stacks - 1)])
        self.cnvs_ = nn.ModuleList([_merge_mod() for _ in range(stacks - 1)])

    def forward(self, x):
        inter = self.pre(x)
        outs = []
        last = len(self.hgs) - 1
        for i, (hourglass, cnv_mod) in enumerate(zip(self.hgs, self.cnvs)):
            cnv = cnv_mod(hourglass(inter))
            outs.append(cnv)
            if i < last:
                inter = self.inters[i](F.relu(self.inters_[i](inter) + self.cnvs_[i](cnv)))
        return outs


def prediction_module(int_features, features_out):
    return nn.Sequential(convolution(1, 256, int_features),
                         convolution(1, int_features, 32),
                         nn.Conv2d(32, features_out, 1, bias=True))


class _TwoStackHead(nn.Module):
    def __init__(self, features, out_channels):
        super().__init__()
        self.output_head1 = prediction_module(features, out_channels)
        self.output_head2 = prediction_module(features, out_channels)

    def forward(self, feats):
        return self.output_head1(feats[0]), self.output_head2(feats[1])


class HeatmapHead(_TwoStackHead):
    def __init__(self, features, heatmaps):
        super().__init__(features, heatmaps)
        for head in (self.output_head1, self.output_head2):
            head[-1].bias.data.fill_(0.01 / 0.99)


class DepthHead(_TwoStackHead):
    pass


class CenterHead(_TwoStackHead):
    def __init__(self, features, heatmaps):
        self.outputs = heatmaps - 1
        super().__init__(features, self.outputs * 2)

    def forward(self, feats):
        a, b = super().forward(feats)
        n, _, h, w = b.shape
        return a.reshape(n, self.outputs, 2, h, w), b.reshape(n, self.outputs, 2, h, w)


class KeypointNet(nn.Module):
    def __init__(self, output_size=None, features=128, heatmaps_out=2, dropout=0.1):
        super().__init__()
        self.backbone = hg()
        self.heatmap_head = HeatmapHead(features, heatmaps_out)
        self.depth_head = DepthHead(features, heatmaps_out)
        self.center_head = CenterHead(features, heatmaps_out)
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, x):
        feats = [self.dropout(f) for f in self.backbone(x)]
        return self.heatmap_head(feats), self.depth_head(feats), self.center_head(feats)


def deployed_forward(net, frames):
    """What the packaged TorchScript model returns: stack-2 outputs, sigmoid on the heatmap only."""
    with torch.no_grad():
        heat, depth, centers = net(frames)
    return torch.sigmoid(heat[-1]), depth[-1], centers[-1]


def nms(x, size=5):
    """x * (x == maxpool_{size x size, stride 1}(x)) (perception/models.py:55-58)."""
    pooled = F.max_pool2d(x, size, stride=1, padding=size // 2)
    return x * (x == pooled).to(x.dtype)


def load_synthetic(module, seed=0, **kw):
    """Fill `module` (any of the classes above) with object_keypoints_amd.synth weights."""
    from object_keypoints_amd import synth
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    values = synth.fill_state_dict(shapes, seed=seed, **kw)
    module.load_state_dict({k: torch.from_numpy(v.copy()) if v.ndim else torch.tensor(int(v)) for k, v in values.items()})
    return module.eval()

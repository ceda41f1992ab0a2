"""TEST INFRASTRUCTURE (GPU + oracle): stage-by-stage comparison of the mixed configuration on the device with its CPU rounding-point
model - a debugging aid, run by hand on the GPU box.  usage: python tests/precision/debug_mix_stages.py [valve_k3|cups_k4]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tests", "golden")); sys.path.insert(0, os.path.join(REPO, "tests", "precision"))
import numpy as np, torch
import cases, emulate
from object_keypoints_amd import ops, synth
from object_keypoints_amd.perception.models import KeypointNet
name = sys.argv[1] if len(sys.argv) > 1 else "cups_k4"
case = cases.NET_CASES[name]
net = KeypointNet(features=128, heatmaps_out=case["heatmaps_out"], compute_dtype=ops.F32MIX)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.fill_state_dict(shapes, seed=case["weight_seed"]).items()})
net.eval().cuda()
xh = synth.frames(1, seed=case["frame_seed"], start=case["frame_index"])
_, emu = emulate.build(case["heatmaps_out"], case["weight_seed"])
pol = emulate.mixed_policy(ops.MIX_FP16_LEVELS, ops.MIX_BRANCH_SINGLE, ops.MIX_STEM_FP16)
trace = {}
orig_act = pol.act
def act(n, t):
    r = orig_act(n, t); trace[n] = r; return r
pol.act = act
emu.forward(torch.from_numpy(xh), pol)
bb = net.backbone
def cmp(tag, a, key):
    got = a.to_nchw().cpu(); want = trace[key]
    print(f"{tag:28s} max|diff| {float((got - want).abs().max()):.3e}   scale {float(want.abs().max()):.2f}")
with torch.no_grad(), ops.f32_split(True, True):
    x = torch.from_numpy(xh).cuda()
    inter = bb.pre[0].forward_frames(x, torch.float32); cmp("pre.0", inter, "backbone.pre.0")
    inter = bb.pre[1](inter); cmp("pre.1", inter, "backbone.pre.1")
    inter = bb.pre[2](inter); cmp("pre.2", inter, "backbone.pre.2")
    hg0 = bb.hgs[0]
    up1 = hg0.up1(inter); cmp("hgs.0.up1", up1, "backbone.hgs.0.up1.1")
    low1 = hg0.low1(inter); cmp("hgs.0.low1", low1, "backbone.hgs.0.low1.1")
    m3 = hg0.low2
    l1 = m3.low1(low1); cmp("hgs.0.low2.low1", l1, "backbone.hgs.0.low2.low1.1")
    m2 = m3.low2
    o2 = m2(l1); cmp("hgs.0.low2.low2 (fp16 dom.)", o2, "backbone.hgs.0.low2.low2")
    l3 = m3.low3(o2); cmp("hgs.0.low2.low3", l3, "backbone.hgs.0.low2.low3.1")
    u1 = m3.up1(low1); cmp("hgs.0.low2.up1", u1, "backbone.hgs.0.low2.up1.1")
    o3 = m3.up2(l3, u1); cmp("hgs.0.low2", o3, "backbone.hgs.0.low2")
    h = hg0(inter); cmp("hgs.0", h, "backbone.hgs.0")
    cnv = bb.cnvs[0](h); cmp("cnvs.0", cnv, "backbone.cnvs.0")
    # inside pre.1
    from object_keypoints_amd.ops import Act
    inter0 = bb.pre[0].forward_frames(x, torch.float32)
    r = bb.pre[1]
    p1, p2 = r._plan(("p", inter0.dtype), lambda: r._build(inter0.dtype))
    print("p1 terms", p1.tap_terms, "p2 terms", p2.tap_terms, "split", p1.split)
    t = Act.empty(1, 128, 128, 256, torch.float32, x.device)
    p1([inter0], t, 128, 128)
    tg = t.to_nchw().cpu()
    print("t  vs model t (fp16-rounded)", float((tg.half().float() - trace["backbone.pre.1.t"]).abs().max()), "unrounded", float((tg - trace["backbone.pre.1.t"]).abs().max()))
    import torch.nn.functional as F
    sd = emu.sd
    w1, b1 = emulate._fold(sd["backbone.pre.1.conv1.weight"], "backbone.pre.1.bn1", "", sd)
    x0 = trace["backbone.pre.0"]
    ref1 = F.relu(F.conv2d(x0.half().float(), w1.half().float(), b1, stride=2, padding=1))
    ref3 = F.relu(F.conv2d(x0, w1, b1, stride=2, padding=1))
    print("t vs single-term ref", float((tg - ref1).abs().max()), " vs fp32 ref", float((tg - ref3).abs().max()))

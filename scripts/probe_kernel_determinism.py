"""Which launch of the 64-frame bf16 step is not bit-reproducible?  Every launch is repeated R times on the same inputs
(serially, one stream) and its output compared with the first run; then whole-network runs with and without the side streams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops, synth
from object_keypoints_amd.perception.models import KeypointNet
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[sys.argv[3] if len(sys.argv) > 3 else "bf16"]
net = KeypointNet(features=128, heatmaps_out=3, compute_dtype=dt)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.fill_state_dict(shapes, seed=0).items()})
net.eval().cuda()
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
x = torch.randn(n, 3, 511, 511, device="cuda", generator=gen)
bad = []

def check(name, outs, rerun):
    torch.cuda.synchronize()
    first = [o.clone() for o in outs]
    for r in range(R):
        rerun(); torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(first, outs)):
            if not torch.equal(a, b):
                d = (a.float() - b.float()).abs()
                bad.append(name)
                print(f"MISMATCH {name} out{i} rep {r}: {int((d > 0).sum())} elements differ, max {float(d.max()):.4g}", flush=True)
                return

orig_call = ops.ConvPlan.__call__
def conv_call(self, srcs, out, ho, wo, **kw):
    orig_call(self, srcs, out, ho, wo, **kw)
    outs = [out.t] + ([kw["dw"][2].t] if kw.get("dw") else [])
    tile = ops._lib.lib()  # noqa
    check(f"conv cout={self.cout} taps_k={self.alg_k} n_src={self.n_src} {out.n}x{ho}x{wo} classes={kw.get('n_classes',1)} dw={bool(kw.get('dw'))}", outs,
          lambda: orig_call(self, srcs, out, ho, wo, **kw))
ops.ConvPlan.__call__ = conv_call
orig_fire = ops.fire_fused
def fire(squeeze, expand, wd, bd, x_, out, stride, skip):
    orig_fire(squeeze, expand, wd, bd, x_, out, stride, skip)
    check(f"fire2 cin={squeeze.cins[0]} mid={squeeze.cout} {x_.n}x{x_.h}x{x_.w} s{stride}", [out.t], lambda: orig_fire(squeeze, expand, wd, bd, x_, out, stride, skip))
ops.fire_fused = fire
orig_chain = ops.fire_chain
def chain(mods, x_, out):
    orig_chain(mods, x_, out)
    check(f"chain {len(mods)} x {x_.t.shape}", [out.t], lambda: orig_chain(mods, x_, out))
ops.fire_chain = chain
orig_heads = ops.heads_fused
def heads(l1, l2, x_, outputs, w, b):
    orig_heads(l1, l2, x_, outputs, w, b)
    ts = []
    for o in outputs:
        if not any(o[2] is t for t in ts): ts.append(o[2])
    check("heads", ts, lambda: orig_heads(l1, l2, x_, outputs, w, b))
ops.heads_fused = heads
orig_stem = ops.StemPlan.from_nchw
def stem(self, frames, out):
    orig_stem(self, frames, out)
    check("stem", [out.t], lambda: orig_stem(self, frames, out))
ops.StemPlan.from_nchw = stem

ops.SIDE_STREAMS = False
with torch.no_grad():
    net.deployed(x)
print("per-launch check done; non-reproducible launches:", len(bad), flush=True)
ops.ConvPlan.__call__ = orig_call; ops.fire_fused = orig_fire; ops.fire_chain = orig_chain; ops.heads_fused = orig_heads; ops.StemPlan.from_nchw = orig_stem
with torch.no_grad():
    for side in (False, True):
        ops.SIDE_STREAMS = side
        base = [t.clone() for t in net.deployed(x)]
        diff = 0
        for r in range(5):
            out = net.deployed(x); torch.cuda.synchronize()
            diff += sum(0 if torch.equal(a, b) else 1 for a, b in zip(base, out))
        print(f"whole net side={side}: {diff} of 15 output comparisons differ", flush=True)

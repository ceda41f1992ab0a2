"""Per-layer-shape timing of one forward (HIP events around every implicit-GEMM launch; side streams off so the
launches serialise).  usage: layer_times.py [bf16|f32] [batch]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops, synth
from object_keypoints_amd.perception.models import KeypointNet
dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f32x3": ops.F32X3, "f32mix": ops.F32MIX}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ops.SIDE_STREAMS = False
net = KeypointNet(features=128, heatmaps_out=3, compute_dtype=dtype)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
vals = synth.fill_state_dict(shapes, seed=0)
net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in vals.items()})
net.eval()
x = torch.randn(n, 3, 511, 511, device="cuda")
for _ in range(2): net.deployed(x)
torch.cuda.synchronize()

class Hook:
    def __init__(self): self.rec = []
    def before(self, plan, tile, macs):
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        return (plan, tile, macs, e0)
    def after(self, tok):
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        plan, tile, macs, e0 = tok
        self.rec.append((tuple(plan.cins), plan.cout, plan.alg_k, plan.last_launch, tile, macs, e0, e1))
h = Hook(); ops.LAUNCH_HOOK = h
REP = 5
for _ in range(REP): net.deployed(x)
torch.cuda.synchronize()
agg = collections.OrderedDict()
esz = 2 if dtype in (torch.bfloat16, torch.float16) else 4
for cins, cout, k, (nn, ho, wo, dw, res, ncls), tile, macs, e0, e1 in h.rec:
    key = (cins, cout, k, ho, wo, dw, res, ncls, tile)
    a = agg.setdefault(key, [0, 0.0, macs])
    a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
tot = sum(a[1] for a in agg.values()) / REP
print(f"{'cins':>12} {'cout':>5} {'K':>5} {'HoxWo':>9} dw res cls tile   n   us/launch  total_us  TFLOP/s  min-GB  GB/s")
for key, (cnt, us, macs) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    cins, cout, k, ho, wo, dw, res, ncls, tile = key
    P = n * ho * wo
    # compulsory bytes: sources once (at conv stride 1 approx), output once, residual once, dw: out + res again
    b = P * (sum(cins) + cout * (1 + res) + (cout * (1 + 1) if dw else 0)) * esz
    per = us / cnt
    print(f"{str(cins):>12} {cout:5d} {k:5d} {ho:4d}x{wo:<4d} {int(dw):2d} {int(res):3d} {ncls:3d} {tile:4d} {cnt//REP:3d} {per:10.1f} {us/REP:9.1f} {2*macs/per/1e6:8.1f} {b/1e9:7.3f} {b/per/1e3:6.0f}")
print("sum", tot, "us")

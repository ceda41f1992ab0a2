"""The resident chains with hot and with cold weights: back-to-back launches (weights stay in every XCD's L2) against launches
separated by a 512 MB fill that streams through L2 / MALL (what the activations do between two uses in the network)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception import backbone as bb
def mods_of(shapes):
    return [bb.fire_module(a, b, stride=s).eval() for a, b, s in shapes]
cases = {"innermost level (8 modules)": ([(384, 512, 2)] + [(512, 512, 1)] * 6 + [(512, 384, 1)], (64, 8, 8, 384)),
         "pair of fire(384, 384) at 8x8": ([(384, 384, 1)] * 2, (64, 8, 8, 384)),
         "fire(384, 384) at 16x16 (okp_fire2)": ([(384, 384, 1)], (64, 16, 16, 384)),
         "fire(256, 256) at 32x32 (okp_fire2)": ([(256, 256, 1)], (64, 32, 32, 256))}
junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for name, (shapes, xs) in cases.items():
    mods = mods_of(shapes)
    x = ops.Act(torch.randn(xs, device="cuda").bfloat16())
    for _ in range(3): bb.run_fire_modules(mods, x)
    torch.cuda.synchronize()
    res = {}
    for mode in ("hot", "cold"):
        ts = []
        for _ in range(10):
            if mode == "cold": junk.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); bb.run_fire_modules(mods, x); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res[mode] = sorted(ts)[len(ts) // 2]
    print(f"{name}: hot {res['hot']:.1f} us, cold {res['cold']:.1f} us (events around one launch: +~10 us of launch latency in both)")

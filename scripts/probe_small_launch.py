"""Fixed cost against K-dependent cost of the small split-product launches of the deep hourglass levels: a 1x1 convolution cin -> cout on
hw x hw maps of n frames, back to back on one stream (each launch depends on nothing but the stream order).
usage: probe_small_launch.py [hw=4] [n=64] [cout=256] [tile=0] [dtype=f32x3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops
kw = dict(hw=4, n=64, cout=256, tile=0, dtype="f32x3")
for a in sys.argv[1:]:
    k, v = a.split("="); kw[k] = v if k == "dtype" else int(v)
n, hw, cout = kw["n"], kw["hw"], kw["cout"]
rng = np.random.default_rng(0)
for cin in (32, 64, 128, 256, 512, 1024):
    w = (rng.standard_normal((cout, cin)) / np.sqrt(cin)).astype(np.float32)
    if kw["dtype"] == "f32x3":
        with ops.f32_split():
            plan = ops.ConvPlan(torch.float32, [cin], [1], cout, [(0, 0, 0, w)], np.zeros(cout, np.float32), relu=True)
        dt = torch.float32
    else:
        dt = torch.bfloat16
        plan = ops.ConvPlan(dt, [cin], [1], cout, [(0, 0, 0, w)], np.zeros(cout, np.float32), relu=True)
    x = ops.Act(torch.randn(n, hw, hw, cin, device="cuda").to(dt))
    out = ops.Act.empty(n, hw, hw, cout, dt, x.t.device)
    for _ in range(20): plan([x], out, hw, hw, tile=kw["tile"])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): plan([x], out, hw, hw, tile=kw["tile"])
    e1.record(); torch.cuda.synchronize()
    print(f"{kw['dtype']} {cin:5d} -> {cout} at {hw}x{hw} x {n}: {e0.elapsed_time(e1) / 200 * 1e3:6.1f} us per launch (200 back to back)")

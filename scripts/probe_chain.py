"""Time okp_fire_chain_forward for different chain lengths (4x4 maps, 64 frames)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception import backbone as bb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mods = [bb.fire_module(512, 512).eval() for _ in range(8)]
x = ops.Act(torch.randn(n, 4, 4, 512, device="cuda").bfloat16())
for count in (1, 2, 4, 6, 8):
    for chain in (True, False):
        ops.FUSE_FIRE_CHAIN = chain
        ops.FUSE_FIRE_MIN_HW = 8
        for _ in range(3): y = bb.run_fire_modules(mods[:count], x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): y = bb.run_fire_modules(mods[:count], x)
        e1.record(); torch.cuda.synchronize()
        print(f"count {count} chain={chain}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")

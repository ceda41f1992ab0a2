"""Time the stem on a 64 x 511 x 511 batch: dedicated kernel vs the generic tap-list kernel (tile 4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception import backbone as bb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(n, 3, 511, 511, device="cuda")
for kernel in (True, False, False, True):
    bb.STEM_KERNEL = kernel
    m = bb.convolution(7, 3, 128, stride=2).eval()
    packed = ops.pack_frames(x, torch.bfloat16)
    for _ in range(3): y = m(packed)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = m(packed)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gb = y.t.numel() * 2 / 1e9
    print(f"stem kernel={kernel}: {us:.1f} us  ({gb / us * 1e6:.0f} GB/s of output)")

"""Time the split-product stem (fp32 NCHW frames -> fp32 NHWC, 7x7/s2, 3 -> 128) against the pack + generic split-product kernel path.
usage: probe_stem_x3.py [n=32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception import backbone as bb
n = int(sys.argv[1].split("=")[1]) if len(sys.argv) > 1 else 32
m = bb.convolution(7, 3, 128, stride=2).eval()
x = torch.randn(n, 3, 511, 511, device="cuda")
for flag in (True, False, True):
    bb.STEM_X3_KERNEL = flag
    with ops.f32_split():
        for _ in range(5): out = m.forward_frames(x, torch.float32)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): out = m.forward_frames(x, torch.float32)
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gb = (out.t.numel() * 4 + x.numel() * 4) / 1e9
    print(f"{'stem kernel (x3)' if flag else 'pack + generic tile 4'}: {us:7.1f} us  {gb / us * 1e6:6.0f} GB/s of frames + output ({gb:.2f} GB)")

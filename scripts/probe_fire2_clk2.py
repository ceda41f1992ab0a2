"""Steady-state phase times of okp_fire2 (debug build -DOKP_FIRE2_CLK=2, OKP_LIB pointing at it): wave 0 of every workgroup adds up, over
all its tiles, the time between consecutive stamps: prologue | - | - | squeeze k-loop | squeeze tile -> LDS + barrier | expand branch |
depth-wise branch; printed per tile (median over workgroups), 100 MHz wall clock."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops, _lib
from object_keypoints_amd.perception import backbone as bb
lib = _lib.lib()
lib.okp_fire2_clk_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
for (c, co, hw, stride, n) in [(256, 256, 64, 1, 64), (256, 256, 32, 1, 64), (256, 256, 64, 2, 64), (384, 384, 16, 1, 64)]:
    m = bb.fire_module(c, co, stride=stride).eval()
    x = ops.Act(torch.randn((n, hw, hw, c), device="cuda").bfloat16())
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    lib.okp_fire2_clk_read(np.zeros(8, np.int64).ctypes.data, 8)
    m(x); torch.cuda.synchronize()
    buf = np.zeros(1024 * 8, np.int64)
    lib.okp_fire2_clk_read(buf.ctypes.data, buf.size)
    t = buf.reshape(1024, 8)
    t = t[t[:, 7] > 0].astype(np.float64)
    tiles = t[:, 7]
    per = t[:, :7] / 100.0 / tiles[:, None]
    print(f"{c}->{co} {hw}x{hw} s{stride} n={n}: {len(t)} workgroups, {tiles.mean():.1f} tiles each; us per tile: pre {np.median(per[:, 0] + per[:, 1] + per[:, 2]):.2f} | "
          f"k-loop {np.median(per[:, 3]):.2f} | s->LDS {np.median(per[:, 4]):.2f} | expand {np.median(per[:, 5]):.2f} | depth-wise {np.median(per[:, 6]):.2f} | sum {np.median(per.sum(axis=1)):.2f}")

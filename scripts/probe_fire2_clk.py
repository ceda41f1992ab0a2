"""Phase time stamps of okp_fire2 (debug build with -DOKP_FIRE2_CLK, OKP_LIB pointing at it): where one tile's latency goes.
Stamps of wave 0 on each workgroup's first tile, 100 MHz wall clock: 0 entry, 1 prologue issued, 2 after the first barrier
(vmcnt/lgkmcnt drained), 3 squeeze k-loop done, 4 squeeze tile in LDS (barrier), 5 expand branch issued, 6 depth-wise branch
issued, 7 its stores retired."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops, _lib
from object_keypoints_amd.perception import backbone as bb
lib = _lib.lib()
lib.okp_fire2_clk_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
for (c, co, hw, stride, n) in [(256, 256, 64, 1, 64), (256, 256, 32, 1, 64), (384, 384, 16, 1, 64), (384, 256, 16, 1, 64), (256, 256, 64, 2, 64), (384, 384, 16, 2, 64),
                               (256, 256, 32, 1, 1), (384, 384, 16, 1, 1), (512, 512, 8, 1, 64)]:
    m = bb.fire_module(c, co, stride=stride).eval()
    x = ops.Act(torch.randn((n, hw, hw, c), device="cuda").bfloat16())
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    lib.okp_fire2_clk_read(np.zeros(8, np.int64).ctypes.data, 8)      # (also clears the buffer)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); m(x); e1.record(); torch.cuda.synchronize()
    buf = np.zeros(1024 * 8, np.int64)
    rc = lib.okp_fire2_clk_read(buf.ctypes.data, buf.size)
    t = buf.reshape(1024, 8)
    live = t[:, 7] > 0
    t = t[live].astype(np.float64) / 100.0          # us
    t0 = t[:, 0].min()
    d = np.diff(t, axis=1)
    print(f"{c}->{co} {hw}x{hw} s{stride} n={n}: event {e0.elapsed_time(e1) * 1e3:6.1f} us, {live.sum()} workgroups, entry skew {t[:, 0].max() - t0:5.1f} us, "
          f"last stamp {t[:, 7].max() - t0:6.1f} us | median phase us: " + " ".join(f"{v:5.1f}" for v in np.median(d, axis=0)) +
          " | p90: " + " ".join(f"{v:5.1f}" for v in np.percentile(d, 90, axis=0)))

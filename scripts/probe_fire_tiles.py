import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception import backbone as bb
for (c, hw) in ((384, 16), (384, 8), (512, 4)):
    m = bb.fire_module(c, c).eval().cuda()
    x = ops.Act(torch.randn(64, hw, hw, c, device="cuda"))
    for tile in (0, 1, 2, 4):
        ops.FORCE_TILE = tile
        try:
            with ops.f32_split(), torch.no_grad():
                for _ in range(5): y = m(x)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50): y = m(x)
                e1.record(); torch.cuda.synchronize()
            print(f"fire({c},{c}) at {hw}x{hw} x 64, float32x3, tile {tile}: {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us per module (two launches)")
        except Exception as e:
            print(f"fire({c},{c}) at {hw}x{hw} tile {tile}: {str(e)[:80]}")
        finally:
            ops.FORCE_TILE = 0

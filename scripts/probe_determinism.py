"""Run the batched pipeline several times on the same frames and compare every output bit for bit
(eager, with and without side streams, and hipGraph replay)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops, synth
from object_keypoints_amd.perception.models import KeypointNet
from object_keypoints_amd.perception.pipeline import BatchedKeypointPipeline
from object_keypoints_amd.perception.utils import camera_utils as cu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net = KeypointNet(features=128, heatmaps_out=3, compute_dtype=torch.bfloat16)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.fill_state_dict(shapes, seed=0).items()})
net.eval().cuda()
params = cu.load_calibration_params(os.path.join(REPO, "config", "calibration.yaml"))
camera = cu.FisheyeCamera(params["K"], params["D"], params["image_size"]).scale(511 / 720)
camera = camera.cut(np.array([(511 / 720 * 1280 - 511.0) / 2.0, 0.0])).scale(64 / 511)
pipe = BatchedKeypointPipeline(net, {"keypoint_config": [1, 3]}, camera, capacity=64)
x = torch.randn(n, 3, 511, 511, device="cuda")
KEYS = ("heat", "depth", "centers", "count", "xyc")
def snap(out): return {k: out[k].clone() for k in KEYS}
def same(a, b): return {k: bool(torch.equal(torch.nan_to_num(a[k].float()), torch.nan_to_num(b[k].float()))) for k in KEYS}
ops.SIDE_STREAMS = False
base = snap(pipe.forward_device(x)); torch.cuda.synchronize()
for side in (False, True):
    ops.SIDE_STREAMS = side
    for rep in range(3):
        out = snap(pipe.forward_device(x)); torch.cuda.synchronize()
        print(f"eager side={side} rep {rep}:", same(base, out))
for side in (True, False):
    ops.SIDE_STREAMS = side
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gout = pipe.forward_device(x)
    for rep in range(3):
        g.replay(); torch.cuda.synchronize()
        print(f"graph side={side} rep {rep}:", same(base, {k: gout[k] for k in KEYS}))
    del g, gout

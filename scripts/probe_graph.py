import sys, time, os
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
for side in (False, True):
    ops.SIDE_STREAMS = side
    for _ in range(3): out = pipe.forward_device(x)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(10): out = pipe.forward_device(x)
    torch.cuda.synchronize()
    print(f"eager side={side}: {(time.time()-t)/10*1e3:.3f} ms")
    ref = out["points"].clone()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            gout = pipe.forward_device(x)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        print(f"graph side={side}: {(time.time()-t)/10*1e3:.3f} ms  same={torch.equal(torch.nan_to_num(gout['points']), torch.nan_to_num(ref))}")
    except Exception as e:
        print("graph capture failed:", repr(e)[:300])

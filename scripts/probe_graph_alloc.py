"""Root-cause probe for 'hipGraph replays with forked hourglass branches differ from the eager step' (round 1).

Three captures of the 64-frame bf16 step with the side streams ON, each replayed R times against the eager result:
  A  plain capture (the round-1 failure mode)
  B  capture with every activation tensor kept alive until the capture ends (no block of the graph's private pool is ever
     reused inside the graph) - if B is clean and A is not, the mismatch is allocator reuse across the forked streams
  C  capture without record_stream() calls but with keep-alive (isolates the deferred-free path)
Prints one line per variant: number of replays that differ and which outputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops, synth
from object_keypoints_amd.perception.models import KeypointNet
from object_keypoints_amd.perception import pipeline as pp
from object_keypoints_amd.perception.utils import camera_utils as cu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 20
net = KeypointNet(features=128, heatmaps_out=3, compute_dtype=torch.bfloat16)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.fill_state_dict(shapes, seed=0).items()})
net.eval().cuda()
params = cu.load_calibration_params(os.path.join(REPO, "config", "calibration.yaml"))
camera = cu.FisheyeCamera(params["K"], params["D"], params["image_size"]).scale(511 / 720)
camera = camera.cut(np.array([(511 / 720 * 1280 - 511.0) / 2.0, 0.0])).scale(64 / 511)
pipe = pp.BatchedKeypointPipeline(net, {"keypoint_config": [1, 3]}, camera, capacity=128)
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
x = torch.randn(n, 3, 511, 511, device="cuda", generator=gen)
KEYS = ("heat", "depth", "centers", "count", "xyc")
ops.SIDE_STREAMS = True
with torch.no_grad():
    for _ in range(2): base = pipe.forward_device(x)
    base = {k: base[k].clone() for k in KEYS}
    again = pipe.forward_device(x)
    print("eager side=True reproducible:", all(torch.equal(base[k], again[k]) for k in KEYS))

orig_empty = ops.Act.empty
keepalive = []
def empty_keep(n_, h, w, c, dtype, device):
    a = orig_empty(n_, h, w, c, dtype, device)
    keepalive.append(a.t)
    return a

def run(tag, keep):
    keepalive.clear()
    if keep: ops.Act.empty = staticmethod(empty_keep)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.no_grad(), torch.cuda.graph(g):
            out = pipe.forward_device(x)
    finally:
        ops.Act.empty = staticmethod(orig_empty)
    bad = {}
    for r in range(R):
        g.replay(); torch.cuda.synchronize()
        for k in KEYS:
            if not torch.equal(out[k], base[k]):
                bad.setdefault(k, 0); bad[k] += 1
    print(f"{tag}: {R} replays, mismatches per output: {bad or 'none'}  (tensors kept alive: {len(keepalive)})", flush=True)
    del g, out

run("A plain capture, side streams", keep=False)
run("B keep-alive capture, side streams", keep=True)
ops.SIDE_STREAMS = False
run("D plain capture, no side streams", keep=False)

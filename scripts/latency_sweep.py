"""Step time of the whole hot path (frames resident in HBM -> 3D keypoints) by batch size, eager launches and a captured
hipGraph (BatchedKeypointPipeline.capture).  The reference processes ONE frame per call; small batches are its operating point.
usage: latency_sweep.py [batches=1,2,4,8,16,32,64] [dtype=bf16|f16|f32|f32x3|f32mix]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from object_keypoints_amd.perception.pipeline import BatchedKeypointPipeline
from object_keypoints_amd.perception.utils import camera_utils as cu

batches = [1, 2, 4, 8, 16, 32, 64]
precision = "bf16"
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k == "batches": batches = [int(x) for x in v.split(",")]
    if k == "dtype": precision = v
dev = torch.device("cuda", 0)
from object_keypoints_amd import ops
compute = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f32x3": ops.F32X3, "f32mix": ops.F32MIX}[precision]
net = bench.build_net(compute).to(dev)
print(f"compute dtype {precision}")
params = cu.load_calibration_params(os.path.join(bench.REPO, "config", "calibration.yaml"))
camera = cu.FisheyeCamera(params["K"], params["D"], params["image_size"]).scale(511 / 720)
camera = camera.cut(np.array([(511 / 720 * 1280 - 511.0) / 2.0, 0.0])).scale(64 / 511)
pipe = BatchedKeypointPipeline(net, {"keypoint_config": [1, 3]}, camera, capacity=64)
print(f"{'batch':>5} {'eager ms':>9} {'graph ms':>9} {'frames/s eager':>15} {'frames/s graph':>15}")
with torch.no_grad():
    for b in batches:
        frames = torch.randn((b, 3, 511, 511), device=dev)
        for _ in range(3): pipe.forward_device(frames)
        torch.cuda.synchronize()
        reps = 50 if b <= 8 else 20
        t0 = time.perf_counter()
        for _ in range(reps): pipe.forward_device(frames)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / reps * 1e3
        graph, static_in, _ = pipe.capture(frames)
        for _ in range(3): graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            static_in.copy_(frames); graph.replay()
        torch.cuda.synchronize()
        g = (time.perf_counter() - t0) / reps * 1e3
        print(f"{b:5d} {eager:9.3f} {g:9.3f} {b / eager * 1e3:15.0f} {b / g * 1e3:15.0f}")

"""Batch-1 step: eager and graph timings plus (under rocprofv3 --kernel-trace) the per-kernel durations of one frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from object_keypoints_amd.perception.pipeline import BatchedKeypointPipeline
from object_keypoints_amd.perception.utils import camera_utils as cu
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
net = bench.build_net(torch.bfloat16).to(dev)
params = cu.load_calibration_params(os.path.join(bench.REPO, "config", "calibration.yaml"))
camera = cu.FisheyeCamera(params["K"], params["D"], params["image_size"]).scale(511 / 720)
camera = camera.cut(np.array([(511 / 720 * 1280 - 511.0) / 2.0, 0.0])).scale(64 / 511)
pipe = BatchedKeypointPipeline(net, {"keypoint_config": [1, 3]}, camera, capacity=64)
frames = torch.randn((b, 3, 511, 511), device=dev)
with torch.no_grad():
    for _ in range(3): pipe.forward_device(frames)
    torch.cuda.synchronize()
    step = pipe.capture(frames)
    for _ in range(5): step.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): step.replay()
    torch.cuda.synchronize()
    print(f"batch {b}: graph {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")

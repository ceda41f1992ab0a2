"""End-to-end latency of the reference's production call (`objects, heatmap = pipeline(frame)`, host frame in, host objects out) through
the drop-in LearnedKeypointTrackingPipeline, fp32 (the reference's precision) and bf16, plus where the host time goes (cProfile)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from object_keypoints_amd import synth
from object_keypoints_amd.perception import pipeline as pp
from object_keypoints_amd.perception.utils import camera_utils as cu
REPO = bench.REPO
params = cu.load_calibration_params(os.path.join(REPO, "config", "calibration.yaml"))
camera = cu.FisheyeCamera(params["K"], params["D"], params["image_size"]).scale(511 / 720)
camera = camera.cut(np.array([(511 / 720 * 1280 - 511.0) / 2.0, 0.0])).scale(64 / 511)
cfg = {"keypoint_config": [1, 3]}
scene = synth.bump_scene([1, 3], n_objects=2, seed=7, index=0)
frame = torch.from_numpy(synth.frames(1, seed=3))
for dtype in (torch.float32, torch.bfloat16):
    net = bench.build_net(torch.float32)
    path = "/tmp/okp_probe_model.pt"
    torch.save(net.state_dict(), path)
    pipe = pp.LearnedKeypointTrackingPipeline(path, True, [64, 64], None, cfg, compute_dtype=dtype)
    pipe.reset(camera)
    for _ in range(5): pipe(frame)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): objs, heat = pipe(frame)
    torch.cuda.synchronize()
    print(f"{dtype}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per frame end to end ({len(objs)} objects)")
pr = cProfile.Profile(); pr.enable()
for _ in range(50): pipe(frame)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(18); print(s.getvalue()[:4500])

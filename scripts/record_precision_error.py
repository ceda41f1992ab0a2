"""Measured error of the 16-bit configurations against the reference's golden outputs (tests/golden/net_*.npz), written as JSON:
per precision and golden case max / mean / p99 |error| of heat, depth and centre maps, and how the peak sets of the heat
maps (the quantity with a bit-exact contract in fp32) compare.  Run on the GPU box; the result is committed as
tests/golden/precision_measured.json and tests/test_gpu_net.py bounds the 16-bit paths at 2x these figures."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
import numpy as np, torch
import cases
from object_keypoints_amd import ops, synth
from object_keypoints_amd.perception.models import KeypointNet

def stats(got, ref):
    e = np.abs(got.astype(np.float64) - ref.astype(np.float64)).ravel()
    return {"max": float(e.max()), "mean": float(e.mean()), "p99": float(np.quantile(e, 0.99)), "ref_absmax": float(np.abs(ref).max())}

from object_keypoints_amd.perception.utils import camera_utils as cu
_p = cu.load_calibration_params(os.path.join(REPO, "config", "calibration.yaml"))
_c = cu.FisheyeCamera(_p["K"], _p["D"], _p["image_size"]).scale(511 / 720)
CAM = _c.cut(np.array([(511 / 720 * 1280 - 511.0) / 2.0, 0.0])).scale(64 / 511).okp()
out = {"device": torch.cuda.get_device_name(0), "note": "HIP path vs golden outputs of the reference (fp32 CPU), one 511x511 frame per case"}
for name in sorted(cases.NET_CASES):
    case = cases.NET_CASES[name]
    g = np.load(os.path.join(REPO, "tests", "golden", f"net_{name}.npz"))
    x = torch.from_numpy(synth.frames(1, seed=case["frame_seed"], start=case["frame_index"])).cuda()
    ref_peaks = None
    for tag, dt in (("f32", torch.float32), ("f16", torch.float16), ("bf16", torch.bfloat16)):
        net = KeypointNet(features=128, heatmaps_out=case["heatmaps_out"], compute_dtype=dt)
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.fill_state_dict(shapes, seed=case["weight_seed"]).items()})
        net.eval().cuda()
        with torch.no_grad():
            heat, depth, centers = net.deployed(x)
            count, yx, _ = ops.peak_nms(heat, cap=4096)
            gcount, gyx, _ = ops.peak_nms(torch.from_numpy(g["heat"]).cuda(), cap=4096)
        row = {"heat": stats(heat.cpu().numpy(), g["heat"]), "depth": stats(depth.cpu().numpy(), g["depth"]),
               "centers": stats(centers.cpu().numpy(), g["centers"])}
        # peak sets on this precision's heat map vs on the golden heat map (same device kernel, so only the maps differ),
        # and the 3D points lifted at the peaks both have in common (this precision's centroid + depth vs the golden's)
        with torch.no_grad():
            _, _, xyc = ops.peak_nms(heat, cap=4096)
            _, _, gxyc = ops.peak_nms(torch.from_numpy(g["heat"]).cuda(), cap=4096)
            pts = ops.lift_peaks(CAM, count, xyc, depth, 63, 63).cpu().numpy()
            gpts = ops.lift_peaks(CAM, gcount, gxyc, torch.from_numpy(g["depth"]).cuda(), 63, 63).cpu().numpy()
        inter = union = 0
        d3 = []
        for k in range(heat.shape[1]):
            la = yx[0, k, :int(count[0, k])].cpu().numpy().tolist(); lb = gyx[0, k, :int(gcount[0, k])].cpu().numpy().tolist()
            a = {tuple(p): i for i, p in enumerate(la)}; b = {tuple(p): i for i, p in enumerate(lb)}
            inter += len(a.keys() & b.keys()); union += len(a.keys() | b.keys())
            for key in a.keys() & b.keys():
                d3.append(float(np.abs(pts[0, k, a[key], :3] - gpts[0, k, b[key], :3]).max()))
        row["peaks"] = {"golden": int(gcount.sum()), "found": int(count.sum()), "jaccard": inter / max(union, 1),
                        "p_C_max_m_at_common_peaks": max(d3), "p_C_mean_m_at_common_peaks": float(np.mean(d3))}
        out.setdefault(name, {})[tag] = row
        print(name, tag, json.dumps(row), flush=True)
dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "precision_measured.json")
os.makedirs(os.path.dirname(dst), exist_ok=True)
with open(dst, "w") as f:
    json.dump(out, f, indent=1)

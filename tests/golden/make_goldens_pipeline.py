"""Golden vectors of the post-network pipeline, produced by the REFERENCE classes
(perception/pipeline.py) on procedurally generated maps (cases.pipeline_case).

cv2 is not installed in the build container, so the one cv2 call on this path
(FisheyeCamera.undistort -> cv2.fisheye.undistortPoints) is served by oracle.geometry's
restatement, patched into the reference camera class; everything else that runs is reference code.
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def _f(a):
    return np.asarray(a, dtype=np.float64).tolist()


def main(ref):
    import torch
    import cases
    from oracle import geometry as og
    pl = ref.pipeline
    out = {"extraction": {}, "objects": {}, "pipeline": {}}
    for name in cases.PIPELINE_CASES:
        c = cases.pipeline_case(name)
        cfg = {"keypoint_config": c["config"]}
        comp = pl.KeypointExtractionComponent(cfg, [64, 64])
        points, conf = comp(c["heat"][None])
        maps = []
        for k in range(c["heat"].shape[0]):
            # indices exactly as perception/pipeline.py:69-73 computes them
            prob = torch.tensor(c["heat"][k].astype(np.float32))[None, None]
            box = torch.nn.functional.conv2d(prob, comp.kernel, bias=None, stride=1, padding=2)
            sup = ref.models.nms(box)
            idx = comp.image_indices[sup[0, 0] > 0.5].numpy()
            maps.append({"indices": idx.astype(int).tolist(),
                         "points": [_f(p) for p in points[0][k]],
                         "confidence": [float(x) for x in conf[0][k]]})
        out["extraction"][name] = maps
        print(f"  extraction {name}: peaks per map {[len(m['indices']) for m in maps]}")
    K_path = os.path.join(REPO, "config", "calibration.yaml")
    cam_ref = ref.camera_utils
    cam_ref.FisheyeCamera.undistort = lambda self, xy: og.fisheye_undistort(xy, self.K, self.D, P=self.K)
    params = cam_ref.load_calibration_params(K_path)
    camera = cam_ref.FisheyeCamera(params["K"], params["D"], params["image_size"]).scale(511 / 720)
    offset = np.array([(511 / 720 * 1280 - 511.0) / 2.0, 0.0])
    camera_small = camera.cut(offset).scale(64 / 511)
    out["camera_small"] = {"K": _f(camera_small.K), "D": _f(camera_small.D), "image_size": _f(camera_small.image_size)}
    for name in cases.OBJECT_CASES:
        c = cases.pipeline_case(name)
        cfg = {"keypoint_config": c["config"]}
        comp = pl.KeypointExtractionComponent(cfg, [64, 64])
        points, conf = comp(c["heat"][None])
        objs = pl.ObjectExtraction(cfg, [64, 64])(points[0], conf[0], c["centers"])
        out["objects"][name] = [{"center": _f(o["center"]),
                                 "heatmap_points": [_f(p) for p in o["heatmap_points"]],
                                 "p_centers": [_f(p) for p in o["p_centers"]]} for o in objs]
        pipe = pl.ObjectKeypointPipeline([64, 64], None, cfg)
        pipe.reset(camera_small)
        res = pipe(torch.from_numpy(c["heat"][None]), torch.from_numpy(c["depth"][None]), torch.from_numpy(c["centers"][None]))
        out["pipeline"][name] = [{"keypoints": [_f(k) for k in o["keypoints"]],
                                  "p_C": [None if p is None else _f(p) for p in o["p_C"]]} for o in res]
        print(f"  objects {name}: {len(objs)} objects, keypoints per object {[[len(k) for k in o['keypoints']] for o in res]}")
    with open(os.path.join(HERE, "pipeline.json"), "w") as f:
        json.dump(out, f)
    print("  wrote pipeline.json", os.path.getsize(os.path.join(HERE, "pipeline.json")), "bytes")

"""GPU parity of the post-network path (peak-NMS, lifting, DLT) through the C ABI:
against golden vectors produced by the reference, against the oracle on seeded inputs, and by
size-independent properties at the benchmark's batch size."""
import json
import os

import numpy as np
import pytest
import torch

import cases
from oracle import geometry as og
from oracle import pipeline as op

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALIB = os.path.join(REPO, "config", "calibration.yaml")


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(REPO, "tests", "golden", "pipeline.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def known():
    with open(os.path.join(REPO, "tests", "golden", "known_answers.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", cases.PIPELINE_CASES)
def test_peak_nms_matches_reference_golden(golden, name):
    from object_keypoints_amd import ops
    c = cases.pipeline_case(name)
    heat = torch.from_numpy(c["heat"][None]).cuda()
    count, yx, xyc = [t.cpu().numpy() for t in ops.peak_nms(heat, cap=128)]
    for k, g in enumerate(golden["extraction"][name]):
        n = len(g["indices"])
        assert int(count[0, k]) == n
        assert yx[0, k, :n].tolist() == g["indices"]          # bit-exact indices, row-major order
        if n:
            np.testing.assert_allclose(xyc[0, k, :n, :2], np.array(g["points"]), rtol=2e-6, atol=2e-5)   # fp32 centroid: ~4 ulp at 64 px
            np.testing.assert_allclose(xyc[0, k, :n, 2], np.array(g["confidence"]), rtol=1e-6, atol=1e-6)


def test_peak_capacity_overflow_is_explicit(golden):
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.pipeline import KeypointExtractionComponent
    c = cases.pipeline_case("noise")
    heat = torch.from_numpy(c["heat"][None]).cuda()
    count, yx, _ = [t.cpu().numpy() for t in ops.peak_nms(heat, cap=16)]
    for k, g in enumerate(golden["extraction"]["noise"]):
        assert int(count[0, k]) == len(g["indices"])           # total is reported even when it exceeds the capacity
        assert yx[0, k].tolist() == g["indices"][:16]          # the first `cap` peaks in row-major order are kept
    comp = KeypointExtractionComponent({"keypoint_config": c["config"]}, [64, 64], capacity=16)
    with pytest.raises(ops.OkpError):
        comp(c["heat"][None])


def test_batched_pipeline_flags_capacity_overflow():
    """The batched pipeline's fixed capacities can truncate where the reference keeps every peak: `overflow` (device flag,
    okp_capacity_overflow) says so and objects() raises instead of returning a truncated frame."""
    from object_keypoints_amd import ops, synth
    from object_keypoints_amd.perception import pipeline as pp
    from object_keypoints_amd.perception.utils import camera_utils as cu
    cam_o = op.eval_camera(CALIB)
    cam = cu.FisheyeCamera(cam_o.K, cam_o.D, cam_o.image_size)
    c = cases.pipeline_case("noise")                                   # about a hundred peaks per map
    k = c["heat"].shape[0]
    heat = torch.from_numpy(c["heat"][None]).cuda()
    depth = torch.ones_like(heat)
    centers = torch.zeros((1, k - 1, 2, 64, 64), device="cuda")
    cfg = {"keypoint_config": c["config"]}
    small = pp.BatchedKeypointPipeline(None, cfg, cam, capacity=16)
    out = small.postprocess_device(heat, depth, centers)
    assert bool(out["overflow"])
    with pytest.raises(ops.OkpError):
        small.objects(out, 0)
    s = synth.bump_scene([1, 3], n_objects=2, seed=7, index=0)
    fine = pp.BatchedKeypointPipeline(None, {"keypoint_config": [1, 3]}, cam, capacity=64)
    out = fine.postprocess_device(*[torch.from_numpy(s[key][None]).cuda() for key in ("heat", "depth", "centers")])
    assert not bool(out["overflow"])
    many = pp.BatchedKeypointPipeline(None, {"keypoint_config": [1, 3]}, cam, capacity=64, max_objects=1)    # two centres, one object slot
    assert bool(many.postprocess_device(*[torch.from_numpy(s[key][None]).cuda() for key in ("heat", "depth", "centers")])["overflow"])


def test_nms_function_matches_oracle():
    from object_keypoints_amd.perception.models import nms
    from oracle import net as onet
    x = torch.from_numpy(np.stack([cases.pipeline_case("special")["heat"], cases.pipeline_case("noise")["heat"]]))
    got = nms(x.cuda()).cpu()
    assert torch.equal(got, onet.nms(x))
    assert torch.equal(nms(x.cuda(), size=3).cpu(), onet.nms(x, size=3))


@pytest.mark.parametrize("name", cases.PIPELINE_CASES)
def test_extraction_component_matches_reference_golden(golden, name):
    from object_keypoints_amd.perception.pipeline import KeypointExtractionComponent
    c = cases.pipeline_case(name)
    comp = KeypointExtractionComponent({"keypoint_config": c["config"]}, [64, 64], capacity=128)
    points, conf = comp(c["heat"][None])
    assert len(points) == 1 and len(points[0]) == c["heat"].shape[0]
    for k, g in enumerate(golden["extraction"][name]):
        assert len(points[0][k]) == len(g["points"])
        for p, gp in zip(points[0][k], g["points"]):
            assert p.dtype == np.float32 and p.shape == (2,)
            np.testing.assert_allclose(p, gp, rtol=2e-6, atol=2e-5)


@pytest.mark.parametrize("name", cases.OBJECT_CASES)
def test_object_pipeline_matches_reference_golden(golden, name):
    from object_keypoints_amd.perception import pipeline as pp
    from object_keypoints_amd.perception.utils import camera_utils as cu
    c = cases.pipeline_case(name)
    cfg = {"keypoint_config": c["config"]}
    params = cu.load_calibration_params(CALIB)
    camera = cu.FisheyeCamera(params["K"], params["D"], params["image_size"]).scale(511 / 720)
    camera_small = camera.cut(np.array([(511 / 720 * 1280 - 511.0) / 2.0, 0.0])).scale(64 / 511)
    np.testing.assert_allclose(camera_small.K, np.array(golden["camera_small"]["K"]), rtol=1e-12)
    pipe = pp.ObjectKeypointPipeline([64, 64], None, cfg)
    pipe.reset(camera_small)
    res = pipe(torch.from_numpy(c["heat"][None]), torch.from_numpy(c["depth"][None]), torch.from_numpy(c["centers"][None]))
    gp = golden["pipeline"][name]
    assert len(res) == len(gp)
    for o, go in zip(res, gp):
        assert set(o.keys()) == {"p_centers", "keypoints", "p_C"}
        for a, b in zip(o["keypoints"], go["keypoints"]):
            np.testing.assert_allclose(a, np.array(b).reshape(np.asarray(a).shape), atol=2e-5)
        for a, b in zip(o["p_C"], go["p_C"]):
            if b is None:
                assert a is None
            else:
                assert a.dtype == np.float64
                np.testing.assert_allclose(a, np.array(b), rtol=0, atol=1e-4)      # north_star: 1e-4 m
                assert np.abs(a - np.array(b)).max() < 1e-6
    # the batched, device-resident form gives the same objects
    batched = pp.BatchedKeypointPipeline(None, cfg, camera_small)
    out = batched.postprocess_device(torch.from_numpy(c["heat"][None]).cuda(), torch.from_numpy(c["depth"][None]).cuda(),
                                     torch.from_numpy(c["centers"][None]).cuda())
    objs = batched.objects(out, 0)           # device-side grouping (okp_group_objects)
    assert len(objs) == len(res)
    for o, r in zip(objs, res):
        for a, b in zip(o["p_C"], r["p_C"]):
            assert (a is None) == (b is None)
            if a is not None:
                np.testing.assert_array_equal(a, b)
        for a, b in zip(o["keypoints"], r["keypoints"]):
            assert np.asarray(a).shape == np.asarray(b).shape
            np.testing.assert_array_equal(a, b)
        assert len(o["p_centers"]) == len(r["p_centers"])
        for a, b in zip(o["p_centers"], r["p_centers"]):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)


def test_triangulation_known_answer_and_oracle(known):
    from object_keypoints_amd.perception.pipeline import TriangulationComponent
    from object_keypoints_amd.perception.utils import camera_utils as cu
    stereo = cu.StereoCamera.from_file(CALIB)
    tri = TriangulationComponent()
    tri.reset(stereo)
    kp = np.array(known["keypoints_distinct"])
    pts = np.concatenate([kp.mean(axis=0)[None], kp])
    p_w = tri(np.array(known["points_left_distinct"]), np.array(known["points_right_distinct"]))
    assert p_w.shape == (4, 3)
    assert np.linalg.norm(p_w - pts, axis=1).max() < known["triangulation_tolerance_m"]      # the reference test's bound
    assert np.linalg.norm(p_w - pts, axis=1).max() < 1e-4                                      # north_star bound
    # noisy correspondences: Hartley-Sturm + DLT against the oracle restatement
    ostereo = og.StereoCamera.from_file(CALIB)
    rng = np.random.default_rng(3)
    X = np.stack([rng.uniform(-0.4, 0.4, 64), rng.uniform(-0.25, 0.25, 64), rng.uniform(0.6, 2.0, 64)], axis=1)
    pl = ostereo.left_camera.project(X) + rng.normal(0, 0.5, (64, 2))
    pr = ostereo.right_camera.project(X, ostereo.T_RL) + rng.normal(0, 0.5, (64, 2))
    got = stereo.triangulate(pl, pr)
    want = ostereo.triangulate(pl, pr)
    assert np.abs(got - want).max() < 1e-4
    got2 = stereo.triangulate(pl, pr, correct_matches=False)
    want2 = ostereo.triangulate(pl, pr, correct=False)
    assert np.abs(got2 - want2).max() < 1e-6


@pytest.mark.parametrize("m", [1, 7, 8, 9, 61, 256, 257, 700])
def test_triangulation_wavefront_and_thread_kernels_agree_with_the_oracle(m):
    """okp_triangulate_dlt has two kernels: up to 256 pairs the wavefront form (eight lanes per pair: Durand-Kerner with one root per lane,
    the seven Hartley-Sturm candidates reduced over the octet, the 4x4 Jacobi with both column rotations of a round in parallel), above
    that one thread per pair.  Both against the oracle (cv2.correctMatches + cv2.triangulatePoints restated, camera_utils.py:92-110) on
    noisy correspondences, on pair counts around the octet / wave / switch-over boundaries; with and without the correction."""
    import os
    from object_keypoints_amd.perception.utils import camera_utils as cu
    from oracle import geometry as og
    calib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config", "calibration.yaml")
    stereo, ostereo = cu.StereoCamera.from_file(calib), og.StereoCamera.from_file(calib)
    rng = np.random.default_rng(100 + m)
    X = np.stack([rng.uniform(-0.4, 0.4, m), rng.uniform(-0.25, 0.25, m), rng.uniform(0.5, 2.0, m)], axis=1)
    pl = ostereo.left_camera.project(X) + rng.normal(0, 0.5, (m, 2))
    pr = ostereo.right_camera.project(X, ostereo.T_RL) + rng.normal(0, 0.5, (m, 2))
    got, want = stereo.triangulate(pl, pr), ostereo.triangulate(pl, pr)
    assert got.shape == (m, 3) and np.isfinite(got).all()
    assert np.abs(got - want).max() < 1e-6                                  # (measured 3e-14: both kernels find the same roots / null vector)
    assert np.abs(stereo.triangulate(pl, pr, correct_matches=False) - ostereo.triangulate(pl, pr, correct=False)).max() < 1e-6
    assert np.array_equal(stereo.triangulate(pl, pr), got)                  # run to run


def test_stereo_triangulate_undistorts_with_the_fisheye_model_whatever_the_camera_class():
    """The reference's StereoCamera.triangulate calls cv2.fisheye.undistortPoints with the cameras' K / D unconditionally
    (camera_utils.py:92-97): a stereo pair of RadTanPinholeCamera objects therefore gives the same 3D points as FisheyeCamera
    objects with the same K / D.  per_camera_model=True is this build's opt-in for undistorting by each camera's own model."""
    from object_keypoints_amd.perception.utils import camera_utils as cu
    p = cu.load_calibration_params(CALIB)
    fish = cu.StereoCamera(cu.FisheyeCamera(p["K"], p["D"], p["image_size"]), cu.FisheyeCamera(p["Kp"], p["Dp"], p["image_size"]), p["T_RL"])
    rad = cu.StereoCamera(cu.RadTanPinholeCamera(p["K"], p["D"], p["image_size"]), cu.RadTanPinholeCamera(p["Kp"], p["Dp"], p["image_size"]), p["T_RL"])
    rng = np.random.default_rng(5)
    X = np.stack([rng.uniform(-0.4, 0.4, 32), rng.uniform(-0.25, 0.25, 32), rng.uniform(0.6, 2.0, 32)], axis=1)
    pl = fish.left_camera.project(X)
    pr = fish.right_camera.project(X, p["T_RL"])
    a = fish.triangulate(pl, pr)
    b = rad.triangulate(pl, pr)
    assert np.array_equal(a, b)                                  # same kernel, same model, same numbers
    assert np.abs(a - X).max() < 1e-4
    c = rad.triangulate(pl, pr, per_camera_model=True)           # the cameras' own (radtan) model: a different answer
    assert np.abs(c - a).max() > 1e-3


def test_undistort_matches_oracle():
    from object_keypoints_amd.perception.utils import camera_utils as cu
    p = cu.load_calibration_params(CALIB)
    cam = cu.FisheyeCamera(p["K"], p["D"], p["image_size"])
    rng = np.random.default_rng(0)
    xy = np.stack([rng.uniform(0, 1280, 500), rng.uniform(0, 720, 500)], axis=1)
    got = cam.undistort(xy)
    want = og.fisheye_undistort(xy.astype(np.float32).astype(np.float64), p["K"], p["D"], P=p["K"])
    assert got.dtype == np.float64
    assert np.abs(got - want).max() < 1e-9
    assert cam.undistort(xy.astype(np.float32)).dtype == np.float32


def test_radtan_camera_matches_oracle():
    """RadTanPinholeCamera: the device undistortion (okp_camera_undistort, model radtan) against the oracle's restatement of
    cv2.undistortPoints, and depth lifting (DetectionToPoint) through such a camera."""
    from object_keypoints_amd.perception import pipeline as pp
    from object_keypoints_amd.perception.utils import camera_utils as cu
    K = np.array([[42.0, 0.0, 32.0], [0.0, 41.5, 32.0], [0.0, 0.0, 1.0]])
    D = np.array([-0.28, 0.07, 0.0006, -0.0002])
    cam, ocam = cu.RadTanPinholeCamera(K, D, [64, 64]), og.RadTanPinholeCamera(K, D, [64, 64])
    rng = np.random.default_rng(2)
    xy = np.stack([rng.uniform(0, 64, 300), rng.uniform(0, 64, 300)], axis=1)
    got = cam.undistort(xy)
    want = ocam.undistort(xy.astype(np.float32).astype(np.float64))
    assert got.dtype == np.float64 and np.abs(got - want).max() < 1e-10
    X = np.array([[0.1, -0.05, 1.0], [-0.2, 0.1, 1.5]])
    np.testing.assert_allclose(cam.project(X), ocam.project(X), rtol=0, atol=1e-12)
    depth = rng.uniform(0.3, 1.5, (64, 64)).astype(np.float32)
    pts = xy[:40].astype(np.float32)
    d2p, od2p = pp.DetectionToPoint(), op.DetectionToPoint()
    d2p.reset(cam); od2p.reset(ocam)
    assert np.abs(d2p(pts, depth) - od2p(pts, depth)).max() < 1e-6


def test_batch64_properties():
    """BASELINE batch size: every frame of a 64-frame batch gives the peaks it gives alone (frames are
    independent), peaks are sorted row-major, and lifted points agree with the oracle on sampled frames."""
    from object_keypoints_amd import ops, synth
    from object_keypoints_amd.perception.utils import camera_utils as cu
    scenes = [synth.bump_scene([1, 1, 1], n_objects=1 + (i % 4), seed=11, index=i) for i in range(64)]
    heat = torch.from_numpy(np.stack([s["heat"] for s in scenes])).cuda()
    depth = torch.from_numpy(np.stack([s["depth"] for s in scenes])).cuda()
    count, yx, xyc = ops.peak_nms(heat, cap=64)
    cam_o = op.eval_camera(CALIB)
    cam = cu.FisheyeCamera(cam_o.K, cam_o.D, cam_o.image_size)
    pts = ops.lift_peaks(cam.okp(), count, xyc, depth, 63, 63).cpu().numpy()
    count, yx, xyc = count.cpu().numpy(), yx.cpu().numpy(), xyc.cpu().numpy()
    assert count.sum() > 64 * 4
    lin = yx[..., 0].astype(np.int64) * 64 + yx[..., 1]
    for n in range(64):
        for k in range(4):
            c = int(count[n, k])
            assert (np.diff(lin[n, k, :c]) > 0).all()                       # row-major order
            assert np.isnan(pts[n, k, c:]).all() and not np.isnan(pts[n, k, :c]).any()
    d2p = op.DetectionToPoint(); d2p.reset(cam_o)
    for n in (0, 17, 63):
        single = ops.peak_nms(heat[n:n + 1], cap=64)
        assert torch.equal(single[1].cpu(), torch.from_numpy(yx[n:n + 1]))
        for k in range(4):
            idx = op.peak_indices(scenes[n]["heat"][k])
            assert yx[n, k, :len(idx)].tolist() == idx.tolist()
            if len(idx):
                want = d2p(xyc[n, k, :len(idx), :2], scenes[n]["depth"][k])
                assert np.abs(pts[n, k, :len(idx), :3] - want).max() < 1e-6


def test_device_grouping_matches_oracle_on_a_batch():
    """okp_group_objects on 64 multi-object frames vs the oracle's ObjectExtraction, frame by frame; the k-means
    branch (more votes than configured for a multi-instance type) is the device reduction and is compared as a set."""
    from object_keypoints_amd import synth
    from object_keypoints_amd.perception import pipeline as pp
    from object_keypoints_amd.perception.utils import camera_utils as cu
    cfg = {"keypoint_config": [1, 3]}
    scenes = [synth.bump_scene([1, 3], n_objects=1 + (i % 3), seed=23, index=i) for i in range(64)]
    heat = torch.from_numpy(np.stack([s["heat"] for s in scenes])).cuda()
    depth = torch.from_numpy(np.stack([s["depth"] for s in scenes])).cuda()
    centers = torch.from_numpy(np.stack([s["centers"] for s in scenes])).cuda()
    cam_o = op.eval_camera(CALIB)
    batched = pp.BatchedKeypointPipeline(None, cfg, cu.FisheyeCamera(cam_o.K, cam_o.D, cam_o.image_size))
    out = batched.postprocess_device(heat, depth, centers)
    opipe = op.ObjectKeypointPipeline([64, 64], None, cfg)
    opipe.reset(cam_o)
    checked = 0
    for n in range(64):
        want = opipe(scenes[n]["heat"][None], scenes[n]["depth"][None], scenes[n]["centers"][None])
        got = batched.objects(out, n)
        assert len(got) == len(want)
        for g, w_ in zip(got, want):
            for a, b in zip(g["keypoints"], w_["keypoints"]):
                a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
                assert a.shape == b.shape
                if a.size:                 # same SET of points (the k-means branch has no defined order)
                    d = np.linalg.norm(a[:, None, :] - b[None, :, :], axis=2)
                    assert d.min(axis=1).max() < 1e-4 and d.min(axis=0).max() < 1e-4
            for a, b in zip(g["p_C"], w_["p_C"]):
                assert (a is None) == (b is None)
                if a is not None:
                    assert np.abs(np.sort(a, axis=0) - np.sort(b, axis=0)).max() < 1e-4
            checked += 1
    assert checked > 64


def test_device_kmeans_reduction_on_double_detections():
    """64 valve frames ([1, 3]) in which every instance of the three-instance type is detected TWICE: the object receives six votes for
    three instances, the case the reference reduces with an unseeded sklearn KMeans (pipeline.py:143-148).  okp_group_objects reduces
    them on the device (deterministic Lloyd iteration); BatchedKeypointPipeline.objects() never reaches scikit-learn, and the centres are
    the oracle's k-means centres as a set, their 3D points the oracle's (DetectionToPoint on the centres), launch after launch the same bits."""
    import sys
    from object_keypoints_amd import synth
    from object_keypoints_amd.perception import pipeline as pp
    from object_keypoints_amd.perception.utils import camera_utils as cu
    cfg = {"keypoint_config": [1, 3]}
    scenes = [synth.add_double_detections(synth.bump_scene([1, 3], n_objects=1 + (i % 8 == 7), seed=47, index=i, max_radius=18.0 if i % 8 != 7 else 9.0), 2,
                                          offset=(5.0, 3.0)) for i in range(64)]
    to = lambda key: torch.from_numpy(np.stack([s[key] for s in scenes])).cuda()
    cam_o = op.eval_camera(CALIB)
    batched = pp.BatchedKeypointPipeline(None, cfg, cu.FisheyeCamera(cam_o.K, cam_o.D, cam_o.image_size))
    out = batched.postprocess_device(to("heat"), to("depth"), to("centers"))
    again = batched.postprocess_device(to("heat"), to("depth"), to("centers"))
    assert torch.equal(torch.nan_to_num(out["reduced"], nan=-1.0), torch.nan_to_num(again["reduced"], nan=-1.0))
    assert not bool(out["overflow"])
    opipe = op.ObjectKeypointPipeline([64, 64], None, cfg)
    opipe.reset(cam_o)
    votes = out["n_votes"].cpu().numpy()
    reduced = out["reduced"].cpu().numpy()
    banned = sys.modules.get("sklearn")
    sys.modules["sklearn"] = None                       # importing scikit-learn inside objects() would now raise ImportError
    try:
        got_all = [batched.objects(out, n) for n in range(64)]
    finally:
        if banned is not None:
            sys.modules["sklearn"] = banned
        else:
            del sys.modules["sklearn"]
    n_reduced = 0
    for n in range(64):
        want = opipe(scenes[n]["heat"][None], scenes[n]["depth"][None], scenes[n]["centers"][None])
        got = got_all[n]
        assert len(got) == len(want)
        for o, (g, w_) in enumerate(zip(got, want)):
            # the reduction ran exactly where an object has more votes than instances, and nowhere else
            for i, c in enumerate(cfg["keypoint_config"]):
                ran = not np.isnan(reduced[n, o, i, 0, 0])
                assert ran == (c > 1 and votes[n, o, i] > c)
                n_reduced += ran
            for a, b in zip(g["keypoints"], w_["keypoints"]):
                a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
                assert a.shape == b.shape
                if a.size:
                    d = np.linalg.norm(a[:, None, :] - b[None, :, :], axis=2)
                    assert d.min(axis=1).max() < 1e-4 and d.min(axis=0).max() < 1e-4
            for a, b in zip(g["p_C"], w_["p_C"]):
                assert (a is None) == (b is None)
                if a is not None:
                    assert np.abs(np.sort(a, axis=0) - np.sort(b, axis=0)).max() < 1e-4
    assert n_reduced >= 40
    # a type with more instances than max_per_type cannot be represented: refused, not truncated
    small = pp.BatchedKeypointPipeline(None, cfg, cu.FisheyeCamera(cam_o.K, cam_o.D, cam_o.image_size), max_per_type=2)
    with pytest.raises(pp.OkpError):
        small.objects(small.postprocess_device(to("heat")[:1], to("depth")[:1], to("centers")[:1]), 0)


def test_association_component_matches_oracle_and_reference_contract(known):
    """Product AssociationComponent (device undistort) vs the oracle on random point sets and on the known-answer
    vectors of the reference's tests (test/test_pipeline.py:208-261)."""
    from object_keypoints_amd.perception import pipeline as pp
    from object_keypoints_amd.perception.utils import camera_utils as cu
    from oracle import geometry as og
    from oracle import pipeline as op
    calib = os.path.join(REPO, "config", "calibration.yaml")
    a = known["association"]
    p = og.load_calibration_params(calib)
    stereo, ostereo = cu.StereoCamera.from_file(calib), og.StereoCamera.from_file(calib)
    assoc, oassoc = pp.AssociationComponent(), op.AssociationComponent()
    assoc.reset(stereo); oassoc.reset(ostereo)
    got = assoc(np.array(a["two_same"]["left"]), np.array(a["two_same"]["right"]))
    assert got.tolist() == a["two_same"]["expected"]
    X = np.array(a["keypoints_X"])
    left = ostereo.left_camera.project(X, np.eye(4)) * a["simple_point_scale"]
    right = ostereo.right_camera.project(X, p["T_RL"]) * a["simple_point_scale"]
    rng = np.random.default_rng(1)
    for _ in range(5):
        shuffled = right[rng.permutation(3)]
        np.testing.assert_equal(right, shuffled[assoc(left, shuffled)])
    # random scenes: 3D points in front of the rig, right points shuffled, two distractors added
    for trial in range(4):
        Xr = np.stack([rng.uniform(-0.3, 0.3, 7), rng.uniform(-0.2, 0.2, 7), rng.uniform(0.6, 1.4, 7)], axis=1)
        l = ostereo.left_camera.project(Xr, np.eye(4))
        r = ostereo.right_camera.project(Xr, p["T_RL"])
        perm = rng.permutation(7)
        r2 = np.concatenate([r[perm], rng.uniform(100, 1100, (2, 2))])
        np.testing.assert_allclose(assoc.cost(l, r2), op.epipolar_cost(ostereo, l, r2), rtol=1e-4, atol=5e-3)   # device undistort takes fp32 pixels
        assert assoc(l, r2).tolist() == oassoc(l, r2).tolist()


def test_stereo_keypoints_end_to_end(known):
    """BASELINE config 5 in miniature: known 3D keypoints seen by both cameras of the rig -> Gaussian-bump heat maps in
    the two 160x90... (quarter-resolution) prediction spaces -> device peak extraction (okp_peak_nms) in each view ->
    AssociationComponent -> TriangulationComponent (device undistort + Hartley-Sturm + DLT) -> the 3D points again.
    Sub-pixel centroids of 2 px-wide bumps at quarter resolution bound the error: centimetres at one metre."""
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception import pipeline as pp
    from object_keypoints_amd.perception.utils import camera_utils as cu
    from oracle import geometry as og
    calib = os.path.join(REPO, "config", "calibration.yaml")
    p = og.load_calibration_params(calib)
    scale = 0.25
    oleft = og.FisheyeCamera(p["K"], p["D"], p["image_size"]).scale(scale)
    oright = og.FisheyeCamera(p["Kp"], p["Dp"], p["image_size"]).scale(scale)
    X = np.array([[0.05, 0.02, 1.0], [0.21, 0.10, 1.1], [-0.18, -0.12, 0.9], [0.10, -0.20, 1.2], [-0.25, 0.15, 1.05]])
    pl = oleft.project(X, np.eye(4))
    pr = oright.project(X, p["T_RL"])
    H, W = 180, 320
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")

    def render(points):
        m = np.zeros((H, W), np.float32)
        for x, y in points:
            m += np.exp(-((xs - x) ** 2 + (ys - y) ** 2) / 4.0).astype(np.float32)
        return np.clip(m, 0, 1)

    heat = torch.from_numpy(np.stack([render(pl), render(pr)])[:, None]).cuda()      # [2 views, 1 map, H, W]
    count, yx, xyc = [t.cpu().numpy() for t in ops.peak_nms(heat, cap=16)]
    assert count[:, 0].tolist() == [5, 5]
    left2d, right2d = xyc[0, 0, :5, :2].astype(np.float64), xyc[1, 0, :5, :2].astype(np.float64)
    # the extraction orders peaks row-major: the two views see them in different orders
    stereo = cu.StereoCamera(cu.FisheyeCamera(oleft.K, oleft.D, oleft.image_size), cu.FisheyeCamera(oright.K, oright.D, oright.image_size), p["T_RL"])
    assoc = pp.AssociationComponent(max_distance=3.0)
    assoc.reset(stereo)
    match = assoc(left2d, right2d)
    assert sorted(match.tolist()) == [0, 1, 2, 3, 4]
    tri = pp.TriangulationComponent()
    tri.reset(stereo)
    X_hat = tri(left2d, right2d[match])
    # match each recovered point to the nearest ground-truth point
    d = np.linalg.norm(X_hat[:, None] - X[None], axis=2)
    assert sorted(d.argmin(axis=1).tolist()) == [0, 1, 2, 3, 4]
    assert d.min(axis=1).max() < 0.05, d.min(axis=1)
    # and the 2D centroids are sub-pixel accurate in both views
    e_l = np.linalg.norm(left2d[:, None] - pl[None], axis=2).min(axis=1).max()
    e_r = np.linalg.norm(right2d[:, None] - pr[None], axis=2).min(axis=1).max()
    assert e_l < 0.25 and e_r < 0.25


@pytest.mark.parametrize("h,w", [(180, 320), (97, 33), (300, 7), (5, 700)])
def test_peak_nms_on_large_maps_matches_oracle(h, w):
    """Maps that do not fit LDS in one piece are processed in strips: indices, order, centroids and confidences must
    equal the oracle's (the reference's own tests extract from 180x320 predictions, test/test_pipeline.py:97)."""
    from object_keypoints_amd import ops
    from oracle import pipeline as op
    rng = np.random.default_rng(h * 1000 + w)
    maps = np.zeros((3, h, w), np.float32)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    for k in range(3):
        for _ in range(12):
            cy, cx = rng.uniform(0, h - 1), rng.uniform(0, w - 1)
            maps[k] += np.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / 4.0).astype(np.float32)
    maps = np.clip(maps, 0, 1)
    maps[2] += (rng.random((h, w)) < 0.0005).astype(np.float32) * 0.9         # isolated hot pixels (each a 5x5 plateau of tied box sums = 25 peaks)
    maps = np.clip(maps, 0, 1).astype(np.float32)
    count, yx, xyc = [t.cpu().numpy() for t in ops.peak_nms(torch.from_numpy(maps[None]).cuda(), cap=4096)]
    for k in range(3):
        idx = op.peak_indices(maps[k])
        n = idx.shape[0]
        assert int(count[0, k]) == n and n > 0
        assert yx[0, k, :n].tolist() == idx.tolist()
        pts, conf = op.refine_peaks(maps[k], idx)
        np.testing.assert_allclose(xyc[0, k, :n, :2], np.stack(pts), rtol=2e-6, atol=1e-4)
        np.testing.assert_allclose(xyc[0, k, :n, 2], np.array(conf), rtol=1e-6, atol=1e-6)

"""CPU-side checks that need no GPU: the C-ABI library loads and exports every symbol include/okp.h
declares, the product's host logic (ObjectExtraction, weight folding, stem tap layout) agrees with the
reference goldens / the oracle, and the product refuses to run without a device."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import cases

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ensure_built():
    from object_keypoints_amd import build
    return build.build()


def test_library_exports_every_declared_symbol():
    path = _ensure_built()
    header = open(os.path.join(REPO, "include", "okp.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(okp_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 15
    lib = ctypes.CDLL(path)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/okp.h but not exported"
    from object_keypoints_amd import _lib
    bound = {n for n, _, _ in _lib.SIGNATURES}
    assert declared == bound, f"ctypes binding and header disagree: {declared ^ bound}"
    assert _lib.lib().okp_abi_version() == 7 == _lib.OKP_ABI


def test_struct_layouts_match_the_header():
    from object_keypoints_amd import _lib
    assert ctypes.sizeof(_lib.okp_tensor) == 32
    assert ctypes.sizeof(_lib.okp_tap) == 24
    assert ctypes.sizeof(_lib.okp_camera) == 72 and _lib.okp_camera.model.offset == 64
    assert _lib.okp_conv_args.src.offset == 16 and _lib.okp_conv_args.out.offset == 80
    assert _lib.okp_conv_args.res.offset == 128 and _lib.okp_conv_args.dw_out.offset == 184 and _lib.okp_conv_args.n_classes.offset == 248 and _lib.okp_conv_args.out16.offset == 256 and _lib.okp_conv_args.res_is_f16.offset == 288 and _lib.okp_conv_args.out_subsample.offset == 292 and _lib.okp_conv_args.src_pairs.offset == 296 and _lib.okp_conv_args.out_pairs.offset == 300 and ctypes.sizeof(_lib.okp_conv_args) == 304


def test_no_cpu_fallback():
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.models import KeypointNet
    with pytest.raises(ops.OkpError):
        ops.Act(torch.zeros(1, 4, 4, 8))
    if not torch.cuda.is_available():
        net = KeypointNet(features=16, heatmaps_out=3).eval()
        with pytest.raises(ops.OkpError):
            net.deployed(torch.zeros(1, 3, 63, 63))
        from object_keypoints_amd.perception.pipeline import InferenceComponent
        with pytest.raises(ops.OkpError):
            InferenceComponent(net, cuda=False)


def test_bad_plan_arguments_are_reported_without_a_gpu():
    from object_keypoints_amd import _lib
    L = _lib.lib()
    cin = (ctypes.c_int32 * 2)(6, 0)
    st = (ctypes.c_int32 * 2)(1, 1)
    w = np.zeros((8, 6), np.float32)
    tap = (_lib.okp_tap * 1)(_lib.okp_tap(0, 0, 0, w.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
    assert not L.okp_conv_create(0, 1, cin, st, 8, 1, tap, None, 0)
    assert b"16 bytes" in L.okp_last_error()
    assert L.okp_peak_nms(None, 1, 64, 64, 8, None, None, None, None) == -1


def test_fold_bn_and_stem_taps_reproduce_the_oracle_convolution():
    """Host-side weight preparation: folding BN into (w, b) and the 8px x 4ch stem tap layout are linear
    algebra that can be checked on the CPU against the oracle's conv + BN."""
    from object_keypoints_amd import synth
    from object_keypoints_amd.perception import backbone as bb
    from oracle import net as onet
    o = onet.load_synthetic(onet.convolution(7, 3, 16, stride=2), seed=5)
    shapes = {k: tuple(v.shape) for k, v in o.state_dict().items()}
    m = bb.convolution(7, 3, 16, stride=2)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.fill_state_dict(shapes, seed=5).items()})
    w, b = bb.fold_bn(m.conv.weight, m.bn)
    x = torch.from_numpy(synth.normal_like("stemx", (1, 3, 21, 21), 2))
    want = o(x)
    got = torch.relu(torch.nn.functional.conv2d(x, torch.from_numpy(w), torch.from_numpy(b), stride=2, padding=3))
    assert torch.allclose(got, want, atol=1e-5)
    # stem taps: out[co] = sum_r sum_{kx<8,c<4} T_r[co, kx*4+c] * packed[2*ho + r, 2*wo + kx, c]
    packed = torch.zeros(1, 21 + 6, 32, 4)
    packed[0, 3:24, 3:24, :3] = x[0].permute(1, 2, 0)
    taps = []
    for r in range(7):
        t = np.zeros((16, 8, 4), np.float32)
        t[:, :7, :3] = np.transpose(w[:, :, r, :], (0, 2, 1))
        taps.append(torch.from_numpy(t.reshape(16, 32)))
    ho = wo = 11
    acc = torch.zeros(16, ho, wo)
    for r in range(7):
        for i in range(ho):
            for j in range(wo):
                acc[:, i, j] += taps[r] @ packed[0, 2 * i + r, 2 * j:2 * j + 8, :].reshape(32)
    assert torch.allclose(torch.relu(acc + torch.from_numpy(b)[:, None, None]), want[0], atol=1e-4)


@pytest.mark.parametrize("name", cases.OBJECT_CASES)
def test_product_object_extraction_matches_reference_golden(name):
    """ObjectExtraction is host logic in the product too; feed it oracle-extracted peaks."""
    from object_keypoints_amd.perception.pipeline import ObjectExtraction
    from oracle import pipeline as op
    with open(os.path.join(REPO, "tests", "golden", "pipeline.json")) as f:
        golden = json.load(f)
    c = cases.pipeline_case(name)
    cfg = {"keypoint_config": c["config"]}
    points, conf = op.KeypointExtractionComponent(cfg, [64, 64])(c["heat"][None])
    objs = ObjectExtraction(cfg, [64, 64])(points[0], conf[0], c["centers"])
    g = golden["objects"][name]
    assert len(objs) == len(g)
    for o, go in zip(objs, g):
        np.testing.assert_allclose(o["center"], go["center"], atol=2e-5)
        for a, b in zip(o["heatmap_points"], go["heatmap_points"]):
            assert np.asarray(a).shape == np.asarray(b).shape
            if np.asarray(b).size:
                np.testing.assert_allclose(a, b, atol=2e-5)
        np.testing.assert_allclose(np.array(o["p_centers"]), np.array(go["p_centers"]), atol=2e-5)


def test_kmeans_branch_set_level():
    """More detections of a multi-instance type than configured: the reference clusters them with an unseeded
    KMeans (pipeline.py:143-148), so only the set of cluster centres is comparable."""
    from object_keypoints_amd.perception.pipeline import ObjectExtraction
    from oracle import pipeline as op
    cfg = {"keypoint_config": [1, 2]}
    center = [np.array([30.0, 30.0], np.float32)]
    pts = [np.array(p, np.float32) for p in ([20.0, 30.0], [20.6, 30.2], [40.0, 31.0], [40.5, 30.5])]
    keypoints = [center, [np.array([30.0, 20.0], np.float32)], pts]
    conf = [[np.float32(5)], [np.float32(4)], [np.float32(c) for c in (3, 2, 4, 1)]]
    centers = np.zeros((2, 2, 64, 64), np.float32)
    for p in pts + [keypoints[1][0]]:
        x, y = int(round(float(p[0]))), int(round(float(p[1])))
        centers[:, 0, y, x] = 30.0 - (x + 0.5)
        centers[:, 1, y, x] = 30.0 - (y + 0.5)
    got = ObjectExtraction(cfg, [64, 64])(keypoints, conf, centers)[0]["heatmap_points"][1]
    want = op.ObjectExtraction(cfg, [64, 64])(keypoints, conf, centers)[0]["heatmap_points"][1]
    key = lambda a: sorted(map(tuple, np.round(np.asarray(a, dtype=np.float64), 4).tolist()))
    assert key(got) == key(want)


def test_cornernet_backbone_import():
    """load_cornernet_backbone: `module.hg.*` keys of a CornerNet-Squeeze checkpoint land in KeypointNet.backbone
    (the reference does this inside KeypointNet._build_hourglass, perception/models.py:69-78); detector-only keys are ignored."""
    from object_keypoints_amd.perception import models, pipeline
    net = models.KeypointNet(features=32, heatmaps_out=3)
    own = net.backbone.state_dict()
    gen = torch.Generator().manual_seed(3)
    fake = {"module.hg." + k: (torch.rand(v.shape, generator=gen).to(v.dtype) if v.dtype.is_floating_point else v.clone())
            for k, v in own.items()}
    fake["module.tl_modules.0.0.conv.weight"] = torch.zeros(4, 4)       # the detector's corner heads: not ours
    n = pipeline.load_cornernet_backbone(net, fake)
    assert n == len(own)
    for k, v in net.backbone.state_dict().items():
        assert torch.equal(v, fake["module.hg." + k])
    del fake["module.hg.pre.0.conv.weight"]
    with pytest.raises(models.OkpError):
        pipeline.load_cornernet_backbone(net, fake)


def test_evaluation_results_table():
    """perception.evaluation.Results (reference scripts/eval_model.py:129-232): association of predictions with ground
    truth and the statistics of the reference's table, on a synthetic scene with known errors."""
    from object_keypoints_amd.perception.evaluation import Results

    class Cam:                                      # pinhole stand-in: 64x64 frame, everything near the axis is in view
        image_size = np.array([64.0, 64.0])

        def project(self, X):
            return np.stack([32 + 60 * X[:, 0] / X[:, 2], 32 + 60 * X[:, 1] / X[:, 2]], axis=1)

        def in_frame(self, x):
            return np.bitwise_or((x <= 0.0).any(axis=1), (x >= self.image_size).any(axis=1)) == False   # noqa: E712

    scene = np.array([[[0.0, 0.0, 1.0], [0.1, 0.0, 1.0], [0.0, 0.1, 1.0]],
                      [[-0.3, 0.2, 1.2], [-0.2, 0.2, 1.2], [-0.3, 0.3, 1.2]]])
    res = Results()
    res.set_calibration(Cam())
    errs = []
    objects = []
    for o in range(2):
        pts = scene[o].copy()
        pts[1] += np.array([0.01, 0.0, 0.0]); errs.append(0.01)
        pts[2] += np.array([0.0, 0.0, 0.05]); errs.append(0.05)
        errs.append(0.0)
        objects.append({"p_C": [pts[0:1], pts[1:3]]})
    objects[1]["p_C"][1] = [objects[1]["p_C"][1][0], None]         # one keypoint not lifted
    res.add(np.eye(4), objects, scene)
    s = res.summary()
    assert s["points"] == 6 and abs(s["missing"] - 100.0 / 6) < 1e-9
    found = np.array([0.0, 0.01, 0.05, 0.0, 0.01]) * 100
    assert abs(s["mean"] - found.mean()) < 1e-9 and abs(s["std"] - found.std()) < 1e-9
    assert abs(s["< 3cm"] - 4 / 6) < 1e-12
    assert abs(s["25th percentile"] - np.percentile(found, 25)) < 1e-9


def test_product_camera_project_reproduces_reference_known_answers():
    """FisheyeCamera.project of the product (host NumPy) against the vectors of the reference's own test
    (test/test_pipeline.py:9-33 via tests/golden/known_answers.json)."""
    from object_keypoints_amd.perception.utils import camera_utils as cu
    with open(os.path.join(REPO, "tests", "golden", "known_answers.json")) as f:
        known = json.load(f)
    p = cu.load_calibration_params(os.path.join(REPO, "config", "calibration.yaml"))
    left = cu.FisheyeCamera(p["K"], p["D"], p["image_size"])
    right = cu.FisheyeCamera(p["Kp"], p["Dp"], p["image_size"])
    kp = np.array(known["keypoints_distinct"])
    pts = np.concatenate([kp.mean(axis=0)[None], kp])
    np.testing.assert_allclose(left.project(pts, np.eye(4)), np.array(known["points_left_distinct"]), atol=1e-6)
    np.testing.assert_allclose(right.project(pts, p["T_RL"]), np.array(known["points_right_distinct"]), atol=1e-6)


def test_bench_launcher_starts_ranks_and_relays_one_json_line():
    """`python bench.py --gpus N` without a torchrun environment must start its own ranks (the driver's 8-GPU command),
    never touch the GPU in the parent, relay rank 0's JSON line and return the child's exit code."""
    import io
    import bench

    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    args = bench.parse(argv)
    assert bench.needs_launcher(args, {}) and not bench.needs_launcher(args, {"WORLD_SIZE": "8"})
    assert not bench.needs_launcher(bench.parse(["--gpus", "1"]), {}) and bench.needs_launcher(bench.parse(["--gpus", "1", "--spawn"]), {})
    cmd = bench.launcher_command(args, argv, 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-6:] == argv and cmd[-7].endswith("bench.py")

    seen = {}

    class FakeProc:
        def __init__(self, cmd, stdout=None, text=None, env=None):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = io.StringIO('W0000 torchrun chatter\n{"metric": "frames/sec", "value": 1.0}\n')

        def wait(self):
            return seen.get("rc", 0)

    out, err = io.StringIO(), io.StringIO()
    assert bench.self_launch(args, argv, device_count=8, popen=FakeProc, out=out, err=err) == 0
    assert out.getvalue() == '{"metric": "frames/sec", "value": 1.0}\n' and "chatter" in err.getvalue()
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "--spawn" not in seen["cmd"]
    seen["rc"] = 5                                                     # a failing rank fails the launcher
    assert bench.self_launch(args, argv, device_count=8, popen=FakeProc, out=io.StringIO(), err=io.StringIO()) == 5
    err = io.StringIO()                                                # fewer devices than ranks: clean refusal, nothing started
    seen.clear()
    assert bench.self_launch(args, argv, device_count=1, popen=FakeProc, out=io.StringIO(), err=err) == 2
    assert "needs 8 devices, 1 visible" in err.getvalue() and not seen
    # the driver's configs[3] run: `python bench.py --gpus 8 --workload cups512` -> eight ranks, the workload flag handed on unchanged
    argv3 = ["--gpus", "8", "--steps", "20", "--warmup", "5", "--workload", "cups512"]
    args3 = bench.parse(argv3)
    assert args3.workload == "cups512" and bench.needs_launcher(args3, {})
    cmd3 = bench.launcher_command(args3, argv3, 29512)
    assert "--nproc-per-node=8" in cmd3 and cmd3[-8:] == argv3 and cmd3[cmd3.index("--master-addr") + 1] == "127.0.0.1"
    assert "configs[3]" in bench.workload_string(args3.batch, args3.dtype, 8, "cups512") and "batch=64/GPU" in bench.workload_string(args3.batch, args3.dtype, 8, "cups512")
    seen.clear(); seen["rc"] = 0
    assert bench.self_launch(args3, argv3, device_count=8, popen=FakeProc, out=io.StringIO(), err=io.StringIO()) == 0 and seen["cmd"][-2:] == ["--workload", "cups512"]
    seen.clear()
    err = io.StringIO()
    assert bench.self_launch(args3, argv3, device_count=7, popen=FakeProc, out=io.StringIO(), err=err) == 2 and "needs 8 devices, 7 visible" in err.getvalue() and not seen


def test_launcher_counts_gpus_without_the_hip_runtime(tmp_path):
    """The launcher parent must not initialise the GPU before it starts torchrun: devices are counted from the KFD topology
    (nodes with SIMDs), with /dev/dri/renderD* as fall-back and the *_VISIBLE_DEVICES lists as caps."""
    import bench
    nodes = tmp_path / "nodes"
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):            # two CPU nodes, three GPUs
        d = nodes / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    dri = tmp_path / "dri"
    dri.mkdir()
    for n in ("card0", "renderD128", "renderD129"):
        (dri / n).write_text("")
    assert bench.gpu_count_without_hip({}, str(nodes), str(dri)) == 3
    assert bench.gpu_count_without_hip({"HIP_VISIBLE_DEVICES": "0,2"}, str(nodes), str(dri)) == 2
    assert bench.gpu_count_without_hip({"ROCR_VISIBLE_DEVICES": ""}, str(nodes), str(dri)) == 0
    assert bench.gpu_count_without_hip({}, str(tmp_path / "absent"), str(dri)) == 2          # no KFD topology: render nodes
    assert bench.gpu_count_without_hip({}, str(tmp_path / "absent"), str(tmp_path / "absent2")) == 0
    import inspect
    src = inspect.getsource(bench.main)
    assert "import torch" not in src.split("rank_main")[0]          # nothing in the launcher branch imports torch


def test_capacity_default_is_uncapped_like_the_reference():
    from object_keypoints_amd.perception.pipeline import KeypointExtractionComponent
    comp = KeypointExtractionComponent({"keypoint_config": [1, 3]}, [64, 64])
    assert comp.capacity is None and comp.keypoint_config == [1, 1, 3]


def test_torch_ops_are_registered_over_the_c_abi():
    """torch.ops.okp.* (csrc/okp_torch.cpp) is a dispatcher shim over the same extern "C" symbols: every launch entry point the
    network uses is registered, with the mutated tensors annotated in the schema; without a GPU a call fails with the
    no-CPU-fallback error instead of computing anything."""
    import torch
    from object_keypoints_amd import _lib, ops
    T = _lib.torch_ops()
    assert T is not None, "libokp_torch.so is built by object_keypoints_amd.build alongside libokp_hip.so"
    names = ["conv_forward", "conv_select_tile", "fire_forward", "fire_chain_forward", "heads_forward", "head_out_forward",
             "stem_forward_nchw", "peak_nms", "lift_peaks", "group_objects",
             # round 6: the rest of include/okp.h - a torch-only caller builds and runs the whole path
             "stem_forward_nchw_pairs", "stem_forward", "pack_frames", "pack_frames_u8", "preprocess_u8", "cast", "add_f16_f32", "dwconv3x3_forward",
             "nms_maxpool", "capacity_overflow", "camera_undistort", "unproject_depth", "triangulate_dlt"]
    for n in names:
        op = getattr(T, n)
        schema = str(op.default._schema)
        assert schema.startswith(f"okp::{n}(")
        if n != "conv_select_tile":
            assert "(a!)" in schema, schema
    for n in ("conv_create", "conv_bn_create", "fold_bn", "conv_destroy", "conv_macs", "conv_picks_patch", "stem_create", "stem_destroy", "stream_wait_stream"):
        assert str(getattr(T, n).default._schema).startswith(f"okp::{n}(")
    # the ABI-5/6 arguments of okp_conv_args are arguments of the op
    schema = str(T.conv_forward.default._schema)
    for arg in ("out16", "write_out", "out_subsample", "src_pairs", "out_pairs"):
        assert arg in schema
    # every launch / creation entry point the header declares has an op (by the C name without its okp_ prefix, or a listed alias)
    import re
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "okp.h")).read()
    declared = set(re.findall(r"\b(okp_[a-z0-9_]+)\s*\(", header))
    alias = {"okp_conv_create_x3": "conv_create", "okp_stem_create_dtype": "stem_create", "okp_stem_create": "stem_create", "okp_fisheye_undistort": "camera_undistort",
             "okp_conv_patch_applies": "conv_picks_patch"}
    not_ops = {"okp_last_error", "okp_abi_version", "okp_device_count", "okp_device_arch"}      # (errors surface as exceptions; devices are torch's business)
    for c_name in sorted(declared - not_ops):
        op_name = alias.get(c_name, c_name[len("okp_"):])
        assert hasattr(T, op_name), f"{c_name} has no torch.ops.okp counterpart"
    # plan creation needs no GPU call before the upload: the argument checks run first, on the host
    with pytest.raises(RuntimeError, match="weight is"):
        T.conv_create(0, [64], [1], 128, [0], [0], [0], [torch.zeros(128, 32)], None, 0, [])
    fw, fb = T.fold_bn(torch.ones(4, 2, 1, 1), torch.full((4,), 2.0), torch.zeros(4), torch.ones(4), torch.full((4,), 4.0 - 1e-5), 1e-5, None)
    assert torch.equal(fw, torch.ones(4, 2, 1, 1)) and torch.equal(fb, torch.full((4,), -1.0))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        T.peak_nms(torch.zeros(1, 1, 8, 8), 4, torch.zeros(1, dtype=torch.int32), torch.zeros(8, dtype=torch.int32), torch.zeros(12), 0)
    with pytest.raises(ops.OkpError):
        ops.peak_nms(torch.zeros(1, 1, 8, 8))


def test_from_calibration_picks_the_camera_model(tmp_path):
    """camera_utils.from_calibration (reference :131-144): equidistant -> FisheyeCamera, radtan -> RadTanPinholeCamera,
    anything else -> ValueError."""
    from object_keypoints_amd.perception.utils import camera_utils as cu
    base = "cam0:\n  camera_model: pinhole\n  intrinsics: [420.0, 415.0, 320.0, 240.0]\n  distortion_coeffs: [-0.28, 0.07, 0.0006, -0.0002]\n  resolution: [640, 480]\n  distortion_model: {}\n"
    for model, cls in (("equidistant", cu.FisheyeCamera), ("radtan", cu.RadTanPinholeCamera)):
        f = tmp_path / f"{model}.yaml"
        f.write_text(base.format(model))
        cam = cu.from_calibration(str(f))
        assert type(cam) is cls and cam.image_size.tolist() == [480, 640] and cam.okp().model == (0 if model == "equidistant" else 1)
    f = tmp_path / "other.yaml"
    f.write_text(base.format("fov"))
    with pytest.raises(ValueError):
        cu.from_calibration(str(f))
    # as in the reference, PinholeCamera.scale()/cut() construct a FisheyeCamera whatever the receiver is (camera_utils.py:18-29)
    assert type(cu.from_calibration(str(tmp_path / "radtan.yaml")).scale(0.5)) is cu.FisheyeCamera


def test_evaluation_results_match_the_reference_table():
    """perception.evaluation.Results against the row the REFERENCE's Results class (scripts/eval_model.py:129-232) produced on
    the same six synthetic frames (tests/golden/evaluation.json, written by tests/golden/make_goldens_eval.py): detections
    missed, keypoint types without detection, a keypoint not lifted, one beyond the 2 m range."""
    import json
    from object_keypoints_amd.perception.evaluation import Results
    from object_keypoints_amd.perception.utils import camera_utils as cu
    with open(os.path.join(REPO, "tests", "golden", "evaluation.json")) as f:
        g = json.load(f)
    params = cu.load_calibration_params(os.path.join(REPO, "config", "calibration.yaml"))
    cam = cu.FisheyeCamera(params["K"], params["D"], params["image_size"]).scale(511 / 720)
    cam = cam.cut(np.array([(511 / 720 * 1280 - 511.0) / 2.0, 0.0])).scale(64 / 511)

    def groups(p_C):
        out = []
        for grp in p_C:
            if grp is None:
                out.append(None)
            elif any(p is None for p in grp):
                out.append([None if p is None else np.array(p) for p in grp])
            else:
                out.append(np.array(grp))
        return out

    res = Results()
    res.set_calibration(cam)
    truth = np.array(g["truth"])
    for fr in g["frames"]:
        res.add(np.array(fr["T_WC"]), [{"p_C": groups(o["p_C"])} for o in fr["objects"]], truth)
    s = res.summary()
    row = g["row"]
    assert s["points"] == int(row["points"])
    assert abs(s["missing"] - float(row["missing"].rstrip("%"))) < 0.006          # the reference prints two decimals
    for col in ("mean", "mean xy", "std", "< 3cm", "25th percentile", "75th percentile"):
        assert abs(s[col] - float(row[col])) < 1e-9, col


def test_reads_a_pytorch_lightning_1_2_1_shaped_checkpoint_without_running_its_pickle(tmp_path):
    """The reference trains with pytorch-lightning==1.2.1 (requirements.txt; scripts/train.py:170 adds a ModelCheckpoint):
    such a checkpoint keys `callbacks` by the callback CLASS and carries hyper-parameter objects.  torch's weights-only
    loader rejects those globals; the loader must still return the tensors, without importing or executing anything the
    pickle names, and ignore non-tensor / non-network entries."""
    import sys
    import types
    import torch
    from object_keypoints_amd.perception import pipeline as pp

    mod = types.ModuleType("pytorch_lightning_fake_model_checkpoint")

    class ModelCheckpoint:                                  # stands in for pytorch_lightning.callbacks.ModelCheckpoint
        pass

    class HParams:                                          # an argparse.Namespace-like hyper-parameter object
        def __init__(self):
            self.lr = 1e-3

    class Tripwire:                                         # reconstructing this would run arbitrary code
        def __reduce__(self):
            return (mod.boom, ())

    def boom():
        raise AssertionError("the loader executed a pickled callable")

    for obj in (ModelCheckpoint, HParams, Tripwire):
        obj.__module__ = mod.__name__
        obj.__qualname__ = obj.__name__
        setattr(mod, obj.__name__, obj)
    boom.__module__, boom.__qualname__ = mod.__name__, "boom"
    mod.boom = boom
    sys.modules[mod.__name__] = mod
    tensors = {"model.heatmap_head.output_head2.2.weight": torch.arange(12.).reshape(3, 4, 1, 1), "model.backbone.pre.0.conv.weight": torch.ones(2, 3, 7, 7),
               "loss.weights": torch.zeros(3)}
    ckpt = {"epoch": 7, "global_step": 1234, "pytorch-lightning_version": "1.2.1",
            "callbacks": {ModelCheckpoint: {"best_model_score": torch.tensor(0.5), "best_model_path": "x.ckpt"}},
            "optimizer_states": [{"state": {0: {"exp_avg": torch.zeros(2)}}, "param_groups": [{"lr": 1e-3}]}], "lr_schedulers": [],
            "hyper_parameters": HParams(), "extra": Tripwire(), "state_dict": tensors}
    path = str(tmp_path / "epoch=7.ckpt")
    torch.save(ckpt, path)
    del sys.modules[mod.__name__]                           # the loading side has no pytorch_lightning at all

    import pickle
    with pytest.raises(pickle.UnpicklingError):
        torch.load(path, map_location="cpu", weights_only=True)          # what the plain safe loader does with this file
    sd = pp.read_checkpoint_state_dict(path)
    assert set(sd) == {"heatmap_head.output_head2.2.weight", "backbone.pre.0.conv.weight", "loss.weights"}
    # the load stays inside torch's weights-only VM: the foreign globals were safe globals (inert stubs) for that one call only ...
    assert not any("okp_opaque" in getattr(g[0] if isinstance(g, tuple) else g, "__module__", "") for g in torch.serialization.get_safe_globals())
    assert sorted(torch.serialization.get_unsafe_globals_in_checkpoint(path)) == sorted(f"{mod.__name__}.{n}" for n in ("ModelCheckpoint", "HParams", "boom"))
    # ... and what the VM trusts by default is torch's own rebuilders plus a few containers (the residual trust of this loader, pinned)
    from torch import _weights_only_unpickler as wo
    foreign = sorted(k for k in wo._get_allowed_globals() if not k.startswith("torch."))
    assert all(k.split(".")[0] in ("collections", "builtins", "_codecs") for k in foreign), foreign
    assert torch.equal(sd["heatmap_head.output_head2.2.weight"], tensors["model.heatmap_head.output_head2.2.weight"])
    # a plain state_dict and a {"state_dict": ...} dict go through the weights-only loader as before
    torch.save({"a.weight": torch.ones(2)}, str(tmp_path / "sd.pt"))
    assert set(pp.read_checkpoint_state_dict(str(tmp_path / "sd.pt"))) == {"a.weight"}
    with open(tmp_path / "junk.pt", "wb") as f:
        f.write(b"not a checkpoint")
    with pytest.raises((pp.OkpError, pickle.UnpicklingError, RuntimeError)):
        pp.read_checkpoint_state_dict(str(tmp_path / "junk.pt"))


def test_compute_mode_is_per_thread_and_parses():
    """ops.f32_split() (what KeypointNet wraps its passes in for "float32x3" / "float32mix") sets per-thread state: a network of another
    configuration running on a second host thread is not affected."""
    import threading
    import torch
    from object_keypoints_amd import ops
    assert ops.parse_compute_dtype("float32x3") == (torch.float32, True, False)
    assert ops.parse_compute_dtype(ops.F32MIX) == (torch.float32, True, True)
    assert ops.parse_compute_dtype(torch.float16) == (torch.float16, False, False) and ops.parse_compute_dtype("bfloat16")[0] == torch.bfloat16
    with pytest.raises(ops.OkpError):
        ops.parse_compute_dtype(torch.float64)
    assert (ops.F32_SPLIT, ops.F32_MIX) == (False, False)
    seen = []
    with ops.f32_split(True, True):
        assert (ops.F32_SPLIT, ops.F32_MIX) == (True, True)
        with ops.f32_split(True, False):
            assert (ops.F32_SPLIT, ops.F32_MIX) == (True, False)
        assert ops.F32_MIX
        t = threading.Thread(target=lambda: seen.append((ops.F32_SPLIT, ops.F32_MIX)))
        t.start(); t.join()
    assert seen == [(False, False)] and (ops.F32_SPLIT, ops.F32_MIX) == (False, False)
    assert not ops.f32_split(False, True).mixed                   # mixed implies split


def test_bench_power_probe_parses_rocm_smi_and_tolerates_its_absence(tmp_path):
    """bench.power_probe samples `rocm-smi --showpower --showclocks` next to the running step (after the timed region): it returns the
    mean board power / shader clock of what the tool printed, and None where the tool is missing or prints something else."""
    import bench
    fake = tmp_path / "fake-smi"
    fake.write_text("#!/bin/sh\n"
                    "case \"$*\" in\n"
                    "  *showmaxpower*) echo 'GPU[0]\t\t: Max Graphics Package Power (W): 1400.0';;\n"
                    "  *) echo 'GPU[0]\t\t: sclk clock level: 3: (2100Mhz)'; echo 'GPU[0]\t\t: Current Socket Graphics Package Power (W): 1210.0';;\n"
                    "esac\n")
    fake.chmod(0o755)
    calls = []
    got = bench.power_probe(lambda: calls.append(1), seconds=0.5, smi=str(fake), sync=lambda: None)
    assert got is not None and got["samples"] >= 1 and got["steps"] == len(calls) - 20
    assert got["board_W_mean"] == 1210.0 and got["sclk_MHz_mean"] == 2100.0 and got["cap_W"] == 1400.0
    assert bench.power_probe(lambda: None, seconds=0.1, smi=str(tmp_path / "no-such-tool"), sync=lambda: None) is None
    silent = tmp_path / "silent-smi"
    silent.write_text("#!/bin/sh\necho nothing useful\n")
    silent.chmod(0o755)
    assert bench.power_probe(lambda: None, seconds=0.2, smi=str(silent), sync=lambda: None) is None


def test_mixed_plan_loaded_without_an_audit_says_so():
    """perception.pipeline._audited (the tail of load_keypoint_net): a mixed-precision network that skips the audit (audit_frames=None)
    raises a RuntimeWarning naming the plan as unverified on these weights; fp32-grade and 16-bit configurations pass silently; the
    default is the automatic audit, and any other string is refused.  (Host logic only: the audit itself needs the device.)"""
    import inspect
    import warnings
    from object_keypoints_amd.perception import pipeline as pp

    class Stub:
        def __init__(self, mixed, split=True):
            self.mixed, self.mfma_split = mixed, split

        def configuration(self):
            return "float32mix" if self.mixed else ("float32x3" if self.mfma_split else "float32")

    with pytest.warns(RuntimeWarning, match="UNVERIFIED on these weights"):
        assert pp._audited(Stub(True), None, pp.HEAT_BAR).mixed
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        pp._audited(Stub(False), None, pp.HEAT_BAR)
        pp._audited(Stub(False, split=False), pp.AUDIT_AUTO, pp.HEAT_BAR)          # no split products: nothing to audit, no device touched
    with pytest.raises(pp.OkpError):
        pp._audited(Stub(True), "sometimes", pp.HEAT_BAR)
    assert inspect.signature(pp.load_keypoint_net).parameters["audit_frames"].default == pp.AUDIT_AUTO
    assert inspect.signature(pp.load_keypoint_net).parameters["on_overflow"].default == "raise"
    with pytest.raises(pp.OkpError):
        pp.load_keypoint_net({}, on_overflow="sometimes", device="cpu")


def test_bench_parity_fields_report_point_error_quantiles(tmp_path):
    """bench.parity_fields: `p_C_err_m` (the maximum) comes with median / p99 / n_over_1e-4, so that one flipped depth pixel under one
    peak (a 1.5 m outlier on a noise-like random-weight depth map) reads as an outlier and not as a bias."""
    import bench
    from oracle import pipeline as op
    from object_keypoints_amd import synth
    sc = synth.bump_scene([1, 3], n_objects=2, seed=7, index=5)
    heat, depth = sc["heat"][None, :1].copy(), sc["depth"][None, :1].copy()         # one map (the two centres)
    oracle = {"heat": heat, "depth": depth}
    cal = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config", "calibration.yaml")
    cam = op.eval_camera(cal)
    d2p = op.DetectionToPoint(); d2p.reset(cam)
    idx = op.peak_indices(heat[0, 0])
    pts, _ = op.refine_peaks(heat[0, 0], idx)
    want = d2p(np.stack(pts), depth[0, 0])
    points = np.full((1, 1, bench.SAMPLE_CAP, 4), np.nan)
    points[0, 0, :len(idx), :3] = want
    points[0, 0, 1, 2] += 1.5                                        # one point 1.5 m off, the others exact
    yx = np.zeros((1, 1, bench.SAMPLE_CAP, 2), np.int64); yx[0, 0, :len(idx)] = idx
    sample = {"heat": heat, "depth": depth, "count": np.array([[len(idx)]]), "yx": yx, "points": points}
    got = bench.parity_fields(sample, oracle, cal)
    st = got["p_C_err_m_stats"]
    assert len(idx) >= 2
    assert abs(got["p_C_err_m"] - 1.5) < 1e-9 and st["max"] == got["p_C_err_m"] and st["n"] == len(idx)
    assert st["n_over_1e-4"] == 1 and (st["median"] < 1e-9 or len(idx) == 2) and got["meets"]["p_C_1e-4_m"] is False and got["meets"]["peaks_identical"]
    # no peak matched: nothing was compared - zero points, null statistics, and the 3D bar is not "met" on a made-up sample
    empty = dict(sample, count=np.array([[0]]))
    got0 = bench.parity_fields(empty, {"heat": heat, "depth": depth}, cal)
    assert got0["p_C_points"] == 0 and got0["p_C_err_m"] is None and got0["p_C_err_m_stats"]["n"] == 0 and got0["p_C_err_m_stats"]["median"] is None
    assert got0["meets"]["p_C_1e-4_m"] is False and not got0["meets"]["peaks_identical"]


@pytest.mark.parametrize("config,n_objects", [([1, 1, 1], 4), ([1, 3], 2), ([1, 3], 1)])
def test_vectorised_voting_equals_the_oracle_on_multi_object_scenes(config, n_objects):
    """The product's ObjectExtraction votes for all peaks of a frame at once; the oracle walks them one by one as the reference does
    (pipeline.py:104-133).  Same objects, same order of the received votes, identical numbers (no k-means branch in these scenes)."""
    from object_keypoints_amd import synth
    from object_keypoints_amd.perception.pipeline import ObjectExtraction
    from oracle import pipeline as op
    cfg = {"keypoint_config": config}
    for index in range(12):
        s = synth.bump_scene(config, n_objects=n_objects, seed=91, index=index)
        points, conf = op.KeypointExtractionComponent(cfg, [64, 64])(s["heat"][None])
        got = ObjectExtraction(cfg, [64, 64])(points[0], conf[0], s["centers"])
        want = op.ObjectExtraction(cfg, [64, 64])(points[0], conf[0], s["centers"])
        assert len(got) == len(want) >= 1
        for g, w_ in zip(got, want):
            assert set(g) == set(w_) == {"center", "heatmap_points", "p_centers", "confidence"}
            assert np.array_equal(g["center"], w_["center"])
            assert len(g["p_centers"]) == len(w_["p_centers"])
            for a, b in zip(g["p_centers"], w_["p_centers"]):
                assert np.array_equal(a, b) and a.dtype == b.dtype
            for a, b in zip(g["heatmap_points"], w_["heatmap_points"]):
                multi = np.asarray(b).ndim == 2 and np.asarray(b).dtype == np.float64      # a k-means result is unordered and unseeded
                if not multi:
                    assert np.array_equal(np.asarray(a), np.asarray(b)) and np.asarray(a).dtype == np.asarray(b).dtype
            for a, b in zip(g["confidence"], w_["confidence"]):
                assert [float(v) for v in a] == [float(v) for v in b]
    # no centre peak: no objects; centre peaks but no votes: empty groups
    ex = ObjectExtraction(cfg, [64, 64])
    assert ex([[]] + [[] for _ in config], [[]] + [[] for _ in config], np.zeros((len(config), 2, 64, 64), np.float32)) == []
    lone = ex([[np.array([5.0, 6.0], np.float32)]] + [[] for _ in config], [[np.float32(3)]] + [[] for _ in config], np.zeros((len(config), 2, 64, 64), np.float32))
    assert len(lone) == 1 and lone[0]["p_centers"] == [] and all(np.asarray(p).size == 0 for p in lone[0]["heatmap_points"])


def test_camera_helpers_agree_with_the_oracle_restatement():
    """camera_utils' host helpers (own formulation: closed-form K^-1, index-assigned matrices) against oracle/geometry.py, which follows the
    reference line by line (camera_utils.py:7-43,119-189; linalg.py:4-20)."""
    from object_keypoints_amd.perception.utils import camera_utils as cu, linalg
    from oracle import geometry as og
    rng = np.random.default_rng(4)
    K = cu.camera_matrix([421.5, 418.25, 322.0, 238.5])
    assert np.array_equal(K, og.camera_matrix([421.5, 418.25, 322.0, 238.5]))
    D = np.array([-0.02, 0.01, -0.003, 0.0004])
    mine, theirs = cu.FisheyeCamera(K, D, [480, 640]), og.FisheyeCamera(K, D, [480, 640])
    np.testing.assert_allclose(mine.Kinv, theirs.Kinv, rtol=0, atol=1e-15)
    skewed = K.copy(); skewed[0, 1] = 0.7
    np.testing.assert_allclose(cu._upper_triangular_inverse(skewed), np.linalg.inv(skewed), rtol=0, atol=1e-15)
    xy, z = rng.uniform(0, 600, (9, 2)), rng.uniform(0.3, 2.0, 9)
    np.testing.assert_allclose(mine.unproject(xy, z), theirs.unproject(xy, z), rtol=0, atol=1e-13)
    probe = np.array([[1.0, 1.0], [0.0, 5.0], [500.0, 100.0], [100.0, 500.0], [479.0, 639.0], [480.0, 10.0], [np.nan, 3.0]])
    assert mine.in_frame(probe).tolist() == theirs.in_frame(probe).tolist()
    for a, b in ((mine.scale(0.125), theirs.scale(0.125)), (mine.cut(np.array([12.0, 7.0])), theirs.cut(np.array([12.0, 7.0])))):
        assert type(a) is cu.FisheyeCamera and np.array_equal(a.K, b.K) and np.array_equal(a.image_size, b.image_size)
    assert np.array_equal(cu.scale_camera_matrix(skewed, (0.5, 0.25)), og.scale_camera_matrix(skewed, (0.5, 0.25)))
    T = np.eye(4); T[:3, :3] = np.linalg.qr(rng.standard_normal((3, 3)))[0]; T[:3, 3] = [-0.11, 0.004, 0.002]
    np.testing.assert_allclose(cu.fundamental_matrix(T, K, skewed), og.fundamental_matrix(T, K, skewed), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(cu.projection_matrix(K, T), og.projection_matrix(K, T), rtol=0, atol=1e-13)
    v = rng.standard_normal(3)
    assert np.array_equal(linalg.skew_matrix(v), og.skew_matrix(v))
    np.testing.assert_allclose(linalg.inv_transform(T), og.inv_transform(T), rtol=0, atol=1e-15)
    pts = rng.standard_normal((5, 3))
    np.testing.assert_allclose(linalg.transform_points(T, pts), og.transform_points(T, pts), rtol=0, atol=1e-15)
    calib = os.path.join(REPO, "config", "calibration.yaml")
    p, q = cu.load_calibration_params(calib), og.load_calibration_params(calib)
    assert set(p) == set(q)
    for k in p:
        np.testing.assert_allclose(np.asarray(p[k], dtype=np.float64), np.asarray(q[k], dtype=np.float64), rtol=0, atol=1e-15)
    stereo = cu.StereoCamera.from_file(calib)
    np.testing.assert_allclose(stereo.F, og.StereoCamera.from_file(calib).F, rtol=1e-12, atol=1e-18)


def test_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
    """include/okp.h is the drop-in boundary for ANY host language: it must be valid ISO C99 (no C++-isms, no HIP / torch types), and a C
    program that includes it links against libokp_hip.so and reads the ABI version - no GPU needed for that call."""
    import shutil
    import subprocess
    from object_keypoints_amd import _lib
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler in this environment")
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include "okp.h"\n'
                   "int main(void) {\n"
                   "  okp_conv_args a; okp_fire_args f; okp_head_out_args h; okp_camera c; okp_tap t; okp_tensor v;\n"
                   "  (void)a; (void)f; (void)h; (void)c; (void)t; (void)v;\n"
                   '  printf("%d %d %d %d %d %d %d %d\\n", okp_abi_version(), OKP_ABI_VERSION, (int)sizeof(okp_conv_args), (int)sizeof(okp_fire_args),\n'
                   '         (int)sizeof(okp_head_out_args), (int)sizeof(okp_camera), (int)sizeof(okp_tap), (int)sizeof(okp_tensor));\n'
                   "  return okp_abi_version() == OKP_ABI_VERSION ? 0 : 1;\n}\n")
    inc = os.path.join(REPO, "include")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I" + inc, str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = tmp_path / "abi"
    r = subprocess.run([gcc, "-std=c99", "-I" + inc, str(src), "-o", str(exe), "-L" + libdir, "-lokp_hip", "-Wl,-rpath," + libdir,
                        "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)
    version, declared, *sizes = [int(x) for x in r.stdout.split()]
    import ctypes
    mirrors = (_lib.okp_conv_args, _lib.okp_fire_args, _lib.okp_head_out_args, _lib.okp_camera, _lib.okp_tap, _lib.okp_tensor)
    assert version == declared == _lib.OKP_ABI and sizes == [ctypes.sizeof(m) for m in mirrors]      # the ctypes mirrors of the structs have the C layout

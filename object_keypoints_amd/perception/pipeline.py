"""Drop-in for the reference's perception/pipeline.py on the HIP path.

Same classes, constructor / __call__ signatures and return nesting as the reference
(perception/pipeline.py:13-209): InferenceComponent, KeypointExtractionComponent, ObjectExtraction,
DetectionToPoint, ObjectKeypointPipeline, LearnedKeypointTrackingPipeline, plus the
TriangulationComponent the reference's test expects (test/test_pipeline.py:171-177).

Device work (network, peak extraction, undistort + depth lifting, DLT) goes through libokp_hip.so;
the object-grouping glue (ObjectExtraction) is host logic as in the reference.  BatchedKeypointPipeline
is the data-parallel form of the same path: a whole batch of frames stays on the device from the
packed input to the per-peak 3D points (one D2H copy at the end).
"""
import numpy as np
import torch

from .. import ops
from ..ops import OkpError
from . import models
from .utils import camera_utils  # noqa: F401  (re-exported like the reference module does)

DEFAULT_PEAK_CAPACITY = 64


def _device():
    if not torch.cuda.is_available():
        raise OkpError("the keypoint pipeline runs on the HIP device; no GPU is visible and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _is_torchscript_archive(path):
    """A TorchScript file is a zip archive with a `<name>/constants.pkl` and `<name>/code/` entries; a torch.save'd
    checkpoint is a zip without them (or a legacy pickle)."""
    import zipfile
    if not zipfile.is_zipfile(path):
        return False
    with zipfile.ZipFile(path) as z:
        names = z.namelist()
    return any(n.endswith("/constants.pkl") for n in names) and any("/code/" in n for n in names)


class _OpaqueGlobal:
    """Stand-in for a pickled global that is not a tensor-building function: callable, constructible, stateless.  Nothing of
    the module the pickle names is imported or run."""

    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, *args, **kwargs):
        return _OpaqueGlobal()

    def __setstate__(self, state):
        pass


MAX_OPAQUE_GLOBALS = 256        # a training checkpoint names a handful of foreign classes; a file that names more is not one
# torch.serialization.safe_globals is PROCESS-global, not thread-local: while the stubs of one checkpoint are registered, a concurrent
# torch.load(weights_only=True) on another thread would resolve those foreign names to inert stubs instead of refusing them.  The
# loads of this module take the lock; other threads' own torch.load calls cannot be held off from here - load model files before
# starting threads that unpickle untrusted data.
import threading
_OPAQUE_LOAD_LOCK = threading.Lock()


def _load_with_opaque_globals(path):
    """torch.load(weights_only=True) - torch's own restricted unpickling VM, with its limits on what REDUCE / BUILD / NEWOBJ may touch -
    for a checkpoint that names globals the VM does not allow: each such global (torch.serialization.get_unsafe_globals_in_checkpoint
    lists them without unpickling anything) is registered as a SAFE global under its pickled name, bound to an inert stub class
    (_OpaqueGlobal: constructible, callable, stateless), for the duration of this one load.  Nothing the pickle names is imported or
    run; what a stub 'constructs' is another stub.  Needed for the reference's training checkpoints: pytorch-lightning 1.2.1
    (requirements.txt) keys the `callbacks` dict of a checkpoint by the callback CLASS (ModelCheckpoint, scripts/train.py:170) and
    stores hyper-parameter objects.  Residual trust: torch's VM itself and the tensor / storage rebuilders on its default allow-list
    (tests/test_host_logic.py pins that the allow-list names nothing outside torch, collections and builtins' containers).
    Zip-format files only (torch.save's format since 1.6; get_unsafe_globals_in_checkpoint reads nothing else): a legacy-format
    checkpoint that names foreign globals is refused with that explanation - re-save it with a current torch."""
    try:
        names = list(torch.serialization.get_unsafe_globals_in_checkpoint(path))
    except Exception as e:
        raise OkpError(f"{path}: cannot list the globals of this file ({type(e).__name__}: {e}); checkpoints that name classes outside torch "
                       "must be in torch.save's zip format (re-save a legacy-format file with a current torch)") from e
    if len(names) > MAX_OPAQUE_GLOBALS:
        raise OkpError(f"{path}: names {len(names)} globals outside torch's weights-only allow-list; not a training checkpoint")
    stubs = []
    for full in names:
        module, _, name = full.rpartition(".")
        stubs.append((type(name or "opaque", (_OpaqueGlobal,), {"__module__": "okp_opaque." + module}), full))
    with _OPAQUE_LOAD_LOCK, torch.serialization.safe_globals(stubs):
        return torch.load(path, map_location="cpu", weights_only=True)


def read_checkpoint_state_dict(path):
    """Tensors of a model file: TorchScript archive (scripts/package_model.py:40-42), torch.save'd state_dict, or a
    (pytorch-lightning) training checkpoint {"state_dict": ..., "callbacks": ..., ...}.  Never executes pickled code and never leaves
    torch's weights-only unpickler: a checkpoint that carries other globals is re-read with those bound to inert stubs
    (_load_with_opaque_globals) and only its tensors are kept.  Returns {name: tensor} with the module prefixes stripped."""
    import pickle
    if _is_torchscript_archive(path):
        sd = torch.jit.load(path, map_location="cpu").state_dict()
    else:
        try:
            sd = torch.load(path, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError:
            try:
                sd = _load_with_opaque_globals(path)
            except Exception as e:
                raise OkpError(f"{path}: not a state_dict, Lightning checkpoint or TorchScript archive this loader can read ({e})") from e
        if isinstance(sd, dict) and isinstance(sd.get("state_dict"), dict):
            sd = sd["state_dict"]
    if not isinstance(sd, dict):
        raise OkpError(f"{path}: no state_dict found in the file")
    return strip_module_prefixes({k: v for k, v in sd.items() if isinstance(k, str) and isinstance(v, torch.Tensor)})


def strip_module_prefixes(sd):
    clean = {}
    for k, v in sd.items():
        for prefix in ("model.model.", "model."):          # TorchScript wrapper / Lightning module prefixes
            if k.startswith(prefix):
                k = k[len(prefix):]
                break
        clean[k] = v
    return clean


HEAT_BAR = 1e-3          # north_star: heat maps within 1e-3 of the fp32 reference


AUDIT_AUTO = "auto"      # load_keypoint_net(audit_frames=...): audit a mixed plan on two synthetic frames when the caller gives none


AUDIT_BATCH = 16         # the synthetic audit runs at a batch that takes the kernels a deployment takes (patch-resident kernel, pair tensors, fused heads)


def _audited(net, audit_frames, heat_bar, _source=None):
    """float32mix is a plan derived on ONE weight family: its single-term fp16 branches are cheap because BatchNorm
    gains attenuate them there.  With `audit_frames` the loaded network is priced on the device against float32x3 (fp32-grade, 2e-6)
    on the caller's frames (KeypointNet.precision_audit: two passes, no CPU reference) and, where the heat maps are further than
    `heat_bar` apart - or a value left the fp16 range: fp16 operands overflow above 65504 - the network falls back to float32x3 with a
    warning.  float32x3 itself is audited for that range (its one failure mode, KeypointNet.on_overflow): where the audit frames leave
    it, the network falls back to exact float32, again with a warning.
    The default (AUDIT_AUTO) audits a split-product configuration on synthetic frames - two N(0,1) frames repeated to a batch of
    AUDIT_BATCH, so that the launch heuristics pick the kernels a deployment runs - so that no such plan ships unverified on the weights it
    was loaded with; `net.audit` records the source of the frames and the batch.  Synthetic frames say nothing about activations on REAL
    frames: the per-pass guard (on_overflow) stays on.  16-bit configurations are the caller's explicit choice of a precision outside the
    bar and are audited only on request.  audit_frames=None skips the audit; for a mixed configuration that is said with a RuntimeWarning."""
    import warnings
    if audit_frames is None:
        if getattr(net, "mixed", False):
            warnings.warn(f"{net.configuration()}: loaded without an audit - this mixed-precision plan was derived on one weight family and is "
                          "UNVERIFIED on these weights (pass audit_frames, or leave the default, to price it against float32x3)", RuntimeWarning, stacklevel=3)
        return net
    source = _source or "caller"
    if isinstance(audit_frames, str):
        if audit_frames != AUDIT_AUTO:
            raise OkpError(f"audit_frames: a [n,3,H,W] tensor, None or '{AUDIT_AUTO}'")
        if not net.mfma_split:
            return net
        from .. import synth
        audit_frames = torch.from_numpy(synth.frames(2, seed=20251)).repeat(AUDIT_BATCH // 2, 1, 1, 1)
        source = "synthetic"
    if not (net.mfma_split or net.compute_dtype in ops.HALF_DTYPES):
        net.audit = {"configuration": net.configuration(), "checked": False, "reason": "exact float32: nothing to audit"}
        return net
    frames = audit_frames.to(next(net.parameters()).device, torch.float32)
    requested = net.configuration()
    base = {"configuration": requested, "checked": True, "frames": int(frames.shape[0]), "batch": int(frames.shape[0]), "frames_source": source, "heat_bar": heat_bar}
    if net.mfma_split and not net.mixed:
        # float32x3: one pass; its only failure mode is the fp16 range of its operands
        with torch.no_grad():
            outs = net.deployed(frames, check_range=False)
        ok = not net.range_overflow() and all(bool(torch.isfinite(t).all()) for t in outs)
        net.audit = dict(base, report={"range_ok": ok}, fell_back=not ok, fell_back_to=None if ok else "float32")
        if not ok:
            warnings.warn(f"{requested}: an activation leaves the fp16 range on the audit frames ({source}); falling back to exact float32", RuntimeWarning, stacklevel=3)
            net.set_compute_dtype(torch.float32)
        return net
    report = net.precision_audit(frames)
    ok = report["heat"]["finite"] and report["heat"]["max"] <= heat_bar
    net.audit = dict(base, report=report, fell_back=not ok, fell_back_to=None if ok else ops.F32X3)
    if not ok:
        warnings.warn(f"{requested}: heat maps differ from float32x3 by {report['heat']['max']:.2e} on the audit frames ({source}, batch {frames.shape[0]}; "
                      f"bar {heat_bar:.0e}, in range and finite: {report['heat']['finite']}); falling back to float32x3", RuntimeWarning, stacklevel=3)
        net.set_compute_dtype(ops.F32X3)
        if net.mixed is False and not report["heat"]["finite"]:
            return _audited(net, audit_frames, heat_bar, _source=source)          # the range may be the reason: float32x3 is audited for it in turn
    return net


def load_keypoint_net(model, compute_dtype=torch.float32, device=None, audit_frames=AUDIT_AUTO, heat_bar=HEAT_BAR, on_overflow="raise"):
    """Accepts a KeypointNet, a state_dict, or a path to a torch.save'd state_dict / Lightning checkpoint /
    TorchScript file produced by the reference's scripts/package_model.py, and returns an eval KeypointNet.
    Entries of the file that are not KeypointNet parameters (loss buffers, metrics of the Lightning module) are ignored;
    a missing network tensor is an error.
    audit_frames ([n,3,H,W] float32, a handful of representative frames): price a mixed / 16-bit `compute_dtype` on these weights and
    fall back to float32x3 when it misses `heat_bar` (see _audited; the result is in `net.audit`).  Default: a split-product configuration
    ("float32x3", "float32mix") is audited on synthetic frames; None skips the audit (with a RuntimeWarning for a mixed configuration).
    on_overflow ("raise" | "float32" | "defer"): what a pass of a split-product configuration does when an activation leaves the fp16
    range (KeypointNet.on_overflow): raise OkpError, re-run the pass with the exact float32 kernels, or leave the flag to the caller."""
    device = device or _device()
    if on_overflow not in ("raise", "float32", "defer"):
        raise OkpError("on_overflow is 'raise', 'float32' or 'defer'")
    if isinstance(model, models.KeypointNet):
        model.on_overflow = on_overflow
        return _audited(model.to(device).eval(), audit_frames, heat_bar)
    if isinstance(model, (str, bytes)) or hasattr(model, "__fspath__"):
        clean = read_checkpoint_state_dict(model)
    elif isinstance(model, dict):
        clean = strip_module_prefixes(model)
    else:
        raise OkpError("cannot interpret model argument")
    for need in ("heatmap_head.output_head2.2.weight", "heatmap_head.output_head2.0.conv.weight"):
        if need not in clean:
            raise OkpError(f"model file holds no KeypointNet: '{need}' is missing")
    heat_out = clean["heatmap_head.output_head2.2.weight"].shape[0]
    features = clean["heatmap_head.output_head2.0.conv.weight"].shape[0]
    net = models.KeypointNet(features=features, heatmaps_out=heat_out, compute_dtype=compute_dtype)
    own = net.state_dict()
    missing = [k for k in own if k not in clean]
    if missing:
        raise OkpError(f"model file lacks {len(missing)} KeypointNet tensors, e.g. '{missing[0]}'")
    net.load_state_dict({k: clean[k] for k in own})
    net.on_overflow = on_overflow
    return _audited(net.to(device).eval(), audit_frames, heat_bar)


def load_cornernet_backbone(net, pretrained):
    """Copies the hourglass of a CornerNet-Squeeze checkpoint into `net.backbone`, as the reference's
    KeypointNet._build_hourglass does with ./models/corner_net.pkl (perception/models.py:69-78: the pickle holds the
    DataParallel-wrapped detector, keys `module.hg.*`, loaded by core/nnet/py_factory.py:119-123); the detector's own
    corner heads (`module.tl_*`, `module.br_*`, ...) are not part of KeypointNet and are skipped.
    `pretrained`: a state_dict or a path to a torch.save'd one.  Returns the number of tensors copied."""
    if isinstance(pretrained, (str, bytes)) or hasattr(pretrained, "__fspath__"):
        pretrained = torch.load(pretrained, map_location="cpu", weights_only=True)      # tensors only: never runs pickled code
    own = net.backbone.state_dict()
    picked = {}
    for k, v in pretrained.items():
        for prefix in ("module.hg.", "hg."):
            if k.startswith(prefix):
                picked[k[len(prefix):]] = v
                break
    missing = [k for k in own if k not in picked]
    if missing:
        raise OkpError(f"CornerNet checkpoint lacks {len(missing)} hourglass tensors, e.g. {missing[0]}")
    bad = [k for k in own if tuple(picked[k].shape) != tuple(own[k].shape)]
    if bad:
        raise OkpError(f"CornerNet checkpoint: shape mismatch at {bad[0]}")
    net.backbone.load_state_dict({k: picked[k] for k in own})
    return len(own)


class InferenceComponent:
    name = "inference"

    def __init__(self, model, cuda=True, compute_dtype=torch.float32, on_overflow="float32"):
        """on_overflow (split-product compute dtypes only): the reference's fp32 model returns numbers for any input; the drop-in does too -
        a pass whose activations leave the fp16 range is re-run with the exact float32 kernels ("float32", with a RuntimeWarning the first
        time) - unless the caller prefers an OkpError ("raise")."""
        if not cuda:
            raise OkpError("InferenceComponent(cuda=False): this build has no CPU path (the CPU restatement is oracle/, test-only)")
        self.cuda = cuda
        self.model = load_keypoint_net(model, compute_dtype, on_overflow=on_overflow)

    def __call__(self, frames):
        frames = frames.to(_device())
        with torch.no_grad():
            heatmaps, depth, centers = [t.cpu() for t in self.model.deployed(frames)]
        return heatmaps, depth, centers


class KeypointExtractionComponent:
    name = "keypoints"
    PROBABILITY_CUTOFF = 0.1

    def __init__(self, keypoint_config, prediction_size, bandwidth=1.0, capacity=None):
        """capacity: peak slots per map; None (default) = one per pixel, i.e. no cap, like the reference
        (pipeline.py:73 keeps every peak).  The batched pipeline passes its fixed capacity explicitly."""
        self.keypoint_config = [1] + keypoint_config['keypoint_config']     # centre map first
        self.n_keypoints = sum(self.keypoint_config)
        self.prediction_size = prediction_size
        self.capacity = capacity

    def extract_device(self, frames):
        """frames N x K x H x W (device or host) -> device tensors (count, yx, xyc)."""
        if isinstance(frames, np.ndarray):
            frames = torch.from_numpy(np.ascontiguousarray(frames.astype(np.float32)))
        frames = frames.to(_device(), torch.float32)
        assert frames.shape[1] == len(self.keypoint_config)
        return ops.peak_nms(frames, cap=self.capacity or frames.shape[2] * frames.shape[3])

    def __call__(self, frames):
        count, _, xyc = self.extract_device(frames)
        count, xyc = count.cpu().numpy(), xyc.cpu().numpy()
        if self.capacity is not None and int(count.max(initial=0)) > self.capacity:
            raise OkpError(f"a heat map has {int(count.max())} peaks, above the capacity {self.capacity}; "
                           "raise `capacity` or leave it None (the reference has no cap)")
        keypoints, confidence = [], []
        for n in range(count.shape[0]):
            kp, cf = [], []
            for k in range(count.shape[1]):
                c = int(count[n, k])
                kp.append([xyc[n, k, j, :2].copy() for j in range(c)])          # (x, y) float32
                cf.append([xyc[n, k, j, 2] for j in range(c)])
            keypoints.append(kp)
            confidence.append(cf)
        return keypoints, confidence


class ObjectExtraction:
    """Centre-vector voting (interface and results of the reference's pipeline.py:93-153): the peaks of map 0 are object centres; every
    other peak votes with `pixel centre + predicted offset` at its rounded position for the nearest centre, votes further than
    VOTE_RADIUS pixels are dropped; per object and keypoint type the surplus over the configured count is reduced (most confident
    detection for a single-instance type, k-means centres for a multi-instance one).  Returns one dict per centre peak:
    {'center', 'heatmap_points': per type (n, 2) or empty, 'confidence': per type list, 'p_centers': the votes received}."""

    VOTE_RADIUS = 20.0

    def __init__(self, keypoint_config, prediction_size):
        self.keypoint_config = keypoint_config['keypoint_config']
        self.prediction_size = prediction_size
        h, w = prediction_size
        self.min = np.zeros(2, dtype=np.int32)
        self.max = np.array([w - 1, h - 1], dtype=np.int32)                      # (x, y) bounds of a rounded peak
        cols, rows = np.arange(w) + 0.5, np.arange(h) + 0.5
        self.image_indices = np.stack([np.broadcast_to(cols[None, :], (h, w)), np.broadcast_to(rows[:, None], (h, w))])   # (2, h, w): x, y of pixel centres

    def _votes(self, keypoints, centers, center_points):
        """All non-centre peaks of the frame at once -> (type index, index within its type, predicted centre (fp64), voted object or -1),
        in the reference's visiting order (type by type, peak by peak)."""
        types = [np.full(len(pts), i, dtype=np.int64) for i, pts in enumerate(keypoints[1:])]
        flat = [p for pts in keypoints[1:] for p in pts]
        if not flat:
            return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros((0, 2)), np.zeros(0, np.int64)
        type_of = np.concatenate(types)
        within = np.concatenate([np.arange(len(pts)) for pts in keypoints[1:]])
        pts = np.stack(flat)
        cell = np.clip(np.round(pts).astype(np.int32), self.min, self.max)      # (x, y), half to even like ndarray.round
        predicted = (self.image_indices[:, cell[:, 1], cell[:, 0]] + centers[type_of, :, cell[:, 1], cell[:, 0]].T).T          # (n, 2) fp64
        delta = center_points[None, :, :] - predicted[:, None, :]
        dist = np.sqrt(delta[..., 0] * delta[..., 0] + delta[..., 1] * delta[..., 1])                                            # (n, n_centres)
        nearest = dist.argmin(axis=1)
        voted = np.where(dist[np.arange(len(pts)), nearest] > self.VOTE_RADIUS, -1, nearest)
        return type_of, within, predicted, voted

    def _reduce(self, points, confidences, wanted):
        if points.shape[0] <= wanted:
            return points
        if wanted == 1:
            return points[np.argmax(confidences)][None]
        from sklearn import cluster
        # the reference passes no n_init (pipeline.py:146) under its pinned scikit-learn 0.24.1, whose default is 10 restarts; newer
        # releases changed the default ('auto' -> 1 for init='random' from 1.4), so it is spelled out
        return cluster.KMeans(init='random', n_clusters=wanted, n_init=10).fit(points).cluster_centers_

    def __call__(self, keypoints, confidence, centers):
        if len(keypoints[0]) == 0:
            return []
        center_points = np.stack(keypoints[0])
        n_types = len(keypoints) - 1
        type_of, within, predicted, voted = self._votes(keypoints, np.asarray(centers), center_points)
        objects = []
        for o, c in enumerate(center_points):
            mine = voted == o
            obj = {'center': c, 'heatmap_points': [], 'p_centers': [p for p in predicted[mine]], 'confidence': []}
            for i in range(n_types):
                picked = within[mine & (type_of == i)]
                conf = [confidence[i + 1][j] for j in picked]
                obj['confidence'].append(conf)
                if len(picked) == 0:
                    obj['heatmap_points'].append(np.array([]))
                else:
                    obj['heatmap_points'].append(self._reduce(np.stack([keypoints[i + 1][j] for j in picked]), np.stack(conf), self.keypoint_config[i]))
            objects.append(obj)
        return objects


class DetectionToPoint:
    def reset(self, camera):
        self.camera = camera
        self.min_index = np.zeros(2, dtype=int)
        self.max_index = camera.image_size.astype(int) - 1

    def __call__(self, xy, p_depth):
        """xy (n,2) pixels + one depth map (H,W) -> (n,3) float64 camera-frame points (device kernel)."""
        if xy.shape[0] == 0:
            return None
        dev = _device()
        depth = torch.from_numpy(np.ascontiguousarray(p_depth, dtype=np.float32))[None].to(dev)
        pts = torch.from_numpy(np.ascontiguousarray(xy, dtype=np.float32)).to(dev)
        ids = torch.zeros(xy.shape[0], dtype=torch.int32, device=dev)
        # the reference clips x against image_size[0]-1 and y against image_size[1]-1 (pipeline.py:161-169)
        out = ops.unproject_depth(self.camera.okp(), pts, ids, depth, int(self.max_index[0]), int(self.max_index[1]))
        return out.cpu().numpy()

    def lift_groups(self, groups, p_depth):
        """Every keypoint group of a frame in ONE launch: groups = [(xy (n_i, 2), depth-map index k_i), ...], p_depth (K, H, W) ->
        [(n_i, 3) float64 or None, ...], each exactly what __call__(xy, p_depth[k_i]) returns (the kernel treats points one by one;
        one upload, one launch and one read-back per frame instead of three tiny transfers and a launch per group)."""
        sizes = [int(np.asarray(xy).shape[0]) for xy, _ in groups]
        if sum(sizes) == 0:
            return [None] * len(groups)
        dev = _device()
        xy_all = np.concatenate([np.asarray(xy, dtype=np.float32).reshape(-1, 2) for xy, _ in groups if np.asarray(xy).shape[0]], axis=0)
        ids_all = np.concatenate([np.full(n, k, dtype=np.int32) for n, (_, k) in zip(sizes, groups) if n])
        depth = torch.from_numpy(np.ascontiguousarray(p_depth, dtype=np.float32)).to(dev)
        out = ops.unproject_depth(self.camera.okp(), torch.from_numpy(np.ascontiguousarray(xy_all)).to(dev), torch.from_numpy(ids_all).to(dev),
                                  depth, int(self.max_index[0]), int(self.max_index[1])).cpu().numpy()
        res, at = [], 0
        for n in sizes:
            res.append(out[at:at + n] if n else None)
            at += n
        return res


class ObjectKeypointPipeline:
    def __init__(self, prediction_size, points_3d, keypoint_config):
        self.keypoint_extraction = KeypointExtractionComponent(keypoint_config, prediction_size)
        self.object_extraction = ObjectExtraction(keypoint_config, prediction_size)
        self.detection_to_point = DetectionToPoint()

    def reset(self, camera):
        self.detection_to_point.reset(camera)

    def __call__(self, heatmap, p_depth, p_centers):
        assert heatmap.shape[0] == 1, "One at the time, please."
        heatmap = heatmap.numpy() if isinstance(heatmap, torch.Tensor) else np.asarray(heatmap)
        p_centers = (p_centers[0].numpy() if isinstance(p_centers, torch.Tensor) else np.asarray(p_centers)[0])
        p_depth = (p_depth[0].numpy() if isinstance(p_depth, torch.Tensor) else np.asarray(p_depth)[0])
        points, confidence = self.keypoint_extraction(heatmap)
        detected_objects = self.object_extraction(points[0], confidence[0], p_centers)
        # the reference lifts group by group (pipeline.py:190-200); here all groups of the frame go through one launch
        groups = []
        for obj in detected_objects:
            groups.append((obj['center'][None], 0))
            for i in range(len(obj['heatmap_points'])):
                groups.append((obj['heatmap_points'][i], 1 + i))
        lifted = self.detection_to_point.lift_groups(groups, p_depth) if groups else []
        objects, at = [], 0
        for obj in detected_objects:
            n_groups = 1 + len(obj['heatmap_points'])
            objects.append({'p_centers': obj['p_centers'],
                            'keypoints': [obj['center'][None]] + obj['heatmap_points'],
                            'p_C': lifted[at:at + n_groups]})
            at += n_groups
        return objects


class LearnedKeypointTrackingPipeline(ObjectKeypointPipeline):
    def __init__(self, model, cuda=True, *args, **kwargs):
        compute_dtype = kwargs.pop("compute_dtype", torch.float32)
        on_overflow = kwargs.pop("on_overflow", "float32")
        super().__init__(*args, **kwargs)
        self.inference = InferenceComponent(model, cuda, compute_dtype=compute_dtype, on_overflow=on_overflow)

    def __call__(self, frame):
        heatmap, depth, centers = self.inference(frame)
        return super().__call__(heatmap, depth, centers), heatmap


class TriangulationComponent:
    """reset(stereo_camera); __call__(left (n,2), right (n,2)) -> (n,3) in the left camera frame."""

    def reset(self, stereo_camera):
        self.stereo_camera = stereo_camera

    def __call__(self, left_keypoints, right_keypoints):
        return self.stereo_camera.triangulate(left_keypoints, right_keypoints)


class AssociationComponent:
    """Left/right keypoint matching before triangulation: reset(stereo_camera); __call__(points_left (n,2),
    points_right (m,2)) -> int array (n,), -1 = unmatched.  The reference tree has no implementation; the contract is
    its test (test/test_pipeline.py:208-261).  Minimum-cost one-to-one assignment on the symmetric epipolar distance
    of the (device-)undistorted points; matches beyond `max_distance` pixels are dropped."""

    def __init__(self, max_distance=20.0):
        self.max_distance = float(max_distance)

    def reset(self, stereo_camera):
        self.stereo_camera = stereo_camera

    def cost(self, points_left, points_right):
        cam = self.stereo_camera
        ul = cam.left_camera.undistort(np.asarray(points_left, dtype=np.float64))
        ur = cam.right_camera.undistort(np.asarray(points_right, dtype=np.float64))
        hl = np.concatenate([ul, np.ones((ul.shape[0], 1))], axis=1)
        hr = np.concatenate([ur, np.ones((ur.shape[0], 1))], axis=1)
        lines_r, lines_l = hl @ cam.F.T, hr @ cam.F
        num = np.abs(lines_r @ hr.T)
        d_r = num / np.maximum(np.linalg.norm(lines_r[:, :2], axis=1), 1e-300)[:, None]
        d_l = num / np.maximum(np.linalg.norm(lines_l[:, :2], axis=1), 1e-300)[None, :]
        return 0.5 * (d_r + d_l)

    def __call__(self, points_left, points_right):
        from scipy.optimize import linear_sum_assignment
        points_left, points_right = np.asarray(points_left), np.asarray(points_right)
        out = np.full(points_left.shape[0], -1, dtype=np.int64)
        if points_left.shape[0] == 0 or points_right.shape[0] == 0:
            return out
        cost = self.cost(points_left, points_right)
        # gate first: a point without any admissible partner must not steal one (all its pairs cost the same)
        rows, cols = linear_sum_assignment(np.where(cost <= self.max_distance, cost, 1e6))
        keep = cost[rows, cols] <= self.max_distance
        out[rows[keep]] = cols[keep]
        return out


# Graph capture keeps the hourglass' forked side streams (fork / join by events captures as graph edges).  Round 1 captured
# without them because replays differed from the eager step at batch 64; the cause was a WAR race inside okp_fire2_kernel
# (fixed in round 2, see its phase-1 comment), not the graph: replays are bit-equal to the eager step either way
# (tests/test_gpu_bench_shapes.py::test_graph_replay_equals_eager_at_batch64).  False = capture the branches serially.
GRAPH_SIDE_STREAMS = True


def _live_plans(net):
    """Every device plan object currently cached by the modules of `net` (ConvPlan / StemPlan / tuples of them)."""
    found = []

    def walk(v):
        if isinstance(v, (tuple, list)):
            for e in v:
                walk(e)
        elif isinstance(v, (ops.ConvPlan, ops.StemPlan)) or isinstance(v, torch.Tensor):
            found.append(v)

    for m in net.modules():
        for v in getattr(m, "_plans", {}).values():
            walk(v)
    return found


class CapturedStep:
    """A captured hipGraph of BatchedKeypointPipeline.forward_device plus everything its kernel nodes point at.

    The graph holds raw device pointers to the plans' packed weights: this object keeps those plan objects alive (a
    load_state_dict() or .to() on the network clears the modules' plan caches, which would otherwise free them under
    the graph), and replay() refuses to run once the network's plans are no longer the captured ones - the graph would
    silently compute with the old weights."""

    def __init__(self, net, graph, static_in, static_out):
        self.net, self.graph, self.static_in, self.static_out = net, graph, static_in, static_out
        self._pinned = _live_plans(net)
        # every module that caches plans, with its plan epoch at capture time: the per-replay staleness test is ~150 integer
        # compares (walking and comparing the plan objects themselves cost a replay ~0.4 ms of host time)
        self._epochs = [(m, m._plan_epoch) for m in net.modules() if hasattr(m, "_plan_epoch")]

    def stale(self):
        return any(m._plan_epoch != e for m, e in self._epochs)

    def replay(self, frames=None):
        if self.stale():
            raise OkpError("the network's weights / plans changed after capture(): capture the step again")
        if frames is not None:
            self.static_in.copy_(frames)
        self.graph.replay()
        return self.static_out

    def __iter__(self):                      # graph, static_in, static_out = pipe.capture(x)
        return iter((self, self.static_in, self.static_out))


class BatchedKeypointPipeline:
    """Frames -> heat/depth/centre maps -> peaks -> per-peak 3D points for a whole batch, device-resident.

    forward_device(frames) returns device tensors
        heat [N,K,H,W], depth [N,K,H,W], centers [N,K-1,2,H,W],
        count [N,K] int32, xyc [N,K,cap,3] fp32 (x, y, confidence), points [N,K,cap,4] fp64 (X, Y, Z, confidence),
        overflow (0-d int32, non-zero: some map exceeded `capacity` peaks or `max_objects` centres - results are truncated)
    `points` is the fixed-capacity payload that is all-gathered across ranks (object_keypoints_amd.distributed).
    objects(...) groups one frame's peaks exactly as ObjectKeypointPipeline does, reusing the lifted points (surplus votes of a
    multi-instance type reduced by the device k-means of okp_group_objects).
    """

    def __init__(self, net, keypoint_config, camera, prediction_size=(64, 64), capacity=DEFAULT_PEAK_CAPACITY,
                 max_objects=16, max_per_type=4):
        self.net = net
        self.max_objects = max_objects
        self.max_per_type = max_per_type
        self.config = keypoint_config
        self.capacity = capacity
        self.prediction_size = list(prediction_size)
        self.object_extraction = ObjectExtraction(keypoint_config, self.prediction_size)
        self.camera = camera
        self.cam = camera.okp()
        self.max_index = camera.image_size.astype(int) - 1

    def forward_device(self, frames):
        """Split-product configurations: the network's fp16-range flag is not read here (no host sync in the batch loop) but folded into
        `overflow` on the device (bit ops.RANGE_OVERFLOW), which objects() checks - except with on_overflow = "float32", where the pass has
        to be judged on the host before it can be re-run."""
        defer = getattr(self.net, "on_overflow", "raise") != "float32"
        heat, depth, centers = self.net.deployed(frames, check_range=not defer)
        return self.postprocess_device(heat, depth, centers, range_flag=self.net.range_flag(frames.device) if defer else None)

    def capture(self, frames):
        """Capture forward_device for this frame shape into a hipGraph (HIP stream capture through torch): the ~75
        launches of a step become one graph launch, which removes the host launch cost that dominates small
        batches (batch 8: 2.7 -> 2.2 ms per step).  Returns a CapturedStep; it also unpacks as
        (step, static_input, static_outputs): replay with step.replay(new_frames) or static_input.copy_(...); step.replay()."""
        static_in = frames.clone()
        for _ in range(2):                                   # warm-up: plans, allocator pools, kernel attributes
            self.forward_device(static_in)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        keep, ops.SIDE_STREAMS = ops.SIDE_STREAMS, ops.SIDE_STREAMS and GRAPH_SIDE_STREAMS
        try:
            with torch.cuda.graph(graph):
                static_out = self.forward_device(static_in)
        finally:
            ops.SIDE_STREAMS = keep
        return CapturedStep(self.net, graph, static_in, static_out)

    def postprocess_device(self, heat, depth, centers, range_flag=None):
        """Peaks, per-peak 3D points and the object grouping from (post-sigmoid) heat, depth and centre maps, all on
        the device: three small launches, nothing crosses PCIe."""
        count, yx, xyc = ops.peak_nms(heat, cap=self.capacity)
        points = ops.lift_peaks(self.cam, count, xyc, depth, int(self.max_index[0]), int(self.max_index[1]))
        out = {"heat": heat, "depth": depth, "centers": centers, "count": count, "yx": yx, "xyc": xyc, "points": points}
        out.update(ops.group_objects(count, xyc, centers, self.config['keypoint_config'], max_obj=self.max_objects,
                                     max_sel=self.max_per_type))
        # The reference has no capacity (pipeline.py:73 keeps every peak); the fixed-capacity tensors do.  `overflow` is a
        # device-side flag (no sync here): a map with more peaks than `capacity`, or more centre peaks than `max_objects`,
        # means `points` / the grouping are truncated - callers check it (objects() raises, bench.py asserts).
        out["overflow"] = ops.capacity_overflow(count, self.capacity, self.max_objects, range_flag=range_flag)
        return out

    def objects(self, out, n):
        """Frame n of a forward_device() result as the reference's list of object dicts
        ({'p_centers', 'keypoints', 'p_C'}, pipeline.py:190-200), assembled from the device-side grouping.  A multi-instance type that
        received more votes than it has instances comes back as the cluster centres of the device k-means reduction (`reduced`,
        okp_group_objects; the reference: an unseeded sklearn KMeans, pipeline.py:143-148), lifted to 3D by one okp_unproject_depth
        launch for the frame - no scikit-learn, no host clustering."""
        cfg = self.config['keypoint_config']
        count = out["count"][n].cpu().numpy()
        xyc = out["xyc"][n].cpu().numpy()
        pts3 = out["points"][n].cpu().numpy()
        n_obj = int(out["n_obj"][n])
        votes = out["n_votes"][n].cpu().numpy()
        if int(count.max(initial=0)) > self.capacity or int(count[0]) > self.max_objects:
            raise OkpError("peak / object capacity exceeded; raise `capacity` / `max_objects`")
        if int(out["overflow"]) & ops.RANGE_OVERFLOW:
            raise OkpError("an activation of the split-product network left the fp16 range in this batch: its maps are not fp32-grade "
                           "(load the network with on_overflow='float32', or run it in torch.float32)")
        if any(c > self.max_per_type for c in cfg):
            raise OkpError(f"a keypoint type has {max(cfg)} instances, above max_per_type = {self.max_per_type}")
        sel = out["sel"][n].cpu().numpy()
        assign = out["assign"][n].cpu().numpy()
        pred = out["pred"][n].cpu().numpy()
        surplus = [(o, i) for o in range(n_obj) for i in range(len(cfg)) if cfg[i] > 1 and votes[o, i] > cfg[i]]
        centres, lifted = {}, {}
        if surplus:
            reduced = out["reduced"][n].cpu().numpy()
            for o, i in surplus:
                centres[(o, i)] = reduced[o, i, :cfg[i]].copy()
            xy = np.concatenate([centres[key] for key in surplus])
            ids = np.concatenate([np.full(cfg[i], i + 1, dtype=np.int32) for _, i in surplus])
            dev = out["depth"].device
            world = ops.unproject_depth(self.cam, torch.from_numpy(xy).to(dev), torch.from_numpy(ids).to(dev), out["depth"][n],
                                        int(self.max_index[0]), int(self.max_index[1])).cpu().numpy()
            at = 0
            for o, i in surplus:
                lifted[(o, i)] = world[at:at + cfg[i]]
                at += cfg[i]
        objects = []
        for o in range(n_obj):
            keypoints = [xyc[0, o, :2].copy()[None]]
            world = [pts3[0, o, :3][None]]
            for i in range(len(cfg)):
                if (o, i) in centres:
                    keypoints.append(centres[(o, i)])
                    world.append(lifted[(o, i)])
                    continue
                idx = [int(j) for j in sel[o, i] if j >= 0][:max(cfg[i], 1)]
                if idx:
                    keypoints.append(xyc[i + 1, idx, :2].copy())
                    world.append(pts3[i + 1, idx, :3])
                else:
                    keypoints.append(np.array([]))
                    world.append(None)
            p_centers = [pred[k, j] for k in range(1, count.shape[0]) for j in range(int(count[k])) if assign[k, j] == o]
            objects.append({'p_centers': p_centers, 'keypoints': keypoints, 'p_C': world})
        return objects

    def objects_reference(self, out, n):
        """The same frame through the host ObjectExtraction (the reference's loop with scikit-learn's k-means): for comparisons only -
        objects() never comes here."""
        return self._objects_host(out, n, out["count"][n].cpu().numpy(), out["xyc"][n].cpu().numpy(), out["points"][n].cpu().numpy())

    def _objects_host(self, out, n, count, xyc, pts3):
        centers = out["centers"][n].cpu().numpy()
        keypoints = [[xyc[k, j, :2].copy() for j in range(int(count[k]))] for k in range(count.shape[0])]
        confidence = [[xyc[k, j, 2] for j in range(int(count[k]))] for k in range(count.shape[0])]
        lifted = {}
        for k in range(count.shape[0]):
            for j in range(int(count[k])):
                lifted[(k, xyc[k, j, 0].tobytes(), xyc[k, j, 1].tobytes())] = pts3[k, j, :3]
        detected = self.object_extraction(keypoints, confidence, centers)
        d2p = None
        objects = []
        for obj in detected:
            groups = [obj['center'][None]] + obj['heatmap_points']
            world = []
            for k, pts in enumerate(groups):
                if pts.shape[0] == 0:
                    world.append(None)
                    continue
                rows = []
                for p in pts:
                    key = (k, np.float32(p[0]).tobytes(), np.float32(p[1]).tobytes())
                    if key in lifted:
                        rows.append(lifted[key])
                    else:                      # k-means centre: not a detected peak, lift it on demand
                        if d2p is None:
                            d2p = DetectionToPoint(); d2p.reset(self.camera)
                        rows.append(d2p(np.asarray(p, dtype=np.float32)[None], out["depth"][n, k].cpu().numpy())[0])
                world.append(np.stack(rows))
            objects.append({'p_centers': obj['p_centers'], 'keypoints': groups, 'p_C': world})
        return objects


class HostFrameFeed:
    """Host -> HBM upload of camera frames on a COPY stream, double-buffered, so that the frames of tick t + 1 cross PCIe while tick t is
    computed (the reference feeds its model from a DataLoader on the host, one frame per call: scripts/eval_model.py:274-293; raw frames
    are 1280 x 720 uint8, perception/datasets/video.py:83-100).  Frames come as a pinned host tensor of the shape / dtype given at
    construction (pin_memory() once, reuse it: the copy is only asynchronous from pinned memory).

        feed = HostFrameFeed(host.shape, host.dtype)
        slot = feed.submit(host)                 # async H2D on the copy stream into the next free device buffer
        frames = feed.acquire(slot)              # the compute stream waits for that copy (device-side wait, no host sync)
        ...                                      # launches that read `frames`
        feed.release(slot)                       # the buffer may be overwritten once the work enqueued so far has run
    """

    def __init__(self, shape, dtype=torch.uint8, device=None, depth=2):
        self.device = device or _device()
        self.buffers = [torch.empty(tuple(shape), dtype=dtype, device=self.device) for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.ready = [torch.cuda.Event() for _ in range(depth)]
        self.done = [None] * depth
        self.next = 0

    def submit(self, host_frames):
        if host_frames.is_cuda or tuple(host_frames.shape) != tuple(self.buffers[0].shape) or host_frames.dtype != self.buffers[0].dtype:
            raise OkpError("HostFrameFeed.submit: a host tensor of the shape / dtype the feed was built for")
        slot = self.next
        self.next = (self.next + 1) % len(self.buffers)
        with torch.cuda.stream(self.copy_stream):
            if self.done[slot] is not None:
                self.copy_stream.wait_event(self.done[slot])          # the buffer's previous tick has been consumed
            self.buffers[slot].copy_(host_frames, non_blocking=True)
            self.ready[slot].record(self.copy_stream)
        return slot

    def acquire(self, slot):
        torch.cuda.current_stream().wait_event(self.ready[slot])
        return self.buffers[slot]

    def release(self, slot):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.done[slot] = ev


class StereoStreamPipeline:
    """BASELINE configs[4] / SURVEY 8(d) config 5: a rig of stereo cameras streamed frame-synchronously.  One TICK takes the 2P
    frames of P stereo pairs (left frame of pair p at index 2p, right at 2p + 1) through the network as ONE batch, finds the heat-map
    peaks of all 2P x K maps in one launch, matches left and right peaks per pair and key-point type (AssociationComponent's
    epipolar assignment, test/test_pipeline.py:208-261), and triangulates every matched pair of the tick in ONE
    okp_triangulate_dlt launch (StereoCamera.triangulate, camera_utils.py:92-110).  The reference's harness processes one frame
    of one camera per call at a 30 Hz cap (scripts/eval_model.py:274-293).  Geometry is fp32 / fp64 whatever the network's
    compute dtype.

    tick(frames) -> list over pairs of {type k: (n_k, 3) float64 points in the left camera frame}, plus timing-free device work:
    three small device -> host copies per tick (peaks, undistorted peaks, 3D points).
    `frames`: fp32 NCHW crops [2P, 3, 511, 511] in HBM, or raw camera frames uint8 [2P, H, W, 3] (resized, cropped and normalised on the
    device, okp_preprocess_u8).  Host-fed operation: stream(host_ticks) uploads the raw frames of tick t + 1 on a copy stream while tick t
    is computed (HostFrameFeed) and yields the same results, bit for bit, as tick() on resident frames."""

    def __init__(self, net, stereo_camera, keypoint_config, capacity=16, max_distance=3.0):
        self.net = net
        self.stereo = stereo_camera
        self.config = [1] + list(keypoint_config["keypoint_config"])
        self.capacity = capacity
        self.assoc = AssociationComponent(max_distance=max_distance)
        self.assoc.reset(stereo_camera)
        self.cam_l = ops.make_camera(stereo_camera.left_camera.K, stereo_camera.left_camera.D)      # fisheye model, as the reference
        self.cam_r = ops.make_camera(stereo_camera.right_camera.K, stereo_camera.right_camera.D)
        self._graph = None

    def capture(self, frames):
        """Capture the network pass of a tick for this frame shape into a hipGraph (one graph launch instead of ~66)."""
        static_in = frames.clone()
        for _ in range(2):
            self.net.deployed(static_in)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        keep, ops.SIDE_STREAMS = ops.SIDE_STREAMS, ops.SIDE_STREAMS and GRAPH_SIDE_STREAMS
        try:
            with torch.cuda.graph(graph):
                static_out = self.net.deployed(static_in)
        finally:
            ops.SIDE_STREAMS = keep
        self._graph = CapturedStep(self.net, graph, static_in, static_out)
        return self._graph

    def stream(self, host_ticks, heat_override=None, use_graph=False):
        """host_ticks: an iterable of pinned host tensors, one per tick (all of one shape / dtype: raw uint8 [2P, H, W, 3] frames or fp32
        [2P, 3, 511, 511] crops).  Generator of tick() results; the upload of the next tick overlaps the current one."""
        it = iter(host_ticks)
        try:
            first = next(it)
        except StopIteration:
            return
        feed = HostFrameFeed(first.shape, first.dtype, device=next(self.net.parameters()).device)
        slot = feed.submit(first)
        while slot is not None:
            frames = feed.acquire(slot)
            upcoming = next(it, None)
            nxt = feed.submit(upcoming) if upcoming is not None else None      # crosses PCIe under this tick's kernels
            result = self.tick(frames, heat_override=heat_override, use_graph=use_graph)
            feed.release(slot)
            yield result
            slot = nxt

    def tick(self, frames, heat_override=None, use_graph=False):
        if frames.shape[0] % 2:
            raise OkpError("a tick holds the left and right frame of every stereo pair: an even number of frames")
        with torch.no_grad():
            if use_graph:
                if self._graph is None:
                    raise OkpError("capture() the tick first")
                heat, depth, centers = self._graph.replay(frames)
            else:
                heat, depth, centers = self.net.deployed(frames)
            if heat_override is not None:
                heat = heat_override
            count, _, xyc = ops.peak_nms(heat, cap=self.capacity)
            count_h = count.cpu().numpy()                                   # D2H 1 (tiny)
            if int(count_h.max(initial=0)) > self.capacity:
                raise OkpError("peak capacity exceeded in a tick; raise `capacity`")
            n_pairs, K = frames.shape[0] // 2, heat.shape[1]
            # undistort every left and every right peak of the tick in two launches (the cost of the assignment needs them)
            xy = xyc[..., :2].reshape(n_pairs, 2, K, self.capacity, 2)
            ul = ops.camera_undistort(self.cam_l, xy[:, 0].reshape(-1, 2))
            ur = ops.camera_undistort(self.cam_r, xy[:, 1].reshape(-1, 2))
            und = torch.stack([ul, ur]).cpu().numpy().reshape(2, n_pairs, K, self.capacity, 2)      # D2H 2
            raw = xy.cpu().numpy()
        F = self.stereo.F
        left_sel, right_sel, owner = [], [], []
        from scipy.optimize import linear_sum_assignment
        for p in range(n_pairs):
            for k in range(K):
                cl, cr = int(count_h[2 * p, k]), int(count_h[2 * p + 1, k])
                if cl == 0 or cr == 0:
                    continue
                hl = np.concatenate([und[0, p, k, :cl], np.ones((cl, 1))], axis=1)
                hr = np.concatenate([und[1, p, k, :cr], np.ones((cr, 1))], axis=1)
                lines_r, lines_l = hl @ F.T, hr @ F
                num = np.abs(lines_r @ hr.T)
                cost = 0.5 * (num / np.maximum(np.linalg.norm(lines_r[:, :2], axis=1), 1e-300)[:, None]
                              + num / np.maximum(np.linalg.norm(lines_l[:, :2], axis=1), 1e-300)[None, :])
                rows, cols = linear_sum_assignment(np.where(cost <= self.assoc.max_distance, cost, 1e6))
                keep = cost[rows, cols] <= self.assoc.max_distance
                for i, j in zip(rows[keep], cols[keep]):
                    left_sel.append(raw[p, 0, k, i]); right_sel.append(raw[p, 1, k, j]); owner.append((p, k))
        result = [{k: np.zeros((0, 3)) for k in range(K)} for _ in range(n_pairs)]
        if left_sel:
            dev = frames.device
            pts = ops.triangulate_dlt(self.cam_l, self.cam_r, self.stereo.T_RL, torch.from_numpy(np.stack(left_sel)).to(dev),
                                      torch.from_numpy(np.stack(right_sel)).to(dev), F=F).cpu().numpy()          # D2H 3
            for (p, k), X in zip(owner, pts):
                result[p][k] = np.concatenate([result[p][k], X[None]])
        return result

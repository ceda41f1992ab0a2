"""Case table shared by the golden-vector generator (runs the reference) and the tests
(run the oracle and the HIP path).  A case names a block of the reference network, its
constructor arguments, the input shape and the seeds of the procedural weights / input
(object_keypoints_amd.synth).  Only outputs are stored; weights and inputs are regenerated.
"""

# kind -> (constructor args, input NCHW shape)
BLOCK_CASES = {
    # convolution(k, inp, out, stride, with_bn)  utils.py:143-156
    "conv3x3_bn_relu":      dict(kind="convolution", args=(3, 8, 16), kwargs={}, x=(2, 8, 13, 11)),
    "conv1x1_bn_relu":      dict(kind="convolution", args=(1, 16, 8), kwargs={}, x=(2, 16, 7, 9)),
    "conv7x7_s2_stem":      dict(kind="convolution", args=(7, 3, 16), kwargs=dict(stride=2), x=(2, 3, 31, 31)),
    "conv3x3_nobn":         dict(kind="convolution", args=(3, 8, 8), kwargs=dict(with_bn=False), x=(1, 8, 6, 6)),
    # residual(inp, out, k=3, stride)  utils.py:158-185
    "residual_same":        dict(kind="residual", args=(16, 16), kwargs={}, x=(2, 16, 12, 10)),
    "residual_s2_proj":     dict(kind="residual", args=(8, 16), kwargs=dict(stride=2), x=(2, 8, 14, 14)),
    "residual_s2_odd":      dict(kind="residual", args=(8, 16), kwargs=dict(stride=2), x=(1, 8, 15, 13)),
    # fire_module(inp, out, sr=2, stride)  CornerNet_Squeeze.py:10-30
    "fire_skip":            dict(kind="fire_module", args=(16, 16), kwargs={}, x=(2, 16, 9, 12)),
    "fire_widen":           dict(kind="fire_module", args=(16, 32), kwargs={}, x=(2, 16, 8, 8)),
    "fire_s2":              dict(kind="fire_module", args=(16, 32), kwargs=dict(stride=2), x=(2, 16, 12, 12)),
    "fire_s2_odd":          dict(kind="fire_module", args=(16, 16), kwargs=dict(stride=2), x=(1, 16, 9, 7)),
    # ConvTranspose2d(dim, dim, 4, 2, 1) + merge  CornerNet_Squeeze.py:35-36, utils.py:139-141
    "unpool_merge":         dict(kind="unpool_merge", args=(16,), kwargs={}, x=(2, 16, 5, 6)),
    # hg_module(n, dims, modules) with the Squeeze factories  modules.py:25-66
    "hg_module_2level":     dict(kind="hg_module", args=(2, [16, 16, 32], [1, 1, 2]), kwargs={}, x=(2, 16, 16, 16)),
    "hg_module_1level":     dict(kind="hg_module", args=(1, [16, 32], [2, 2]), kwargs={}, x=(1, 16, 8, 12)),
    # prediction_module(int_features, features_out)  models.py:13-18
    "prediction_module":    dict(kind="prediction_module", args=(16, 3), kwargs={}, x=(2, 256, 6, 5)),
}
BLOCK_WEIGHT_SEED = 11
BLOCK_INPUT_SEED = 12

# whole network: KeypointNet(features=128, heatmaps_out=K) + deployed wrapper
NET_CASES = {
    "valve_k3": dict(heatmaps_out=3, weight_seed=0, frame_seed=1, frame_index=0),   # config/valve.json [1,3]
    "cups_k4":  dict(heatmaps_out=4, weight_seed=0, frame_seed=1, frame_index=1),   # config/cups.json [1,1,1]
}


# ---------------------------------------------------------------------------------------------
# post-network pipeline cases: heat / depth / centre maps generated procedurally
# ---------------------------------------------------------------------------------------------
import numpy as _np


def _bump(size, cx, cy, length_scale=2.0, gain=1.0):
    ys, xs = _np.meshgrid(_np.arange(size, dtype=_np.float32), _np.arange(size, dtype=_np.float32), indexing="ij")
    d2 = (xs - _np.float32(cx)) ** 2 + (ys - _np.float32(cy)) ** 2
    return (_np.float32(gain) * _np.exp(-d2 / _np.float32(length_scale ** 2))).astype(_np.float32)


def pipeline_case(name):
    """-> dict(config=..., heat (K,64,64), depth (K,64,64), centers (K-1,2,64,64)) float32."""
    from object_keypoints_amd import synth
    size = 64
    if name == "valve_one":
        s = synth.bump_scene([1, 3], n_objects=1, seed=3, index=0)
        return dict(config=[1, 3], heat=s["heat"], depth=s["depth"], centers=s["centers"])
    if name == "valve_two":
        s = synth.bump_scene([1, 3], n_objects=2, seed=3, index=2)
        return dict(config=[1, 3], heat=s["heat"], depth=s["depth"], centers=s["centers"])
    if name == "cups_four":
        s = synth.bump_scene([1, 1, 1], n_objects=4, seed=3, index=1)
        return dict(config=[1, 1, 1], heat=s["heat"], depth=s["depth"], centers=s["centers"])
    if name == "special":
        heat = _np.zeros((4, size, size), dtype=_np.float32)
        heat[0] = _bump(size, 20.5, 30.0)                              # symmetric: two tied maxima survive
        heat[1] = _np.clip(_bump(size, 0.3, 0.2) + _bump(size, 63.0, 31.0) + _bump(size, 40.0, 63.0), 0, 1)   # border peaks
        heat[2] = _np.clip(3.0 * _bump(size, 33.0, 17.0, length_scale=3.0), 0, 1)    # saturated 1.0 plateau
        heat[3] = 0.0                                                   # empty map
        return dict(config=[1, 1, 1], heat=heat, depth=_np.ones_like(heat), centers=_np.zeros((3, 2, size, size), _np.float32))
    if name == "noise":
        heat = _np.stack([synth.uniform(f"noise{k}", (size, size), 7, 0.0, 0.12) for k in range(4)])
        return dict(config=[1, 1, 1], heat=heat.astype(_np.float32), depth=_np.ones_like(heat), centers=_np.zeros((3, 2, size, size), _np.float32))
    if name == "weak":
        heat = _np.stack([_bump(size, 10 + 9 * k, 12 + 7 * k, gain=g) for k, g in enumerate((0.03, 0.11, 0.12, 0.3))])
        return dict(config=[1, 1, 1], heat=heat, depth=_np.ones_like(heat), centers=_np.zeros((3, 2, size, size), _np.float32))
    raise KeyError(name)


PIPELINE_CASES = ["valve_one", "valve_two", "cups_four", "special", "noise", "weak"]
OBJECT_CASES = ["valve_one", "valve_two", "cups_four"]       # cases with consistent centre / depth maps

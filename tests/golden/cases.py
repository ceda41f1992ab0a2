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

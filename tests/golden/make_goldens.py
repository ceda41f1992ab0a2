#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation (read-only at
/root/reference) on procedurally generated weights and inputs.

Runs only in the build container (the reference does not exist on the GPU box).
Stores OUTPUTS only: weights and inputs are regenerated from seeds by
object_keypoints_amd.synth, and no reference source text is copied.

    python tests/golden/make_goldens.py [net] [blocks] [pipeline] [geometry]
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

from _ref_import import import_reference  # noqa: E402
from object_keypoints_amd import synth    # noqa: E402
import cases                              # noqa: E402


def _load_synth(module, seed):
    import torch
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    vals = synth.fill_state_dict(shapes, seed=seed)
    module.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in vals.items()})
    return module.eval()


def _build_block(ref, case):
    import torch
    from perception.corner_net_lite.core.models import CornerNet_Squeeze as sq
    from perception.corner_net_lite.core.models.py_utils import utils as cu
    from perception.corner_net_lite.core.models.py_utils import modules as cm
    kind, args, kwargs = case["kind"], case["args"], case["kwargs"]
    if kind == "convolution":
        return cu.convolution(*args, **kwargs)
    if kind == "residual":
        return cu.residual(*args, **kwargs)
    if kind == "fire_module":
        return sq.fire_module(*args, **kwargs)
    if kind == "prediction_module":
        return ref.models.prediction_module(*args)
    if kind == "hg_module":
        return cm.hg_module(*args, make_pool_layer=sq.make_pool_layer, make_unpool_layer=sq.make_unpool_layer,
                            make_up_layer=sq.make_layer, make_low_layer=sq.make_layer,
                            make_hg_layer_revr=sq.make_layer_revr, make_hg_layer=sq.make_hg_layer)
    if kind == "unpool_merge":
        class UnpoolMerge(torch.nn.Module):
            def __init__(self, dim):
                super().__init__()
                self.up2 = sq.make_unpool_layer(dim)
                self.merg = cu.merge()

            def forward(self, x):
                low, up1 = x
                return self.merg(up1, self.up2(low))
        return UnpoolMerge(*args)
    raise KeyError(kind)


def block_inputs(name, case):
    """Shared with the tests: the input tensor(s) of a block case."""
    x = synth.normal_like(f"{name}/x", case["x"], cases.BLOCK_INPUT_SEED)
    if case["kind"] == "unpool_merge":
        n, c, h, w = case["x"]
        up1 = synth.normal_like(f"{name}/up1", (n, c, 2 * h, 2 * w), cases.BLOCK_INPUT_SEED)
        return (x, up1)
    return x


def make_blocks(ref):
    import torch
    out = {}
    for name, case in cases.BLOCK_CASES.items():
        mod = _load_synth(_build_block(ref, case), cases.BLOCK_WEIGHT_SEED)
        x = block_inputs(name, case)
        with torch.no_grad():
            if isinstance(x, tuple):
                y = mod(tuple(torch.from_numpy(t) for t in x))
            else:
                y = mod(torch.from_numpy(x))
        out[name] = y.numpy().astype(np.float32)
        print(f"  block {name:22s} -> {out[name].shape} |y|max={np.abs(out[name]).max():.3f}")
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **out)


def make_net(ref):
    import torch
    for name, case in cases.NET_CASES.items():
        torch.manual_seed(0)
        net = ref.models.KeypointNet([64, 64], features=128, heatmaps_out=case["heatmaps_out"])
        _load_synth(net, case["weight_seed"])
        x = torch.from_numpy(synth.frames(1, seed=case["frame_seed"], start=case["frame_index"]))
        with torch.no_grad():
            heat, depth, centers = net(x)
            # scripts/package_model.py:26-28 — the deployed wrapper
            dep = (torch.sigmoid(heat[-1]), depth[-1], centers[-1])
        np.savez_compressed(os.path.join(HERE, f"net_{name}.npz"),
                            heat=dep[0].numpy(), depth=dep[1].numpy(), centers=dep[2].numpy(),
                            logits=heat[-1].numpy(),
                            stack1_heat=heat[0].numpy())
        print(f"  net {name}: heat {tuple(dep[0].shape)} depth {tuple(dep[1].shape)} centers {tuple(dep[2].shape)}")


def main():
    what = set(sys.argv[1:]) or {"net", "blocks", "pipeline", "geometry"}
    ref = import_reference()
    if "blocks" in what:
        make_blocks(ref)
    if "net" in what:
        make_net(ref)
    if "pipeline" in what:
        import make_goldens_pipeline
        make_goldens_pipeline.main(ref)
    meta = {"generator": "tests/golden/make_goldens.py", "reference": "ethz-asl/object_keypoints @ /root/reference",
            "torch": __import__("torch").__version__, "numpy": np.__version__}
    with open(os.path.join(HERE, "META.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()

"""Parity of the HIP path with the REFERENCE (golden vectors from tests/golden/make_goldens.py) and
with the oracle, block by block, through the product's mirrored modules and the C ABI."""
import numpy as np
import pytest
import torch

import cases
import golden_util as gu

pytestmark = pytest.mark.gpu


def _product_block(case):
    from object_keypoints_amd.perception import backbone as bb
    from object_keypoints_amd.perception import models as pm
    kind, args, kwargs = case["kind"], case["args"], case["kwargs"]
    if kind == "unpool_merge":
        class M(torch.nn.Module):
            def __init__(self, dim):
                super().__init__()
                self.up2 = bb.unpool_merge(dim)
        return M(*args)
    if kind == "prediction_module":
        return pm.prediction_module(*args)
    return getattr(bb, kind)(*args, **kwargs)


def _run(name, dtype):
    from object_keypoints_amd import ops, synth
    dev = torch.device("cuda:0")
    case = cases.BLOCK_CASES[name]
    mod = _product_block(case)
    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    vals = synth.fill_state_dict(shapes, seed=cases.BLOCK_WEIGHT_SEED)
    mod.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in vals.items()})
    mod.eval()
    x = gu.block_inputs(name, case)
    if case["kind"] == "unpool_merge":
        low, up1 = [ops.Act.from_nchw(torch.from_numpy(t).to(dev), dtype) for t in x]
        return mod.up2(low, up1).to_nchw().cpu().numpy()
    if case["kind"] == "convolution" and case["args"][1] == 3:
        y = mod(ops.pack_frames(torch.from_numpy(x).to(dev), dtype))
    else:
        y = mod(ops.Act.from_nchw(torch.from_numpy(x).to(dev), dtype))
    if isinstance(y, torch.Tensor):
        return y.cpu().numpy()
    return y.to_nchw().cpu().numpy()


BLOCKS = sorted(n for n, c in cases.BLOCK_CASES.items() if c["kind"] != "convolution" or c["kwargs"].get("with_bn", True) or True)


@pytest.mark.parametrize("name", BLOCKS)
def test_block_fp32_matches_reference_golden(name):
    got = _run(name, torch.float32)
    ref = gu.golden_blocks()[name]
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    assert err <= 1e-3, f"{name}: max |err| {err}"          # north_star: 1e-3 on fp32 heat maps; blocks are O(1)
    assert err <= 2e-4 * max(1.0, np.abs(ref).max()), f"{name}: fp32 path should be near round-off, got {err}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("name", BLOCKS)
def test_block_16bit_tracks_reference_golden(name, dtype):
    got = _run(name, dtype)
    ref = gu.golden_blocks()[name]
    scale = np.abs(ref).max()
    err = np.abs(got - ref)
    # 16-bit storage between layers: 8 (bf16) or 11 (fp16) significant bits per tensor, a handful of layers per block
    f = 1.0 if dtype == torch.bfloat16 else 0.15
    assert err.max() <= f * 0.06 * scale, f"{name}: max |err| {err.max()} vs scale {scale}"
    assert err.mean() <= f * 0.012 * scale

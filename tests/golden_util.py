"""Helpers shared by the tests: rebuild the oracle module and the inputs of a golden case."""
import os

import numpy as np
import torch

import cases
from object_keypoints_amd import synth
from oracle import net as onet

GOLDEN = os.path.dirname(os.path.abspath(cases.__file__))


class _UnpoolMerge(torch.nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.up2 = torch.nn.ConvTranspose2d(dim, dim, 4, 2, 1)

    def forward(self, x):
        low, up1 = x
        return up1 + self.up2(low)


def build_oracle_block(case):
    kind, args, kwargs = case["kind"], case["args"], case["kwargs"]
    if kind == "unpool_merge":
        return _UnpoolMerge(*args)
    ctor = getattr(onet, kind)
    return ctor(*args, **kwargs)


def block_inputs(name, case):
    x = synth.normal_like(f"{name}/x", case["x"], cases.BLOCK_INPUT_SEED)
    if case["kind"] == "unpool_merge":
        n, c, h, w = case["x"]
        up1 = synth.normal_like(f"{name}/up1", (n, c, 2 * h, 2 * w), cases.BLOCK_INPUT_SEED)
        return (x, up1)
    return x


def load_block(name):
    case = cases.BLOCK_CASES[name]
    mod = onet.load_synthetic(build_oracle_block(case), seed=cases.BLOCK_WEIGHT_SEED)
    return case, mod, block_inputs(name, case)


def golden_blocks():
    return np.load(os.path.join(GOLDEN, "blocks.npz"))


def golden_net(name):
    return np.load(os.path.join(GOLDEN, f"net_{name}.npz"))

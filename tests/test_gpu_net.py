"""Whole-network parity on the GPU: HIP KeypointNet vs the reference's golden outputs (fp32 within the
north_star tolerance 1e-3 on heat maps), plus the MAC count the roofline is computed from."""
import numpy as np
import pytest
import torch

import cases
import golden_util as gu

pytestmark = pytest.mark.gpu


def _net(case, dtype):
    from object_keypoints_amd import synth
    from object_keypoints_amd.perception.models import KeypointNet
    net = KeypointNet(features=128, heatmaps_out=case["heatmaps_out"], compute_dtype=dtype)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    vals = synth.fill_state_dict(shapes, seed=case["weight_seed"])
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in vals.items()})
    return net.eval()


@pytest.mark.parametrize("name", sorted(cases.NET_CASES))
def test_fp32_network_matches_reference(name):
    from object_keypoints_amd import ops, synth
    case = cases.NET_CASES[name]
    net = _net(case, torch.float32)
    x = torch.from_numpy(synth.frames(1, seed=case["frame_seed"], start=case["frame_index"])).cuda()
    ops.COUNTERS["macs"] = 0
    heat, depth, centers = net.deployed(x)
    torch.cuda.synchronize()
    g = gu.golden_net(name)
    assert tuple(heat.shape) == g["heat"].shape and tuple(centers.shape) == g["centers"].shape
    e_heat = np.abs(heat.cpu().numpy() - g["heat"]).max()
    e_depth = np.abs(depth.cpu().numpy() - g["depth"]).max()
    e_cent = np.abs(centers.cpu().numpy() - g["centers"]).max()
    print(f"{name}: heat err {e_heat:.2e} depth err {e_depth:.2e} centers err {e_cent:.2e}")
    assert e_heat <= 1e-3                                   # north_star tolerance
    assert e_depth <= 1e-3 * max(1.0, np.abs(g["depth"]).max())
    assert e_cent <= 1e-3 * max(1.0, np.abs(g["centers"]).max())
    if case["heatmaps_out"] == 3:
        assert ops.COUNTERS["macs"] == 37_282_609_152       # deployed path, K=3


def test_full_forward_returns_both_stacks():
    from object_keypoints_amd import synth
    case = cases.NET_CASES["valve_k3"]
    net = _net(case, torch.float32)
    x = torch.from_numpy(synth.frames(1, seed=case["frame_seed"], start=case["frame_index"])).cuda()
    (h1, h2), (d1, d2), (c1, c2) = net(x)
    g = gu.golden_net("valve_k3")
    assert np.abs(h2.cpu().numpy() - g["logits"]).max() <= 2e-3
    assert np.abs(h1.cpu().numpy() - g["stack1_heat"]).max() <= 2e-3
    assert tuple(c1.shape) == (1, 2, 2, 64, 64)


def test_bf16_network_tracks_reference():
    from object_keypoints_amd import synth
    case = cases.NET_CASES["valve_k3"]
    net = _net(case, torch.bfloat16)
    x = torch.from_numpy(synth.frames(2, seed=case["frame_seed"], start=case["frame_index"])).cuda()
    heat, depth, centers = net.deployed(x)
    g = gu.golden_net("valve_k3")
    err = np.abs(heat[0].cpu().numpy() - g["heat"][0])
    print(f"bf16 heat err max {err.max():.3e} mean {err.mean():.3e}")
    # bf16 activations through ~60 layers: stated tolerance for the throughput configuration
    assert err.mean() <= 0.02 and err.max() <= 0.25


def test_batch_independence_and_eval_only():
    from object_keypoints_amd import ops, synth
    case = cases.NET_CASES["valve_k3"]
    net = _net(case, torch.float32)
    x = torch.from_numpy(synth.frames(3, seed=5)).cuda()
    h3, _, _ = net.deployed(x)
    h1, _, _ = net.deployed(x[1:2])
    assert torch.equal(h3[1:2], h1)                        # frames are independent: same bits at any batch
    net.train()
    with pytest.raises(ops.OkpError):
        net.deployed(x)
    with pytest.raises(ops.OkpError):
        net.eval().deployed(x.cpu())                        # no CPU fallback

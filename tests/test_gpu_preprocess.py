"""On-device pre-processing (okp_preprocess_u8: resize + centre crop + normalise + pack) against the oracle's restatement
of the reference's data path (albumentations SmallestMaxSize + CenterCrop over cv2.resize, then video.py:215).
Integer resize arithmetic and IEEE-ordered normalisation: bit-exact in fp32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _expected_packed(frames_u8, size):
    from oracle import pipeline as op
    from oracle import preprocess as pre
    crops = np.stack([pre.resize_center_crop(f, size) for f in frames_u8])
    return op.normalize_frames(crops)                                   # [N,3,size,size] fp32


@pytest.mark.parametrize("shape,size", [((2, 720, 1280), 511), ((1, 97, 61), 33), ((3, 40, 40), 40), ((1, 33, 200), 21)])
def test_preprocess_matches_oracle_bit_exact(shape, size):
    from object_keypoints_amd import ops
    n, h, w = shape
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    ref = _expected_packed(frames, size)
    act = ops.preprocess_u8(torch.from_numpy(frames).cuda(), torch.float32, size=size)
    got = act.t.cpu().numpy()                                           # [N, size+6, Wp, 4] with the image at (3,3)
    assert got.shape[1] == size + 6 and got.shape[3] == 4
    inner = got[:, 3:3 + size, 3:3 + size, :3].transpose(0, 3, 1, 2)
    assert np.array_equal(inner, ref)
    halo = got.copy()
    halo[:, 3:3 + size, 3:3 + size, :3] = 0
    assert not halo.any()                                               # zero halo and zero 4th channel
    # the already-cropped path gives the same packed tensor
    crops = np.stack([__import__("oracle.preprocess", fromlist=["x"]).resize_center_crop(f, size) for f in frames])
    same = ops.pack_frames_u8(torch.from_numpy(crops).cuda(), torch.float32).t.cpu().numpy()
    assert np.array_equal(same, got)


def test_network_takes_raw_camera_frames():
    """KeypointNet on raw 720x1280 uint8 frames == KeypointNet on the oracle-preprocessed fp32 frames."""
    from object_keypoints_amd import synth
    from object_keypoints_amd.perception.models import KeypointNet
    net = KeypointNet(features=128, heatmaps_out=3, compute_dtype=torch.bfloat16)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    vals = synth.fill_state_dict(shapes, seed=0)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in vals.items()})
    net.eval()
    rng = np.random.default_rng(6)
    frames = rng.integers(0, 256, (1, 720, 1280, 3), dtype=np.uint8)
    a = net.deployed(torch.from_numpy(frames).cuda())
    b = net.deployed(torch.from_numpy(_expected_packed(frames, 511)).cuda())
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_preprocess_rejects_bad_input():
    from object_keypoints_amd import ops
    with pytest.raises(ops.OkpError):
        ops.preprocess_u8(torch.zeros(1, 3, 20, 20, dtype=torch.uint8).cuda(), torch.float32, size=16)   # NCHW
    with pytest.raises(ops.OkpError):
        ops.preprocess_u8(torch.zeros(1, 20, 20, 3, dtype=torch.float32).cuda(), torch.float32, size=16)

"""Properties of the oracle's cv::resize(INTER_LINEAR, 8U) restatement (oracle/preprocess.py).  OpenCV is not importable
here, so this stage is parity-unpinned against the library itself; the properties below are what any correct
implementation of the published algorithm satisfies."""
import numpy as np

from oracle import preprocess as pre


def test_geometry_of_the_reference_frames():
    assert pre.smallest_max_size(720, 1280, 511) == (511, 908)       # video.py:63-69: 1280x720 -> 908x511 -> crop at x = 198
    img = np.zeros((720, 1280, 3), np.uint8)
    assert pre.resize_center_crop(img).shape == (511, 511, 3)


def test_constant_and_identity():
    img = np.full((37, 53, 3), 173, np.uint8)
    assert (pre.resize_linear_u8(img, (19, 31)) == 173).all()        # weights sum to 2048: constants are preserved exactly
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (23, 29, 3), dtype=np.uint8)
    assert (pre.resize_linear_u8(img, (23, 29)) == img).all()        # same size: fx = 0 everywhere


def test_close_to_float_bilinear():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (72, 128, 3), dtype=np.uint8)
    out = pre.resize_linear_u8(img, (51, 91)).astype(np.float64)
    sy = np.clip((np.arange(51) + 0.5) * (72 / 51) - 0.5, 0, 71); sx = np.clip((np.arange(91) + 0.5) * (128 / 91) - 0.5, 0, 127)
    y0 = np.floor(sy).astype(int); x0 = np.floor(sx).astype(int)
    y1 = np.minimum(y0 + 1, 71); x1 = np.minimum(x0 + 1, 127)
    fy = (sy - y0)[:, None, None]; fx = (sx - x0)[None, :, None]
    f = img.astype(np.float64)
    ref = (f[y0][:, x0] * (1 - fx) + f[y0][:, x1] * fx) * (1 - fy) + (f[y1][:, x0] * (1 - fx) + f[y1][:, x1] * fx) * fy
    assert np.abs(out - ref).max() <= 1.0                            # 11-bit weights + two truncations: within one grey level


def test_monotone_ramp_stays_monotone():
    ramp = np.tile(np.linspace(0, 255, 200).astype(np.uint8)[None, :, None], (8, 1, 3))
    out = pre.resize_linear_u8(ramp, (8, 77)).astype(int)
    assert (np.diff(out[0, :, 0]) >= 0).all()

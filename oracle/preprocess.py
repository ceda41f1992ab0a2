"""ORACLE (test infrastructure, not product code): NumPy restatement of the frame pre-processing in front of the network.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.

Reference being restated: perception/datasets/video.py:95-96 (albumentations SmallestMaxSize(511) + CenterCrop(511, 511),
both thin wrappers over cv2.resize(INTER_LINEAR) and array slicing) and :215 (normalisation, see oracle/pipeline.py).

Third-party arithmetic: cv2.resize lives in OpenCV (vendored env pins opencv 3.4.2, corner_net_lite/conda_packagelist.txt:53),
which is NOT importable here.  `resize_linear_u8` restates the published algorithm of cv::resize for CV_8U / INTER_LINEAR
(modules/imgproc/src/resize.cpp: resizeGeneric_, HResizeLinear<uchar,int,short,INTER_RESIZE_COEF_SCALE=2048>,
VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,INTER_RESIZE_COEF_BITS*2>>).  PARITY UNPINNED against OpenCV itself
(no golden vectors can be generated); checked by properties in tests/test_oracle_preprocess.py.  An OpenCV build that
dispatches this call to IPP may differ from the generic path by one grey level.
"""
import numpy as np


def _coeffs(dsize, ssize):
    scale = 1.0 / (float(dsize) / float(ssize))                      # cv::resize: inv_scale = dsize / ssize; scale = 1 / inv_scale
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    low, high = s < 0, s >= ssize - 1
    f = np.where(low | high, np.float32(0), f).astype(np.float32)
    s = np.where(low, 0, np.where(high, ssize - 1, s))
    w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)     # saturate_cast<short>(cvRound(.)): ties to even
    w1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, np.minimum(s + 1, ssize - 1), w0, w1


def resize_linear_u8(img, dsize_hw):
    """uint8 [H,W,C] -> uint8 [h,w,C], cv::resize(INTER_LINEAR) fixed-point arithmetic."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    dh, dw = dsize_hw
    sy0, sy1, b0, b1 = _coeffs(dh, img.shape[0])
    sx0, sx1, a0, a1 = _coeffs(dw, img.shape[1])
    src = img.astype(np.int64)
    h = src[:, sx0, :] * a0[None, :, None] + src[:, sx1, :] * a1[None, :, None]       # horizontal pass, int
    h0, h1 = h[sy0], h[sy1]
    out = (((b0[:, None, None] * (h0 >> 4)) >> 16) + ((b1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def smallest_max_size(h, w, max_size):
    """albumentations.SmallestMaxSize geometry (py3round = round half to even)."""
    scale = max_size / min(h, w)
    return int(round(h * scale)), int(round(w * scale))


def resize_center_crop(img, size=511):
    """SmallestMaxSize(size) + CenterCrop(size, size) as the reference's SceneDataset applies them (video.py:95-96)."""
    rh, rw = smallest_max_size(img.shape[0], img.shape[1], size)
    r = resize_linear_u8(img, (rh, rw))
    y0, x0 = (rh - size) // 2, (rw - size) // 2
    return r[y0:y0 + size, x0:x0 + size]

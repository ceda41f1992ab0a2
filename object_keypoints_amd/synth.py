"""Procedural (counter-based) weights, frames and heat/depth/centre maps.

There is no network access for checkpoints or datasets, so every test, golden
vector and benchmark uses values generated here from an integer seed.  The
generator is a pure-NumPy splitmix64 hash of (seed, crc32(name), element index):
no torch RNG, no NumPy distribution code, so the same (seed, name, shape) gives
the same bits in the build container, on the GPU box and in the golden-vector
generator that runs the reference (tests/golden/make_goldens.py).

Weight scales are chosen so activations stay O(1) through the ~60-layer
stacked hourglass (the reference's default init leaves the logits in
[0.005, 0.013], useless as a parity signal — SURVEY.md §8(c)).
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def unit_uniform(name, shape, seed):
    """float32 array in [0, 1), a pure function of (name, shape, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    tag = np.uint64(zlib.crc32(name.encode("utf-8")))
    base = (np.uint64(seed & 0xFFFFFFFF) << np.uint64(32)) | tag
    with np.errstate(over="ignore"):
        h = _splitmix64(_splitmix64(np.full(n, base, dtype=np.uint64))
                        + np.arange(n, dtype=np.uint64))
    u = (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return u.reshape(shape)


def uniform(name, shape, seed, lo, hi):
    return (np.float32(lo) + np.float32(hi - lo) * unit_uniform(name, shape, seed)).astype(np.float32)


def normal_like(name, shape, seed):
    """Approximately N(0,1) float32 (sum of 4 uniforms, variance-corrected)."""
    acc = np.zeros(shape, dtype=np.float32)
    for i in range(4):
        acc += unit_uniform(f"{name}#{i}", shape, seed)
    return ((acc - np.float32(2.0)) * np.float32(np.sqrt(3.0))).astype(np.float32)


# ----------------------------------------------------------------------------
# state_dict filling
# ----------------------------------------------------------------------------

def _is_bn_group(prefix, keys):
    return (prefix + ".running_mean") in keys


def _is_head_output(prefix):
    parts = prefix.split(".")
    return len(parts) >= 2 and parts[-1] == "2" and parts[-2].startswith("output_head")


def fill_state_dict(shapes, seed=0, branch_gain=0.3, head_gain=0.45):
    """shapes: {state_dict key: shape tuple} -> {key: np.ndarray}.

    Conv / conv-transpose weights: uniform with variance gain^2/fan_in.
    BatchNorm: gamma in [0.8,1.2] (x branch_gain for the last BN of a residual
    branch, so `relu(branch + skip)` keeps O(1) scale through both hourglasses), beta in [-0.2,0.2],
    running_mean in [-0.2,0.2], running_var in [0.6,1.4].
    """
    keys = set(shapes)
    out = {}
    for key, shape in shapes.items():
        shape = tuple(shape)
        prefix, _, leaf = key.rpartition(".")
        if leaf == "num_batches_tracked":
            out[key] = np.array(100, dtype=np.int64)
        elif leaf == "running_mean":
            out[key] = uniform(key, shape, seed, -0.2, 0.2)
        elif leaf == "running_var":
            out[key] = uniform(key, shape, seed, 0.6, 1.4)
        elif _is_bn_group(prefix, keys):
            if leaf == "weight":
                g = uniform(key, shape, seed, 0.8, 1.2)
                last = prefix.rsplit(".", 1)[-1]
                # bn2 closes fire/residual branches; skip.1 is the projected skip
                if last == "bn2":
                    g = g * np.float32(branch_gain)
                out[key] = g.astype(np.float32)
            else:
                out[key] = uniform(key, shape, seed, -0.2, 0.2)
        elif leaf == "weight":
            if len(shape) != 4:
                raise ValueError(f"unexpected weight {key} {shape}")
            if prefix.endswith("up2"):           # ConvTranspose2d (Cin, Cout, 4, 4), stride 2
                fan_in = shape[0] * 4            # 2x2 taps reach each output pixel
                gain = 1.0
            else:
                fan_in = shape[1] * shape[2] * shape[3]
                gain = np.sqrt(2.0)
                if _is_head_output(prefix):      # last 1x1 of a head: keeps logits ~N(0, 1.5^2)
                    gain = head_gain
            b = gain * np.sqrt(3.0 / fan_in)
            out[key] = uniform(key, shape, seed, -b, b)
        elif leaf == "bias":
            out[key] = uniform(key, shape, seed, -0.1, 0.1)
        else:
            raise ValueError(f"unexpected state_dict entry {key}")
    return out


def frames(n, seed=1, start=0, size=511):
    """(n,3,size,size) float32 ~N(0,1): stands in for the normalised RGB crop
    (reference perception/datasets/video.py:215)."""
    return np.stack([normal_like(f"frame{start + i}", (3, size, size), seed) for i in range(n)])


# ----------------------------------------------------------------------------
# synthetic heat / depth / centre maps (SURVEY.md §8(d) "Heatmaps for NMS/3D stages")
# ----------------------------------------------------------------------------

def bump_scene(keypoint_config, n_objects=1, seed=0, index=0, size=64, length_scale=2.0,
               margin=6.0, max_radius=9.0):
    """One frame of post-network maps for `n_objects` objects.

    Returns dict with heat (K,size,size), depth (K,size,size), centers (K-1,2,size,size),
    and the generating 2D points/depths.  Gaussian bumps exp(-|d|^2/l^2) as in the
    reference's target rendering (perception/datasets/video.py:22-25,44-53), clipped to [0,1].
    """
    config = [1] + list(keypoint_config)
    K = len(config)
    heat = np.zeros((K, size, size), dtype=np.float32)
    depth = np.zeros((K, size, size), dtype=np.float32)
    centers = np.zeros((K - 1, 2, size, size), dtype=np.float32)
    ys, xs = np.meshgrid(np.arange(size, dtype=np.float32), np.arange(size, dtype=np.float32), indexing="ij")
    objects = []
    tag = f"scene{index}"
    # object centres on a jittered grid so objects stay > 20 px apart when possible
    g = int(np.ceil(np.sqrt(n_objects)))
    cell = (size - 2 * margin) / g
    for o in range(n_objects):
        gy, gx = divmod(o, g)
        j = unit_uniform(f"{tag}/o{o}/c", (2,), seed)
        cx = margin + (gx + 0.3 + 0.4 * j[0]) * cell
        cy = margin + (gy + 0.3 + 0.4 * j[1]) * cell
        obj = {"center": np.array([cx, cy], dtype=np.float32), "points": [], "depths": []}
        radius = min(0.35 * cell, max_radius)
        for k, count in enumerate(config):
            pts = []
            for i in range(count):
                if k == 0:
                    p = np.array([cx, cy], dtype=np.float32)
                else:
                    a = unit_uniform(f"{tag}/o{o}/k{k}/{i}", (2,), seed)
                    ang = 2 * np.pi * (a[0] + (i + 0.37 * k) / max(count, 1))
                    r = radius * (0.55 + 0.45 * a[1])
                    p = np.array([cx + r * np.cos(ang), cy + r * np.sin(ang)], dtype=np.float32)
                    p = np.clip(p, 2.0, size - 3.0).astype(np.float32)
                z = np.float32(0.3 + 1.2 * unit_uniform(f"{tag}/o{o}/k{k}/{i}/z", (1,), seed)[0])
                d2 = (xs - p[0]) ** 2 + (ys - p[1]) ** 2
                heat[k] += np.exp(-d2 / np.float32(length_scale ** 2)).astype(np.float32)
                near = d2 < np.float32(16.0)
                depth[k][near] = z
                if k > 0:
                    centers[k - 1, 0][near] = (cx - (xs + 0.5))[near]
                    centers[k - 1, 1][near] = (cy - (ys + 0.5))[near]
                pts.append((p, z))
            obj["points"].append([p for p, _ in pts])
            obj["depths"].append([z for _, z in pts])
        objects.append(obj)
    heat = np.clip(heat, 0.0, 1.0).astype(np.float32)
    return {"heat": heat, "depth": depth, "centers": centers, "objects": objects}


def add_double_detections(scene, k, offset=(4.0, 3.0), size=64, length_scale=2.0):
    """Every keypoint of map `k` (>= 1) of a bump_scene gets a second bump at `offset` pixels from it, with the same depth and a centre
    vector that points at the same object: each instance is detected twice, so the object receives twice as many votes for that type as it
    has instances - the case the reference reduces with k-means (perception/pipeline.py:143-148).  Returns a new scene dict."""
    heat, depth, centers = scene["heat"].copy(), scene["depth"].copy(), scene["centers"].copy()
    ys, xs = np.meshgrid(np.arange(size, dtype=np.float32), np.arange(size, dtype=np.float32), indexing="ij")
    extra = np.zeros((size, size), dtype=np.float32)
    for obj in scene["objects"]:
        cx, cy = obj["center"]
        for p, z in zip(obj["points"][k], obj["depths"][k]):
            q = np.clip(p + np.array(offset, dtype=np.float32), 2.0, size - 3.0).astype(np.float32)
            d2 = (xs - q[0]) ** 2 + (ys - q[1]) ** 2
            extra += np.exp(-d2 / np.float32(length_scale ** 2)).astype(np.float32)
            near = d2 < np.float32(16.0)
            depth[k][near] = z
            centers[k - 1, 0][near] = (cx - (xs + 0.5))[near]
            centers[k - 1, 1][near] = (cy - (ys + 0.5))[near]
    heat[k] = np.clip(heat[k] + extra, 0.0, 1.0).astype(np.float32)
    return {"heat": heat, "depth": depth, "centers": centers, "objects": scene["objects"]}

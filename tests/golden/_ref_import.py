"""Import shim for the *reference* package (read-only at /root/reference).

Only used by the golden-vector generators in this directory, which run in the
build container.  Nothing on the GPU box imports this file: /root/reference
does not exist there.  Recipe from SURVEY.md §8(c): stub the optional
third-party modules the reference imports at module scope but never executes
on the keypoint path, and run with cwd=/root/reference because
perception/models.py:71 opens its config with a cwd-relative path.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("OKP_REFERENCE_ROOT", "/root/reference")


def import_reference():
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    import numpy as np
    if not hasattr(np, "int"):
        np.int = int  # perception/pipeline.py:161-162,168 uses the removed alias

    def stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    stub("timm")
    stub("cv2")
    stub("numba", jit=lambda *a, **k: (lambda f: f))
    stub("h5py")
    sk = stub("skvideo")
    sk.io = stub("skvideo.io")
    stub("albumentations")
    for pool in ("top_pool", "bottom_pool", "left_pool", "right_pool"):
        stub(pool)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    os.chdir(REFERENCE_ROOT)
    import perception.models as models
    import perception.pipeline as pipeline
    from perception.utils import camera_utils, linalg
    return types.SimpleNamespace(models=models, pipeline=pipeline,
                                 camera_utils=camera_utils, linalg=linalg)

"""Build recipe for libokp_hip.so (gfx950 only, in-tree so it travels with the repo snapshot).

    python -m object_keypoints_amd.build [--force]

hipcc cross-compiles without a GPU; each .hip is compiled to build/*.o in parallel and linked
into object_keypoints_amd/lib/libokp_hip.so.
"""
import concurrent.futures
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libokp_hip.so")
TORCH_LIB_PATH = os.path.join(LIB_DIR, "libokp_torch.so")      # torch.ops.okp.*: dispatcher shim over the same C ABI (csrc/okp_torch.cpp)
OBJ_DIR = os.path.join(REPO, "build", "okp")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I" + os.path.join(REPO, "include"), "-I" + CSRC,
          "-Wall", "-Wno-unused-function"]
# keep MFMA accumulators in arch VGPRs: with 256-thread workgroups hipcc otherwise allocates them as
# AGPRs and copies all of them to and from VGPRs around every K-step (v_accvgpr_write/read x 128).
VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
AGPR_FILES = set()


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    return [os.path.join(REPO, "include", "okp.h"), os.path.join(CSRC, "okp_internal.h"),
            os.path.join(CSRC, "okp_igemm_kernel.h"), os.path.abspath(__file__)]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
    extra = [] if os.path.basename(src) in AGPR_FILES else VGPR_FORM
    flags = CFLAGS + extra + os.environ.get("OKP_EXTRA_CFLAGS", "").split()
    stamp = obj + ".flags"                      # an object built with other flags (experiment switches) is stale
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
    if not same_flags or _stale(obj, [src] + _deps()):
        cmd = [HIPCC] + flags + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        with open(stamp, "w") as f:
            f.write(" ".join(flags))
        return obj, True, r.stderr
    return obj, False, ""


def build_torch_ops(verbose=False):
    """g++ (host only) build of csrc/okp_torch.cpp against this interpreter's torch: registers torch.ops.okp.* on top of the
    extern "C" symbols of libokp_hip.so (found next to it through an $ORIGIN rpath).  No HIP code, no GPU needed."""
    import torch
    from torch.utils import cpp_extension
    src = os.path.join(CSRC, "okp_torch.cpp")
    if not _stale(TORCH_LIB_PATH, [src, os.path.join(REPO, "include", "okp.h"), LIB_PATH, os.path.abspath(__file__)]):
        return TORCH_LIB_PATH
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", src, "-o", TORCH_LIB_PATH,
           "-I" + os.path.join(REPO, "include"), f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += ["-I" + p for p in cpp_extension.include_paths()]
    cmd += ["-L" + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-L" + LIB_DIR, "-lokp_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"building libokp_torch.so failed:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return TORCH_LIB_PATH


def build(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    srcs = _sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [o for o, _, _ in results]
    rebuilt = any(r for _, r, _ in results)
    for f in os.listdir(OBJ_DIR):                 # objects (and flag stamps) of sources that no longer exist
        if f.endswith((".o", ".o.flags")) and os.path.join(OBJ_DIR, f.split(".o")[0] + ".o") not in objs:
            os.remove(os.path.join(OBJ_DIR, f))
    if verbose:
        for _, _, err in results:
            if err.strip():
                print(err, file=sys.stderr)
    if rebuilt or _stale(LIB_PATH, objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    try:
        build_torch_ops(verbose)
    except Exception as e:        # the dispatcher shim is optional: the ctypes binding serves every launch without it
        if os.path.exists(TORCH_LIB_PATH):
            os.remove(TORCH_LIB_PATH)          # never leave a stale shim built against another torch / ABI
        print(f"warning: libokp_torch.so (torch.ops.okp.*) not built, the ctypes binding will be used: {e}", file=sys.stderr)
    return LIB_PATH


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path)

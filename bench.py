#!/usr/bin/env python3
"""Benchmark of the keypoint-inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torchrun environment: this process never touches the GPU; it starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` as a child (one rank
per GPU over RCCL), relays rank 0's JSON line and exits with the child's code.  Under torchrun (WORLD_SIZE set)
the process is a rank and runs the benchmark directly.

One step = one pass of the hot path over one batch of synthetic 511x511 frames already resident in HBM:
fp32 NCHW frames -> stem -> CornerNet-Squeeze hourglass + heads (HIP implicit-GEMM convs) -> per-map peak-NMS ->
per-peak depth lifting to 3D -> object grouping -> (N>1) ONE all-gather of the fixed-capacity 3D-keypoint tensor.
Workload = BASELINE.json configs[2]: batch 64 per GPU, bf16 activations/weights with fp32 accumulation, K=3 maps
(config/valve.json).  Frames shard across ranks (weak scaling, weights replicated); value = frames of all ranks /
max-over-ranks time.  Random-weight networks give flat heat maps, so the stages after the heads are measured on
injected Gaussian-bump scenes (`synth.bump_scene`, 1-2 objects per frame, seed-indexed per global
frame); the network itself runs in full on the frames in every step.

Prints ONE JSON line (rank 0).  `roofline` = the dominant kernel (the patch-resident 3x3 implicit-GEMM kernel), timed
live with HIP events on the launch stream during the timed steps; `cpu_baseline` = the oracle (torch-CPU fp32
restatement + NumPy post-processing) on the host cores, N=1 only, bounded sample.  At N=1 the line also carries `f32mix`,
`f32x3`, `fp32` (BASELINE configs[1] precision) and `fp16` (configs[4] precision) objects, each {value, ms_per_step, roofline,
collective, heat_err_vs_oracle, depth_err_vs_oracle, peak_jaccard, p_C_err_m, meets}: every north_star tolerance of that
precision measured on the cpu_baseline's frames against the oracle (parity_fields).  `collective` describes the one
data-path collective as this run saw it (backend, ranks, payload, median all-gather time).  `roofline.counter_commit` /
`stale` say which tree the committed PMC counters were collected on.  `power` (N=1): board power and shader clock sampled with rocm-smi
while the step runs back to back AFTER the timed region, and the dominant kernel alone on normally distributed and on all-zero operands
(the same instructions at different power: the kernel runs at the board's power limit); `--no-probes` skips these
after-the-fact probes (the profiling passes of scripts/profile_all.sh use it: their per-kernel averages then hold the timed steps only).
`--workload cups512` = BASELINE configs[3] (config/cups.json: K = 4 maps, four objects per frame, 64 frames per GPU, the all-gather
payload [64 N, 4, cap, 4]); the default N = 1 line carries one rank's share of it as the `cups64` object.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

GFLOP_PER_FRAME = 74.565218304          # 37 282 609 152 MACs x 2 (deployed path, K=3)
PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3, "f32x3": 2500.0, "f32mix": 2500.0}   # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
# f32x3 (ops.F32X3): fp32 tensors, every product as three fp16 MFMA terms.  `achieved` counts the ALGORITHMIC FLOPs once, the peak is
# the fp16 pipe's: a perfect kernel of this kind would read frac = 1/3 (roofline.mfma_terms says so in the line).
PEAK_HBM_GBPS = 8000.0                  # HBM3E spec (6 290 GB/s measured with a float4 copy), same guide
ERR_FRAMES = 4                          # frames of the cpu_baseline sample on which every precision is compared with the oracle


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--dtype", choices=["bf16", "f16", "f32", "f32x3", "f32mix"], default="bf16", help="precision of the headline line")
    ap.add_argument("--extra-dtypes", default="f32mix,f32x3,f32,f16", help="N=1: further precisions reported as extra objects ('' = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--spawn", action="store_true", help="go through the launcher even for --gpus 1 (tests)")
    ap.add_argument("--workload", choices=["batch64", "cups512", "stream8"], default="batch64",
                    help="cups512: BASELINE configs[3] (K = 4, config/cups.json, four objects per frame) as the headline, at any N; "
                         "stream8: print only the BASELINE configs[4] object (the default run carries both as `cups64` / `stream8` next to the headline)")
    ap.add_argument("--no-stream8", action="store_true", help="N=1: skip the configs[4] object")
    ap.add_argument("--no-cups", action="store_true", help="N=1: skip the configs[3] object (`cups64`)")
    ap.add_argument("--no-host-fed", action="store_true", help="N=1: skip the `host_fed` object (the batch-64 step with its frames starting in pinned host memory)")
    ap.add_argument("--no-probes", action="store_true", help="skip the power / operand / collective probes behind the timed region (profiling passes)")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` starts its own ranks (no GPU call happens in this process)
# --------------------------------------------------------------------------------------------------

def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(args, argv, port):
    child_args = [a for a in argv if a != "--spawn"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py")] + child_args


def needs_launcher(args, environ):
    return "WORLD_SIZE" not in environ and (args.gpus > 1 or args.spawn)


def gpu_count_without_hip(environ=os.environ, sysfs="/sys/class/kfd/kfd/topology/nodes", dri="/dev/dri"):
    """GPUs this process could use, counted WITHOUT the HIP / HSA runtime: the launcher parent forks + execs torchrun, and a
    process that has initialised the GPU must not exec on this pool.  KFD topology nodes with SIMDs are GPUs (CPU nodes have
    simd_count 0); /dev/dri/renderD* is the fall-back; *_VISIBLE_DEVICES lists cap the count."""
    import glob
    import re
    n = 0
    for path in glob.glob(os.path.join(sysfs, "*", "properties")):
        try:
            m = re.search(r"^simd_count\s+(\d+)", open(path).read(), re.M)
        except OSError:
            continue
        n += 1 if m and int(m.group(1)) > 0 else 0
    if n == 0:
        n = len(glob.glob(os.path.join(dri, "renderD*")))
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip()]))
    return n


def self_launch(args, argv, device_count, popen=subprocess.Popen, out=sys.stdout, err=sys.stderr):
    """Start the ranks as a child process, relay rank 0's JSON line, return the exit code.  `device_count` comes from
    gpu_count_without_hip(): this process never touches the HIP runtime."""
    if device_count < args.gpus:
        print(f"bench.py --gpus {args.gpus} needs {args.gpus} devices, {device_count} visible", file=err)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("NCCL_DEBUG", "VERSION")               # one "RCCL version ..." line per run in the log (stderr)
    proc = popen(launcher_command(args, argv, free_port()), stdout=subprocess.PIPE, text=True, env=env)
    relayed = 0
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            out.write(line)
            out.flush()
            relayed += 1
        else:
            err.write(line)
    rc = proc.wait()
    if rc == 0 and relayed != 1:
        print(f"launcher: expected one JSON line from rank 0, got {relayed}", file=err)
        return 3
    return rc


# --------------------------------------------------------------------------------------------------
# rank body
# --------------------------------------------------------------------------------------------------

class KernelTimer:
    """ops.LAUNCH_HOOK: brackets the launches of named implicit-GEMM populations with HIP events recorded on the launch
    stream (torch's current stream is the stream the C ABI launches on).  A population = (plan dtype, tile codes, n_src or
    None, split-product plans only?)."""

    def __init__(self, populations):
        self.pops = populations            # {label: (torch dtype, tiles, n_src | None, split | None)}
        self.records = {k: [] for k in populations}
        self.enabled = False
        self.step = 0
        # a HIP event record costs the queue 6-7 us (a barrier packet): bracketing all 12 launches of every step took 1.3 % off the
        # step rate, so the launches of every 8th timed step are bracketed (still live, still inside the timed region)
        self.every = int(os.environ.get("OKP_BENCH_TIMER_EVERY", "8"))

    def before(self, plan, tile, macs):
        import torch
        if not self.enabled or self.step % self.every:
            return None
        for label, (dtype, tiles, n_src, split) in self.pops.items():
            if plan.dtype == dtype and tile in tiles and (n_src is None or plan.n_src == n_src) and (split is None or plan.split == split):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                return (label, e0, e1, macs, getattr(plan, "last_bytes", 0))
        return None

    def after(self, token):
        if token is not None:
            token[2].record()
            self.records[token[0]].append(token[1:])

    def summary(self, label):
        rec = self.records[label]
        ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in rec)
        return len(rec), ms, 2.0 * sum(m for _, _, m, _ in rec), float(sum(b for _, _, _, b in rec))


def git_head():
    try:
        return subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def counter_provenance(path):
    """Which tree a committed counter file was measured on: the `commit` its profile script stamped into `<file>.meta.json`
    (scripts/profile_all.sh), compared with the kernels' sources as they are now.  stale = some file under csrc/ or include/ changed since."""
    meta_path = path[:-5] + ".meta.json" if path.endswith(".json") else path + ".meta.json"
    commit = None
    if os.path.exists(meta_path):
        try:
            with open(meta_path) as f:
                commit = json.load(f).get("commit")
        except (OSError, ValueError):
            commit = None
    if commit is None:
        return {"counter_commit": None, "stale": True, "stale_reason": "no commit stamp next to the counter file (a pre-round-4 set)"}
    try:
        r = subprocess.run(["git", "-C", REPO, "diff", "--quiet", commit, "--", "object_keypoints_amd/csrc", "include"], capture_output=True, timeout=20)
        if r.returncode == 0:
            return {"counter_commit": commit, "stale": False}
        if r.returncode == 1:
            return {"counter_commit": commit, "stale": True, "stale_reason": "kernel sources changed since the counters were collected"}
    except Exception:
        pass
    # no git history on this box (a gpurun snapshot): compare with the source hash the profile script stamped
    try:
        with open(meta_path) as f:
            want = json.load(f).get("csrc_sha16")
        return {"counter_commit": commit, "stale": want != csrc_sha16(), **({} if want == csrc_sha16() else {"stale_reason": "kernel sources differ from the stamped hash"})}
    except Exception:
        return {"counter_commit": commit, "stale": None, "stale_reason": "cannot compare (no git history, no source hash)"}


def csrc_sha16():
    import hashlib
    h = hashlib.sha256()
    for d in ("object_keypoints_amd/csrc", "include"):
        for f in sorted(os.listdir(os.path.join(REPO, d))):
            if f.endswith((".hip", ".h", ".cpp")):
                with open(os.path.join(REPO, d, f), "rb") as fh:
                    h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def kernel_time_share(kernel_sig, precision):
    """Share of the step's kernel time the named kernel takes in the newest committed rocprofv3 kernel-stats table of this precision
    (profiles/r*_<precision>_kernel_stats.csv), as a sentence naming the file; says so when there is none."""
    import csv
    import glob
    import re
    tag = "" if precision == "bf16" else precision + "_"
    pat = re.compile(r"^r\d+[a-z]*_" + tag + r"kernel_stats\.csv$")
    files = sorted(f for f in glob.glob(os.path.join(REPO, "profiles", f"r*_{tag}kernel_stats.csv")) if pat.match(os.path.basename(f)))
    if not files:
        return "no committed kernel-stats table for this precision"
    try:
        with open(files[-1]) as f:
            rows = list(csv.DictReader(f))
        total = sum(float(r["TotalDurationNs"]) for r in rows)
        plain = lambda name: name.replace("(anonymous namespace)::", "")
        mine = sum(float(r["TotalDurationNs"]) for r in rows if any(sig in plain(r["Name"]) for sig in kernel_sig))
        return f"{100.0 * mine / total:.1f} % of the kernel time in profiles/{os.path.basename(files[-1])}"
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return f"profiles/{os.path.basename(files[-1])} could not be read"


def committed_counters(kernel_sig, precision="bf16"):
    """Counter figures of the dominant kernel from the committed PMC passes of this same command at this precision (bench.py
    cannot run the profiler on itself): HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) and, when present, MFMA-busy / LDS-wait fractions from
    the SQ pass.  Files: profiles/r*_pmc_hbm_traffic.json / r*_sq_counters.json for the bf16 headline,
    profiles/r*_<precision>_pmc_hbm_traffic.json / ..._sq_counters.json for the others (scripts/profile_all.sh).  Null when absent."""
    import glob
    import re
    res = {"traffic": None, "traffic_source": None, "mfma_busy": None, "lds_wait": None, "hbm_GBps": None, "counter_source": None,
           "counter_commit": None, "stale": None}
    tag = "" if precision == "bf16" else precision + "_"
    pat = re.compile(r"^r\d+[a-z]*_" + tag + r"(pmc_hbm_traffic|sq_counters)\.json$")
    def newest(kind):
        return sorted(f for f in glob.glob(os.path.join(REPO, "profiles", f"r*_{tag}{kind}.json")) if pat.match(os.path.basename(f)))
    files = newest("pmc_hbm_traffic")
    if files:
        with open(files[-1]) as f:
            data = json.load(f)
        for name, row in data.items():
            if any(sig in name for sig in kernel_sig):
                res["traffic"] = (row["fetch_MB_per_launch_corrected"] + row["write_MB_per_launch"]) * 1e6
                res["traffic_source"] = os.path.basename(files[-1])
                res.update(counter_provenance(files[-1]))
                break
    files = newest("sq_counters")
    if files:
        with open(files[-1]) as f:
            data = json.load(f)
        for name, row in data.items():
            if any(sig in name for sig in kernel_sig):
                res["mfma_busy"] = row.get("mfma_busy_frac")
                res["lds_wait"] = row.get("lds_wait_frac")
                res["hbm_GBps"] = row.get("hbm_GBps")
                res["counter_source"] = os.path.basename(files[-1])
                break
    return res


def build_net(dtype, heatmaps_out=3):
    import numpy as np
    import torch
    from object_keypoints_amd import synth
    from object_keypoints_amd.perception.models import KeypointNet
    net = KeypointNet(features=128, heatmaps_out=heatmaps_out, compute_dtype=dtype)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    vals = synth.fill_state_dict(shapes, seed=0)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in vals.items()})
    return net.eval()


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(seconds):
    """The oracle (a port: torch-CPU fp32 restatement + NumPy post-processing) on the host cores.  Also returns the
    oracle's heat / depth maps of the sample frames: the checker for the `*_vs_oracle` fields of every precision."""
    import numpy as np
    import torch
    from oracle import net as onet
    from oracle import pipeline as op
    from object_keypoints_amd import synth
    cores = usable_cores()
    torch.set_num_threads(cores)
    net = onet.load_synthetic(onet.KeypointNet(features=128, heatmaps_out=3), seed=0)
    cam = op.eval_camera(os.path.join(REPO, "config", "calibration.yaml"))
    d2p = op.DetectionToPoint(); d2p.reset(cam)
    x = torch.from_numpy(synth.frames(ERR_FRAMES, seed=1))
    keep = {}

    def step():
        heat, depth, _ = onet.deployed_forward(net, x)
        heat, depth = heat.numpy(), depth.numpy()
        keep["heat"], keep["depth"] = heat, depth
        for n in range(heat.shape[0]):
            for k in range(heat.shape[1]):
                idx = op.peak_indices(heat[n, k])[:64]
                pts, _ = op.refine_peaks(heat[n, k], idx)
                if pts:
                    d2p(np.stack(pts), depth[n, k])
    step()                                   # warm-up
    frames, t0 = 0, time.perf_counter()
    while True:
        step()
        frames += ERR_FRAMES
        el = time.perf_counter() - t0
        if el >= seconds and frames >= 2 * ERR_FRAMES:
            break
    return ({"value": frames / el, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
             "sample": f"{frames} frames (batches of {ERR_FRAMES}, 3x511x511 fp32): oracle torch-CPU forward + NumPy peak-NMS/centroid/depth lifting, {el:.1f} s"},
            keep)


def oracle_maps(heatmaps_out):
    """The oracle's heat / depth maps of the error-sample frames for a network of `heatmaps_out` maps (the checker of the configs[3]
    object's parity fields; the K = 3 maps come out of cpu_baseline's timed sample)."""
    import torch
    from oracle import net as onet
    from object_keypoints_amd import synth
    torch.set_num_threads(usable_cores())
    net = onet.load_synthetic(onet.KeypointNet(features=128, heatmaps_out=heatmaps_out), seed=0)
    heat, depth, _ = onet.deployed_forward(net, torch.from_numpy(synth.frames(ERR_FRAMES, seed=1)))
    return {"heat": heat.numpy(), "depth": depth.numpy()}


TORCH_DTYPES = {"bf16": "bfloat16", "f16": "float16", "f32": "float32", "f32x3": "float32", "f32mix": "float32"}
OBJECT_KEY = {"f32": "fp32", "f16": "fp16", "bf16": "bf16", "f32x3": "f32x3", "f32mix": "f32mix"}
# substrings of the dominant kernel's name in the rocprofv3 CSVs (mangled where the tool's demangler gives up on _Float16 / __bf16)
KERNEL_SIG = {"bf16": ("okp_igemm_patch_kernel",), "f16": ("okp_igemm_patch_kernel",),
              "f32": ("okp_igemm_kernel<float, 256, 256, 4, 2, 2, 128, 32, 1>", "okp_igemm_kernelIfLi256ELi256ELi4ELi2ELi2ELi128ELi32ELi1E"),
              "f32x3": ("okp_igemm_patch_x3_kernel",), "f32mix": ("okp_igemm_patch_x3_kernel",),
              "f32mix_hbm": ("okp_igemm_kernel<F32S, 128, 128,", "F32SELi128ELi128E")}
KERNEL_NAME = {"bf16": "okp_igemm_patch_kernel<bf16,256co x 16x16px>", "f16": "okp_igemm_patch_kernel<f16,256co x 16x16px>",
               "f32": "okp_igemm_kernel<f32,256x256,src1>", "f32x3": "okp_igemm_patch_x3_kernel<f32 split into 3 fp16 MFMA terms,256co x 16x16px>",
               "f32mix": "okp_igemm_patch_x3_kernel<f32 split into 3 fp16 MFMA terms,256co x 16x16px> (`cnvs`, transposed convolutions)",
               "f32mix_hbm": "okp_igemm_kernel<f32 split,128x128> (1x1 convolutions / fire modules on fp32 tensors)"}


# The two batch workloads: BASELINE configs[2] (the headline: valve, K = 3 maps, 1-2 objects per frame) and configs[3] (cups: K = 4 maps -
# centre + three single-instance keypoint types, config/cups.json - and four objects per frame; 64 frames per GPU, 512 at 8 GPUs)
WORKLOADS = {
    "batch64": {"config": "configs[2]", "heatmaps_out": 3, "keypoint_config": (1, 3), "objects": lambda i: 1 + i % 2, "seed": 7, "what": "K=3 (valve)", "scenes": "1-2 objects/frame"},
    "cups512": {"config": "configs[3]", "heatmaps_out": 4, "keypoint_config": (1, 1, 1), "objects": lambda i: 4, "seed": 31, "what": "K=4 (cups, config/cups.json)", "scenes": "4 objects/frame"},
}


def bump_maps(start, count, dev, wl):
    """Injected post-network maps for global frames start..start+count of workload `wl`."""
    import numpy as np
    import torch
    from object_keypoints_amd import synth
    scenes = [synth.bump_scene(list(wl["keypoint_config"]), n_objects=wl["objects"](start + i), seed=wl["seed"], index=start + i) for i in range(count)]
    to = lambda key: torch.from_numpy(np.stack([s[key] for s in scenes])).to(dev)
    n_peaks = sum(sum(len(p) for p in o["points"]) for s in scenes for o in s["objects"])
    return to("heat"), to("depth"), to("centers"), n_peaks


def run_precision(name, ctx, steps, warmup, workload="batch64"):
    """Timed steps of the hot path at one precision on this rank.  Returns (result dict, heat maps of the error sample)."""
    import torch
    from object_keypoints_amd import distributed as dist_, ops
    from object_keypoints_amd.perception.pipeline import BatchedKeypointPipeline
    dev, world, batch = ctx["dev"], ctx["world"], ctx["batch"]
    wl = WORKLOADS[workload]
    dtype = getattr(torch, TORCH_DTYPES[name])
    net = build_net({"f32x3": ops.F32X3, "f32mix": ops.F32MIX}.get(name, dtype), heatmaps_out=wl["heatmaps_out"]).to(dev)
    pipe = BatchedKeypointPipeline(net, {"keypoint_config": list(wl["keypoint_config"])}, ctx["camera"], capacity=64)
    frames = ctx["frames"]
    pool = ctx.get("frames_pool") or [frames]     # the timed steps walk through several different batches (see rank_main)
    if workload not in ctx["bumps"]:
        ctx["bumps"][workload] = bump_maps(ctx["start"], batch, dev, wl)
    b_heat, b_depth, b_centers, n_peaks = ctx["bumps"][workload]
    # dominant kernel: 16-bit = the patch-resident 3x3 kernel (tile 13, one symbol for one and two sources); fp32 = 256x256 gather tile;
    # float32x3 = the split-product patch kernel; float32mix has three populations (below)
    # split-product configurations: the patch-resident split kernel (tile 13 of fp32 plans: every 3x3 / transposed convolution of
    # float32x3, `cnvs` and the transposed convolutions of float32mix - whose fp16 branches are float16 plans and do not match)
    gather = name == "f32"
    pops = {"mfma": (dtype, (3,) if gather else (13,), 1 if gather else None, None)}
    if name == "f32mix":
        pops["hbm"] = (dtype, (2,), None, True)
        pops["mfma16"] = (torch.float16, (13,), None, None)        # the single-term fp16 branches: the fp16 patch kernel, the largest share of its kernel time
    timer = KernelTimer(pops)
    ops.LAUNCH_HOOK = timer

    def step(i=0):
        # split-product configurations: the network's fp16-range flag is folded into the step's `overflow` word on the device (no host read
        # in the timed loop; asserted clear behind it)
        net.deployed(pool[i % len(pool)], check_range=False)          # heat / depth / centre maps of the network (in full)
        out = pipe.postprocess_device(b_heat, b_depth, b_centers, range_flag=net.range_flag(dev))     # peaks -> 3D -> objects on the injected scenes
        return out, dist_.all_gather_keypoints(out["points"], total_frames=batch * world)

    with torch.no_grad():
        for i in range(warmup):
            out, gathered = step(i)
        m0 = ops.COUNTERS["macs"]
        net.deployed(frames)
        gflop_per_frame = 2.0 * (ops.COUNTERS["macs"] - m0) / batch / 1e9       # what this network's launches multiply (K = 3: GFLOP_PER_FRAME)
        dist_.barrier(); torch.cuda.synchronize()
        timer.enabled = True
        t0 = time.perf_counter()
        for i in range(steps):
            timer.step = i
            out, gathered = step(i)
        dist_.barrier(); torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        timer.enabled = False
        ops.LAUNCH_HOOK = None
        elapsed = dist_.max_over_ranks(elapsed, dev)
        assert gathered.shape[0] == batch * world
        assert not bool(out["overflow"]), "peak / object capacity exceeded, or an activation left the fp16 range (split-product configurations), in the timed step"
        found = int(out["count"].sum())
        assert batch <= found <= n_peaks, (found, n_peaks)      # at least the centre of every frame; bumps closer than the 5x5 window merge
        sample = None
        if ctx["err_frames"] is not None:
            # the error sample through the product path: maps, peak indices (okp_peak_nms) and per-peak 3D points (okp_lift_peaks)
            h, d, _ = net.deployed(ctx["err_frames"])
            cnt, yx, xyc = ops.peak_nms(h, cap=SAMPLE_CAP)
            pts = ops.lift_peaks(pipe.cam, cnt, xyc, d, int(pipe.max_index[0]), int(pipe.max_index[1]))
            sample = {k: v.cpu().numpy() for k, v in (("heat", h), ("depth", d), ("count", cnt), ("yx", yx), ("points", pts))}
        probes = world == 1 and not ctx.get("no_probes", False)
        coll = collective_probe(out["points"], batch, world, dev, own_group=probes and ctx.get("probe_collective", False)) if (probes or world > 1) else None
        power = power_probe(step) if probes and ctx.get("probe_power", False) else None
        if power is not None and name in ("bf16", "f16"):
            power["dominant_kernel_alone"] = operand_probe(dtype, dev)
        ctx["probe_collective"] = ctx["probe_power"] = False        # (once per run: the headline precision; the configs[3] object asks for its own collective probe)
    value = batch * world * steps / elapsed
    n_launch, k_ms, k_flops, _ = timer.summary("mfma")
    achieved = k_flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    peak = PEAK_TFLOPS[name]
    ctr = committed_counters(KERNEL_SIG[name], name)
    roof = {"bound": "mfma", "kernel": KERNEL_NAME[name], "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
            "frac": achieved / peak, "traffic": ctr["traffic"], "traffic_unit": "HBM bytes/launch",
            "traffic_source": ctr["traffic_source"], "mfma_busy": ctr["mfma_busy"], "lds_wait": ctr["lds_wait"],
            "hbm_GBps": ctr["hbm_GBps"], "counter_source": ctr["counter_source"], "counter_commit": ctr["counter_commit"], "stale": ctr["stale"],
            "launches_timed": n_launch, "avg_launch_us": (k_ms * 1e3 / n_launch) if n_launch else None,
            "avg_gflop_per_launch": (k_flops / n_launch / 1e9) if n_launch else None}
    # ... and the WHOLE STEP against the same peak (every launch, every gap): algorithmic FLOPs of the network x frames / step time / peak
    roof["step_frac"] = (GFLOP_PER_FRAME if workload == "batch64" else gflop_per_frame) * value / world / 1e3 / peak
    # algorithmic FLOPs per frame: SURVEY 8(d)'s figure for the K = 3 network; the K = 4 network of configs[3] by its own launches' MAC
    # count (the mixed configuration launches part of the stem twice: not an algorithmic count, so never taken from there)
    gf = GFLOP_PER_FRAME if workload == "batch64" else gflop_per_frame
    res = {"value": value, "ms_per_step": elapsed / steps * 1e3, "steps": steps, "gflop_per_frame": gf,
           "conv_stack_tflops_per_gpu": gf * value / world / 1e3, "roofline": roof}
    if name == "f32mix":
        # three populations: `roofline` = the kernel with the largest share of this configuration's kernel time, the fp16 patch kernel of its
        # single-term residual branches (conv1 / conv2 of pre[1], pre[2], inters[0]); `roofline_mfma` = the split-product patch kernel (`cnvs`,
        # transposed convolutions: three MFMA terms); `roofline_hbm` = the 128x128 split tiles on fp32 tensors (1x1 convolutions)
        n_h, ms_h, _, bytes_h = timer.summary("hbm")
        ctr_h = committed_counters(KERNEL_SIG["f32mix_hbm"], name)
        ach = bytes_h / (ms_h * 1e-3) / 1e9 if ms_h > 0 else 0.0
        res["roofline_mfma"] = roof
        res["roofline_hbm"] = {"bound": "hbm", "kernel": KERNEL_NAME["f32mix_hbm"], "achieved": ach, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                               "frac": ach / PEAK_HBM_GBPS, "traffic": ctr_h["traffic"], "traffic_unit": "HBM bytes/launch",
                               "traffic_source": ctr_h["traffic_source"], "hbm_GBps": ctr_h["hbm_GBps"], "counter_source": ctr_h["counter_source"],
                               "counter_commit": ctr_h["counter_commit"], "stale": ctr_h["stale"],
                               "launches_timed": n_h, "avg_launch_us": (ms_h * 1e3 / n_h) if n_h else None,
                               "avg_algorithmic_MB_per_launch": (bytes_h / n_h / 1e6) if n_h else None,
                               "share_of_kernel_time": kernel_time_share(KERNEL_SIG["f32mix_hbm"], name)}
        n_6, ms_6, fl_6, _ = timer.summary("mfma16")
        ctr_6 = committed_counters(KERNEL_SIG["f16"], name)
        ach6 = fl_6 / (ms_6 * 1e-3) / 1e12 if ms_6 > 0 else 0.0
        res["roofline"] = {"bound": "mfma", "kernel": KERNEL_NAME["f16"] + " (single-term residual branches)", "achieved": ach6, "peak": peak, "unit": "TFLOP/s",
                           "frac": ach6 / peak, "traffic": ctr_6["traffic"], "traffic_unit": "HBM bytes/launch", "traffic_source": ctr_6["traffic_source"],
                           "mfma_busy": ctr_6["mfma_busy"], "lds_wait": ctr_6["lds_wait"], "hbm_GBps": ctr_6["hbm_GBps"], "counter_source": ctr_6["counter_source"],
                           "counter_commit": ctr_6["counter_commit"], "stale": ctr_6["stale"], "launches_timed": n_6,
                           "avg_launch_us": (ms_6 * 1e3 / n_6) if n_6 else None, "avg_gflop_per_launch": (fl_6 / n_6 / 1e9) if n_6 else None, "mfma_terms": 1,
                           "share_of_kernel_time": kernel_time_share(KERNEL_SIG["f16"], name), "step_frac": roof["step_frac"]}
    if name == "f32x3":
        res["roofline"]["mfma_terms"] = 3       # MFMA FLOPs issued per algorithmic FLOP (hi*hi + lo*hi + hi*lo): frac 1/3 = the pipe saturated
        res["roofline"]["issued_frac"] = 3.0 * res["roofline"]["frac"] if res["roofline"].get("frac") is not None else None   # ... i.e. what the matrix pipe itself sees
    if name == "f32mix":
        res["roofline_mfma"]["issued_frac"] = 3.0 * res["roofline_mfma"]["frac"] if res["roofline_mfma"].get("frac") is not None else None
        res["roofline_mfma"]["mfma_terms"] = 3       # (the single-term branches of this configuration run on the fp16 patch kernel: float16 plans, not in this population)
    res["collective"] = coll
    if power is not None and name in ("bf16", "f16"):
        # the dense peak is quoted at 2.4 GHz; what the matrix pipe could deliver at the clock this board held under the step
        power["peak_at_sustained_clock_TFLOPs"] = peak * power["sclk_MHz_mean"] / power["sclk_MHz_peak_is_quoted_at"]
        # NOT a roofline fraction (the roofline's peak is the dense peak at 2.4 GHz, `roofline.frac`): the same rate against the peak
        # rescaled to the clock rocm-smi reported under the step - a power diagnostic; the in-kernel clock reads up to 10 % lower
        power["achieved_over_peak_at_smi_clock"] = {"value": achieved / power["peak_at_sustained_clock_TFLOPs"],
                                                    "note": "not a roofline fraction: dominant kernel's rate / (2.5 PFLOP/s x rocm-smi sclk / 2.4 GHz)"}
    res["power"] = power
    del pipe, net
    torch.cuda.empty_cache()
    return res, sample


def power_probe(step, seconds=3.0, smi="rocm-smi", sync=None):
    """Board power and shader clock while the step runs back to back, sampled with rocm-smi AFTER the timed region (never inside it).
    The dominant 16-bit kernel sits at the board's power limit: the clock it sustains, not the 2.4 GHz the 2.5 PFLOP/s
    peak is quoted at, sets its rate.  Returns None where rocm-smi is missing or prints something else."""
    import re
    import shutil
    import threading
    if shutil.which(smi) is None:
        return None
    if sync is None:
        import torch
        sync = torch.cuda.synchronize
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                txt = subprocess.run([smi, "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            except Exception:
                return
            w = re.search(r"Package Power \(W\):\s*([0-9.]+)", txt)
            c = re.search(r"sclk clock level:\s*\S+\s*\((\d+)Mhz\)", txt)
            if w and c:
                samples.append((float(w.group(1)), float(c.group(1))))
    th = threading.Thread(target=sampler, daemon=True)
    t0 = time.perf_counter()
    for _ in range(20):
        step()                                   # the clock settles before the first sample
    sync()
    th.start()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            step()
        sync()
        n += 10
    stop.set()
    th.join(timeout=15)
    if len(samples) > 1:
        samples = samples[:-1]                   # the last call may have straddled the end of the loop
    if not samples:
        return None
    cap = None
    try:
        m = re.search(r"Max Graphics Package Power \(W\):\s*([0-9.]+)", subprocess.run([smi, "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout)
        cap = float(m.group(1)) if m else None
    except Exception:
        pass
    watts, mhz = [a for a, _ in samples], [b for _, b in samples]
    return {"board_W_mean": sum(watts) / len(watts), "board_W_max": max(watts), "cap_W": cap, "sclk_MHz_mean": sum(mhz) / len(mhz),
            "sclk_MHz_min": min(mhz), "sclk_MHz_peak_is_quoted_at": 2400, "samples": len(samples), "steps": n, "source": "rocm-smi --showpower --showclocks"}


def operand_probe(dtype, dev, n=64, hw=64):
    """The dominant kernel alone (3x3 256 -> 256 at 64 x 64, the batch of the step) on normally distributed operands and on all-zero ones:
    the same instructions, the same memory traffic - what differs is the power the matrix pipe draws, and with it the clock the board
    sustains.  After the timed region; 60 warm-up launches each."""
    import numpy as np
    import torch
    from object_keypoints_amd import ops
    from object_keypoints_amd.perception.backbone import conv_taps
    rng = np.random.default_rng(0)
    wt = (rng.standard_normal((256, 256, 3, 3)) / 48.0).astype(np.float32)
    out = ops.Act.empty(n, hw, hw, 256, dtype, dev)
    res = {"kernel": "okp_igemm_patch_kernel, 3x3 256 -> 256 at %d x %d, %d frames" % (hw, hw, n), "gflop": 2.0 * n * hw * hw * 256 * 2304 / 1e9}
    for key, scale in (("random_operands", 1.0), ("zero_operands", 0.0)):
        plan = ops.ConvPlan(dtype, [256], [1], 256, conv_taps(wt * scale), np.zeros(256, np.float32), relu=True)
        x = ops.Act((torch.randn(n, hw, hw, 256, device=dev) * scale).to(dtype))
        for _ in range(60):
            plan([x], out, hw, hw, tile=13)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            plan([x], out, hw, hw, tile=13)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res[key] = {"us_per_launch": us, "TFLOPs": res["gflop"] / us * 1e3}
        del plan, x
    return res


def run_stream8(ctx, ticks=200, warmup=10):
    """BASELINE configs[4] as a measurement (never the headline): 8 camera streams = 4 stereo pairs, one TICK = the 8 frames of
    one instant (resident in HBM) -> fp16 network as one batch -> peak-NMS on all maps -> left/right association -> ONE DLT
    triangulation launch -> 3D points on the host (StereoStreamPipeline; reference harness: scripts/eval_model.py:274-293, one
    frame of one camera per call, capped at 30 Hz).  Reports the per-tick latency eager and with the network pass replayed from a
    captured hipGraph, and the sustained rate against the 8 x 30 Hz requirement.  As in the main workload the stages behind the
    heads run on injected maps (random weights give flat heat maps): Gaussian bumps rendered from known 3D points through the
    left and right camera of every pair, so the triangulated points can be checked against those points in the same run."""
    import numpy as np
    import torch
    from object_keypoints_amd.perception import pipeline as pp
    from object_keypoints_amd.perception.utils import camera_utils as cu
    dev = ctx["dev"]
    params = cu.load_calibration_params(os.path.join(REPO, "config", "calibration.yaml"))
    offset = np.array([(511 / 720 * 1280 - 511.0) / 2.0, 0.0])
    left = cu.FisheyeCamera(params["K"], params["D"], params["image_size"]).scale(511 / 720).cut(offset).scale(64 / 511)
    right = cu.FisheyeCamera(params["Kp"], params["Dp"], params["image_size"]).scale(511 / 720).cut(offset).scale(64 / 511)
    stereo = cu.StereoCamera(left, right, params["T_RL"])
    net = build_net(torch.float16).to(dev)
    cfg = {"keypoint_config": [1, 3]}
    pipe = pp.StereoStreamPipeline(net, stereo, cfg, capacity=16, max_distance=1.5)
    n_pairs, K, size = 4, 3, 64
    ys, xs = np.meshgrid(np.arange(size, dtype=np.float32), np.arange(size, dtype=np.float32), indexing="ij")
    # per pair: 1 centre, 1 point of type 1, 3 points of type 2, 0.45-0.7 m in front of the rig, well apart in the image
    base = {0: [[0.00, 0.00, 0.55]], 1: [[0.12, -0.10, 0.60]], 2: [[-0.16, 0.12, 0.50], [0.14, 0.04, 0.65], [-0.12, -0.16, 0.70]]}
    heat = np.zeros((2 * n_pairs, K, size, size), np.float32)
    truth = []
    for p in range(n_pairs):
        t = {}
        for k, pts in base.items():
            X = np.array(pts) * np.array([1.0 - 0.05 * p, 1.0 + 0.04 * p, 1.0]) + np.array([0.01 * p, -0.01 * p, 0.03 * p])
            t[k] = X
            for side, (cam, T) in enumerate(((left, np.eye(4)), (right, params["T_RL"]))):
                for px, py in cam.project(X, T):
                    heat[2 * p + side, k] += np.exp(-((xs - px) ** 2 + (ys - py) ** 2) / 4.0)
        truth.append(t)
    heat_dev = torch.from_numpy(np.clip(heat, 0.0, 1.0)).to(dev)
    gen = torch.Generator(device=dev); gen.manual_seed(99)
    frames = torch.randn((2 * n_pairs, 3, 511, 511), generator=gen, device=dev, dtype=torch.float32)
    pipe.capture(frames)

    def timed(use_graph):
        for _ in range(warmup):
            out = pipe.tick(frames, heat_override=heat_dev, use_graph=use_graph)
        torch.cuda.synchronize()
        lat = []
        for _ in range(ticks):
            t0 = time.perf_counter()
            out = pipe.tick(frames, heat_override=heat_dev, use_graph=use_graph)      # ends with the D2H copy of the 3D points
            lat.append((time.perf_counter() - t0) * 1e3)
        return np.array(lat), out

    lat_e, out = timed(False)
    lat_g, out_g = timed(True)
    worst, matched = 0.0, 0
    for p in range(n_pairs):
        for k in range(K):
            got, want = out[p][k], truth[p][k]
            assert got.shape == want.shape, (p, k, got.shape, want.shape)                # every point of every pair triangulated
            assert np.array_equal(got, out_g[p][k])                                        # graph replay = eager, bit for bit
            d = np.linalg.norm(got[:, None] - want[None], axis=2).min(axis=1)
            worst = max(worst, float(d.max())); matched += got.shape[0]
    # 64 x 64 maps and a 6.2 cm baseline: depth moves 13 cm per pixel of disparity at 0.7 m, and the reference's centroid (5 x 5 window
    # around the peak, pipeline.py:53-61) is biased by up to ~0.3 px between pixel centres - centimetres, by construction of the path
    assert worst < 0.06, worst
    med_g = float(np.median(lat_g))
    # ---- fed: "host-u8" - the same ticks with every tick's frames starting in HOST memory as raw camera frames (uint8 1280 x 720 RGB,
    # pinned): H2D on a copy stream (HostFrameFeed), resize + crop + normalise on the device (okp_preprocess_u8), then the tick as above.
    # Latency = one tick alone, upload included; sustained = ticks back to back, the upload of tick t + 1 under the kernels of tick t.
    g = torch.Generator().manual_seed(17)
    host_u8 = torch.randint(0, 256, (2 * n_pairs, 720, 1280, 3), generator=g, dtype=torch.uint8).pin_memory()
    pipe_h = pp.StereoStreamPipeline(net, stereo, cfg, capacity=16, max_distance=1.5)
    pipe_h.capture(host_u8.to(dev))
    feed = pp.HostFrameFeed(host_u8.shape, host_u8.dtype, device=dev)
    resident = pipe_h.tick(host_u8.to(dev), heat_override=heat_dev, use_graph=True)
    lat_h = []
    for i in range(warmup + ticks):
        t0 = time.perf_counter()
        slot = feed.submit(host_u8)
        out_h = pipe_h.tick(feed.acquire(slot), heat_override=heat_dev, use_graph=True)
        feed.release(slot)
        if i >= warmup:
            lat_h.append((time.perf_counter() - t0) * 1e3)
    lat_h = np.array(lat_h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_stream = 0
    for out_s in pipe_h.stream((host_u8 for _ in range(ticks)), heat_override=heat_dev, use_graph=True):
        n_stream += 1
    tick_s = (time.perf_counter() - t0) / max(n_stream, 1) * 1e3
    for p in range(n_pairs):
        for k in range(K):
            assert np.array_equal(out_h[p][k], resident[p][k]) and np.array_equal(out_s[p][k], resident[p][k])      # host-fed = resident, bit for bit
    fed = {"fed": "host-u8", "frame": "uint8 1280x720 RGB, pinned host memory -> copy stream -> okp_preprocess_u8 (resize, centre crop, normalise) -> fp16 network",
           "MB_per_tick": host_u8.numel() / 1e6, "tick_ms_graph": {"median": float(np.median(lat_h)), "p99": float(np.quantile(lat_h, 0.99))},
           "tick_ms_sustained": tick_s, "sustained_fps": 2 * n_pairs * 1e3 / tick_s, "headroom_x": (1e3 / 30.0) / float(np.median(lat_h)),
           "equals_resident_tick": True}
    return {"workload": "BASELINE configs[4]: 8 camera streams (4 stereo pairs) x 30 fps, fp16 convolutions + fp32/fp64 geometry: 8 frames per tick "
                        "(HBM-resident) -> hourglass + heads -> peak-NMS -> left/right association -> one DLT triangulation launch -> 3D points on the host",
            "ticks": ticks, "frames_per_tick": 2 * n_pairs,
            "tick_ms_eager": {"median": float(np.median(lat_e)), "p99": float(np.quantile(lat_e, 0.99))},
            "tick_ms_graph": {"median": med_g, "p99": float(np.quantile(lat_g, 0.99))},
            "sustained_fps": 2 * n_pairs * 1e3 / med_g, "required_fps": 240.0, "tick_budget_ms": 1e3 / 30.0,
            "headroom_x": (1e3 / 30.0) / med_g, "points_per_tick": matched, "triangulation_err_m_max": worst, "host_fed": fed}


def run_host_fed(ctx, name="bf16", steps=12, warmup=3):
    """The batch-64 step fed from HOST memory (never `value`, which starts with the frames resident in HBM): the reference's harness
    hands its model frames from a DataLoader on the host (scripts/eval_model.py:274-293).  Two host layouts, both pinned, both uploaded by
    HostFrameFeed on a copy stream under the previous step's kernels: (a) raw camera frames, uint8 64 x 720 x 1280 x 3 (177 MB per batch;
    resize, centre crop and normalisation on the device: okp_preprocess_u8, perception/datasets/video.py:83-100,215); (b) the reference's
    model input, fp32 NCHW 64 x 3 x 511 x 511 (201 MB).  Whole step: network + peaks + lifting + grouping on the injected scenes."""
    import torch
    from object_keypoints_amd.perception.pipeline import BatchedKeypointPipeline, HostFrameFeed
    dev, batch = ctx["dev"], ctx["batch"]
    wl = WORKLOADS["batch64"]
    net = build_net(getattr(torch, TORCH_DTYPES[name])).to(dev)
    pipe = BatchedKeypointPipeline(net, {"keypoint_config": list(wl["keypoint_config"])}, ctx["camera"], capacity=64)
    if "batch64" not in ctx["bumps"]:
        ctx["bumps"]["batch64"] = bump_maps(ctx["start"], batch, dev, wl)
    b_heat, b_depth, b_centers, _ = ctx["bumps"]["batch64"]
    g = torch.Generator().manual_seed(5)
    hosts = {"u8_1280x720": torch.randint(0, 256, (batch, 720, 1280, 3), generator=g, dtype=torch.uint8).pin_memory(),
             "f32_nchw_511": ctx["frames"].cpu().pin_memory()}
    res = {"note": "PCIe-inclusive: every step's frames start in pinned host memory; upload of step i + 1 on a copy stream under step i", "dtype": name, "steps": steps}
    with torch.no_grad():
        for label, host in hosts.items():
            feed = HostFrameFeed(host.shape, host.dtype, device=dev)

            def run(n):
                slot = feed.submit(host)
                for i in range(n):
                    frames = feed.acquire(slot)
                    nxt = feed.submit(host) if i + 1 < n else None
                    net.deployed(frames, check_range=False)
                    out = pipe.postprocess_device(b_heat, b_depth, b_centers)
                    feed.release(slot)
                    slot = nxt
                return out

            run(warmup)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = run(steps)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            assert not bool(out["overflow"])
            t0 = time.perf_counter()
            for _ in range(4):
                feed.buffers[0].copy_(host, non_blocking=True)
            torch.cuda.synchronize()
            t_copy = (time.perf_counter() - t0) / 4
            mb = host.numel() * host.element_size() / 1e6
            res[label] = {"frames_per_s": batch / dt, "ms_per_step": dt * 1e3, "MB_per_step": mb, "h2d_alone_ms": t_copy * 1e3, "h2d_GBps": mb / t_copy / 1e3}
            del feed
    del pipe, net, hosts
    torch.cuda.empty_cache()
    return res


FRAME_BATCHES = int(os.environ.get("OKP_BENCH_FRAME_BATCHES", "4"))      # different resident frame batches the timed steps walk through (1: A/B against one re-read batch)


SAMPLE_CAP = 1024      # peak slots per map of the error sample (random-weight networks give flat maps: ~100 peaks each)


def collective_probe(points, batch, world, dev, reps=8, own_group=False):
    """The one data-path collective (all_gather_keypoints), bracketed by HIP events on a few calls AFTER the timed region: which
    backend and how many ranks it saw, the payload per rank and its median duration.  A plain `python bench.py --gpus 1` has no
    process group - all_gather_keypoints passes the tensor through in the timed steps - so with own_group the probe creates a
    one-rank RCCL group for itself (tcp://127.0.0.1, no torchrun), measures the real collective on it and destroys it again: the
    N = 1 record then shows the communicator set-up and the all-gather working on this box."""
    import torch
    import torch.distributed as dist
    from object_keypoints_amd import distributed as dist_
    payload = int(points.numel() * points.element_size())
    made = False
    note = None
    if not dist.is_initialized():
        if not own_group:
            return {"backend": None, "world": 1, "payload_bytes_per_rank": payload, "allgather_us_median": None,
                    "note": "no process group (not started by torchrun): all_gather_keypoints passes the tensor through"}
        try:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            fd_guard = _StdoutToStderr()          # RCCL prints its version banner to C stdout: the line this process prints must stay alone there
            fd_guard.__enter__()
            dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1, device_id=dev)
            made = True
            note = "one-rank RCCL group created by the probe (the timed steps ran without a process group)"
        except Exception as e:       # no RCCL on this box: say so, the benchmark itself is unaffected
            fd_guard.__exit__(None, None, None)
            return {"backend": None, "world": 1, "payload_bytes_per_rank": payload, "allgather_us_median": None,
                    "note": f"no process group, and a one-rank RCCL group could not be created: {type(e).__name__}: {str(e)[:120]}"}
    try:
        us = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dist_.barrier()
            e0.record()
            g = dist_.all_gather_keypoints(points, total_frames=batch * world)
            e1.record()
            torch.cuda.synchronize()
            us.append(e0.elapsed_time(e1) * 1e3)
        assert g.shape[0] == batch * world and (made is False or torch.equal(torch.nan_to_num(g), torch.nan_to_num(points)))
        us.sort()
        res = {"backend": dist.get_backend(), "world": dist.get_world_size(), "payload_bytes_per_rank": payload,
               "gathered_bytes": int(g.numel() * g.element_size()), "allgather_us_median": us[len(us) // 2], "allgather_us_min": us[0], "calls": reps}
        if note:
            res["note"] = note
        return res
    finally:
        if made:
            dist.destroy_process_group()
            fd_guard.__exit__(None, None, None)


class _StdoutToStderr:
    """File descriptor 1 -> 2 for the duration (C-level prints of a library included: the C stdio buffers are flushed before fd 1 comes back)."""

    def __enter__(self):
        import ctypes
        sys.stdout.flush()
        self.libc = ctypes.CDLL(None)
        self.libc.fflush(None)
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self.libc.fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)


def parity_fields(sample, oracle, camera_file):
    """Every north_star tolerance of one precision on the error sample, against the oracle (= the reference, tests/golden):
    heat maps (bar 1e-3), depth maps, peak-index sets (bar: identical, i.e. Jaccard 1.0) and the 3D points of the peaks both
    sides found (bar 1e-4 m).  The oracle's peaks / points come from its own heat / depth maps through oracle.pipeline."""
    import numpy as np
    from oracle import pipeline as op
    cam = op.eval_camera(camera_file)
    d2p = op.DetectionToPoint(); d2p.reset(cam)
    heat, depth = sample["heat"], sample["depth"]
    inter = union = 0
    dists = []
    for n in range(heat.shape[0]):
        for k in range(heat.shape[1]):
            idx = op.peak_indices(oracle["heat"][n, k])
            c = int(sample["count"][n, k])
            assert c <= SAMPLE_CAP, "error sample: more peaks than SAMPLE_CAP"
            mine = {tuple(v): j for j, v in enumerate(sample["yx"][n, k, :c].tolist())}
            theirs = [tuple(v) for v in idx.tolist()]
            inter += len(set(theirs) & set(mine)); union += len(set(theirs) | set(mine))
            pts, _ = op.refine_peaks(oracle["heat"][n, k], idx)
            if not pts:
                continue
            want = d2p(np.stack(pts), oracle["depth"][n, k])
            for j, key in enumerate(theirs):
                if key in mine:
                    dists.append(float(np.linalg.norm(sample["points"][n, k, mine[key], :3] - want[j])))
    n_pts = len(dists)                        # (no matched peak: nothing was compared - null statistics, and the 3D bar is NOT met on an empty sample)
    dists = np.array(dists if dists else [np.nan])
    worst = float(dists.max()) if n_pts else None
    e_d = np.abs(depth.astype(np.float64) - oracle["depth"].astype(np.float64))
    return {"heat_err_vs_oracle": heat_error(heat, oracle["heat"]),
            "depth_err_vs_oracle": {"max": float(e_d.max()), "mean": float(e_d.mean())},
            "peak_jaccard": inter / union if union else 1.0, "peaks_compared": union,
            # (a random-weight network's depth maps are noise-like: ONE flipped depth pixel under a peak moves that point by the
            #  depth difference - the max alone cannot tell such an outlier from a bias, the quantiles can)
            "p_C_err_m": worst, "p_C_points": n_pts,
            "p_C_err_m_stats": {"median": float(np.median(dists)) if n_pts else None, "p99": float(np.quantile(dists, 0.99)) if n_pts else None, "max": worst,
                                "n_over_1e-4": int((dists > 1e-4).sum()) if n_pts else 0, "n": n_pts},
            "meets": {"heat_1e-3": bool(np.abs(heat - oracle["heat"]).max() <= 1e-3), "peaks_identical": inter == union,
                      "p_C_1e-4_m": bool(n_pts) and worst <= 1e-4}}


def heat_error(got, want):
    import numpy as np
    e = np.abs(got.astype(np.float64) - want.astype(np.float64)).ravel()
    return {"max": float(e.max()), "mean": float(e.mean()), "p99": float(np.quantile(e, 0.99)), "frames": int(got.shape[0])}


def workload_string(batch, name, world, workload="batch64"):
    wl = WORKLOADS[workload]
    return (f"BASELINE {wl['config']}: batch={batch}/GPU synthetic 511x511 frames, {name} activations+weights, fp32 accumulate, "
            f"CornerNet-Squeeze {wl['what']} random-init procedural weights; fp32 NCHW frames -> stem -> hourglass+heads; "
            f"peak-NMS -> depth lifting -> object grouping on injected bump scenes (SURVEY 8(d), {wl['scenes']})"
            + (" -> all-gather of 3D keypoints" if world > 1 else ""))


def rank_main(args):
    import numpy as np
    import torch
    from object_keypoints_amd import distributed as dist_, synth
    from object_keypoints_amd.perception.utils import camera_utils as cu
    rank, local_rank, world = dist_.init()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    params = cu.load_calibration_params(os.path.join(REPO, "config", "calibration.yaml"))
    camera = cu.FisheyeCamera(params["K"], params["D"], params["image_size"]).scale(511 / 720)
    camera = camera.cut(np.array([(511 / 720 * 1280 - 511.0) / 2.0, 0.0])).scale(64 / 511)

    # synthetic frames generated ON the device (seeded per global frame block): N x 3 x 511 x 511 fp32 ~N(0,1)
    start, _ = dist_.shard(args.batch * world, rank, world)
    gen = torch.Generator(device=dev); gen.manual_seed(1234 + start)
    frames = torch.randn((args.batch, 3, 511, 511), generator=gen, device=dev, dtype=torch.float32)
    # the timed steps do not re-run ONE resident batch: a 64-frame batch is 201 MB, less than the 256 MB of Infinity Cache, so a batch that is
    # read again every 5 ms could be served from there.  FRAME_BATCHES different batches (seeded per batch) are walked round robin.
    pool = [frames] + [torch.randn((args.batch, 3, 511, 511), generator=gen, device=dev, dtype=torch.float32) for _ in range(FRAME_BATCHES - 1)]
    with_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    ctx = {"dev": dev, "world": world, "batch": args.batch, "camera": camera, "frames": frames, "frames_pool": pool, "probe_collective": True, "probe_power": True,
           "no_probes": args.no_probes, "start": start, "bumps": {},
           "err_frames": torch.from_numpy(synth.frames(ERR_FRAMES, seed=1)).to(dev) if with_cpu else None}

    if args.workload == "stream8":
        print(json.dumps({"metric": "stream8 tick latency (BASELINE configs[4])", "stream8": run_stream8(ctx)}), flush=True)
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return
    workload = args.workload                      # batch64 (configs[2], the headline) or cups512 (configs[3])
    head, head_heat = run_precision(args.dtype, ctx, args.steps, args.warmup, workload)
    result = {
        "metric": "frames/sec keypoint inference (511x511 -> heatmaps+3D)",
        "value": head["value"], "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": workload_string(args.batch, args.dtype, world, workload),
                   "frames_per_gpu": args.batch, "global_batch": args.batch * world, "parallelism": f"frame-dp{world}",
                   "resident_frame_batches": FRAME_BATCHES},
        "conv_stack_tflops_per_gpu": head["conv_stack_tflops_per_gpu"],
        "roofline": head["roofline"],
        "collective": head["collective"],
        "power": head["power"],
    }
    extra_heat = {}
    cups_heat = None
    if world == 1 and workload == "batch64":
        for name in [d for d in args.extra_dtypes.split(",") if d and d != args.dtype]:
            # fp32 runs ~12x longer per step (157 TFLOP/s MFMA peak): fewer steps keep the default run within minutes
            # (with the driver's --steps 20 every precision still times at least 10 steps)
            steps = max(10, args.steps // 10) if name == "f32" else max(10, args.steps // (4 if name in ("f32x3", "f32mix") else 2))
            res, extra_heat[name] = run_precision(name, ctx, steps, min(args.warmup, 2))
            res["workload"] = workload_string(args.batch, name, world)
            res.pop("power", None)               # (probed once, on the headline precision)
            result[OBJECT_KEY[name]] = res
        if not args.no_cups:
            # BASELINE configs[3], one rank's share (64 frames of the 512): the workload a `--workload cups512 --gpus 8` run weak-scales
            ctx["probe_collective"] = True       # its payload ([64, 4, cap, 4]) through a one-rank RCCL group of its own
            res, cups_heat = run_precision(args.dtype, ctx, max(10, args.steps // 2), min(args.warmup, 2), "cups512")
            res["workload"] = workload_string(args.batch, args.dtype, world, "cups512")
            res["dtype"] = args.dtype
            res.pop("power", None)
            result["cups64"] = res
    if world == 1 and not args.no_stream8 and workload == "batch64":
        result["stream8"] = run_stream8(ctx)
    if world == 1 and not args.no_host_fed and workload == "batch64":
        result["host_fed"] = run_host_fed(ctx, args.dtype if args.dtype in ("bf16", "f16") else "bf16")
    if with_cpu:
        cal = os.path.join(REPO, "config", "calibration.yaml")
        if workload == "batch64":
            result["cpu_baseline"], oracle = cpu_baseline(args.cpu_seconds)
            result.update(parity_fields(head_heat, oracle, cal))
            for name, smp in extra_heat.items():
                result[OBJECT_KEY[name]].update(parity_fields(smp, oracle, cal))
            if cups_heat is not None:
                result["cups64"].update(parity_fields(cups_heat, oracle_maps(WORKLOADS["cups512"]["heatmaps_out"]), cal))
        else:
            result.update(parity_fields(head_heat, oracle_maps(WORKLOADS[workload]["heatmaps_out"]), cal))
    if rank == 0:
        print(json.dumps(result), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if needs_launcher(args, os.environ):
        return self_launch(args, argv, gpu_count_without_hip())     # no torch import, no HIP call in the launcher parent
    rank_main(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())

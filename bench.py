#!/usr/bin/env python3
"""Benchmark of the keypoint-inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the hot path over one batch of synthetic 511x511 frames already resident in
HBM: frame packing -> CornerNet-Squeeze hourglass + heads (HIP implicit-GEMM convs) -> per-map
peak-NMS -> per-peak depth lifting to 3D -> (N>1) ONE all-gather of the fixed-capacity 3D-keypoint
tensor.  Workload = BASELINE.json configs[2]: batch 64 per GPU, bf16 activations/weights with fp32
accumulation, K=3 maps (config/valve.json).  Frames shard across ranks (weak scaling, weights
replicated); value = frames of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = the bf16 256x256-tile implicit-GEMM
instantiation, timed live with HIP events on the launch stream during the timed steps) and
`cpu_baseline` (the oracle: torch-CPU fp32 restatement + NumPy post-processing on the host cores,
N=1 only, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

GFLOP_PER_FRAME = 74.565218304          # 37 282 609 152 MACs x 2 (deployed path, K=3)
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


class KernelTimer:
    """ops.LAUNCH_HOOK: brackets launches of ONE implicit-GEMM instantiation with HIP events recorded on
    the launch stream (torch's current stream is the stream the C ABI launches on)."""

    def __init__(self, dtype, tile, n_src):
        self.key = (dtype, tile, n_src)
        self.records = []
        self.enabled = False

    def before(self, plan, tile, macs):
        if not self.enabled or (plan.dtype, tile) != self.key[:2] or (self.key[2] is not None and plan.n_src != self.key[2]):
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return (e0, e1, macs)

    def after(self, token):
        if token is not None:
            token[1].record()
            self.records.append(token)

    def summary(self):
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in self.records)
        flops = 2.0 * sum(m for _, _, m in self.records)
        return len(self.records), ms, flops


def pmc_traffic(kernel_sig):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate runs of this same command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950).  bench.py cannot run the profiler on itself, so the number comes from profiles/ and is null if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_hbm_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        data = json.load(f)
    for name, row in data.items():
        if kernel_sig in name:
            return (row["fetch_MB_per_launch_corrected"] + row["write_MB_per_launch"]) * 1e6, os.path.basename(files[-1])
    return None, None


def build_net(dtype):
    from object_keypoints_amd import synth
    from object_keypoints_amd.perception.models import KeypointNet
    net = KeypointNet(features=128, heatmaps_out=3, compute_dtype=dtype)
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
    """The oracle (a port: torch-CPU fp32 restatement + NumPy post-processing) on the host cores."""
    from oracle import net as onet
    from oracle import pipeline as op
    from object_keypoints_amd import synth
    cores = usable_cores()
    torch.set_num_threads(cores)
    net = onet.load_synthetic(onet.KeypointNet(features=128, heatmaps_out=3), seed=0)
    cam = op.eval_camera(os.path.join(REPO, "config", "calibration.yaml"))
    d2p = op.DetectionToPoint(); d2p.reset(cam)
    x = torch.from_numpy(synth.frames(4, seed=1))

    def step():
        heat, depth, _ = onet.deployed_forward(net, x)
        heat, depth = heat.numpy(), depth.numpy()
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
        frames += 4
        el = time.perf_counter() - t0
        if el >= seconds and frames >= 8:
            break
    return {"value": frames / el, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{frames} frames (batches of 4, 3x511x511 fp32): oracle torch-CPU forward + NumPy peak-NMS/centroid/depth lifting, {el:.1f} s"}


def main():
    args = parse()
    from object_keypoints_amd import distributed as dist_, ops
    from object_keypoints_amd.perception.pipeline import BatchedKeypointPipeline
    from object_keypoints_amd.perception.utils import camera_utils as cu
    rank, local_rank, world = dist_.init()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    net = build_net(dtype).to(dev)
    params = cu.load_calibration_params(os.path.join(REPO, "config", "calibration.yaml"))
    camera = cu.FisheyeCamera(params["K"], params["D"], params["image_size"]).scale(511 / 720)
    camera = camera.cut(np.array([(511 / 720 * 1280 - 511.0) / 2.0, 0.0])).scale(64 / 511)
    pipe = BatchedKeypointPipeline(net, {"keypoint_config": [1, 3]}, camera, capacity=64)

    # synthetic frames generated ON the device (seeded per global frame block): N x 3 x 511 x 511 fp32 ~N(0,1)
    start, _ = dist_.shard(args.batch * world, rank, world)
    gen = torch.Generator(device=dev); gen.manual_seed(1234 + start)
    frames = torch.randn((args.batch, 3, 511, 511), generator=gen, device=dev, dtype=torch.float32)

    # dominant kernel: bf16 = the patch-resident 3x3 kernel (tile 13, one symbol for one and two sources); fp32 = 256x256 gather tile
    timer = KernelTimer(dtype, 13 if dtype == torch.bfloat16 else 3, None if dtype == torch.bfloat16 else 1)
    ops.LAUNCH_HOOK = timer

    def step():
        out = pipe.forward_device(frames)
        return dist_.all_gather_keypoints(out["points"])

    with torch.no_grad():
        for _ in range(args.warmup):
            gathered = step()
        dist_.barrier(); torch.cuda.synchronize()
        timer.enabled = True
        t0 = time.perf_counter()
        for _ in range(args.steps):
            gathered = step()
        dist_.barrier(); torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        timer.enabled = False
    elapsed = dist_.max_over_ranks(elapsed, dev)
    assert gathered.shape[0] == args.batch * world

    total_frames = args.batch * world * args.steps
    value = total_frames / elapsed
    n_launch, k_ms, k_flops = timer.summary()
    achieved = k_flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    peak = PEAK_TFLOPS[args.dtype]
    traffic, traffic_src = pmc_traffic("okp_igemm_patch_kernel" if args.dtype == "bf16"
                                       else "okp_igemm_kernelIfLi256ELi256ELi4ELi2ELi2ELi128ELi32ELi1E")
    result = {
        "metric": "frames/sec keypoint inference (511x511 -> heatmaps+3D)",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"BASELINE configs[2]: batch={args.batch}/GPU synthetic 511x511 frames, {args.dtype} activations+weights, fp32 accumulate, "
                               "CornerNet-Squeeze K=3 (valve) random-init procedural weights; fp32 NCHW frames -> stem -> hourglass+heads -> peak-NMS -> depth lifting -> object grouping"
                               + (" -> all-gather of 3D keypoints" if world > 1 else ""),
                   "frames_per_gpu": args.batch, "global_batch": args.batch * world, "parallelism": f"frame-dp{world}"},
        "conv_stack_tflops_per_gpu": GFLOP_PER_FRAME * value / world / 1e3,
        "roofline": {"bound": "mfma", "kernel": "okp_igemm_patch_kernel<bf16,256co x 16x16px>" if args.dtype == "bf16" else "okp_igemm_kernel<f32,256x256,src1>", "achieved": achieved, "peak": peak,
                     "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic, "traffic_unit": "HBM bytes/launch", "traffic_source": traffic_src,
                     "launches_timed": n_launch, "avg_launch_us": (k_ms * 1e3 / n_launch) if n_launch else None,
                     "avg_gflop_per_launch": (k_flops / n_launch / 1e9) if n_launch else None},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

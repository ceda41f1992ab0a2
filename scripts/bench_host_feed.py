"""PCIe-inclusive rate of the hot path (never bench.py's `value`: that one starts with the frames resident in HBM).
The boundary handed HOST buffers: (a) the reference's own layout, fp32 NCHW 64 x 3 x 511 x 511 (201 MB per batch);
(b) raw camera frames, uint8 64 x 720 x 1280 x 3 (177 MB), resized / cropped / normalised on the device (okp_preprocess_u8).
Each measured serially (copy, then compute) and double-buffered (the copy of batch i + 1 on a second stream under the compute
of batch i).  usage: bench_host_feed.py [bf16|f16|f32mix] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops, synth
from object_keypoints_amd.perception.models import KeypointNet
name = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32mix": ops.F32MIX}[name]
net = KeypointNet(features=128, heatmaps_out=3, compute_dtype=dtype)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.fill_state_dict(shapes, seed=0).items()})
net.eval().cuda()
N = 64
host = {"fp32 NCHW 511x511": torch.randn(N, 3, 511, 511).pin_memory(),
        "uint8 HWC 720x1280": torch.randint(0, 256, (N, 720, 1280, 3), dtype=torch.uint8).pin_memory()}
copy_stream = torch.cuda.Stream()
with torch.no_grad():
    for label, h in host.items():
        mb = h.numel() * h.element_size() / 1e6
        dev = [torch.empty_like(h, device="cuda") for _ in range(2)]
        for _ in range(3):
            dev[0].copy_(h, non_blocking=True); net.deployed(dev[0])
        torch.cuda.synchronize()
        # copy alone
        t0 = time.perf_counter()
        for _ in range(steps):
            dev[0].copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        t_copy = (time.perf_counter() - t0) / steps
        # resident (compute alone)
        t0 = time.perf_counter()
        for _ in range(steps):
            net.deployed(dev[0])
        torch.cuda.synchronize()
        t_comp = (time.perf_counter() - t0) / steps
        # serial: copy then compute on one stream
        t0 = time.perf_counter()
        for _ in range(steps):
            dev[0].copy_(h, non_blocking=True)
            net.deployed(dev[0])
        torch.cuda.synchronize()
        t_serial = (time.perf_counter() - t0) / steps
        # double-buffered: the copy of the next batch on its own stream under this batch's compute
        main = torch.cuda.current_stream()
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        done = [torch.cuda.Event(), torch.cuda.Event()]
        with torch.cuda.stream(copy_stream):
            dev[0].copy_(h, non_blocking=True); ready[0].record(copy_stream)
        done[1].record(main)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            b, nb = i & 1, (i + 1) & 1
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(done[nb])                 # the buffer's previous batch has been consumed
                dev[nb].copy_(h, non_blocking=True); ready[nb].record(copy_stream)
            main.wait_event(ready[b])
            net.deployed(dev[b])
            done[b].record(main)
        torch.cuda.synchronize()
        t_db = (time.perf_counter() - t0) / steps
        print(f"{name:6s} {label:20s} {mb:6.0f} MB/batch: copy {t_copy * 1e3:6.2f} ms ({mb / t_copy / 1e3:5.1f} GB/s) | resident {N / t_comp:7.0f} frames/s "
              f"({t_comp * 1e3:5.2f} ms) | serial {N / t_serial:7.0f} frames/s ({t_serial * 1e3:5.2f} ms) | double-buffered {N / t_db:7.0f} frames/s ({t_db * 1e3:5.2f} ms)", flush=True)

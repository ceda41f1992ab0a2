"""What the vendor GEMM (hipBLASLt / rocBLAS behind torch.matmul) reaches on this box on the plain GEMMs that the 3x3 convolutions of the
step are equivalent to (M = frames x output pixels, N = 256 output channels, K = 9 x cin; operands dense and contiguous in HBM - the
convolution kernels also have to gather their rows).  A calibration of the achievable MFMA rate for these shapes, not a code path."""
import sys, time, torch
dev = torch.device("cuda", 0)
def run(M, N, K, dtype, reps=20):
    a = torch.randn((M, K), device=dev, dtype=dtype)
    b = torch.randn((K, N), device=dev, dtype=dtype)
    for _ in range(3): torch.matmul(a, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): torch.matmul(a, b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dt * 1e6, 2.0 * M * N * K / dt / 1e12
for dtype in (torch.bfloat16, torch.float16):
    for name, M, K in [("3x3 256->256 @64x64, N=64", 64 * 4096, 2304), ("3x3 256->256 @128x128 (pre[1].conv2)", 64 * 16384, 2304),
                       ("3x3 128->256 s2 -> 128x128 (pre[1].conv1)", 64 * 16384, 1152), ("convT class: 2x2 taps 256->256 @32x32 x 4 classes", 4 * 64 * 1024, 1024),
                       ("square 8192^3 (for scale)", 8192, 8192)]:
        N = 8192 if M == 8192 else 256
        us, tf = run(M, N, K, dtype)
        print(f"{str(dtype):15s} {name:52s} M={M:8d} N={N:5d} K={K:5d}: {us:8.1f} us  {tf:7.1f} TFLOP/s")

# The same layers through torch.nn.functional.conv2d (MIOpen: what stock PyTorch-ROCm - the reference's own stack - would launch on this GPU),
# channels-last and NCHW, after MIOpen's find step (benchmark mode), convolution only (no BatchNorm / ReLU / residual launches).
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
def conv_case(name, n, cin, cout, h, k, stride, dtype, transposed=False, channels_last=True):
    x = torch.randn((n, cin, h, h), device=dev, dtype=dtype)
    w = torch.randn((cin, cout, k, k) if transposed else (cout, cin, k, k), device=dev, dtype=dtype)
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last); w = w.contiguous(memory_format=torch.channels_last)
    fn = (lambda: F.conv_transpose2d(x, w, stride=stride, padding=1)) if transposed else (lambda: F.conv2d(x, w, stride=stride, padding=(k - 1) // 2))
    try:
        for _ in range(3): y = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): y = fn()
        torch.cuda.synchronize()
    except Exception as e:                                   # a shape MIOpen has no solver for in this dtype / layout
        print(f"{str(dtype):15s} {name:52s} {'NHWC' if channels_last else 'NCHW'}: {type(e).__name__}")
        return
    dt = (time.perf_counter() - t0) / 10
    ho = y.shape[2]
    flop = 2.0 * n * ho * ho * cout * cin * k * k / (stride * stride if transposed else 1)
    print(f"{str(dtype):15s} {name:52s} {'NHWC' if channels_last else 'NCHW'}: {dt * 1e6:8.1f} us  {flop / dt / 1e12:7.1f} TFLOP/s")
with torch.no_grad():
    for dtype in (torch.bfloat16, torch.float32):
        for cl in (True, False):
            conv_case("conv2d 3x3 256->256 @64x64, N=64", 64, 256, 256, 64, 3, 1, dtype, channels_last=cl)
            conv_case("conv2d 3x3 256->256 @128x128 (pre[1].conv2)", 64, 256, 256, 128, 3, 1, dtype, channels_last=cl)
            conv_case("conv2d 3x3 128->256 s2 256x256->128x128 (pre[1].conv1)", 64, 128, 256, 256, 3, 2, dtype, channels_last=cl)
            conv_case("conv_transpose2d 4x4 s2 256->256 32x32->64x64", 64, 256, 256, 32, 4, 2, dtype, transposed=True, channels_last=cl)

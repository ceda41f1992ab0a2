import torch, time
x = torch.empty(1 << 29, dtype=torch.bfloat16, device="cuda")   # 1 GiB
y = torch.empty_like(x)
def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
gb = x.numel() * 2 / 1e9
print("fill  %.2f TB/s (write only)" % (gb / t(lambda: x.fill_(1.0)) / 1e3))
print("copy  %.2f TB/s (read+write bytes)" % (2 * gb / t(lambda: y.copy_(x)) / 1e3))
print("sum   %.2f TB/s (read only)" % (gb / t(lambda: x.sum()) / 1e3))
z = torch.empty(1 << 26, dtype=torch.bfloat16, device="cuda")   # 128 MiB: fits the Infinity Cache
gbz = z.numel() * 2 / 1e9
print("fill 128MiB %.2f TB/s" % (gbz / t(lambda: z.fill_(1.0), 50) / 1e3))

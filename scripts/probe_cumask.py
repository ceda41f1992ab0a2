"""Can the chip be partitioned?  Two streams with complementary CU masks (hipExtStreamCreateWithCUMask): the patch kernel on `big`,
a chain of small fire modules on `small`; each alone and both together.  usage: probe_cumask.py [n_small_cus=32] [pattern=mod8|low]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_keypoints_amd import ops
from object_keypoints_amd.perception import backbone as bb
from object_keypoints_amd.perception.backbone import conv_taps
n_small = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pattern = sys.argv[2] if len(sys.argv) > 2 else "mod8"
hip = ctypes.CDLL("libamdhip64.so")
torch.cuda.init(); torch.zeros(1, device="cuda")
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (32 * w + b) in bits) for w in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)
if pattern == "mod8":       # every 8th CU index
    small_bits = {i for i in range(256) if i % 8 == 7 and len([j for j in range(i) if j % 8 == 7]) < n_small}
else:
    small_bits = set(range(n_small))
big_bits = set(range(256)) - small_bits
S_big, S_small = masked_stream(big_bits), masked_stream(small_bits)
S_all = torch.cuda.Stream()
rng = np.random.default_rng(0)
wt = (rng.standard_normal((256, 256, 3, 3)) / 48).astype(np.float32)
plan = ops.ConvPlan(torch.bfloat16, [256], [1], 256, conv_taps(wt), np.zeros(256, np.float32), relu=True)
x = ops.Act(torch.randn(64, 64, 64, 256, device="cuda").bfloat16())
o = ops.Act.empty(64, 64, 64, 256, torch.bfloat16, x.t.device)
fires = [bb.fire_module(384, 384).eval() for _ in range(2)]
x16 = ops.Act(torch.randn(64, 16, 16, 384, device="cuda").bfloat16())
def big(k=8):
    for _ in range(k): plan([x], o, 64, 64, tile=13)
def small(k=40):
    y = x16
    for i in range(k): y = fires[i % 2](y)
def timed(fn_streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for fn, st in fn_streams:
        with torch.cuda.stream(st): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
for fn, st in ((big, S_all), (small, S_all)): timed([(fn, st)])
print(f"small CUs {len(small_bits)} ({pattern})")
for rep in range(2):
    print(f"  big alone  (unmasked) {timed([(big, S_all)]):7.3f} ms | (masked) {timed([(big, S_big)]):7.3f} ms")
    print(f"  small alone (unmasked) {timed([(small, S_all)]):7.3f} ms | (masked) {timed([(small, S_small)]):7.3f} ms")
    print(f"  together: unmasked streams {timed([(big, S_all), (small, torch.cuda.Stream())]):7.3f} ms | masked {timed([(small, S_small), (big, S_big)]):7.3f} ms")

import os, time, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)' ; ldd object_keypoints_amd/lib/libokp_hip.so | grep -i hip; ls /usr/local/lib/python3.10/dist-packages/torch/lib | grep -i hip")
import torch
from oracle import net as onet
from object_keypoints_amd import synth
net = onet.load_synthetic(onet.KeypointNet(features=128, heatmaps_out=3), seed=0)
x = torch.from_numpy(synth.frames(4, seed=1))
for th in (8, 16, 32, 64):
    torch.set_num_threads(th)
    onet.deployed_forward(net, x[:1])
    t = time.time(); onet.deployed_forward(net, x); dt = time.time() - t
    print("threads", th, "batch4", round(dt, 3), "s ->", round(4 / dt, 2), "fps", flush=True)
    if dt > 20: break

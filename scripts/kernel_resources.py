"""Compile csrc/okp_igemm.hip (or the file given) with -Rpass-analysis=kernel-resource-usage and print a table."""
import re, subprocess, sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "object_keypoints_amd/csrc/okp_igemm.hip")
form = [] if src.endswith("_w4.hip") else ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950"] + form + [
       "-I" + os.path.join(REPO, "include"), "-I" + os.path.join(REPO, "object_keypoints_amd/csrc"), "-c", src, "-o", "/tmp/_kres.o",
       "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("OKP_EXTRA_CFLAGS", "").split()      # (the build's experiment switches)
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+(\w[\w \[\]/]*?):\s+(\d+)", line)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if "DF16b" in k or "okp_fire" in k or len(sys.argv) > 2:
        name = re.sub(r"_ZN12_GLOBAL__N_116okp_igemm_kernelI|EEv14OkpIgemmParams", "", k)
        print(f"{name:48s} VGPR {v.get('VGPRs', -1):4d} AGPR {v.get('AGPRs', -1):4d} spill {v.get('VGPRs Spill', -1):3d} scratch {v.get('ScratchSize [bytes/lane]', -1):4d} LDS {v.get('LDS Size [bytes/block]', -1):7d} occ {v.get('Occupancy [waves/SIMD]', -1)}")

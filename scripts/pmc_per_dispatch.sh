#!/bin/bash
# per-dispatch FETCH_SIZE / WRITE_SIZE of the 256x256 implicit-GEMM launches of one bench step (which layer re-fetches?)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pd_f /tmp/pd_w
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pd_f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pd_w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os
FILTER = os.environ.get("OKP_PMC_FILTER", "okp_igemm_kernelIDF16bLi256ELi256E")
def rows(d, c):
    out = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and (FILTER in r["Kernel_Name"]):
                out.append((int(r["Dispatch_Id"]), r["Kernel_Name"][-45:-20] if "fire" in r["Kernel_Name"] else ("src2" if "ELi2EEEv" in r["Kernel_Name"] else "src1"), float(r["Counter_Value"])))
    return sorted(out)
f, w = rows("/tmp/pd_f", "FETCH_SIZE"), rows("/tmp/pd_w", "WRITE_SIZE")
n = len(f) // 2
for (i, k, fv), (_, _, wv) in zip(f[-n:], w[-n:]):
    print(f"dispatch {i:5d} {k}: fetch x2 {2 * fv * 1024 / 1e6:8.1f} MB   write {wv * 1024 / 1e6:8.1f} MB")
PY

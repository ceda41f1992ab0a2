#!/bin/bash
# kernel durations (rocprofv3 timestamps) of a tiny convolution: separates device time from host launch rate
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/small -- python3 $GRAFT_REPO_ROOT/scripts/bench_conv.py k=1 cin=128 cout=64 hw=2 tile=8 iters=200 2>&1 | tail -1
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/small/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "okp_igemm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows][-100:]
g = [int(rows[i+1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) for i in range(len(rows)-101, len(rows)-1)]
print("kernel ns: min %d med %d max %d | gap ns: min %d med %d" % (min(d), sorted(d)[50], max(d), min(g), sorted(g)[50]))
print({k: rows[-1][k] for k in rows[-1] if "Size" in k or "Scratch" in k or "LDS" in k or "GPR" in k})
PY

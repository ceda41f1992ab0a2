#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/gaps; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/raw -- python3 $GRAFT_REPO_ROOT/scripts/probe_gaps.py > $out/log.txt 2>&1
trace=$(ls $out/raw/*/*kernel_trace.csv | head -1)
python3 - $trace <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows:
    n = r["Kernel_Name"]
    if "okp_" not in n: continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = "patch" if "patch" in n else "fire2" if "fire2" in n else "igemm" if "igemm" in n else n[:20]
    gap = (s - prev) / 1e3 if prev else 0
    print(f"{'---' if gap > 500 else ''}{k:6s} dur {(e-s)/1e3:7.1f}  gap {gap:8.1f}")
    prev = e
PY
rm -rf $out/raw

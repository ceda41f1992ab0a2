#!/bin/bash
# usage: trace_b1.sh <tag> [batch]: every kernel of the last graph replay of a batch-b step -> gpurun_out/<tag>/b1_forward.txt
tag=$1; b=${2:-1}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/raw -- python3 $GRAFT_REPO_ROOT/scripts/probe_b1.py $b > $out/b1.log 2>&1
trace=$(ls $out/raw/*/*kernel_trace.csv | head -1)
python3 - $trace > $out/b1_forward.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "okp_stem_kernel" in r["Kernel_Name"]]
sel = rows[st[-1]:]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    n = r["Kernel_Name"]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-t0)/1e3:9.1f} {d/1e3:7.1f}us grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):5d}x{r['Workgroup_Size_X']:>4s} lds {r.get('LDS_Block_Size','?'):>6s} {n[:100]}")
PY
rm -rf $out/raw
cat $out/b1.log | tail -3; cat $out/b1_forward.txt

#!/bin/bash
# usage: trace_timeline.sh <tag> [dtype]  (on the GPU box) -> gpurun_out/<tag>_timeline.txt : every dispatch of the last forward with its queue
tag=$1; dt=${2:-bf16}
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/raw_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --dtype $dt --no-cpu-baseline --no-stream8 --extra-dtypes= > /tmp/tl.json 2> /tmp/tl.err
trace=$(ls /tmp/raw_$tag/*/*kernel_trace.csv | head -1)
python3 - $trace > $out/${tag}_timeline.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "okp_stem_kernel" in r["Kernel_Name"]]
sel = rows[st[-1]:]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    n = r["Kernel_Name"]
    m = re.search(r"okp_\w+?kernel", n)
    name = n[:90]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}us q{r.get('Queue_Id','?'):>3} wg{int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):5d}x{r['Workgroup_Size_X']:>4} lds{r.get('LDS_Block_Size','?'):>6} {name}")
PY
tail -3 /tmp/tl.json

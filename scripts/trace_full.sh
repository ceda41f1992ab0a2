#!/bin/bash
# usage: trace_full.sh <tag> [bench args]: every dispatch of the last forward with its queue -> gpurun_out/<tag>/full_forward.txt
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/raw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --extra-dtypes "" "$@" > $out/bench_profiled.json 2> $out/bench_profiled.err
trace=$(ls $out/raw/*/*kernel_trace.csv | head -1)
python3 - $trace > $out/full_forward.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "okp_stem_kernel" in r["Kernel_Name"]]
sel = rows[st[-1]:]
t0 = int(sel[0]["Start_Timestamp"])
qs = {}
for r in sel:
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    n = r["Kernel_Name"]
    m = re.search(r"okp_\w+?kernel", n)
    k = n[:90]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-t0)/1e3:9.1f} {d/1e3:7.1f}us q{q} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):5d}x{r['Workgroup_Size_X']:>4s} lds {r.get('LDS_Block_Size','?'):>6s} {k}")
PY
rm -rf $out/raw
cat $out/full_forward.txt | head -100

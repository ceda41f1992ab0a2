#!/bin/bash
# Per-kernel in-situ times of the bf16 bench step for several builds (lib/libokp_hip_<tag>.so) in ONE gpurun call: rocprofv3 --kernel-trace --stats
# of the same command per build, palindromic order.  usage: [AB_ARGS="--dtype f32x3" AB_STEPS=20] ab_rocprof.sh outdir A B ...
out=$1; shift
L=$GRAFT_REPO_ROOT/object_keypoints_amd/lib
order="$@"; rev=$(echo $order | tr ' ' '\n' | tac | tr '\n' ' ')
i=0
cd /tmp && export TMPDIR=/tmp
for v in $order $rev; do
  i=$((i+1))
  OKP_LIB=$L/libokp_hip_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_${i}_$v -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps ${AB_STEPS:-40} --warmup 10 ${AB_ARGS:-} --extra-dtypes '' --no-cpu-baseline --no-stream8 --no-cups --no-host-fed --no-probes > $GRAFT_REPO_ROOT/$out/bench_${i}_$v.json 2> $GRAFT_REPO_ROOT/$out/bench_${i}_$v.err
  f=$(find $GRAFT_REPO_ROOT/$out/prof_${i}_$v -name "*kernel_stats.csv" | head -1)
  echo "== build $v (run $i): $(python3 -c "import json,sys;d=json.loads(open('$GRAFT_REPO_ROOT/$out/bench_${i}_$v.json').read().strip().splitlines()[-1]);print('ms_per_step %.3f' % d['ms_per_step'])")"
  python3 - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
def short(n):
    m=re.search(r'okp_\w+',n); s=m.group(0) if m else n[:40]
    t=re.search(r'I(DF16[b_])((?:Li\d+E|Lb\dE)+)',n)
    if t: s+='<'+','.join(re.findall(r'L[ib](\d+)E',t.group(2)))+'>'
    return s
for r in rows[:16]:
    print(f"  {short(r['Name']):60s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
  find $GRAFT_REPO_ROOT/$out/prof_${i}_$v -name "*.db" -delete; find $GRAFT_REPO_ROOT/$out/prof_${i}_$v -name "*kernel_trace.csv" -delete
done

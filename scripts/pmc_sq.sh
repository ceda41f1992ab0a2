#!/bin/bash
# usage: pmc_sq.sh "<bench_conv args>"  : SQ counters of one convolution shape (one pass, 8 SQ slots)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sq
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/sq -- python3 $GRAFT_REPO_ROOT/scripts/bench_conv.py $1 iters=3 > /tmp/sq.log 2>&1
echo "rc=$?"
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "okp_igemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
v = {k: x[-1] for k, x in agg.items()}
print({k: f"{x:.4g}" for k, x in v.items()})
if "SQ_WAVE_CYCLES" in v:
    wc = v["SQ_WAVE_CYCLES"]
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS"):
        if k in v: print(f"{k}/WAVE_CYCLES = {v[k]/wc:.3f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CYCLES" in v:
        print(f"MFMA_BUSY/BUSY_CYCLES = {v['SQ_VALU_MFMA_BUSY_CYCLES']/v['SQ_BUSY_CYCLES']:.3f}")
PY

#!/bin/bash
# usage: pmc_sq3.sh "<bench_conv args>" : vector-memory side counters of one convolution launch (LDS-DMA issue / latency)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sqC
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/sqC -- python3 $GRAFT_REPO_ROOT/scripts/bench_conv.py $1 iters=3 > /tmp/sqC.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/sqC/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "okp_igemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
v = {k: x[-1] for k, x in agg.items()}
wc = v.get("SQ_WAVE_CYCLES", 1)
out = {k: round(x / wc, 4) for k, x in v.items() if k != "SQ_WAVE_CYCLES"}
out["vmem_latency_(LEVEL/INSTS)"] = round(v.get("SQ_INST_LEVEL_VMEM", 0) / max(v.get("SQ_INSTS_VMEM_RD", 1), 1), 1)
out["INSTS_VMEM_RD"] = v.get("SQ_INSTS_VMEM_RD")
print(json.dumps(out))
PY

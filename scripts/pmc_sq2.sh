#!/bin/bash
# usage: pmc_sq2.sh "<bench_conv args>" <tag> : SQ / LDS counters of one convolution launch in two passes (8 SQ slots each);
# prints one JSON object (last dispatch of the okp_igemm* kernel).  The program sits directly behind `--`.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sqA /tmp/sqB
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/sqA -- python3 $GRAFT_REPO_ROOT/scripts/bench_conv.py $1 iters=3 > /tmp/sqA.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d /tmp/sqB -- python3 $GRAFT_REPO_ROOT/scripts/bench_conv.py $1 iters=3 > /tmp/sqB.log 2>&1
python3 - "$2" <<'PY'
import csv, glob, collections, json, sys
v = {}
for d in ("/tmp/sqA", "/tmp/sqB"):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "okp_igemm" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    v.update({k: x[-1] for k, x in agg.items()})
out = {"tag": sys.argv[1], "counters": v}
wc = v.get("SQ_WAVE_CYCLES")
if wc:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS"):
        if k in v: out[k + "/WAVE_CYCLES"] = round(v[k] / wc, 4)
if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CYCLES" in v:
    out["MFMA_BUSY/BUSY_CYCLES"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["SQ_BUSY_CYCLES"], 4)
if "SQ_LDS_BANK_CONFLICT" in v and "SQ_LDS_IDX_ACTIVE" in v and v["SQ_LDS_IDX_ACTIVE"]:
    out["LDS_BANK_CONFLICT/IDX_ACTIVE"] = round(v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], 4)
print(json.dumps(out))
PY
tail -2 /tmp/sqA.log | cut -c1-200

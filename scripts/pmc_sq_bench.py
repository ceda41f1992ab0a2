"""Per-kernel SQ counter figures of the bench command.

usage: pmc_sq_bench.py <dir of SQ pass A> <dir of SQ pass B> <pmc_hbm_traffic.json> <kernel_stats.csv>  > profiles/rNN_sq_counters.json

Counters are per dispatch, summed over the whole chip; a kernel's row is the mean over its dispatches.
  mfma_busy_frac          = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs  over  SQ_BUSY_CYCLES / 32 shader engines: the share of the kernel's
                            duration in which a SIMD's matrix pipe is busy (checked on okp_igemm_patch_kernel: theoretical MFMA
                            cycles / duration x clock gives the same number)
  lds_wait_frac           = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES    (wave-cycles waiting to issue an LDS instruction)
  lds_bank_conflict_frac  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  hbm_GBps                = (corrected FETCH_SIZE + WRITE_SIZE per launch, separate passes) / average duration of the kernel-trace pass
"""
import collections, csv, glob, json, re, sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", n)[:96]
    return re.sub(r"(okp_igemm_patch_x3_kernel|okp_stem_x3_kernel)(<(true|false)>|ILb[01]E)", r"\1", n)    # instantiations that differ in the output format only: one population


def per_kernel(d):
    tot = collections.defaultdict(collections.Counter)
    cnt = collections.defaultdict(collections.Counter)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    return {k: {c: tot[k][c] / cnt[k][c] for c in tot[k]} for k in tot}


a, b = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
traffic = json.load(open(sys.argv[3]))
dur_tot, dur_n = collections.Counter(), collections.Counter()
for r in csv.DictReader(open(sys.argv[4])):          # (instantiations merged by short(): average over all their calls)
    dur_tot[short(r["Name"])] += float(r["TotalDurationNs"]); dur_n[short(r["Name"])] += float(r["Calls"])
dur = {k: dur_tot[k] / dur_n[k] for k in dur_tot if dur_n[k]}
out = {}
for k in sorted(set(a) | set(b)):
    if "okp_" not in k:
        continue
    v = dict(a.get(k, {})); v.update(b.get(k, {}))
    row = {"avg_us_kernel_trace": round(dur[k] / 1e3, 2) if k in dur else None}
    if v.get("SQ_BUSY_CYCLES"):
        row["mfma_busy_frac"] = round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / (v["SQ_BUSY_CYCLES"] / 32.0), 4)
    if v.get("SQ_WAVE_CYCLES"):
        row["lds_wait_frac"] = round(v.get("SQ_WAIT_INST_LDS", 0.0) / v["SQ_WAVE_CYCLES"], 4)
        row["wait_inst_any_frac"] = round(v.get("SQ_WAIT_INST_ANY", 0.0) / v["SQ_WAVE_CYCLES"], 4)
    if v.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_bank_conflict_frac"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"], 4)
    t = traffic.get(k)
    if t and k in dur:
        row["hbm_MB_per_launch"] = round(t["fetch_MB_per_launch_corrected"] + t["write_MB_per_launch"], 2)
        row["hbm_GBps"] = round(row["hbm_MB_per_launch"] * 1e6 / dur[k], 1)       # MB * 1e6 B / ns = GB/s
    row["counters_per_launch"] = {c: round(x, 1) for c, x in sorted(v.items())}
    out[k] = row
json.dump(out, sys.stdout, indent=1)
print()

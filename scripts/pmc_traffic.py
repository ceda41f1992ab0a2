"""Per-kernel HBM traffic per launch from two rocprofv3 PMC passes of the bench command.

usage: pmc_traffic.py <dir of the --pmc FETCH_SIZE pass> <dir of the --pmc WRITE_SIZE pass> > profiles/rNN_pmc_hbm_traffic.json

rocprofv3 reports both counters in KiB per dispatch.  FETCH_SIZE is doubled: on gfx950 it tallies the 128-byte
requests of wide streaming reads at 64 bytes (MI355X_MICROARCH.md, HBM section).  Calibration in this repo's own
access pattern: okp_stem_kernel writes exactly N*256*256*128*2 bytes (1073.74 MB at N=64)."""
import collections, csv, glob, json, re, sys


def per_kernel(d, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", r["Kernel_Name"])[:96]
            k = re.sub(r"(okp_igemm_patch_x3_kernel|okp_stem_x3_kernel)(<(true|false)>|ILb[01]E)", r"\1", k)    # instantiations that differ in the output format only: one population
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, 0) + write.get(k, 0))):
    if not k.startswith("okp_") and "okp_" not in k:
        continue
    out[k] = {"launches": int(nf.get(k, nw.get(k, 0))),
              "fetch_MB_per_launch_corrected": round(2.0 * fetch.get(k, 0.0) * 1024 / 1e6 / max(nf.get(k, 1), 1), 2),
              "write_MB_per_launch": round(write.get(k, 0.0) * 1024 / 1e6 / max(nw.get(k, 1), 1), 2)}
json.dump(out, sys.stdout, indent=1)
print()

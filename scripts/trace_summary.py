"""Summarise a rocprofv3 kernel-trace CSV: per-dispatch durations of the LAST forward (from the
last stem / frame-packing launch on) and totals per kernel family."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
packs = [i for i, r in enumerate(rows) if "okp_pack_frames" in r["Kernel_Name"] or "okp_preprocess" in r["Kernel_Name"]]
if not packs:      # the stem kernels read the frames themselves and are the first launch of a step
    packs = [i for i, r in enumerate(rows) if "okp_stem_kernel" in r["Kernel_Name"] or "okp_stem_x3_kernel" in r["Kernel_Name"]]
first = len(packs) - 1
# (fp32-storage configurations run the stem and pre[1] in two frame chunks: the launches then come in pairs - a short gap inside a step, a long
#  one between steps - and the step starts at the first of its pair)
ts = lambda i: int(rows[packs[i]]["Start_Timestamp"])
if first >= 2 and (ts(first) - ts(first - 1)) < 0.5 * (ts(first - 1) - ts(first - 2)):
    first -= 1
sel = rows[packs[first]:]
def short(n):
    if "okp_igemm_patch_x3_kernel" in n:
        return "okp_igemm_patch_x3_kernel"
    if "okp_igemm_patch_kernel" in n:
        return "okp_igemm_patch_kernel"
    if "okp_igemm_kernel<" in n:
        m = re.search(r"okp_igemm_kernel<(?:\(anonymous namespace\)::)?(\w+), (\d+), (\d+), \d+, \d+, (\d+), (\d+), (\d+), (\d+)>", n)
        if m:
            return f"igemm<{m.group(1)},{m.group(2)}x{m.group(3)},ring{m.group(4)}x{m.group(5)}B,mfma{m.group(6)},src{m.group(7)}>"
    if "okp_igemm_kernelI" in n:
        m = re.search(r"okp_igemm_kernelI(\w+?)Li(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", n)
        return f"igemm<{ 'bf16' if 'DF16b' in m.group(1) else 'f32'},{m.group(2)}x{m.group(3)},ring{m.group(4)}x{m.group(5)}B,mfma{m.group(6)},src{m.group(7)}>"
    for k in ("okp_stem_x3_kernel", "okp_fire_x3_kernel", "okp_dwconv3x3", "okp_pack_frames", "okp_head_out", "okp_heads_x3_kernel", "okp_heads_kernel", "okp_peak_nms", "okp_stem_kernel", "okp_fire_chain_kernel", "okp_fire2_kernel", "okp_group_objects", "okp_lift_peaks"):
        if k in n: return k
    return n[:40]
t0 = int(sel[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in sel)
tot = collections.Counter(); cnt = collections.Counter()
thr = int(sys.argv[2]) * 1000 if len(sys.argv) > 2 else None
for r in sel:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = short(r["Kernel_Name"])
    tot[k] += d; cnt[k] += 1
    if thr is not None and d > thr:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f}us  {d/1e3:8.1f}us  grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d} wg  {k}")
print(f"last forward: wall {(t1-t0)/1e3:.1f} us, sum of kernels {sum(tot.values())/1e3:.1f} us, {len(sel)} dispatches")
for k, v in tot.most_common():
    print(f"  {k:32s} n={cnt[k]:4d}  {v/1e3:9.1f} us")

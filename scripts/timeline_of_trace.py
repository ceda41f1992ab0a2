"""Every dispatch of the last network pass in a rocprofv3 --kernel-trace output directory (from the last stem launch on): start, end, duration (us),
queue, workgroups, kernel.  usage: timeline_of_trace.py <rocprofv3 -d directory>"""
import csv, sys, re, glob
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "okp_stem_kernel" in r["Kernel_Name"]]
sel = rows[st[-1]:]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    n = r["Kernel_Name"]
    m = re.search(r"okp_\w+", n)
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}us q{r.get('Queue_Id','?'):>3} wg{int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):5d} {m.group(0)[:40] if m else n[:40]}")

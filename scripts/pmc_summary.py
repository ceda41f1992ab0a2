import csv, sys, collections, glob
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(list)
        for r in rows:
            if "okp_igemm" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d, {k: f"{v[-1]:.4g}" for k, v in agg.items()})

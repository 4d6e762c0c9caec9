import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if sys.argv[1] not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in agg:
    print(k)
    for c, v in sorted(agg[k].items()): print(f"   {c:32s} {v / n[(k, c)]:.4g}")

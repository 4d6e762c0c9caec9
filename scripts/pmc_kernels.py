import csv, glob, sys, collections
want = ("hist_kernel<0>", "accumulate_stream_kernel<false>", "matcher_cost_f16", "gather_rows", "msda_fused_kernel", "hist_stream_kernel<1>", "select_kernel")
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = next((w for w in want if w in r["Kernel_Name"]), None)
            if k:
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in agg:
            print(k, {c: f"{v / n[(k, c)]:.3g}" for c, v in agg[k].items()}, "launches", max(n[(k, c)] for c in agg[k]))

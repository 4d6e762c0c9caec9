import csv, glob, sys, collections
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if "msda_fused" in r["Kernel_Name"]:
            k = (("head-major" if "true" in r["Kernel_Name"].split("msda_fused_kernel")[1][:12] else "interleaved"), r["Counter_Name"])
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for k, (v, n) in sorted(agg.items()):
        print(d, k, f"{v/n/1024:.1f} MB per launch (raw counter, KB units){'; x2 corrected = %.1f MB' % (v/n/512) if k[1]=='FETCH_SIZE' else ''}  launches {n}")

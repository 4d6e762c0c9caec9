"""rocprofv3 --pmc directories -> per kernel (name substring filter) the mean of every counter per launch, and ratios that matter for a
GEMM-shaped kernel.   python scripts/pmc_sum.py <substring> <dir> [<dir> ...]"""
import csv, glob, sys, collections
want = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"]:
                k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, c in agg.items():
    per = {cn: v / n[(k, cn)] for cn, v in c.items()}
    print(k, "launches", max(n[(k, cn)] for cn in c))
    for cn, v in sorted(per.items()):
        print(f"    {cn:32s} {v:.4g}")
    g = per.get
    if g("SQ_BUSY_CU_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
        print(f"    MFMA busy / CU busy cycles        {g('SQ_VALU_MFMA_BUSY_CYCLES') / g('SQ_BUSY_CU_CYCLES'):.3f}")
    if g("SQ_LDS_BANK_CONFLICT") and g("SQ_LDS_IDX_ACTIVE"):
        print(f"    LDS bank-conflict / active cycles {g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):.3f}")

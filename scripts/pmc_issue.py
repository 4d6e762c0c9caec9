"""instruction-issue utilisation per kernel: (vector + scalar + LDS + VMEM instructions per launch) / (1024 SIMDs x launch cycles at 2.1 GHz),
joined from a rocprofv3 --pmc pass (counter csv) and a --kernel-trace --stats pass (kernel_stats csv) of the same command"""
import csv, glob, sys, collections
pmc_dir, stats_csv = sys.argv[1], sys.argv[2]
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(pmc_dir + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(r["Kernel_Name"], r["Counter_Name"])] += 1
dur = {r["Name"]: (float(r["AverageNs"]), int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(stats_csv))}
rows = []
for k, c in cnt.items():
    if k not in dur: continue
    per = {cn: v / n[(k, cn)] for cn, v in c.items()}
    ins = sum(per.get(x, 0) for x in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"))
    cyc = dur[k][0] * 2.1
    rows.append((dur[k][2] / 1e6, k[:80], ins / (1024 * cyc) / 0.25, per.get("SQ_INSTS_VALU", 0) / max(ins, 1)))
for tot, k, util, vfrac in sorted(rows, reverse=True)[:30]:
    print(f"{tot:8.1f} ms  issue {util:5.2f} of peak  (vector share {vfrac:4.2f})  {k}")

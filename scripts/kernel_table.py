"""per-step table of a rocprofv3 --kernel-trace --stats kernel_stats.csv:  python scripts/kernel_table.py <csv> <steps in the run> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
print(f"total {sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps:.2f} ms per step")
for r in rows[:n]:
    print(f"{r['Name'][:84]:84s} n/step={int(r['Calls']) / steps:7.1f} ms/step={float(r['TotalDurationNs']) / 1e6 / steps:7.2f} avg_us={float(r['AverageNs']) / 1e3:8.1f}")

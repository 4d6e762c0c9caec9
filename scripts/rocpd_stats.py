"""rocprofv3 --kernel-trace (rocpd sqlite output) -> per-kernel stats CSV, the same columns as `--stats` prints.

    python scripts/rocpd_stats.py gpurun_out/prof_x/run_results.db profiles/r1_kernel_stats_x.csv"""
import csv, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("""select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
                         from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                         group by s.kernel_name order by 3 desc"""))
tot = sum(r[2] for r in rows)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100.0 * r[2] / tot, 3), r[4], r[5]])
print(f"{len(rows)} kernels, {tot / 1e6:.1f} ms of kernel time")
for r in rows[:12]:
    print(f"{r[0][:70]:70s} {r[1]:6d} {r[2] / 1e6:9.2f} ms  avg {r[3] / 1e3:9.1f} us")

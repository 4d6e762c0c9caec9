"""rocprofv3 --kernel-trace CSV -> timeline of the dispatches behind the last idle gap of >= 20 ms (the analysed repetition of
scripts/mb_decoder_trace.py): per kernel start offset, duration, gap to the previous end; then totals per kernel name.
    python scripts/trace_timeline.py <..._kernel_trace.csv> [out.txt]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[max(0, i - 8):i]) > 20e6:
        cut = i
ev = ev[cut:]
t0 = ev[0][0]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
busy = sum(e[1] - e[0] for e in ev)
span = max(e[1] for e in ev) - t0
print(f"# {len(ev)} dispatches, span {span / 1e6:.3f} ms, sum of durations {busy / 1e6:.3f} ms, idle between dispatches {(span - busy) / 1e6:.3f} ms", file=out)
agg = collections.defaultdict(lambda: [0, 0, 0])
prev_end = t0
for s, e, n in ev:
    short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    gap = s - prev_end
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap / 1e3:7.1f}  {short}", file=out)
    a = agg[short]; a[0] += 1; a[1] += e - s; a[2] += max(gap, 0)
    prev_end = max(prev_end, e)
print("# per kernel: calls, total us, total gap in front (us)", file=out)
for k, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"# {k:60s} {c:5d} {d / 1e3:10.1f} {g / 1e3:9.1f}", file=out)

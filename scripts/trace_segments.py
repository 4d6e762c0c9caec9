"""rocprofv3 --kernel-trace CSV of scripts/mb_train_trace.py + its stdout (the LABELS line) -> per segment (between two marker kernels of the
last iteration) the GPU time and the top kernels.
    python scripts/trace_segments.py <kernel_trace.csv> <stdout of mb_train_trace.py> [out.txt] [top]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
labels = [l for l in open(sys.argv[2]) if l.startswith("LABELS")][-1].rstrip("\n").split("\t")[1:]
out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
top = int(sys.argv[4]) if len(sys.argv) > 4 else 8
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
start = max(i for i, e in enumerate(ev) if "sin" in e[2] and "double" in e[2])
ev = ev[start + 1:]
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
segs, cur, name = [], [], "(before the first marker)"
li = 0
for s, e, n in ev:
    if "cos" in n and "double" in n:
        segs.append((name, cur)); cur = []; name = labels[li] if li < len(labels) else "?"; li += 1
        continue
    cur.append((s, e, n))
segs.append((name, cur))
assert li == len(labels), (li, len(labels))
tot_all = 0
# nested wrappers: a segment named X runs until the next marker, so "after X" segments hold what the caller did between two wrapped calls
agg = collections.OrderedDict()
for name, ks in segs:
    a = agg.setdefault(name, [0, 0, collections.defaultdict(lambda: [0, 0])])
    for s, e, n in ks:
        a[0] += 1; a[1] += e - s
        k = a[2][short(n)]; k[0] += 1; k[1] += e - s
        tot_all += e - s
print(f"# sum of kernel durations of the traced iteration: {tot_all / 1e6:.2f} ms", file=out)
for name, (c, d, ks) in agg.items():
    if c == 0:
        continue
    print(f"{d / 1e6:8.2f} ms {c:5d} launches  {name}", file=out)
    for k, (cc, dd) in sorted(ks.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"            {dd / 1e3:9.1f} us {cc:4d} x  {k}", file=out)

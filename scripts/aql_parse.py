"""Summarise the AQL packet headers of an AMD_LOG_LEVEL=4 log: per hardware queue, how many dispatches carried which
(barrier, acquire scope, release scope), and the packets around every attn_mask launch."""
import collections, re, sys
pat = re.compile(r"HWq=(0x[0-9a-f]+), id=(\d+), (Dispatch|BarrierAND|BarrierValue) Header\s*=\s*(0x[0-9a-f]+).*?\(type=(\d+), barrier=(\d+), acquire=(\d+), release=(\d+)\)")
name_pat = re.compile(r"ShaderName\s*:\s*(\S+)")
per = collections.defaultdict(collections.Counter)
seq = collections.defaultdict(list)
last_name = None
for line in open(sys.argv[1], errors="replace"):
    m = name_pat.search(line)
    if m:
        last_name = m.group(1)
        continue
    m = pat.search(line)
    if m:
        hwq, _, kind, hdr, typ, bar, acq, rel = m.groups()
        per[hwq][(kind, int(bar), int(acq), int(rel))] += 1
        seq[hwq].append((kind, int(bar), int(acq), int(rel), last_name if kind == "Dispatch" else None))
        if kind == "Dispatch":
            last_name = None
for hwq, c in per.items():
    print("HW queue", hwq)
    for k, n in c.most_common():
        print(f"   {n:7d} x {k[0]:12s} barrier={k[1]} acquire={k[2]} release={k[3]}")
shown = 0
for hwq, s in seq.items():
    for i, e in enumerate(s):
        if e[4] and "attn_mask" in e[4] and shown < 6:
            print("around attn_mask on", hwq)
            for j in range(max(0, i - 3), min(len(s), i + 2)):
                print("     ", s[j])
            shown += 1

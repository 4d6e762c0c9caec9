#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel (static count from the gfx950 assembly): python scripts/isa_mix.py <file.hip> <kernel substring>"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def asm(src):
    from s2d_amd.build import FLAGS, HIPCC
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run([HIPCC] + [f for f in FLAGS if f != "-fPIC"] + ["-S", "--cuda-device-only", src, "-o", out], capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr)
        return open(out).read()


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "trans"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")): return "lane"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    text = asm(sys.argv[1])
    sub = sys.argv[2]
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and sub in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    print(lines[start].split(":")[0], len(body), "lines")
    for l in lines[end:end + 400]:
        if re.search(r"\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size)", l) and "set" in l:
            print("   ", l.strip().split(".")[-1])
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    tot = collections.Counter()
    for l in body:
        s = l.strip()
        if s and not s.startswith((".", ";")) and not s.endswith(":"):
            tot[classify(s.split()[0])] += 1
    print("whole kernel:", dict(tot))
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            c = collections.Counter()
            for s in body[labels[m.group(1)]:i]:
                s = s.strip()
                if s and not s.startswith((".", ";")) and not s.endswith(":"):
                    c[classify(s.split()[0])] += 1
            print(f"loop {m.group(1)} lines {labels[m.group(1)]}..{i}: {dict(c)}")


if __name__ == "__main__":
    main()

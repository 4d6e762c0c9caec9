#!/usr/bin/env python3
"""ISA lint of libs2d_hip's kernels: fail on the instruction shape behind round 1's "two-stream stale read".

What was measured on MI355X (scripts/race_diag.py, gpurun_out/race/*.log, DESIGN.md "Streams"): with a second HIP stream
keeping the CUs busy,

        global_load_dword v23, ...          ; youngest outstanding load
        s_waitcnt vmcnt(0)
        v_pk_mul_f32 v[28:29], v[20:21], v[22:23]      ; packed-f32 op, FIRST reader of v23, straight behind the wait

computed its high lane-half product as if v23 were 0 in lanes 48..63 (the wave's last quarter), although the same register,
stored a few instructions later, held the loaded value; the same code with a single-lane VALU instruction (v_mul_f32,
v_mov_b32) as first reader, or with dwordx4 loads, never did.  The load, the wait and the packed op are all architecturally
correct, so the library avoids the shape instead: the kernels are built with -fno-slp-vectorize (the compiler's SLP pass is
what turns scalar arithmetic on freshly loaded values into v_pk_* instructions), and this lint checks the result.

Rule 1: inside one kernel, a MULTI-ELEMENT consumer -- `v_pk_*`, `v_cvt_pk*`, `v_dot*`, `v_mfma*`: instructions that read a
register as packed halves or as part of a fragment -- that directly follows an `s_waitcnt` carrying a vmcnt field (only `s_nop`
in between) and reads a VGPR that a ONE-register vector-memory load (dword or narrower) wrote earlier in the kernel is an
error.  (Round 2 linted `v_pk_*` only; the measured failure was a `v_pk_mul_f32`, the other three classes are refused on the same
grounds -- first reader of a just-awaited single dword that is not a plain one-lane-one-value VALU op -- without having been
seen to fail.  A plain VALU first reader is what the fixed kernels use and what ran clean.)

Rule 2 (source level): a function that spells out its own counted `s_waitcnt vmcnt(N)` with N > 0 in inline asm (a hand-built
pipeline: today the wave-specialised GEMM) must not issue masked buffer loads -- the `| OOB` / 0xFFFFFFF0 offset trick that lets
the hardware bounds check drop a load -- inside it: round 2's WS kernel read a register set's first rows before they had landed
whenever such dropped loads sat in the counted window (profiles/r2_ws_gemm/README.md), and ran clean once every load was clamped
in range.  Compiler-inserted counted waits are not concerned: the compiler counts what it issued.  (Multi-register loads -- dwordx2/x4, the library's normal access width -- never showed the effect: the 16-B form of
the same kernel ran 0 wrong words in > 500 two-stream launches, and every other kernel of the two-stream schedule is
bitwise equal to its one-stream result.)

    python scripts/isa_lint.py [file.s ...]          (no arguments: disassemble every csrc/*.hip for gfx950)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "s2d_amd", "csrc")
MULTI = ("v_pk_", "v_cvt_pk", "v_dot", "v_mfma")
LOAD = re.compile(r"^\s*(global_load|buffer_load|flat_load|scratch_load)_(\w+)\s+(v\[\d+:\d+\]|v\d+)")
REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    out = set()
    for a, b, c in REG.findall(tok):
        if c:
            out.add(int(c))
        else:
            out.update(range(int(a), int(b) + 1))
    return out


def lint_text(text, name):
    bad = []
    kernel, loaded, prev_wait = None, set(), False
    for ln, line in enumerate(text.split("\n"), 1):
        s = line.split(";")[0].strip()
        if not s:
            continue
        if s.endswith(":") and not s.startswith("."):
            kernel, loaded, prev_wait = s[:-1], set(), False
            continue
        if s.startswith(".") or s.endswith(":"):
            if s.endswith(":"):
                prev_wait = False                 # a branch target: the wait no longer directly precedes
            continue
        m = LOAD.match(s)
        if m and " lds" not in s:
            r = regs(m.group(3))
            if len(r) == 1:                       # the measured shape: one-register (dword or narrower) loads; wider loads never showed it
                loaded |= r
            else:
                loaded -= r
            prev_wait = False
            continue
        if s.startswith("s_waitcnt") and "vmcnt" in s:
            prev_wait = True
            continue
        if s.startswith("s_nop"):
            continue
        ops = s.split(None, 1)[1] if " " in s else ""
        parts = ops.split(",")
        used = set()
        for t in parts[1:]:                       # first operand is the destination
            used |= regs(t)
        if prev_wait and s.startswith(MULTI):
            hit = used & loaded
            if hit:
                bad.append(f"{name}:{ln}: {kernel}: `{s}` straight behind s_waitcnt vmcnt reads loaded v{sorted(hit)}")
        # a register another instruction has read (its data had landed) or overwritten is no longer "fresh from a load"
        loaded -= used
        if parts and not s.startswith(("global_store", "buffer_store", "flat_store", "ds_write", "ds_store", "scratch_store")):
            loaded -= regs(parts[0])
        prev_wait = False
    return bad


def lint_source(path):
    """Rule 2: hand-counted vmcnt(N > 0) and masked (OOB-dropped) buffer loads in one function"""
    bad = []
    text = open(path).read()
    # split at kernel / function heads: good enough for these files (one `__global__` or `__device__` head per definition)
    heads = [m.start() for m in re.finditer(r"^(template\s*<[^>]*>\s*)?(__global__|__device__|static)\b", text, re.M)] + [len(text)]
    for a, b in zip(heads, heads[1:]):
        body = text[a:b]
        counted = [m for m in re.finditer(r'asm\s+volatile\s*\(\s*"s_waitcnt\s+vmcnt\((\d+)\)', body) if int(m.group(1)) > 0]
        counted += [m for m in re.finditer(r"__builtin_amdgcn_s_waitcnt\s*\(", body)]
        if not counted:
            continue
        if re.search(r"\bOOB\b|0xFFFFFFF0", body) and "lint: every load in range" not in body:
            name = re.search(r"(\w+)\s*\(", body[body.find("void"):] if "void" in body else body)
            bad.append(f"{os.path.basename(path)}: {name.group(1) if name else '?'}: hand-counted s_waitcnt vmcnt(N>0) together with masked (OOB) buffer loads")
    return bad


def disassemble(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        sys.path.insert(0, ROOT)
        from s2d_amd.build import FLAGS, HIPCC
        r = subprocess.run([HIPCC] + [f for f in FLAGS if f != "-fPIC"] + ["-S", "--cuda-device-only", src, "-o", out],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
        return open(out).read()


def main(argv):
    files = argv or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    bad = []
    for f in files:
        text = open(f).read() if f.endswith(".s") else disassemble(f)
        bad += lint_text(text, os.path.basename(f))
        if not f.endswith(".s"):
            bad += lint_source(f)
    for b in bad:
        print(b)
    print(f"isa_lint: {len(files)} file(s), {len(bad)} finding(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

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

Rule 3 (round 5): matrix-core hazards that INLINE-ASM MFMAs hide from the compiler.  On gfx950 an MFMA's result registers may not be
read or written by a non-MFMA instruction (VALU, v_accvgpr_read / _mov, a store's data operand), nor read by a later MFMA as its A / B
operand, before NumPasses + 3 wait states have gone by (v_mfma_*_32x32x16_{f16,bf16}: 8 passes -> 11; 16x16x32: 4 -> 7; the fp32-input
32x32x2: 16 -> 19); and a VALU / v_accvgpr_write that writes a register an MFMA reads as SrcC needs 2 wait states in front of that MFMA.
Nothing in the hardware interlocks these: the compiler's hazard recogniser pads them with s_nop -- for instructions it KNOWS to be MFMAs.
An MFMA spelt in inline asm (csrc/ffn.hip: GEMM 1 on "+v" accumulators, so that the 256 output accumulators get the AGPR half) is an
opaque string to it: a build of that kernel with GEMM 2 in asm too (`"+a"` accumulators; profiles/r5_experiments/ffn_asm_hazard.txt)
has none of the 14 `s_nop 11` the builtin build carries between an MFMA and the v_accvgpr_read / v_accvgpr_mov_b32 of its
accumulator, and none of the 8 `s_nop 1` between a v_accvgpr_mov_b32 and the MFMA that accumulates into it -- the deterministic
1e-4 errors of round 4's asm form of GEMM 2 (a lost low-order product in whole output tiles) are what an accumulator read or moved
one MFMA too early gives.  This rule recomputes the distances on the built ISA, whoever wrote the instruction: wait states are
counted conservatively (every instruction 1, `s_nop N` N + 1) along straight-line code: a label or an unconditional branch ends a window.

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


AREG = re.compile(r"([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def xregs(tok):
    """{('v', n) / ('a', n)} named by an operand token"""
    out = set()
    for k, a, b, k1, c in AREG.findall(tok):
        if k1:
            out.add((k1, int(c)))
        else:
            out.update((k, i) for i in range(int(a), int(b) + 1))
    return out


def mfma_passes(op):
    if "32x32x2_f32" in op or "32x32x4" in op:
        return 16
    if "16x16x4_f32" in op or "32x32x16" in op or "32x32x8" in op:
        return 8
    if "16x16x32" in op or "16x16x16" in op:
        return 4
    return 16


def lint_mfma(text, name):
    """Rule 3: MFMA result -> non-MFMA reader / writer (or MFMA A / B reader) distance, VALU write -> MFMA SrcC distance"""
    bad = []
    kernel = None
    pend = []        # [dst regs, wait states still required, wait states seen, line, text]
    lastw = {}       # register -> wait states since a non-MFMA instruction wrote it
    for ln, line in enumerate(text.split("\n"), 1):
        s = line.split(";")[0].strip()
        if not s or s.startswith("."):
            continue
        if s.endswith(":"):
            if not s.startswith(".L") and not s.startswith("BB"):
                kernel = s[:-1]
            pend, lastw = [], {}          # a label: the code in front of it is not (only) what runs in front of it
            continue
        op = s.split()[0]
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            pend, lastw = [], {}
            continue
        ops = s.split(None, 1)[1] if " " in s else ""
        parts = [t.strip() for t in ops.split(",")]
        step = 1
        if op == "s_nop":
            step = int(parts[0], 0) + 1 if parts and parts[0] else 1
        is_mfma = op.startswith("v_mfma") or op.startswith("v_smfmac")
        if op != "s_nop" and not op.startswith("s_") or op.startswith("s_") and False:
            dst = xregs(parts[0]) if parts else set()
            srcs = set()
            for t in parts[1:]:
                srcs |= xregs(t)
            is_store = op.startswith(("global_store", "buffer_store", "flat_store", "ds_write", "ds_store", "scratch_store", "global_atomic", "buffer_atomic"))
            if is_store:
                srcs |= dst
                dst = set()
            for e in pend:
                if is_mfma:
                    ab = set()
                    for t in parts[1:3]:
                        ab |= xregs(t)
                    hit = ab & e[0]
                else:
                    hit = (srcs | dst) & e[0]
                if hit and e[2] < e[1]:
                    bad.append(f"{name}:{ln}: {kernel}: `{s}` touches {sorted(hit)[:4]} {e[2]} wait states behind `{e[4]}` (line {e[3]}; needs {e[1]})")
            if is_mfma and len(parts) >= 4:
                srcc = xregs(parts[3])
                for r in srcc:
                    if lastw.get(r, 99) < 2:
                        bad.append(f"{name}:{ln}: {kernel}: `{s}` reads SrcC {r} {lastw[r]} wait states behind a VALU write (needs 2)")
                        break
        for e in pend:
            e[2] += step
        pend = [e for e in pend if e[2] < e[1]]
        for r in list(lastw):
            lastw[r] += step
            if lastw[r] > 4:
                del lastw[r]
        if is_mfma:
            pend.append([xregs(parts[0]), mfma_passes(op) + 3, 0, ln, s])
        elif op.startswith("v_") and parts:
            for r in xregs(parts[0]):
                lastw[r] = 0
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
        bad += lint_mfma(text, os.path.basename(f))
        if not f.endswith(".s"):
            bad += lint_source(f)
    for b in bad:
        print(b)
    print(f"isa_lint: {len(files)} file(s), {len(bad)} finding(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

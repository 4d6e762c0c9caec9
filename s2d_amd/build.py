"""Build libs2d_hip.so (every HIP kernel + the C ABI) for gfx950, in-tree.

    python -m s2d_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU box.
"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libs2d_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: the SLP pass packs scalar f32 arithmetic on freshly loaded values into v_pk_*_f32; that shape -- a packed
# op as the first reader of a just-awaited dword load -- is what produced wrong values under a two-stream schedule on
# MI355X (DESIGN.md "Streams"; scripts/isa_lint.py checks the built ISA for it), and packed f32 ops buy nothing beside MFMAs
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fno-vectorize", "-Wall", "-Wno-unused-function",
         "-I" + CSRC, "-I" + os.path.join(os.path.dirname(CSRC), "..", "include")]


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in (src,) + tuple(extra))


def _compile(src, force):
    obj = os.path.join(CSRC, "_obj", os.path.basename(src) + ".o")
    hdrs = tuple(glob.glob(os.path.join(CSRC, "*.h"))) + tuple(glob.glob(os.path.join(CSRC, "..", "..", "include", "*.h")))
    if force or _newer(src, obj, hdrs):
        os.makedirs(os.path.dirname(obj), exist_ok=True)
        r = subprocess.run([HIPCC] + FLAGS + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
        # a shared library links with undefined symbols: load it once, so that a kernel whose host stub the compiler dropped (seen: a
        # kernel-body lambda calling another lambda that returns a value) fails HERE, not on the GPU box
        import ctypes
        try:
            ctypes.CDLL(LIB)
        except OSError as e:
            raise RuntimeError(f"{LIB} does not load: {e}")
        if verbose:
            print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

"""Cut one kernel's ISA out of a `hipcc -S --cuda-device-only` listing and count its instruction classes.
    python scripts/kext.py listing.s <substring of the mangled kernel name> out.s"""
import re
import sys

src, pat, out = sys.argv[1:4]
lines = open(src).read().split("\n")
st = None
for i, l in enumerate(lines):
    if l and not l[0].isspace() and not l.startswith(".") and pat in l.split(":")[0] and ":" in l:
        st = i
        break
assert st is not None, "kernel not found"
en = st
while not lines[en].startswith(".Lfunc_end"):
    en += 1
body = lines[st:en + 1]
open(out, "w").write("\n".join(body))
pats = dict(valu=r"^\s+v_", salu=r"^\s+s_", bperm=r".*ds_bpermute", dsread=r".*ds_read", dswrite=r".*ds_write", vmem=r".*(buffer_load|global_load)",
            scratch=r".*scratch_", waits=r".*s_waitcnt", branches=r".*s_cbranch", mfma=r".*v_mfma")
print(pat, "lines", len(body), " ".join(f"{k} {sum(1 for l in body if re.match(r, l))}" for k, r in pats.items()))

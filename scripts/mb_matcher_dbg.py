"""timing experiments on the matcher cost kernel: one process per experiment build (scripts/build_matcher_dbg.sh N ...; S2D_MATCHER_DBG
bits: 1 no target gathers, 2 no staged-row DMA, 4 no tap-table setup, 8 no query sampling).  Results of those builds are wrong by
construction.   python scripts/mb_matcher_dbg.py 0 1 2 4 ..."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for n in sys.argv[1:]:
    env = dict(os.environ, S2D_MATCHER_MIX="1")
    if n != "0":
        env["S2D_HIP_LIB"] = os.path.join(os.path.dirname(HERE), "s2d_amd", "csrc", f"libs2d_hip_mdbg{n}.so")
    r = subprocess.run([sys.executable, os.path.join(HERE, "mb_matcher_time.py"), "once"], env=env, capture_output=True, text=True)
    print(f"dbg={n}:", " ".join(l for l in r.stdout.splitlines() if "ms/call" in l) or r.stderr[-400:], flush=True)

"""timing experiments on the fused MSDeformAttn gather (scripts/build_msda_dbg.sh N ...): one process per build.  python scripts/mb_msda_dbg.py 0 1 2"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
for n in sys.argv[1:]:
    env = dict(os.environ)
    if n != "0":
        env["S2D_HIP_LIB"] = os.path.join(os.path.dirname(HERE), "s2d_amd", "csrc", f"libs2d_hip_sdbg{n}.so")
    r = subprocess.run([sys.executable, os.path.join(HERE, "mb_msda_tiled.py")], env=env, capture_output=True, text=True)
    print(f"dbg={n}:\n" + "\n".join(l for l in r.stdout.splitlines() if "HEAD=0" in l) or r.stderr[-400:], flush=True)

"""timing experiments on the one-launch FFN: one process per experiment build (scripts/build_ffn_dbg.sh N ...; S2D_FFN_DBG bits:
1 no DMA, 2 no barrier, 4 no activation work, 8 no fragment reads).  Results of those builds are wrong by construction.
python scripts/mb_ffn_dbg.py 0 1 2 4 ..."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys
sys.path.insert(0, os.path.dirname(%r))
import torch
from s2d_amd import ops
M = 309120
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((M, 256), device=dev, generator=g)
W1 = torch.nn.Parameter(torch.randn((1024, 256), device=dev, generator=g) * 0.06)
W2 = torch.nn.Parameter(torch.randn((256, 1024), device=dev, generator=g) * 0.03)
b1 = torch.randn((1024,), device=dev, generator=g) * 0.1
b2 = torch.randn((256,), device=dev, generator=g) * 0.1
def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n
out = []
for p in (0.0, 0.3):
    out.append("p=%%.1f %%.3f ms" %% (p, timeit(lambda: ops.ffn_fused(x, W1, b1, W2, b2, dropout=(p, 7, 1, 2) if p > 0 else None))))
print("dbg=%%s: " %% os.environ.get("DBG") + " | ".join(out), flush=True)
''' % HERE
for n in sys.argv[1:]:
    env = dict(os.environ, DBG=n)
    if n != "0":
        env["S2D_HIP_LIB"] = os.path.join(HERE, "..", "s2d_amd", "csrc", f"libs2d_hip_dbg{n}.so")
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)

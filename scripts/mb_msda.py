"""timing harness for the fused MSDA launch at the c4 pyramid (used for the head-major / branch-free / LDS-staged
experiments recorded in DESIGN.md section 5; the shipped kernel is the direct gather)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from s2d_amd import ops
dev = torch.device("cuda")
shapes = [(23, 40), (46, 80), (92, 160)]
S = sum(h * w for h, w in shapes); N, M, D, L, P = 16, 8, 32, 3, 4
torch.manual_seed(0)
both = torch.randn((N, S, 288 + 256), device=dev)
both[..., :192] *= 2.0          # offsets of a few pixels, like the initialised module
value, oa = both[..., 288:], both[..., :288]
def t(n=10):
    for _ in range(2): y = ops.msda_fused_forward(value, np.array(shapes), oa)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): y = ops.msda_fused_forward(value, np.array(shapes), oa)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, y
dt, y = t()
print(f"msda_fused_forward: {dt*1e3:.3f} ms  checksum {float(y.double().sum()):.6f} {float(y.abs().max()):.6f}", flush=True)

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
if os.environ.get("MB_MSDA_HM", "") != "1":
    dt, y = t()
    print(f"msda_fused_forward: {dt*1e3:.3f} ms  checksum {float(y.double().sum()):.6f} {float(y.abs().max()):.6f}", flush=True)
else:
    y = ops.msda_fused_forward(value, np.array(shapes), oa)

# head-major value layout [N][M][S][32]: same arithmetic, a pixel's 128 B of one head next to its
# neighbours' instead of 1 KB apart.  MB_MSDA_HM=1 runs only this variant (for a PMC pass of its own).
from s2d_amd._lib import lib
def hm_forward(value_hm, oa):
    out = torch.empty((N, S, M * D), device=dev, dtype=torch.float32)
    sh = ops._host_i64(np.array(shapes))
    lib().call("s2d_msda_fused_forward_f32", value_hm, -1, sh, oa, oa.stride(1), N, S, M, D, L, P, out, torch.cuda.current_stream().cuda_stream)
    return out
value_hm = value.reshape(N, S, M, D).permute(0, 2, 1, 3).contiguous()
def t_hm(n=10):
    for _ in range(2): y = hm_forward(value_hm, oa)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): y = hm_forward(value_hm, oa)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, y
if os.environ.get("MB_MSDA_HM", "") != "0":
    dth, yh = t_hm()
    print(f"msda_fused_forward, head-major value: {dth*1e3:.3f} ms  bits equal: {bool(torch.equal(y, yh))}", flush=True)

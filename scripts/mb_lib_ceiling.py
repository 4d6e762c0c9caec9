"""Ceiling probe: what a tuned library fp16 GEMM reaches on the S2D shapes with K tripled (the MFMA work of the
split-fp16 x3 contraction, without its conversions and with half its input bytes).  Diagnostic only."""
import torch, time
shapes = [(309120, 1024, 256), (309120, 256, 1024), (309120, 256, 256), (942080, 256, 2304), (942080, 256, 64), (58880, 256, 2304)]
for M, N, K in shapes:
    a = torch.randn((M, 3 * K), device="cuda", dtype=torch.float16)
    b = torch.randn((N, 3 * K), device="cuda", dtype=torch.float16)
    for _ in range(3): c = a @ b.t()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): c = a @ b.t()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"M={M} N={N} K={K}: {dt*1e6:8.1f} us  algorithmic {2*M*N*K/dt/1e12:7.1f} TF  mfma {6*M*N*K/dt/1e12:7.1f} TF", flush=True)

# what the reference's own fp32 path gets from the vendor library on this GPU (torch.matmul fp32 in / fp32 out; the reference
# runs these layers through torch.nn.Linear / F.conv2d), next to this repo's kernels on the same fp32 problem
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2d_amd import ops
torch.backends.cuda.matmul.allow_tf32 = False
for M, N, K in shapes[:3] + [(235520, 256, 512), (58880, 1024, 256)]:
    a = torch.randn((M, K), device="cuda"); w = torch.nn.Parameter(torch.randn((N, K), device="cuda") / K ** 0.5, requires_grad=False)
    def t(fn, n=8):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
    d_lib = t(lambda: a @ w.t())
    line = f"fp32 problem M={M} N={N} K={K}: library fp32 {d_lib*1e6:8.1f} us {2*M*N*K/d_lib/1e12:6.1f} TF"
    for mode in ("f16x3", "bf16x3"):
        ops.set_dense_mode(mode)
        d = t(lambda: ops.gemm_nt(a, w))
        line += f" | {mode} {d*1e6:8.1f} us {2*M*N*K/d/1e12:6.1f} TF"
    ops.set_dense_mode("f16x3")
    print(line, flush=True)

"""cost of the fused dropout epilogue: the encoder's three dropout GEMMs (linear1 + ReLU, linear2 + residual, out_proj + residual) with p = 0.3 and without"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
dev = torch.device("cuda")
torch.manual_seed(0)
for (M, N, K, res, relu) in [(309120, 1024, 256, False, True), (309120, 256, 1024, True, False), (309120, 256, 256, True, False)]:
    A = torch.randn((M, K), device=dev)
    W = torch.nn.Parameter(torch.randn((N, K), device=dev) / K ** 0.5, requires_grad=False)
    b = torch.randn((N,), device=dev)
    R = torch.randn((M, N), device=dev) if res else None
    out = torch.empty((M, N), device=dev)
    for name, drop in (("no dropout", None), ("p = 0.3", (0.3, 1234, 1))):
        fn = lambda: ops.gemm_nt(A, W, bias=b, res=R, relu=relu, dropout=drop, out=out)
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"gemm {M}x{N}x{K} res={int(res)} relu={int(relu)} {name:11s}: {dt*1e3:7.3f} ms  {2.0*M*N*K/dt/1e12:6.1f} TFLOP/s   kept {float((out != 0).float().mean()):.4f}", flush=True)

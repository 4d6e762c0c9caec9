"""short-K 1x1 convolutions that close a bottleneck (GEMM + FrozenBN scale/bias + residual + ReLU): S2D_GEMM_HI_PRE=1 (default: residual tile
prefetched before the first k-tile) vs =0.  Run once per setting."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
dev = torch.device("cuda")
torch.manual_seed(0)
tag = f"PRE={os.environ.get('S2D_GEMM_HI_PRE', '1')}"
for (M, N, K) in [(942080, 256, 64), (235520, 512, 128), (942080, 64, 64), (58880, 1024, 256)]:
    A = torch.randn((M, K), device=dev)
    W = torch.nn.Parameter(torch.randn((N, K), device=dev) / K ** 0.5, requires_grad=False)
    sc = torch.rand((N,), device=dev) + 0.5; bi = torch.randn((N,), device=dev)
    R = torch.randn((M, N), device=dev)
    fn = lambda: ops.gemm_nt(A, W, sc, bi, R, relu=True)
    for _ in range(3): y = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): y = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    i = torch.randint(0, M, (4096,), device=dev)
    ref = torch.relu((A[i].double() @ W.double().t()) * sc.double() + bi.double() + R[i].double())
    err = float((y[i].double() - ref).abs().max() / ref.abs().max())
    gb = 4.0 * (M * K + 2 * M * N) / 1e9
    print(f"{tag} gemm+res {M}x{N}x{K}: {dt*1e3:7.3f} ms  {gb/dt/1e3:5.2f} TB/s (A + res + C once)  rel err {err:.1e}", flush=True)

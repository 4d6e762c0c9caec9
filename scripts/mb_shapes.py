"""dense GEMM shapes of the step under the current dispatch (S2D_GEMM_HI=0 forces the 128x128 kernel)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
dev = torch.device("cuda")
torch.manual_seed(0)
def t(fn, n=6):
    for _ in range(2): y = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): y = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
tag = f"HI={os.environ.get('S2D_GEMM_HI','2')}"
for (M, N, K) in [(309120, 1024, 256), (309120, 256, 1024), (58880, 1024, 256), (58880, 256, 1024), (14720, 2048, 512), (235520, 256, 512), (942080, 128, 256), (58880, 256, 256), (235520, 256, 256), (309120, 288, 256), (200, 256, 256), (200, 2048, 256), (942080, 256, 64), (942080, 64, 256), (942080, 256, 256), (235520, 512, 128), (235520, 128, 512), (309120, 544, 256), (309120, 256, 256), (471040, 100, 256)]:
    A = torch.randn((M, K), device=dev); W = torch.nn.Parameter(torch.randn((N, K), device=dev) / K ** 0.5, requires_grad=False)
    dt = t(lambda: ops.gemm_nt(A, W))
    gb = 4.0 * (M * K + M * N) / 1e9
    print(f"{tag} gemm {M}x{N}x{K}: {dt*1e3:7.3f} ms {2*M*N*K/dt/1e12:6.1f} TF  {gb/dt/1e3:5.2f} TB/s (A+C once)", flush=True)

"""one NT GEMM shape, forward form (static pre-split weights, bias): python scripts/mb_gemm_shape.py M N K [reps] -- for rocprofv3 --pmc passes"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
M, N, K = (int(a) for a in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda")
x = torch.randn((M, K), device=dev)
w = torch.nn.Parameter(torch.randn((N, K), device=dev) / K ** 0.5, requires_grad=False)
b = torch.randn((N,), device=dev)
for _ in range(3): y = ops.gemm_nt(x, w, bias=b)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): y = ops.gemm_nt(x, w, bias=b)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f"gemm_nt M={M} N={N} K={K}: {dt * 1e3:.3f} ms  {2.0 * M * N * K / dt / 1e12:.0f} TFLOP/s  {(M * K + M * N) * 4 / dt / 1e12:.2f} TB/s (A + C)")

import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
ops.set_dense_mode("bf16x3")
for N, K in [(256, 256), (1024, 256), (256, 1024), (256, 2048), (1024, 1024)]:
    for M in [8192, 32768, 131072, 524288, 1572864]:
        if M * max(N, K) * 4 > 3.5e9: continue
        A = torch.randn((M, K), device="cuda"); B = torch.randn((N, K), device="cuda") / K ** 0.5
        for _ in range(2): C = ops.gemm_nt(A, B)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): C = ops.gemm_nt(A, B)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        gb = (M * K + M * N) * 4 / 1e9
        print(f"M={M:8d} N={N:5d} K={K:5d}  {dt*1e3:8.3f} ms  {2.0*M*N*K/dt/1e12:7.1f} TF   {gb/dt/1e3:6.2f} TB/s (A+C {gb:.2f} GB)")
        del A, B, C

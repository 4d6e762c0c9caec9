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

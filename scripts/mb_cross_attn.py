"""masked cross-attention at the c4 shapes, per level"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
dev = torch.device("cuda")
B, Q, T, C = 2, 100, 8, 256
torch.manual_seed(0)
for (hl, wl) in [(23, 40), (46, 80), (92, 160)]:
    K = T * hl * wl
    q = torch.randn((B, Q, C), device=dev); k = torch.randn((B, K, C), device=dev); v = torch.randn((B, K, C), device=dev)
    bits = torch.randint(-2**31, 2**31 - 1, (B, K, 4), device=dev, dtype=torch.int32)
    unm = torch.full((B, 4), -1, device=dev, dtype=torch.int32)
    for _ in range(3): o = ops.masked_attn(q, k, v, bits, unm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): o = ops.masked_attn(q, k, v, bits, unm)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    rd = 2 * B * K * C * 4
    print(f"level {hl}x{wl}: K={K}: {dt*1e6:7.1f} us  K+V {rd/1e6:6.1f} MB -> {rd/dt/1e12:5.2f} TB/s", flush=True)

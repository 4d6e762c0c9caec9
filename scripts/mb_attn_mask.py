"""attn_mask_bits at the c4 shapes, per level"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
dev = torch.device("cuda")
B, Q, T, hm, wm = 2, 100, 8, 184, 320
ml = torch.randn((B, T * hm * wm, Q), device=dev)
for (hl, wl) in [(23, 40), (46, 80), (92, 160)]:
    for _ in range(3): bits, unm = ops.attn_mask_bits(ml, B, Q, T, hm, wm, hl, wl)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): bits, unm = ops.attn_mask_bits(ml, B, Q, T, hm, wm, hl, wl)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    rd = B * T * hl * wl * 4 * Q * 4
    print(f"level {hl}x{wl}: {dt*1e6:7.1f} us  reads {rd/1e6:6.1f} MB -> {rd/dt/1e12:5.2f} TB/s", flush=True)

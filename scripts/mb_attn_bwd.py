"""masked cross-attention backward at the decoder's three levels (B = 2 clips, Q = 100, T = 8 frames): S2D_ATTN_BWD_MFMA=1 (default, matrix cores)
vs 0 (the scalar fp32 kernels), one process per setting (the switch is read once); prints ms per call and the gradients' checksums"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import backward as Bk, ops
dev = torch.device("cuda")
B, Q, C, H = 2, 100, 256, 8
g = torch.Generator(device=dev).manual_seed(0)
for (hl, wl) in ((23, 40), (46, 80), (92, 160)):
    K = 8 * hl * wl
    q = torch.randn((B, Q, C), device=dev, generator=g); k = torch.randn((B, K, C), device=dev, generator=g); v = torch.randn((B, K, C), device=dev, generator=g)
    dout = torch.randn((B, Q, C), device=dev, generator=g)
    bits = torch.randint(0, 2 ** 31, (B, K, 4), device=dev, generator=g, dtype=torch.int64).to(torch.int32) & torch.randint(0, 2 ** 31, (B, K, 4), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
    unm = torch.full((B, 4), -1, device=dev, dtype=torch.int32)
    out, lse = ops.masked_attn(q, k, v, bits, unm, want_lse=True)
    for _ in range(2): dq, dk, dv = Bk.masked_attn_backward(q, k, v, out, lse, dout, bits, unm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): dq, dk, dv = Bk.masked_attn_backward(q, k, v, out, lse, dout, bits, unm)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"MFMA={os.environ.get('S2D_ATTN_BWD_MFMA', '1')} K={K}: {dt * 1e3:.3f} ms  dq {float(dq.double().abs().sum()):.6e} dk {float(dk.double().abs().sum()):.6e} dv {float(dv.double().abs().sum()):.6e}", flush=True)

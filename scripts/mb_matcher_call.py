"""one VideoHungarianMatcher cost pass at config c4 (10 layers x 2 clips, Q = 100, N = 10, P = 160 000, T = 8): time per call and a
checksum of the cost matrices (RNG points, fixed seed)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
NL, B, Q, T, hm, wm, H, W, N, P = 10, 2, 100, 8, 184, 320, 736, 1280, 10, 160000
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(3)
coarse = torch.randn((NL * B * T, Q, hm // 8 + 2, wm // 8 + 2), generator=g, device=dev) * 6
ml = torch.nn.functional.interpolate(coarse, size=(hm, wm), mode="bilinear").permute(0, 2, 3, 1).reshape(NL, B, T * hm * wm, Q).contiguous()
cls = torch.randn((NL, B, Q, 2), generator=g, device=dev)
yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
tgt = torch.zeros((B, N, T, H, W), device=dev, dtype=torch.uint8)
for b in range(B):
    for n in range(N):
        cy, cx, r = 100 + 50 * n, 150 + 100 * n, 40 + 12 * n
        tgt[b, n] = (((yy - cy) ** 2 + (xx - cx - 40 * b) ** 2) < r * r).to(torch.uint8)
cnt = torch.full((B,), N, dtype=torch.int32, device=dev)
for _ in range(2):
    C = ops.matcher_cost(ml, cls, tgt, cnt, (Q, T, hm, wm), P, (0.0, 5.0, 5.0), seed=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    C = ops.matcher_cost(ml, cls, tgt, cnt, (Q, T, hm, wm), P, (0.0, 5.0, 5.0), seed=1)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
iq, it, nm = ops.lsap(C, cnt, B)
print(f"matcher_cost call: {dt*1e3:.3f} ms   sum C {float(C.double().sum()):.6f}   assignment checksum {int((iq.long() * 131 + it.long()).sum())}", flush=True)

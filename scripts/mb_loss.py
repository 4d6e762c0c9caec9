"""time the point-loss launch sequence (ops.point_loss) at the c4 shapes: NL=10 layers x B=2 clips x N=10 matched targets x T=8 frames
= 1600 rows, 184x320 logit maps, 736x1280 targets, P=160000 (3P oversampled + P/4 random points per row).  S2D_HIP_LIB selects an
experiment build (scripts/build_loss_dbg.sh)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
NL, B, Q, T, hm, wm, H, W, N, P = 10, 2, 100, 8, 184, 320, 736, 1280, 10, 160000
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
coarse = torch.randn((NL * B * Q, 1, T * hm // 8, wm // 8), device=dev, generator=g) * 3
ml = torch.nn.functional.interpolate(coarse, size=(T * hm, wm), mode="bilinear").view(NL, B, Q, T * hm * wm).permute(0, 1, 3, 2).contiguous()
tgt = (torch.rand((B, N, T, H // 16, W // 16), device=dev, generator=g) > 0.7).to(torch.uint8).repeat_interleave(16, 3).repeat_interleave(16, 4).contiguous()
cnt = torch.full((B,), N, dtype=torch.int32, device=dev)
ne = ops.target_nonempty(tgt, cnt)
idx_q = torch.stack([torch.randperm(Q, device=dev, generator=g)[:N] for _ in range(NL * B)]).to(torch.int32).contiguous()
idx_t = torch.arange(N, device=dev, dtype=torch.int32).repeat(NL * B, 1).contiguous()
nm = torch.full((NL * B,), N, dtype=torch.int32, device=dev)
def run():
    return ops.point_loss(ml, tgt, cnt, ne, idx_q, idx_t, nm, (Q, T, hm, wm), P, seed=3)
for _ in range(2): L = run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): L = run()
torch.cuda.synchronize()
print(os.path.basename(os.environ.get("S2D_HIP_LIB", "default")), f"{(time.perf_counter() - t0) / 5 * 1e3:.2f} ms/call  losses[0] {L[0].tolist()}", flush=True)

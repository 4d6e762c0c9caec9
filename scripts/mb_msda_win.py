"""fused MSDA launch at the c4 pyramid: LDS-windowed kernel for the finest level's queries (S2D_MSDA_WIN=1, default) against the
4x4-patch global-gather kernel (=0), bit-compared, for offsets identical for all queries (initialisation: bias only), smooth
(bias + 0.3 px noise), with 2 % of the samples thrown far outside the windows, and random (sigma 2 px / 6 px)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from s2d_amd import ops
dev = torch.device("cuda")
shapes = [(23, 40), (46, 80), (92, 160)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
S = sum(h * w for h, w in shapes); N = 16
torch.manual_seed(0)
def run(value, oa, n=20):
    for _ in range(3): y = ops.msda_fused_forward(value, np.array(shapes), oa)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): y = ops.msda_fused_forward(value, np.array(shapes), oa)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, y
for name in ("identical offsets (init)", "smooth offsets (bias + 0.3 px noise)", "smooth + 2 % outliers (30 px)", "random offsets (sigma 2 px)", "random offsets (sigma 6 px)"):
    both = torch.randn((N, S, 288 + 256), device=dev)
    if name.startswith("random"):
        both[..., :192] *= 2.0 if "2 px" in name else 6.0
    else:
        th = torch.arange(8, device=dev) * (2 * np.pi / 8)
        g = torch.stack([th.cos(), th.sin()], -1); g = g / g.abs().max(-1, keepdim=True)[0]
        bias = (g.view(8, 1, 1, 2) * torch.arange(1, 5, device=dev).view(1, 1, 4, 1)).expand(8, 3, 4, 2).reshape(-1)
        both[..., :192] = bias + (0.0 if name.startswith("identical") else 0.3 * torch.randn((N, S, 192), device=dev))
        if "outliers" in name:
            both[..., :192] += 30.0 * (torch.rand((N, S, 192), device=dev) < 0.02)
    value, oa = both[..., 288:], both[..., :288]
    res = {}
    for mode in ("0", "1"):
        os.environ["S2D_MSDA_WIN"] = mode
        res[mode] = run(value, oa)
    same = torch.equal(res["0"][1], res["1"][1])
    print(f"{name:40s} gather {res['0'][0]*1e3:.3f} ms   windowed {res['1'][0]*1e3:.3f} ms   bit-identical {same}"
          + ("" if same else f"  max |diff| {float((res['0'][1] - res['1'][1]).abs().max()):.3e}"), flush=True)

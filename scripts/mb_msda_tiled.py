"""fused MSDA launch at the c4 pyramid: 4x4-patch workgroups (S2D_MSDA_TILED=1, default) vs 4-consecutive-query workgroups (=0);
offsets random per query (as after training) and identical for all queries (as at initialisation: offsets = bias only).
Run once per setting (the switch is read once per process):  S2D_MSDA_TILED=0 python scripts/mb_msda_tiled.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from s2d_amd import ops
dev = torch.device("cuda")
shapes = [(23, 40), (46, 80), (92, 160)]
S = sum(h * w for h, w in shapes); N = 16
torch.manual_seed(0)
for name in ("random offsets (sigma 2 px)", "smooth offsets (bias + 0.3 px noise)", "identical offsets (init)"):
    both = torch.randn((N, S, 288 + 256), device=dev)
    if name.startswith("random"):
        both[..., :192] *= 2.0
    else:
        th = torch.arange(8, device=dev) * (2 * np.pi / 8)
        g = torch.stack([th.cos(), th.sin()], -1); g = g / g.abs().max(-1, keepdim=True)[0]
        bias = (g.view(8, 1, 1, 2) * torch.arange(1, 5, device=dev).view(1, 1, 4, 1)).expand(8, 3, 4, 2).reshape(-1)
        both[..., :192] = bias + (0.3 * torch.randn((N, S, 192), device=dev) if name.startswith("smooth") else 0.0)
    value, oa = both[..., 288:], both[..., :288]
    for head in ("1", "0", "1", "0"):                   # S2D_MSDA_HEAD is read per call: one head per workgroup (coarsest level in LDS) vs the patch kernel
        os.environ["S2D_MSDA_HEAD"] = head
        for _ in range(3): y = ops.msda_fused_forward(value, np.array(shapes), oa)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): y = ops.msda_fused_forward(value, np.array(shapes), oa)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"S2D_MSDA_TILED={os.environ.get('S2D_MSDA_TILED', '1')} S2D_MSDA_HEAD={head}  {name:40s} {dt*1e3:.3f} ms  checksum {float(y.double().sum()):.6f}", flush=True)

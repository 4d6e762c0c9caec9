"""a few launches of the fused MSDA op at the c4 pyramid with initialisation-like offsets (for rocprofv3 passes): S2D_MSDA_WIN selects the kernel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from s2d_amd import ops
dev = torch.device("cuda")
shapes = [(23, 40), (46, 80), (92, 160)]
S = sum(h * w for h, w in shapes); N = 16
torch.manual_seed(0)
both = torch.randn((N, S, 288 + 256), device=dev)
th = torch.arange(8, device=dev) * (2 * np.pi / 8)
g = torch.stack([th.cos(), th.sin()], -1); g = g / g.abs().max(-1, keepdim=True)[0]
bias = (g.view(8, 1, 1, 2) * torch.arange(1, 5, device=dev).view(1, 1, 4, 1)).expand(8, 3, 4, 2).reshape(-1)
both[..., :192] = bias + 0.3 * torch.randn((N, S, 192), device=dev)
value, oa = both[..., 288:], both[..., :288]
for _ in range(5): y = ops.msda_fused_forward(value, np.array(shapes), oa)
torch.cuda.synchronize()
print("ok", float(y.double().sum()))

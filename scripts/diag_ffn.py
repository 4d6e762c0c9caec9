"""where the one-launch FFN differs from a float64 reference (debugging aid)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from s2d_amd import ops

M, F = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = torch.Generator().manual_seed(0)
x = torch.randn((M, 256), generator=g)
W1 = torch.randn((F, 256), generator=g) * 0.06
W2 = torch.randn((256, F), generator=g) * 0.03
b1 = torch.randn((F,), generator=g) * 0.1
b2 = torch.randn((256,), generator=g) * 0.1
ref = (torch.relu(x.double() @ W1.double().t() + b1.double()) @ W2.double().t() + b2.double() + x.double()).numpy()
for rep in range(3):
    out = ops.ffn_fused(x.cuda(), torch.nn.Parameter(W1.cuda()), b1.cuda(), torch.nn.Parameter(W2.cuda()), b2.cuda()).cpu().double().numpy()
    d = np.abs(out - ref) / np.abs(ref).max()
    bad = d > 2e-5
    print(f"rep {rep}: max {d.max():.3e}, bad {bad.sum()} of {bad.size}")
    rows, cols = np.nonzero(bad)
    print(" bad rows (count per row):", {int(r): int((rows == r).sum()) for r in np.unique(rows)})
    print(" bad cols histogram by 16:", np.bincount(cols // 16, minlength=16).tolist())
# per-chunk contribution check: zero all but one chunk of hidden units
for c in (0, 1, 2, 15, 30, 31):
    if c * 32 >= F:
        continue
    W2c = torch.zeros_like(W2); W2c[:, 32 * c:32 * c + 32] = W2[:, 32 * c:32 * c + 32]
    refc = (torch.relu(x.double() @ W1.double().t() + b1.double()) @ W2c.double().t() + b2.double() + x.double()).numpy()
    out = ops.ffn_fused(x.cuda(), torch.nn.Parameter(W1.cuda()), b1.cuda(), torch.nn.Parameter(W2c.cuda()), b2.cuda()).cpu().double().numpy()
    d = np.abs(out - refc) / np.abs(ref).max()
    print(f"only chunk {c}: max {d.max():.3e}, bad {(d > 2e-5).sum()}")

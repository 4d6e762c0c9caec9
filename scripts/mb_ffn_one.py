"""the one-launch encoder FFN at the metric shape, a few launches (for rocprofv3 --pmc / --kernel-trace):
python scripts/mb_ffn_one.py [rows] [p] [ln: 0 none, 1 ln2, 2 ln1+ln2]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from s2d_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 309120
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
ln = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((M, 256), device=dev, generator=g)
W1 = torch.nn.Parameter(torch.randn((1024, 256), device=dev, generator=g) * 0.06)
W2 = torch.nn.Parameter(torch.randn((256, 1024), device=dev, generator=g) * 0.03)
b1 = torch.randn((1024,), device=dev, generator=g) * 0.1
b2 = torch.randn((256,), device=dev, generator=g) * 0.1
g1, be1, g2, be2 = (torch.randn((256,), device=dev, generator=g) * 0.1 + (1 if i % 2 == 0 else 0) for i in range(4))
for _ in range(6):
    y = ops.ffn_fused(x, W1, b1, W2, b2, ln1=(g1, be1) if ln == 2 else None, ln2=(g2, be2) if ln >= 1 else None,
                      dropout=(p, 7, 1, 2) if p > 0 else None)
torch.cuda.synchronize()
print(float(y.abs().mean()))

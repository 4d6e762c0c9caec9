"""one 3x3 conv shape a few times (for rocprofv3 --pmc)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
x = torch.randn((16, 184, 320, 256), device="cuda"); w = torch.randn((256, 3, 3, 256), device="cuda") / 48
for _ in range(3):
    y = ops.conv2d_nhwc(x, w, 1, 1)
torch.cuda.synchronize()

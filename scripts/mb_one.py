"""run one dense shape a few times (for rocprofv3 --pmc)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
M, N, K = [int(v) for v in sys.argv[1:4]]
mode = sys.argv[4] if len(sys.argv) > 4 else "bf16x3"
ops.set_dense_mode(mode)
A = torch.randn((M, K), device="cuda"); B = torch.randn((N, K), device="cuda") / K ** 0.5
for _ in range(3):
    C = ops.gemm_nt(A, B)
torch.cuda.synchronize()

"""3x3 convolutions of the step (R50 res3-5, FPN layer_1): input-halo kernel (default) against the implicit-GEMM kernel (S2D_CONV_HALO=0)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
dev = torch.device("cuda")
torch.manual_seed(0)
def t(fn, n=8):
    for _ in range(3): y = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): y = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
tag = f"HALO={os.environ.get('S2D_CONV_HALO','1')}"
for (n, H, W, C) in [(16, 184, 320, 256), (16, 92, 160, 128), (16, 46, 80, 256), (16, 23, 40, 512), (16, 92, 160, 256)]:
    x = torch.randn((n, H, W, C), device=dev); w = torch.nn.Parameter(torch.randn((C, 3, 3, C), device=dev) / (9 * C) ** 0.5, requires_grad=False)
    b = torch.randn((C,), device=dev)
    dt = t(lambda: ops.conv2d_nhwc(x, w, stride=1, pad=1, bias=b, relu=True))
    y = ops.conv2d_nhwc(x, w, stride=1, pad=1, bias=b, relu=True)
    print(f"{tag} conv3x3 {n}x{H}x{W} {C}->{C}: {dt*1e3:7.3f} ms {2.0*n*H*W*C*9*C/dt/1e12:6.1f} TF  checksum {float(y.double().sum()):.4f}", flush=True)

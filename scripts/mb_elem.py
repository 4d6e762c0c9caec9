"""achieved HBM bandwidth of the forward's streaming kernels at their c4 shapes (algorithmic bytes / time)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
dev = torch.device("cuda")
torch.manual_seed(0)
def t(fn, n=10):
    for _ in range(3): y = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): y = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
def rep(name, dt, nbytes):
    print(f"{name:58s} {dt*1e3:7.3f} ms  {nbytes/dt/1e12:5.2f} TB/s", flush=True)
M, C = 309120, 256
x = torch.randn((M, C), device=dev); r = torch.randn((M, C), device=dev); g = torch.randn((C,), device=dev); b = torch.randn((C,), device=dev)
rep("layernorm 309120x256 + residual", t(lambda: ops.layernorm(x, g, b, res=r)), 3 * 4 * M * C)
rep("layernorm 309120x256", t(lambda: ops.layernorm(x, g, b)), 2 * 4 * M * C)
xs = torch.randn((1600, C), device=dev)
rep("layernorm 1600x256 (decoder)", t(lambda: ops.layernorm(xs, g, b), 50), 2 * 4 * 1600 * C)
f = torch.randn((16, 184, 320, 256), device=dev)
rep("groupnorm 16x184x320x256 G=32 (+relu)", t(lambda: ops.groupnorm_nhwc(f, 32, g, b, relu=True)), 3 * 4 * f.numel())
up = torch.randn((16, 92, 160, 256), device=dev)
rep("groupnorm 16x184x320x256 + upsampled add", t(lambda: ops.groupnorm_nhwc(f, 32, g, b, up=up)), 3 * 4 * f.numel() + 4 * up.numel())
pos = torch.randn((19320, C), device=dev)
xb = torch.randn((16 * 19320, C), device=dev)
rep("add_bcast 309120x256 + pos[19320x256]", t(lambda: ops.add_bcast(xb, pos)), 2 * 4 * xb.numel())
s = torch.randn((16, 368, 640, 64), device=dev)
rep("maxpool3x3s2 16x368x640x64", t(lambda: ops.maxpool3x3s2(s)), 4 * s.numel() * 1.25)

import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
from s2d_amd import ops
dev = torch.device("cuda")
torch.manual_seed(0)
def t(fn, n=8):
    for _ in range(3): y = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): y = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for mode in ("f16x3", "bf16x3", "f16x3"):
    ops.set_dense_mode(mode)
    for (M, N, K) in [(309120, 1024, 256), (309120, 256, 1024), (309120, 256, 256), (235520, 256, 512), (58880, 1024, 256)]:
        A = torch.randn((M, K), device=dev); W = torch.nn.Parameter(torch.randn((N, K), device=dev) / K ** 0.5, requires_grad=False)
        dt = t(lambda: ops.gemm_nt(A, W))
        print(f"{mode} gemm {M}x{N}x{K}: {dt*1e3:7.3f} ms {2*M*N*K/dt/1e12:6.1f} TF", flush=True)
    x = torch.randn((16, 92, 160, 256), device=dev); w = torch.nn.Parameter(torch.randn((256, 3, 3, 256), device=dev) / 48, requires_grad=False)
    dt = t(lambda: ops.conv2d_nhwc(x, w, stride=1, pad=1, relu=True))
    print(f"{mode} conv 16x92x160 256->256 k3: {dt*1e3:7.3f} ms {2.0*16*92*160*256*9*256/dt/1e12:6.1f} TF", flush=True)

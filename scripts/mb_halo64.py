"""res2's 3x3 convolution (64 -> 64 channels, 184 x 320, 16 frames): the input-halo kernel on 16 x 16 patches (default) vs the implicit-GEMM 128 x 64
kernel (S2D_CONV_HALO64=0).  Run once per setting."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
dev = torch.device("cuda"); torch.manual_seed(0)
tag = f"HALO64={os.environ.get('S2D_CONV_HALO64', '1')}"
for (N, H, W, Cin, Cout) in [(16, 184, 320, 64, 64), (16, 92, 160, 64, 64), (4, 184, 320, 64, 48)]:
    x = torch.randn((N, H, W, Cin), device=dev)
    w = ops.mark_static(torch.randn((Cout, 3, 3, Cin), device=dev) / (9 * Cin) ** 0.5)
    sc = torch.rand((Cout,), device=dev) + 0.5; bi = torch.randn((Cout,), device=dev)
    fn = lambda: ops.conv2d_nhwc(x, w, 1, 1, scale=sc, bias=bi, relu=True)
    for _ in range(3): y = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): y = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{tag} conv3x3 {N}x{H}x{W} {Cin}->{Cout}: {dt*1e3:7.3f} ms  {2.0*N*H*W*Cout*9*Cin/dt/1e12:6.1f} TFLOP/s  checksum {float(y.double().sum()):.4f}", flush=True)

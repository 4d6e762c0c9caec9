"""the R50 stem (7x7/2 on NHWC4, 64 channels) at c4: halo kernel (default) vs implicit GEMM (S2D_CONV_STEM=0); run once per setting"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
dev = torch.device("cuda"); torch.manual_seed(0)
tag = f"STEM={os.environ.get('S2D_CONV_STEM', '1')}"
for (N, H, W) in [(16, 736, 1280), (4, 480, 864), (1, 33, 47)]:
    x = torch.randn((N, H, W, 4), device=dev); x[..., 3] = 0
    w = ops.mark_static(torch.randn((64, 7, 7, 4), device=dev) / 12.0)
    sc = torch.rand((64,), device=dev) + 0.5; bi = torch.randn((64,), device=dev)
    fn = lambda: ops.conv2d_nhwc(x, w, 2, 3, scale=sc, bias=bi, relu=True)
    for _ in range(3): y = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): y = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    ref = torch.relu(torch.nn.functional.conv2d(x[:1].double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), stride=2, padding=3).permute(0, 2, 3, 1) * sc.double() + bi.double())
    err = float((y[:1].double() - ref).abs().max() / ref.abs().max())
    print(f"{tag} stem {N}x{H}x{W}: {dt*1e3:7.3f} ms  rel err {err:.1e}", flush=True)

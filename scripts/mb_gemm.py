"""micro-benchmark: dense kernels, fp32-MFMA vs split-bf16, on the shapes that dominate the c4 step"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from s2d_amd import ops
dev = torch.device("cuda")
torch.manual_seed(0)
shapes = [("enc linear1  309120x1024x256", 309120, 1024, 256), ("enc linear2  309120x256x1024", 309120, 256, 1024),
          ("K proj      1884160x256x256", 1884160, 256, 256), ("einsum B=1   471040x100x256", 471040, 100, 256),
          ("res3 1x1    235520x512x128", 235520, 512, 128), ("decoder      200x2048x256", 200, 2048, 256)]
for name, M, N, K in shapes:
    A = torch.randn((M, K), device=dev); B = torch.randn((N, K), device=dev) / K ** 0.5
    res = {}
    for mode in ("f32", "bf16x3", "f16x3"):
        ops.set_dense_mode(mode)
        for _ in range(2): C = ops.gemm_nt(A, B)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): C = ops.gemm_nt(A, B)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        res[mode] = (dt, C)
    ref = (A[:4096].double() @ B.double().T)
    fl = 2.0 * M * N * K
    print(f"{name:34s} " + " | ".join(f"{m} {res[m][0]*1e3:7.3f} ms {fl/res[m][0]/1e12:6.1f} TF err {((res[m][1][:4096].double()-ref).abs().max()/ref.abs().max()).item():.1e}" for m in res))
# conv 3x3
x = torch.randn((16, 92, 160, 128), device=dev); w = torch.randn((128, 3, 3, 128), device=dev) / 34
for mode in ("f32", "bf16x3", "f16x3"):
    ops.set_dense_mode(mode)
    for _ in range(2): y = ops.conv2d_nhwc(x, w, 1, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): y = ops.conv2d_nhwc(x, w, 1, 1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"conv3x3 res3 16x92x160x128->128 {mode:7s} {dt*1e3:7.3f} ms {2*16*92*160*128*9*128/dt/1e12:6.1f} TF")

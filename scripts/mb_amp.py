"""per-shape time of the AMP (single-pass fp16) kernel against the fp32-class split-fp16 x3 dispatch on the shapes the AMP mode covers"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
dev = torch.device("cuda")
torch.manual_seed(0)


def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


convs = [(16, 368, 640, 4, 64, 7, 2, 3), (16, 184, 320, 64, 64, 3, 1, 1), (16, 92, 160, 128, 128, 3, 1, 1), (16, 46, 80, 256, 256, 3, 1, 1), (16, 23, 40, 512, 512, 3, 1, 1),
         (16, 184, 320, 128, 128, 3, 2, 1), (16, 92, 160, 256, 256, 3, 2, 1)]
for (N, H, W, Cin, Cout, k, s, p) in convs:
    x = torch.randn((N, H, W, Cin), device=dev); w = torch.nn.Parameter(torch.randn((Cout, k, k, Cin), device=dev) * 0.05)
    sc = torch.ones((Cout,), device=dev); sh = torch.zeros((Cout,), device=dev)
    t3 = timed(lambda: ops.conv2d_nhwc(x, w, s, p, scale=sc, bias=sh, relu=True))
    with ops.amp_fp16(True):
        t1 = timed(lambda: ops.conv2d_nhwc(x, w, s, p, scale=sc, bias=sh, relu=True))
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    fl = 2.0 * N * Ho * Wo * Cout * k * k * Cin
    print(f"conv {k}x{k}/{s} {N}x{H}x{W}x{Cin}->{Cout}: x3 {t3*1e3:7.3f} ms ({fl/t3/1e12:6.1f} TF)   amp {t1*1e3:7.3f} ms ({fl/t1/1e12:6.1f} TF)   x{t3/t1:.2f}", flush=True)
gemms = [(942080, 256, 64, True), (942080, 64, 256, False), (235520, 512, 128, True), (235520, 128, 512, False), (58880, 1024, 256, True), (58880, 256, 1024, False),
         (14720, 2048, 512, True), (14720, 512, 2048, False), (235520, 768, 256, False), (58880, 768, 256, False), (200, 256, 256, False), (200, 2048, 256, False), (200, 256, 2048, False)]
for (M, N, K, res) in gemms:
    A = torch.randn((M, K), device=dev); B = torch.nn.Parameter(torch.randn((N, K), device=dev) * 0.05)
    R = torch.randn((M, N), device=dev) if res else None
    bias = torch.zeros((N,), device=dev)
    t3 = timed(lambda: ops.gemm_nt(A, B, bias=bias, res=R, relu=True))
    with ops.amp_fp16(True):
        t1 = timed(lambda: ops.gemm_nt(A, B, bias=bias, res=R, relu=True))
    fl = 2.0 * M * N * K
    by = 4.0 * (M * K + M * N * (2 if res else 1))
    print(f"gemm {M}x{N}x{K}{' +res' if res else ''}: x3 {t3*1e3:7.3f} ms ({fl/t3/1e12:6.1f} TF, {by/t3/1e12:4.2f} TB/s)   amp {t1*1e3:7.3f} ms ({fl/t1/1e12:6.1f} TF, {by/t1/1e12:4.2f} TB/s)   x{t3/t1:.2f}", flush=True)
mf = torch.randn((2, 471040, 256), device=dev); e = torch.randn((2, 100, 256), device=dev); out = torch.empty((2, 471040, 100), device=dev)
t3 = timed(lambda: ops.gemm_nt(mf, e, out=out))
with ops.amp_fp16(True):
    t1 = timed(lambda: ops.gemm_nt(mf, e, out=out))
print(f"einsum 2x471040x100x256: x3 {t3*1e3:.3f} ms   amp {t1*1e3:.3f} ms   x{t3/t1:.2f}")

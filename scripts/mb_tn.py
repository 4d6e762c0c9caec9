"""Weight-gradient contractions (the TN kernel, csrc/gemm_tn.hip) at the training iteration's shapes: ms and algorithmic TFLOP/s per call,
next to the forward contraction of the same shape.  MB_TN_ONLY=<substring>: only the matching cases (PMC passes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import backward as B, ops
dev = torch.device("cuda")
only = os.environ.get("MB_TN_ONLY")


def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for (M, N, K) in [(309120, 1024, 256), (309120, 256, 1024), (309120, 544, 256), (942080, 256, 256), (235520, 768, 256), (942080, 64, 256)]:
    name = f"linear M={M} N={N} K={K}"
    if only and only not in name: continue
    x = torch.randn((M, K), device=dev); dy = torch.randn((M, N), device=dev)
    w = torch.nn.Parameter(torch.randn((N, K), device=dev) / K ** 0.5, requires_grad=False)
    fl = 2.0 * M * N * K / 1e12
    a, b = t(lambda: ops.gemm_nt(x, w)), t(lambda: B.weight_grad(dy, x))
    os.environ["S2D_TN_MFMA16"] = "0" if os.environ.get("S2D_TN_MFMA16", "0") == "1" else "1"
    c = t(lambda: B.weight_grad(dy, x)); other = os.environ["S2D_TN_MFMA16"]
    os.environ["S2D_TN_MFMA16"] = "0" if other == "1" else "1"
    print(f"{name}: forward {a:.3f} ms ({fl / a * 1e3:.0f} TFLOP/s) | wgrad {b:.3f} ms ({fl / b * 1e3:.0f} TFLOP/s) | wgrad with S2D_TN_MFMA16={other}: {c:.3f} ms", flush=True)
    del x, dy
for (N_, H, W, Ci, Co, s) in [(16, 184, 320, 256, 256, 1), (16, 184, 320, 64, 64, 1), (16, 92, 160, 128, 128, 1), (16, 46, 80, 256, 256, 1),
                              (16, 23, 40, 512, 512, 1), (16, 184, 320, 128, 128, 2), (16, 92, 160, 256, 256, 2)]:
    name = f"conv3x3 {N_}x{H}x{W} {Ci}->{Co} stride {s}"
    if only and only not in name: continue
    x = torch.randn((N_, H, W, Ci), device=dev)
    w = torch.nn.Parameter(torch.randn((Co, 3, 3, Ci), device=dev) / (9 * Ci) ** 0.5, requires_grad=False)
    y = ops.conv2d_nhwc(x, w, stride=s, pad=1)
    dy = torch.randn_like(y)
    fl = 2.0 * y.numel() * Ci * 9 / 1e12
    a, b = t(lambda: ops.conv2d_nhwc(x, w, stride=s, pad=1)), t(lambda: B.conv_weight_grad(dy, x, 3, 3, s, 1))
    os.environ["S2D_TN_MFMA16"] = "0" if os.environ.get("S2D_TN_MFMA16", "0") == "1" else "1"
    c = t(lambda: B.conv_weight_grad(dy, x, 3, 3, s, 1)); other = os.environ["S2D_TN_MFMA16"]
    os.environ["S2D_TN_MFMA16"] = "0" if other == "1" else "1"
    print(f"{name}: forward {a:.3f} ms ({fl / a * 1e3:.0f} TFLOP/s) | wgrad {b:.3f} ms ({fl / b * 1e3:.0f} TFLOP/s) | wgrad with S2D_TN_MFMA16={other}: {c:.3f} ms", flush=True)
    del x, y, dy

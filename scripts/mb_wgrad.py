"""where the time of a linear layer's backward goes (s2d_amd/backward.py) at the encoder's shapes"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import backward as B, ops
dev = torch.device("cuda")


def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for (M, N, K) in [(309120, 1024, 256), (309120, 256, 1024), (309120, 544, 256), (309120, 256, 256), (942080, 64, 256), (942080, 256, 64), (235520, 768, 256)]:
    x = torch.randn((M, K), device=dev); dy = torch.randn((M, N), device=dev)
    w = torch.nn.Parameter(torch.randn((N, K), device=dev) / K ** 0.5, requires_grad=False)
    S, chunk = B._slices(M, ((N + 127) // 128) * ((K + 127) // 128))
    Mp = S * chunk
    t_fwd = t(lambda: ops.gemm_nt(x, w))
    t_dg = t(lambda: B.input_grad(dy, w))
    t_tr = t(lambda: (B.transpose(dy, Mp), B.transpose(x, Mp)))
    t_wg = t(lambda: B.weight_grad(dy, x))
    t_bg = t(lambda: B.bias_grad(dy))
    print(f"linear M={M} N={N} K={K}: forward {t_fwd:.3f} ms | dgrad {t_dg:.3f} | wgrad {t_wg:.3f} (transposes {t_tr:.3f}, {S} slices of {chunk}) | bias grad {t_bg:.3f}", flush=True)

"""the video decoder's query-side GEMMs (M = 200) through ops.gemm_nt: S2D_GEMM_SMALL=1 (default: gemm_small.hip) vs =0 (tiled kernels);
checks against a float64 product.  Run once per setting."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
dev = torch.device("cuda")
torch.manual_seed(0)
tag = f"SMALL={os.environ.get('S2D_GEMM_SMALL', '1')}"
for (M, N, K, res, relu) in [(200, 256, 256, True, False), (200, 256, 256, False, False), (200, 2048, 256, False, True), (200, 256, 2048, True, False), (200, 512, 256, False, False),
                             (200, 2, 256, False, False), (1, 768, 256, False, False), (100, 256, 256, False, True), (256, 320, 128, False, False), (37, 70, 64, True, True),
                             (126, 288, 256, False, False), (96, 256, 1024, False, False), (24, 256, 2048, False, False), (32, 256, 256, True, False), (32, 2048, 256, False, True)]:
    A = torch.randn((M, K), device=dev)
    W = torch.nn.Parameter(torch.randn((N, K), device=dev) / K ** 0.5, requires_grad=False)
    b = torch.randn((N,), device=dev)
    R = torch.randn((M, N), device=dev) if res else None
    fn = lambda: ops.gemm_nt(A, W, bias=b, res=R, relu=relu)
    for _ in range(5): y = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): y = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    ref = A.double() @ W.double().t() + b.double() + (R.double() if res else 0.0)
    if relu: ref = ref.clamp_min(0)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    print(f"{tag} gemm {M}x{N}x{K} res={int(res)} relu={int(relu)}: {dt*1e6:7.2f} us/launch (back to back)  rel err {err:.2e}", flush=True)

"""wave-specialised GEMM kernel against the current dispatch on the step's large shapes (S2D_GEMM_WS=0/1 per process)"""
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
tag = f"WS={os.environ.get('S2D_GEMM_WS','0')}"
shapes = [(309120, 1024, 256, 0), (309120, 256, 1024, 1), (309120, 544, 256, 0), (309120, 256, 256, 1), (58880, 1024, 256, 0),
          (58880, 256, 1024, 1), (14720, 2048, 512, 0), (235520, 256, 512, 0), (942080, 256, 256, 0), (235520, 128, 512, 0)]
for (M, N, K, r) in shapes:
    A = torch.randn((M, K), device=dev); W = torch.nn.Parameter(torch.randn((N, K), device=dev) / K ** 0.5, requires_grad=False)
    b = torch.randn((N,), device=dev)
    R = torch.randn((M, N), device=dev) if r else None
    dt = t(lambda: ops.gemm_nt(A, W, bias=b, res=R, relu=not r))
    gb = 4.0 * (M * K + M * N * (2 if r else 1)) / 1e9
    print(f"{tag} gemm {M}x{N}x{K} res={r}: {dt*1e3:7.3f} ms {2*M*N*K/dt/1e12:6.1f} TF  {gb/dt/1e3:5.2f} TB/s", flush=True)
if os.environ.get("MB_PRESPLIT", "1") == "1":
    for (M, N, K, r) in shapes:
        A = torch.randn((M, K), device=dev); W = torch.nn.Parameter(torch.randn((N, K), device=dev) / K ** 0.5, requires_grad=False)
        b = torch.randn((N,), device=dev)
        R = torch.randn((M, N), device=dev) if r else None
        As = ops.split_rows(A)
        ref = ops.gemm_nt(A, W, bias=b, res=R, relu=not r)
        out = ops.gemm_nt_presplit(As, M, K, W, bias=b, res=R, relu=not r)
        same = bool(torch.equal(ref, out))
        dt = t(lambda: ops.gemm_nt_presplit(As, M, K, W, bias=b, res=R, relu=not r))
        dts = t(lambda: ops.split_rows(A))
        print(f"{tag} presplit-A gemm {M}x{N}x{K} res={r}: {dt*1e3:7.3f} ms {2*M*N*K/dt/1e12:6.1f} TF  (split pass {dts*1e3:6.3f} ms, bits equal to gemm_nt: {same})", flush=True)
for (M, N, K, r, relu) in [(309120, 256, 1024, 1, 0), (309120, 256, 256, 1, 0), (309120, 1024, 256, 0, 1)]:
    A = torch.randn((M, K), device=dev); W = torch.nn.Parameter(torch.randn((N, K), device=dev) / K ** 0.5, requires_grad=False)
    b = torch.randn((N,), device=dev)
    R = torch.randn((M, N), device=dev) if r else None
    dt = t(lambda: ops.gemm_nt(A, W, bias=b, res=R, relu=bool(relu), dropout=(0.3, 1234, 1)))
    print(f"{tag} gemm+dropout {M}x{N}x{K} res={r}: {dt*1e3:7.3f} ms {2*M*N*K/dt/1e12:6.1f} TF", flush=True)
if os.environ.get("MB_CONV", "1") == "1":
    for (n, H, Wd, Ci, Co, k) in [(16, 92, 160, 256, 256, 3), (16, 46, 80, 256, 256, 3), (16, 184, 320, 64, 64, 3)]:
        x = torch.randn((n, H, Wd, Ci), device=dev); w = torch.nn.Parameter(torch.randn((Co, k, k, Ci), device=dev) / (k * k * Ci) ** 0.5, requires_grad=False)
        b = torch.randn((Co,), device=dev)
        dt = t(lambda: ops.conv2d_nhwc(x, w, stride=1, pad=k // 2, bias=b, relu=True))
        fl = 2.0 * n * H * Wd * Co * k * k * Ci
        print(f"{tag} conv {n}x{H}x{Wd} {Ci}->{Co} k{k}: {dt*1e3:7.3f} ms {fl/dt/1e12:6.1f} TF", flush=True)

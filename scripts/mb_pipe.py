"""128x128 split-fp16 kernel: plain vs software-pipelined main loop (S2D_GEMM_PIPE), on conv / GEMM shapes of the step"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
dev = torch.device("cuda")
torch.manual_seed(0)
def t(fn, n=5):
    for _ in range(2): y = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): y = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, y
tag = f"PIPE={os.environ.get('S2D_GEMM_PIPE','0')} HI={os.environ.get('S2D_GEMM_HI','2')}"
for (N, H, W, Cin, Cout, k) in [(16, 184, 320, 256, 256, 3), (16, 46, 80, 256, 256, 3), (16, 92, 160, 128, 128, 3), (16, 23, 40, 512, 512, 3)]:
    x = torch.randn((N, H, W, Cin), device=dev); w = torch.randn((Cout, k, k, Cin), device=dev) / (k * k * Cin) ** 0.5
    dt, y = t(lambda: ops.conv2d_nhwc(x, w, 1, k // 2))
    ref = torch.nn.functional.conv2d(x[:1].permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), padding=k // 2).permute(0, 2, 3, 1)
    err = ((y[:1].double() - ref).abs().max() / ref.abs().max()).item()
    print(f"{tag} conv{k}x{k} {N}x{H}x{W}x{Cin}->{Cout}: {dt*1e3:7.3f} ms {2*N*H*W*Cout*k*k*Cin/dt/1e12:6.1f} TF err {err:.1e}", flush=True)
for (M, Nn, K) in [(309120, 1024, 256), (309120, 256, 1024), (309120, 256, 256), (58880, 1024, 256), (333, 200, 260)]:
    A = torch.randn((M, K), device=dev); B = torch.randn((Nn, K), device=dev) / K ** 0.5
    dt, C = t(lambda: ops.gemm_nt(A, B))
    ref = A[:2048].double() @ B.double().T
    err = ((C[:2048].double() - ref).abs().max() / ref.abs().max()).item()
    print(f"{tag} gemm {M}x{Nn}x{K}: {dt*1e3:7.3f} ms {2*M*Nn*K/dt/1e12:6.1f} TF err {err:.1e}", flush=True)

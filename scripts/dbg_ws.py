import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from s2d_amd import ops
torch.manual_seed(0)
dev = "cuda"
prev = None
def check(M, N, K, bias, res, relu, reps=4):
    global prev
    A = torch.randn((M, K), device=dev); B = torch.randn((N, K), device=dev)
    b = torch.randn((N,), device=dev) if bias else None
    R = torch.randn((M, N), device=dev) if res else None
    raw = A.double() @ B.double().T
    ref = raw
    if bias: ref = ref + b.double()
    if res: ref = ref + R.double()
    if relu: ref = ref.clamp_min(0)
    for rep in range(reps):
        out = ops.gemm_nt(A, B, bias=b, res=R, relu=relu)
        torch.cuda.synchronize()
        bad = (out.double() - ref).abs() > 1e-3 * ref.abs().max()
        nb = int(bad.sum())
        rows = torch.nonzero(bad.any(1)).flatten().tolist()
        cols = torch.nonzero(bad.any(0)).flatten().tolist()
        print(f"M{M} N{N} K{K} bias{int(bias)} res{int(res)} relu{int(relu)} rep{rep}: bad {nb} rows {rows[:12]} ncols {len(cols)} cols {cols[:3]}", flush=True)
        if nb:
            r0, c0 = rows[0], cols[0]
            print("   out ", out[r0, c0:c0 + 6].tolist())
            print("   ref ", ref[r0, c0:c0 + 6].float().tolist())
            print("   raw ", raw[r0, c0:c0 + 6].float().tolist())
            # is the wrong value the raw accumulator of some other row of this problem?
            v = out[r0, c0].double() - (b[c0].double() if bias else 0)
            hit = torch.nonzero((raw[:, c0] - v).abs() < 1e-3).flatten().tolist()
            print("   value equals raw[row, c0] for rows", hit[:8])
            if prev is not None and prev.shape[1] > c0:
                hitp = torch.nonzero((prev[:, c0].double() - out[r0, c0].double()).abs() < 1e-4).flatten().tolist()
                print("   value equals previous problem's out[row, c0] for rows", hitp[:8])
            # per-lane pattern: which (row, col) inside the 64x64 block are bad
            blk = bad[(r0 // 64) * 64:(r0 // 64) * 64 + 64, (c0 // 64) * 64:(c0 // 64) * 64 + 64]
            print("   bad rows in block", torch.nonzero(blk.any(1)).flatten().tolist(), "bad cols count per row", blk.sum(1)[blk.any(1)].tolist())
    prev = out
for M, N, K in [(128, 128, 256), (256, 128, 512), (1000, 384, 320), (4096, 256, 1024), (40000, 256, 256)]:
    for bias, res, relu in [(0, 0, 0), (1, 0, 1), (1, 1, 0)]:
        check(M, N, K, bias, res, relu)

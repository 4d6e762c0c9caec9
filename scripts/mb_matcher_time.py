"""time the matcher cost launch sequence at the c4 shapes (kernel experiments: S2D_HIP_LIB selects the library)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
NL, B, Q, T, hm, wm, H, W, N, P = 10, 2, 100, 8, 184, 320, 736, 1280, 10, 160000
dev = torch.device("cuda")
ml = torch.randn((NL, B, T * hm * wm, Q), device=dev)
cls = torch.randn((NL, B, Q, 2), device=dev)
tgt = (torch.rand((B, N, T, H, W), device=dev) > 0.7).to(torch.uint8)
cnt = torch.full((B,), N, dtype=torch.int32, device=dev)
res = {}
for mix in (("1",) if "once" in sys.argv else ("1", "0", "1", "0")):
    os.environ["S2D_MATCHER_MIX"] = mix               # read per call
    for _ in range(2):
        C = ops.matcher_cost(ml, cls, tgt, cnt, (Q, T, hm, wm), P, (0.0, 5.0, 5.0), seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        C = ops.matcher_cost(ml, cls, tgt, cnt, (Q, T, hm, wm), P, (0.0, 5.0, 5.0), seed=1)
    torch.cuda.synchronize()
    res[mix] = C
    print(os.environ.get("S2D_HIP_LIB", "default"), f"mix={mix}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms/call", flush=True)
if "0" in res:
    print("max |difference| of the cost matrices / max |cost|:", float((res["0"] - res["1"]).abs().max() / res["0"].abs().max()))

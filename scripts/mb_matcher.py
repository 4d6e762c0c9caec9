"""micro-benchmark: matcher_cost kernel at the metric's shapes, random vs spatially sorted points"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from s2d_amd import ops

NL, B, Q, T, hm, wm, H, W, N, P = 10, 2, 100, 8, 184, 320, 736, 1280, 10, 160000
dev = torch.device("cuda")
ml = torch.randn((NL, B, T * hm * wm, Q), device=dev)
cls = torch.randn((NL, B, Q, 2), device=dev)
tgt = (torch.rand((B, N, T, H, W), device=dev) > 0.7).to(torch.uint8)
cnt = torch.full((B,), N, dtype=torch.int32, device=dev)
rng = np.random.default_rng(0)
c = rng.random((NL, B, P, 2), dtype=np.float32)
key = (np.floor(c[..., 1] * hm).astype(np.int64) * wm + np.floor(c[..., 0] * wm).astype(np.int64))
cs = np.take_along_axis(c, np.argsort(key, axis=-1)[..., None], axis=2)


def run(coords, label):
    for _ in range(2):
        C = ops.matcher_cost(ml, cls, tgt, cnt, (Q, T, hm, wm), P, (0.0, 5.0, 5.0), coords=coords, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        C = ops.matcher_cost(ml, cls, tgt, cnt, (Q, T, hm, wm), P, (0.0, 5.0, 5.0), coords=coords, seed=1)
    torch.cuda.synchronize()
    print(f"{label:30s} {(time.perf_counter()-t0)/3*1e3:8.2f} ms", float(C.sum()))

run(None, "device RNG")
run(torch.from_numpy(c).to(dev), "random coords buffer")
run(torch.from_numpy(np.ascontiguousarray(cs)).to(dev), "sorted coords buffer")
z = np.zeros_like(c); z[:] = 0.5
run(torch.from_numpy(z).to(dev), "constant coords")

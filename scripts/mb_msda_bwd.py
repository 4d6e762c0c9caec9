"""MSDeformAttn backward at config c4 (16 frames, S = 19320, 8 heads x 32, 3 levels x 4 points): the atomic-free sorted form vs
the float-atomic scatter, HIP-event times of the whole call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from s2d_amd import ops

shapes = np.array([(23, 40), (46, 80), (92, 160)])
S = int((shapes[:, 0] * shapes[:, 1]).sum())
N, M, D, L, P = 16, 8, 32, 3, 4
g = torch.Generator(device="cuda").manual_seed(0)
value = torch.randn((N, S, M, D), generator=g, device="cuda")
refs = []
for (H, W) in shapes:
    yy, xx = torch.meshgrid(torch.arange(H, device="cuda") + 0.5, torch.arange(W, device="cuda") + 0.5, indexing="ij")
    refs.append(torch.stack([xx.reshape(-1) / W, yy.reshape(-1) / H], -1))
ref = torch.cat(refs, 0)
norm = torch.tensor([[w, h] for (h, w) in shapes], device="cuda", dtype=torch.float32)
off = torch.randn((N, S, M, L, P, 2), generator=g, device="cuda") * 3.0            # a few pixels, like the initialised offsets
loc = (ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]).contiguous()
aw = torch.softmax(torch.randn((N, S, M, L * P), generator=g, device="cuda"), -1).view(N, S, M, L, P).contiguous()
go = torch.randn((N, S, M * D), generator=g, device="cuda")
lsi = np.concatenate([[0], np.cumsum(shapes[:, 0] * shapes[:, 1])[:-1]])


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


ms_s = t(lambda: ops.msda_backward(value, shapes, lsi, loc, aw, go))
ms_a = t(lambda: ops.msda_backward(value, shapes, lsi, loc, aw, go, atomics=True))
a = ops.msda_backward(value, shapes, lsi, loc, aw, go)
b = ops.msda_backward(value, shapes, lsi, loc, aw, go, atomics=True)
print(f"msda backward c4: sorted {ms_s:.2f} ms, atomics {ms_a:.2f} ms; max |diff| / max: " +
      ", ".join(f"{float((x - y).abs().max() / y.abs().max()):.2e}" for x, y in zip(a, b)))
fwd = t(lambda: ops.msda_forward(value, shapes, lsi, loc, aw))
print(f"msda forward (drop-in form) {fwd:.2f} ms")

# the fused form the pixel decoder's backward calls (raw projection rows): the one-launch query half (round 5) vs loc kernel + chain
from s2d_amd import backward as Bk
oa = torch.cat([(torch.randn((N, S, M * L * P * 2), generator=g, device="cuda") * 3.0), torch.randn((N, S, M * L * P), generator=g, device="cuda")], -1).contiguous()
v3 = value.view(N, S, M * D)
for mode, waves in (("1", "4"), ("1", "8"), ("0", "4")):
    os.environ["S2D_MSDA_BWD_WAVES"] = waves
    Bk._MSDA_BWD_REC = mode != "0"
    ms = t(lambda: Bk.msda_fused_backward(v3, shapes, oa, go))
    print(f"fused backward, one-launch query half={mode != '0'} waves/SIMD={waves}: {ms:.2f} ms per call", flush=True)
os.environ.pop("S2D_MSDA_BWD_WAVES"); Bk._MSDA_BWD_REC = True

"""3x3 convolutions of the step (R50 res2-5 bottlenecks, FPN layer_1) on the halo kernel: fragment reads one MFMA group ahead
(S2D_CONV_HALO_PIPE=1, default) against the round-4 form (0), bitwise comparison, in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd import ops
dev = torch.device("cuda")
torch.manual_seed(0)
def t(fn, n=10):
    for _ in range(3): y = fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): y = fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n, y
for (n, H, W, Ci, Co) in [(16, 184, 320, 256, 256), (16, 92, 160, 128, 128), (16, 46, 80, 256, 256), (16, 23, 40, 512, 512), (16, 184, 320, 64, 64)]:
    x = torch.randn((n, H, W, Ci), device=dev)
    w = torch.nn.Parameter(torch.randn((Co, 3, 3, Ci), device=dev) / (9 * Ci) ** 0.5, requires_grad=False)
    b = torch.randn((Co,), device=dev)
    out = {}
    for mode in ("0", "1", "0", "1"):
        os.environ["S2D_CONV_HALO_PIPE"] = mode
        dt, y = t(lambda: ops.conv2d_nhwc(x, w, stride=1, pad=1, bias=b, relu=True))
        out.setdefault(mode, []).append(dt)
        if mode == "0": y0 = y
        else: same = bool(torch.equal(y, y0))
    fl = 2.0 * n * H * W * Ci * 9 * Co
    print(f"conv3x3 {n}x{H}x{W} {Ci}->{Co}: round-4 form {min(out['0']):.4f} ms ({fl / min(out['0']) / 1e9:.0f} TF) | fragments a group ahead {min(out['1']):.4f} ms "
          f"({fl / min(out['1']) / 1e9:.0f} TF) | bits equal {same}", flush=True)

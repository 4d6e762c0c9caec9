"""The three kernels BASELINE.json's north_star asks per-kernel numbers for, alone, at the metric's (c4) shapes:
  * msda_fused_kernel      -- the deformable-attention bilinear gather (HBM/L2-bound): 16 frames x 19 320 queries x 8 heads x 12 samples
  * mask-logit einsum      -- "bqc,btchw->bqthw" as the pixel-major GEMM [471 040 x 256] x [100 x 256]^T per clip (MFMA)
  * cross_attn_kernel      -- masked cross-attention QK^T / softmax / AV over the three memory levels (MFMA + VALU)
Run plainly for timings, or as the program of a rocprofv3 pass (kernel trace, or --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE,
or --pmc FETCH_SIZE, or --pmc WRITE_SIZE; counters and traces in separate passes); scripts/pmc_northstar.py joins the passes
into profiles/r3_pmc_northstar.json, which bench.py reports under roofline.per_kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from s2d_amd import ops
dev = torch.device("cuda")
REPS = int(os.environ.get("NS_REPS", "10"))
torch.manual_seed(0)


def timed(fn, reps=REPS):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


# ---- MSDeformAttn gather: offsets as the module initialises them (bias grid) + 0.5 px of learned variation
shapes = [(23, 40), (46, 80), (92, 160)]
S = sum(h * w for h, w in shapes); N = 16
both = torch.randn((N, S, 288 + 256), device=dev)
th = torch.arange(8, device=dev) * (2 * np.pi / 8)
g = torch.stack([th.cos(), th.sin()], -1); g = g / g.abs().max(-1, keepdim=True)[0]
bias = (g.view(8, 1, 1, 2) * torch.arange(1, 5, device=dev).view(1, 1, 4, 1)).expand(8, 3, 4, 2).reshape(-1)
both[..., :192] = bias + 0.5 * torch.randn((N, S, 192), device=dev)
value, oa = both[..., 288:], both[..., :288]
dt = timed(lambda: ops.msda_fused_forward(value, np.array(shapes), oa))
alg = 4.0 * N * (2 * S * 256 + S * 8 * 12 * 3)            # value read + output written + offsets/logits read, per launch
print(f"msda_fused: {dt*1e3:.3f} ms  algorithmic {alg/1e6:.1f} MB -> {alg/dt/1e12:.3f} TB/s", flush=True)

# ---- mask-logit einsum
B, T, hm, wm, Q, C = 2, 8, 184, 320, 100, 256
mf = torch.randn((B, T * hm * wm, C), device=dev)
e = torch.randn((B, Q, C), device=dev)
out = torch.empty((B, T * hm * wm, Q), device=dev)
dt = timed(lambda: ops.gemm_nt(mf, e, out=out))
fl = 2.0 * B * T * hm * wm * Q * C
print(f"mask einsum: {dt*1e3:.3f} ms  {fl/1e9:.1f} GFLOP -> {fl/dt/1e12:.1f} TFLOP/s algorithmic", flush=True)

# ---- masked cross-attention, three levels
for (hl, wl) in shapes:
    K = T * hl * wl
    q = torch.randn((B, Q, C), device=dev); k = torch.randn((B, K, C), device=dev); v = torch.randn((B, K, C), device=dev)
    bits = torch.randint(-2**31, 2**31 - 1, (B, K, 4), device=dev, dtype=torch.int32)
    unm = torch.full((B, 4), -1, device=dev, dtype=torch.int32)
    dt = timed(lambda: ops.masked_attn(q, k, v, bits, unm))
    fl = 4.0 * B * Q * K * C
    print(f"cross_attn K={K}: {dt*1e3:.3f} ms  {fl/1e9:.1f} GFLOP -> {fl/dt/1e12:.1f} TFLOP/s algorithmic, K+V {2*B*K*C*4/dt/1e12:.2f} TB/s", flush=True)

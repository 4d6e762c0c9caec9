"""Encoder FFN at the metric shape (309 120 rows x 256 -> 1024 -> 256): the one-launch form (csrc/ffn.hip) against the two GEMM
launches + LayerNorm it replaces.  python scripts/mb_ffn.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from s2d_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 309120
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((M, 256), device=dev, generator=g)
W1 = torch.nn.Parameter(torch.randn((1024, 256), device=dev, generator=g) * 0.06)
W2 = torch.nn.Parameter(torch.randn((256, 1024), device=dev, generator=g) * 0.03)
b1 = torch.randn((1024,), device=dev, generator=g) * 0.1
b2 = torch.randn((256,), device=dev, generator=g) * 0.1
g1, be1, g2, be2 = (torch.randn((256,), device=dev, generator=g) * 0.1 + (1 if i % 2 == 0 else 0) for i in range(4))


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def two_launch(p):
    d2, d3 = ((p, 7, 1), (p, 7, 2)) if p > 0 else (None, None)
    s1 = ops.layernorm(x, g1, be1)
    h = ops.gemm_nt(s1, W1, bias=b1, relu=True, dropout=d2)
    return ops.layernorm(ops.gemm_nt(h, W2, bias=b2, res=s1, dropout=d3), g2, be2)


flops = 4.0 * M * 1024 * 256
S = M // 16 if M % 16 == 0 else M
Wp = ops.mark_static(torch.randn((544, 256), device=dev, generator=g) * 0.05)
bp = torch.randn((544,), device=dev, generator=g) * 0.1
bp[:288] = 0          # (the pos term carries the bias of the offsets / logits columns)
pos = torch.randn((S, 288), device=dev, generator=g) * 0.5
for p in (0.3, 0.0):
    drop = (p, 7, 1, 2) if p > 0 else None
    ref = two_launch(p)
    out = ops.ffn_fused(x, W1, b1, W2, b2, ln1=(g1, be1), ln2=(g2, be2), dropout=drop)
    err = float((out - ref).abs().max() / ref.abs().max())
    t2 = timeit(lambda: two_launch(p))
    tf = timeit(lambda: ops.ffn_fused(x, W1, b1, W2, b2, ln1=(g1, be1), ln2=(g2, be2), dropout=drop))
    s1 = ops.layernorm(x, g1, be1)
    tf2 = timeit(lambda: ops.ffn_fused(s1, W1, b1, W2, b2, ln2=(g2, be2), dropout=drop))
    tf0 = timeit(lambda: ops.ffn_fused(s1, W1, b1, W2, b2, dropout=drop))
    tp = timeit(lambda: ops.gemm_nt(ref, Wp, bias=bp, res=pos, res_rows=S, res_cols=288))
    tfp = timeit(lambda: ops.ffn_fused(x, W1, b1, W2, b2, ln1=(g1, be1), ln2=(g2, be2), dropout=drop, post=(Wp, bp, pos)))
    yp, outp = ops.ffn_fused(x, W1, b1, W2, b2, ln1=(g1, be1), ln2=(g2, be2), dropout=drop, post=(Wp, bp, pos))
    refp = ops.gemm_nt(yp, Wp, bias=bp, res=pos, res_rows=S, res_cols=288)
    print(f"p={p}: next layer's projection as its own launch {tp:.3f} ms | fused LN1+FFN+LN2+projection {tfp:.3f} ms (vs {tf:.3f} + {tp:.3f} = "
          f"{tf + tp:.3f}) | max rel diff of the projection {float((outp - refp).abs().max() / refp.abs().max()):.2e}", flush=True)
    print(f"p={p}: two launches + 2 LN {t2:.3f} ms | fused LN1+FFN+LN2 {tf:.3f} ms ({flops / tf / 1e9:.0f} TFLOP/s alg.) | "
          f"fused FFN+LN2 {tf2:.3f} | fused FFN {tf0:.3f} | max rel diff {err:.2e}", flush=True)
    # the attention's output projection + dropout1 + residual in front
    Wo = torch.nn.Parameter(torch.randn((256, 256), device=dev, generator=g) * 0.07)
    bo = torch.randn((256,), device=dev, generator=g) * 0.1
    samp = torch.randn((M, 256), device=dev, generator=g)
    d1 = (p, 7, 0) if p > 0 else None
    to = timeit(lambda: ops.gemm_nt(samp, Wo, bias=bo, res=x, dropout=d1))
    x1 = ops.gemm_nt(samp, Wo, bias=bo, res=x, dropout=d1)
    refq = ops.ffn_fused(x1, W1, b1, W2, b2, ln1=(g1, be1), ln2=(g2, be2), dropout=drop, post=(Wp, bp, pos))
    tq = timeit(lambda: ops.ffn_fused(samp, W1, b1, W2, b2, ln1=(g1, be1), ln2=(g2, be2), dropout=drop, pre=(Wo, bo, x, 0)))
    tqp = timeit(lambda: ops.ffn_fused(samp, W1, b1, W2, b2, ln1=(g1, be1), ln2=(g2, be2), dropout=drop, post=(Wp, bp, pos), pre=(Wo, bo, x, 0)))
    q = ops.ffn_fused(samp, W1, b1, W2, b2, ln1=(g1, be1), ln2=(g2, be2), dropout=drop, post=(Wp, bp, pos), pre=(Wo, bo, x, 0))
    print(f"p={p}: output_proj as its own launch {to:.3f} ms | out_proj+LN1+FFN+LN2 {tq:.3f} ms (vs {to + tf:.3f}) | with the projection too {tqp:.3f} ms "
          f"(vs {to + tfp:.3f}) | max rel diff {float((q[0] - refq[0]).abs().max() / refq[0].abs().max()):.2e} / "
          f"{float((q[-1] - refq[1]).abs().max() / refq[1].abs().max()):.2e}", flush=True)

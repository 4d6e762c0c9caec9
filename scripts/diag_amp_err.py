"""AMP-mode end-to-end case: per-layer error statistics of the device logits against the oracle with fp16-rounded operands"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle.oracle_np as oracle
from tests.parity import run_case
hip, ref = run_case(oracle, seed=int(os.environ.get("SEED", "3")), B=2, T=2, H0=60, W0=90, Q=16, P=256, ns=(3, 4), amp=True)
for k in ("s_logits", "s_masks"):
    a, b = hip[k].astype(np.float64), ref[k].astype(np.float64)
    sc = np.abs(b).max()
    print(k, "scale", sc)
    for l in range(a.shape[0]):
        d = np.abs(a[l] - b[l]) / sc
        print(f"  layer {l}: max {d.max():.2e}  99% {np.percentile(d, 99):.2e}  median {np.median(d):.2e}  frac > 1e-3: {np.mean(d > 1e-3):.4f}")
print("kd_counts", hip["kd_counts"], ref["kd_counts"])
for k, v in ref["losses"].items():
    print(f"  {k:22s} {hip['losses'][k]:.6f} {float(v):.6f} rel {abs(hip['losses'][k]-float(v))/max(abs(float(v)),1e-9):.2e}")

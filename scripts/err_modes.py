import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle_np
from tests.parity import run_case
from s2d_amd import ops
ref = None
for mode in ("f32", "bf16x3", "f16x3"):
    ops.set_dense_mode(mode)
    hip, r = run_case(oracle_np if ref is None else None, seed=4, B=1, T=1, H0=256, W0=256, Q=10, P=1024, ns=(3,))
    ref = ref or r
    b = ref["s_masks"].astype(np.float64)
    e = np.abs(hip["s_masks"] - b)
    print(mode, "mask logits: max|err|/max|ref| = %.2e, rel-L2 = %.2e, per-layer max:" % (e.max() / np.abs(b).max(), np.linalg.norm(e) / np.linalg.norm(b)),
          ["%.1e" % (e[i].max() / np.abs(b[i]).max()) for i in range(10)])
    le = max(abs(hip["losses"][k] - float(v)) / max(abs(float(v)), 1e-6) for k, v in ref["losses"].items())
    print(mode, "max rel loss err %.2e" % le)

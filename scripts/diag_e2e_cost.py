"""config-2 KD pass: device cost matrix vs float64 evaluation on the device's own inputs, per layer (diagnostic; GPU)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_np as O          # noqa: E402
from tests.parity import run_case           # noqa: E402
from tests.test_gpu_e2e import _cost64      # noqa: E402

hip, _ = run_case(None, seed=5, B=1, T=2, H0=480, W0=854, Q=100, P=12544, ns=(10,), kd_want=10)
model = hip["model"]
Cdev = model.criterion.matcher.last_cost.cpu().numpy()
NL, B = 10, 1
for layer in range(NL):
    Co, scale = _cost64(O, hip["s_logits"][layer][0], hip["s_masks"][layer][0], hip["kd_targets"][0], hip["coords_kd"]["matcher"][layer, 0][None],
                 *hip["matcher_weights"])
    C32 = O.matcher_cost(hip["s_logits"][layer][0], hip["s_masks"][layer][0], hip["kd_targets"][0], hip["coords_kd"]["matcher"][layer, 0][None],
                         *hip["matcher_weights"]).astype(np.float64)
    Cd = Cdev[layer][:, :Co.shape[1]].astype(np.float64)
    e = np.abs(Cd - Co)
    q, n = np.unravel_index(e.argmax(), e.shape)
    print(f"layer {layer}: |x|max {np.abs(hip['s_masks'][layer]).max():9.2f} max|C| {np.abs(Co).max():8.3f} dev-vs-f64 {e.max()/np.abs(Co).max():.3e} (q {q}, n {n}; col-mean err {e.mean(0).round(7)}) "
          f"fp32-oracle-vs-f64 {np.abs(C32 - Co).max()/np.abs(Co).max():.3e}")
print("kd target pixel sums", [int(t.sum()) for t in hip["kd_targets"][0]])

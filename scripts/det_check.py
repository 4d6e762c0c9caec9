"""run-to-run determinism probe at the c4 shapes under the different stream schedules"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from s2d_amd import ops
from s2d_amd.modeling import TargetSet, build_kd_model
dev = torch.device("cuda:0")
B, T, H0, W0, Q, P, N = bench.CONFIGS["c4"]
model = build_kd_model(num_queries=Q, num_frames=T, num_points=P).to(dev)
frames, masks = bench.synth_batch(0, B, T, H0, W0, N, dev)
bench.calibrate_teacher(model, ops.normalize_pad(frames))

def run():
    model.criterion.seed = 0; model.criterion.matcher.seed = 0
    losses = model.forward_losses(ops.normalize_pad(frames), TargetSet.from_list(masks, device=dev))
    torch.cuda.synchronize()
    t = model.last["teacher"]; s = model.last["student"]
    return ({k: float(v) for k, v in losses.items()}, t.mask_logits[-1].clone(), s.mask_logits[-1].clone(),
            [x.clone() for x in model.criterion.last_indices])

for ot, oc in [(False, False), (True, False), (True, True)]:
    model.overlap_teacher, model.overlap_criteria = ot, oc
    run()
    for rep in range(3):
        a = run(); b = run()
        diff = [k for k in a[0] if a[0][k] != b[0][k]]
        print(f"teacher_overlap={ot} criteria_overlap={oc} rep{rep}: loss keys differing {len(diff)} {diff[:4]}; teacher logits equal "
              f"{bool((a[1] == b[1]).all())}; student logits equal {bool((a[2] == b[2]).all())}; "
              f"indices equal {all(bool((x == y).all()) for x, y in zip(a[3], b[3]))}", flush=True)

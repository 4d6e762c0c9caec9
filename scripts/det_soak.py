"""soak: two-stream forward_losses repeated N times against the single-stream result (bitwise)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from s2d_amd import ops
from s2d_amd.modeling import TargetSet, build_kd_model
dev = torch.device("cuda:0")
B, T, H0, W0, Q, P, N = bench.CONFIGS["c4"]
model = build_kd_model(num_queries=Q, num_frames=T, num_points=P).to(dev)
frames, masks = bench.synth_batch(0, B, T, H0, W0, N, dev)
bench.calibrate_teacher(model, ops.normalize_pad(frames))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40

def run():
    model.criterion.seed = 0; model.criterion.matcher.seed = 0
    losses = model.forward_losses(ops.normalize_pad(frames), TargetSet.from_list(masks, device=dev))
    torch.cuda.synchronize()
    return ({k: float(v) for k, v in losses.items()}, model.last["teacher"].mask_logits[-1].clone(),
            model.last["student"].mask_logits.clone())

model.overlap_teacher = model.overlap_criteria = False
ref = run()
assert run()[0] == ref[0]
model.overlap_teacher = model.overlap_criteria = True
bad = 0
for rep in range(reps):
    cur = run()
    d = [k for k in ref[0] if ref[0][k] != cur[0][k]]
    t_ok, s_ok = torch.equal(ref[1], cur[1]), torch.equal(ref[2], cur[2])
    if d or not t_ok or not s_ok:
        bad += 1
        print(f"rep {rep}: {len(d)} loss keys differ {d[:3]}; teacher logits equal {t_ok}; student logits equal {s_ok}", flush=True)
print(f"two-stream soak: {bad} of {reps} runs differ from the single-stream result", flush=True)

"""where the two-stream step's time goes: HIP events at the joins of KDVideoMaskFormer.forward_losses (student forward on the main stream, teacher
forward + pseudo targets on the side stream, then GT criterion on the side stream beside the KD criterion on the main stream), c4, median over steps"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from s2d_amd import ops
from s2d_amd.modeling import TargetSet, build_kd_model
dev = torch.device("cuda", 0)
B, T, H0, W0, Q, P, N = bench.CONFIGS["c4"]
model = build_kd_model(num_queries=Q, num_frames=T, num_points=P, weights=(0.0, 5.0, 5.0), kd_weights=(0.0, 5.0, 5.0), dropout=0.3).to(dev)
model.train()
frames, masks = bench.synth_batch(0, B, T, H0, W0, N, dev)
bench.calibrate_teacher(model, ops.normalize_pad(frames))
mean, std = model.pixel_mean.flatten().cpu().numpy(), model.pixel_std.flatten().cpu().numpy()
main = torch.cuda.current_stream()
side = torch.cuda.Stream(device=dev)
ev = lambda: torch.cuda.Event(enable_timing=True)
rows = []
for it in range(12):
    e = {k: ev() for k in ("t0", "img", "s_done", "t_done", "g_done", "k_done", "end")}
    e["t0"].record(main)
    images = ops.normalize_pad(frames, 32, mean, std)
    gt = TargetSet.from_list(masks, device=dev)
    Hp, Wp = images.shape[1:3]
    e["img"].record(main)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        teacher = model.teacher(images, True, aux_masks=False)
        tgt, cnt, kept, ne = ops.kd_targets(teacher.class_logits[-1], teacher.mask_logits[-1], teacher.dims, Hp, Wp, Q, model.score_threshold_distillation,
                                            model.num_predictions_distillation)
        e["t_done"].record(side)
    student = model.student(images, True)
    e["s_done"].record(main)
    side.wait_stream(main); main.wait_stream(side)
    with torch.cuda.stream(side):
        lg = model.criterion(student, gt, False, None)
        e["g_done"].record(side)
    lk = model.criterion(student, TargetSet(tgt, cnt, ne), True, None)
    e["k_done"].record(main)
    main.wait_stream(side)
    e["end"].record(main)
    torch.cuda.synchronize()
    if it >= 3:
        rows.append([e["t0"].elapsed_time(e[k]) for k in ("img", "s_done", "t_done", "g_done", "k_done", "end")])
r = np.median(np.array(rows), 0)
print(f"ms from step start (median of {len(rows)}): inputs ready {r[0]:.2f} | student forward done {r[1]:.2f} | teacher forward + pseudo targets done {r[2]:.2f} | "
      f"GT criterion done {r[3]:.2f} | KD criterion done {r[4]:.2f} | step end {r[5]:.2f}", flush=True)
print(f"  => side stream idle before the criteria {max(r[1] - r[2], 0):.2f} ms, main idle {max(r[2] - r[1], 0):.2f} ms; criteria phase {r[5] - max(r[1], r[2]):.2f} ms", flush=True)

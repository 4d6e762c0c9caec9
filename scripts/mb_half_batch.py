"""forward (student + teacher, two streams) and the two criteria for ONE clip against the batch of two at c4: what a per-clip pipeline (clip 0's criteria
hidden under clip 1's forward) would have to pay in kernel efficiency"""
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
main = torch.cuda.current_stream(); side = torch.cuda.Stream(device=dev)
def fwd(fr, two):
    images = ops.normalize_pad(fr, 32, mean, std)
    Hp, Wp = images.shape[1:3]
    s_ = side if two else main
    s_.wait_stream(main)
    with torch.cuda.stream(s_):
        teacher = model.teacher(images, True, aux_masks=False)
        kd = ops.kd_targets(teacher.class_logits[-1], teacher.mask_logits[-1], teacher.dims, Hp, Wp, Q, model.score_threshold_distillation, model.num_predictions_distillation)
    student = model.student(images, True)
    main.wait_stream(s_)
    return student, kd
def crit(student, kd, ms, two):
    tgt, cnt, kept, ne = kd
    gt = TargetSet.from_list(ms, device=dev)
    s_ = side if two else main
    s_.wait_stream(main)
    with torch.cuda.stream(s_):
        model.criterion(student, gt, False, None)
    model.criterion(student, TargetSet(tgt, cnt, ne), True, None)
    main.wait_stream(s_)
def timed(fn, n=8, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for two in (False, True):
    for nb in (2, 1):
        fr = frames[: nb * T]; ms = masks[:nb]
        tf = timed(lambda: fwd(fr, two))
        st = fwd(fr, two); torch.cuda.synchronize()
        tc = timed(lambda: crit(st[0], st[1], ms, two))
        print(f"{'two streams' if two else 'one stream '}  {nb} clip(s): forward (student + teacher + pseudo targets) {tf:.2f} ms | both criteria {tc:.2f} ms", flush=True)

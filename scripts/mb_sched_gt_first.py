"""forward + loss at c4 with the GT criterion queued right behind the student's forward on the main stream (it needs nothing from the teacher) and
the KD criterion behind both, against the shipped schedule (GT criterion on the side stream behind the teacher, KD criterion on the main stream),
under stream priorities that let the student finish first"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
model.overlap_teacher = model.overlap_criteria = True


def shipped():
    images = ops.normalize_pad(frames, 32, mean, std)
    return sum(model.forward_losses(images, TargetSet.from_list(masks, device=dev)).values())


def gt_first(side):
    images = ops.normalize_pad(frames, 32, mean, std)
    gt = TargetSet.from_list(masks, device=dev)
    Hp, Wp = images.shape[1:3]
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        teacher = model.teacher(images, True, aux_masks=False)
        tgt, cnt, kept, ne = ops.kd_targets(teacher.class_logits[-1], teacher.mask_logits[-1], teacher.dims, Hp, Wp, Q,
                                            model.score_threshold_distillation, model.num_predictions_distillation)
    student = model.student(images, True)
    lg = model.criterion(student, gt, False, None)
    main.wait_stream(side)
    lk = model.criterion(student, TargetSet(tgt, cnt, ne), True, None)
    return sum(lg.values()) + sum(lk.values())


def timed(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for name, main_p, side_p in (("default", 0, 0), ("main high", -1, 0), ("side low", 0, 1), ("default", 0, 0)):
    side = torch.cuda.Stream(device=dev, priority=side_p)
    model._side = side
    ms = torch.cuda.Stream(device=dev, priority=main_p)
    ms.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(ms):
        a = timed(shipped)
        b = timed(lambda: gt_first(side))
    torch.cuda.current_stream().wait_stream(ms)
    print(f"{name:10s} (main {main_p:2d}, side {side_p:2d}): shipped schedule {a:.2f} ms | GT criterion behind the student, KD behind both {b:.2f} ms", flush=True)

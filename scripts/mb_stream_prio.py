"""two-stream schedule of forward_losses at c4 under different HIP stream priorities: the student (main stream) is the longer chain (it computes
all ten heads' full mask maps), the teacher (side stream) the shorter one -- does favouring either shorten the step?"""
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
print("priority range (least, greatest):", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a", flush=True)
def step():
    images = ops.normalize_pad(frames, 32, mean, std)
    return sum(model.forward_losses(images, TargetSet.from_list(masks, device=dev)).values())
def timed(n=10, w=3):
    for _ in range(w): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
model.overlap_teacher = model.overlap_criteria = True
for name, main_p, side_p in (("default (0, 0)", 0, 0), ("side low", 0, 1), ("main high", -1, 0), ("side high", 0, -1), ("default (0, 0)", 0, 0), ("main high, side low", -1, 1)):
    try:
        model._side = torch.cuda.Stream(device=dev, priority=side_p)
        ms = torch.cuda.Stream(device=dev, priority=main_p)
        ms.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ms):
            t = timed()
        torch.cuda.current_stream().wait_stream(ms)
        print(f"{name:22s} main priority {main_p:2d}, side priority {side_p:2d}: {t:.3f} ms per step", flush=True)
    except Exception as e:
        print(name, "failed:", type(e).__name__, str(e)[:200], flush=True)

"""One full training iteration on one GPU at the metric's config c4 (2 clips x T=8 x 720p, Q=100, P=160000): forward + loss
(student + teacher, GT + KD), backward of the student through the HIP gradient kernels, full-model clip + AdamW + EMA teacher
update.  BASELINE config 4 ("additionally times backward + all-reduce + optimizer + EMA"); the all-reduce is
the identity at one rank."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from s2d_amd import ops
from s2d_amd.modeling import TargetSet, build_kd_model
from s2d_amd.optim import FullModelGradientClippingAdamW, param_groups_like_reference
dev = torch.device("cuda")
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
B, T, H0, W0, Q, P, N = bench.CONFIGS[cfg]
model = build_kd_model(num_queries=Q, num_frames=T, num_points=P, weights=(0.0, 5.0, 5.0), kd_weights=(0.0, 5.0, 5.0)).to(dev)
model.train()
model.overlap_teacher = model.overlap_criteria = os.environ.get('ONE_STREAM') is None
frames, masks = bench.synth_batch(0, B, T, H0, W0, N, dev)
bench.calibrate_teacher(model, ops.normalize_pad(frames))
groups = param_groups_like_reference(model.student, 1e-4, 0.05)
teach = dict(zip((id(p) for p in model.student.parameters()), model.teacher.parameters()))
opt = FullModelGradientClippingAdamW(groups, lr=1e-4, clip_norm=0.01, ema_params=[teach[id(g["params"][0])] for g in groups])
mean, std = model.pixel_mean.flatten().cpu().numpy(), model.pixel_std.flatten().cpu().numpy()


def it():
    images = ops.normalize_pad(frames, 32, mean, std)
    targets = TargetSet.from_list(masks, device=dev)
    opt.zero_grad()
    # gradients accumulate into the optimizer's arena views: forward_backward adds into p.grad
    losses = model.forward_backward(images, targets)
    f = opt.allreduce_grads()
    opt.step(inv_scale=f, ema_momentum=0.999)
    return sum(losses.values())


for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tot = it()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"iteration {i}: {dt*1e3:.1f} ms  loss {float(tot):.4f}  grad norm {opt.grad_norm():.4f}  found_inf {opt.found_inf()}  "
          f"peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
t0 = time.perf_counter(); model.forward_losses(ops.normalize_pad(frames, 32, mean, std), TargetSet.from_list(masks, device=dev)); torch.cuda.synchronize()
print(f"forward + loss alone (one stream): {(time.perf_counter()-t0)*1e3:.1f} ms")

# phase timing of one more iteration (synchronising wrappers: adds the sync cost, shows where the backward goes)
import types
from s2d_amd import ops as _ops
phases = {}
def timed(obj, name, label):
    fn = getattr(obj, name)
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); phases[label] = phases.get(label, 0.0) + (time.perf_counter() - t0) * 1e3
        return r
    setattr(obj, name, w)
head = model.student[1]
timed(model.student[0], "backward", "trunk backward"); timed(head.pixel_decoder, "backward_features", "pixel decoder backward")
timed(head.predictor, "backward", "decoder backward"); timed(_ops, "point_loss_backward", "point loss backward (2 passes)")
timed(model.criterion, "forward", "criteria (2 passes, forward)"); timed(model.teacher, "forward", "teacher forward")
timed(model.student[0], "forward", "trunk forward"); timed(head.pixel_decoder, "forward_features", "pixel decoder forward"); timed(head.predictor, "forward", "decoder forward")
it(); torch.cuda.synchronize()
print("phases (ms): " + ", ".join(f"{k} {v:.1f}" for k, v in phases.items()))

"""One training iteration at c4 (one stream) with MARKER kernels between its phases, for `rocprofv3 --kernel-trace`:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o tr -- python3 scripts/mb_train_trace.py
    python scripts/trace_segments.py DIR/.../tr_kernel_trace.csv OUT.txt
A marker is a float64 `cos_` on one element (a kernel name nothing else in the process produces); trace_segments.py cuts the trace at the
markers of the LAST iteration and prints per segment the kernel totals.  Segment labels are printed here in marker order."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from s2d_amd import ops
from s2d_amd.modeling import TargetSet, build_kd_model
from s2d_amd.optim import FullModelGradientClippingAdamW, param_groups_like_reference
dev = torch.device("cuda")
B, T, H0, W0, Q, P, N = bench.CONFIGS["c4"]
model = build_kd_model(num_queries=Q, num_frames=T, num_points=P, weights=(0.0, 5.0, 5.0), kd_weights=(0.0, 5.0, 5.0)).to(dev)
model.train()
model.overlap_teacher = model.overlap_criteria = False
frames, masks = bench.synth_batch(0, B, T, H0, W0, N, dev)
bench.calibrate_teacher(model, ops.normalize_pad(frames))
groups = param_groups_like_reference(model.student, 1e-4, 0.05)
teach = dict(zip((id(p) for p in model.student.parameters()), model.teacher.parameters()))
opt = FullModelGradientClippingAdamW(groups, lr=1e-4, clip_norm=0.01, ema_params=[teach[id(g["params"][0])] for g in groups])
mean, std = model.pixel_mean.flatten().cpu().numpy(), model.pixel_std.flatten().cpu().numpy()
mk = torch.zeros(1, device=dev, dtype=torch.float64)
labels = []
LIVE = [False]


def mark(label):
    if LIVE[0]:
        mk.cos_()
        labels.append(label)


def wrap(obj, name, label):
    fn = getattr(obj, name)
    def w(*a, **k):
        mark(label)
        r = fn(*a, **k)
        mark("after " + label)
        return r
    setattr(obj, name, w)


head = model.student[1]
wrap(model.teacher, "forward", "teacher forward")
wrap(model.student[0], "forward", "trunk forward"); wrap(head.pixel_decoder, "forward_features", "pixel decoder forward")
wrap(head.predictor, "forward", "decoder forward"); wrap(model.criterion, "forward", "criterion forward")
wrap(ops, "point_loss_backward", "point loss backward")
wrap(head.predictor, "backward", "decoder backward"); wrap(head.pixel_decoder, "backward_features", "pixel decoder backward")
wrap(model.student[0], "backward", "trunk backward")
# finer cuts inside the backward walks
from s2d_amd.modeling import backbone as BBm, pixel_decoder as PDm
for name, *_ in BBm.R50_STAGES:
    for i, blk in enumerate(getattr(model.student[0], name)):
        wrap(blk, "backward", f"trunk {name}.{i} backward")
wrap(model.student[0].stem, "backward", "trunk stem backward")
enc = head.pixel_decoder.transformer.encoder if hasattr(head.pixel_decoder, "transformer") else None
if enc is not None:
    for i, lyr in enumerate(enc.layers):
        wrap(lyr, "backward", f"encoder layer {i} backward")


def it():
    images = ops.normalize_pad(frames, 32, mean, std)
    targets = TargetSet.from_list(masks, device=dev)
    opt.zero_grad()
    losses = model.forward_backward(images, targets)
    mark("all-reduce + optimizer")
    f = opt.allreduce_grads()
    opt.step(inv_scale=f, ema_momentum=0.999)
    mark("end")
    return sum(losses.values())


for i in range(3):
    it()
torch.cuda.synchronize()
LIVE[0] = True
mk.sin_()                                    # start marker of the traced iteration
t0 = time.perf_counter(); it(); torch.cuda.synchronize()
print(f"traced iteration: {(time.perf_counter() - t0) * 1e3:.1f} ms")
print("LABELS\t" + "\t".join(labels))

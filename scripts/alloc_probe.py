import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import bench
from s2d_amd import ops
from s2d_amd.modeling import TargetSet, build_kd_model
from s2d_amd.optim import FullModelGradientClippingAdamW, param_groups_like_reference
dev = torch.device("cuda")
B, T, H0, W0, Q, P, N = bench.CONFIGS["c4"]
model = build_kd_model(num_queries=Q, num_frames=T, num_points=P, dropout=0.3).to(dev); model.train()
frames, masks = bench.synth_batch(0, B, T, H0, W0, N, dev)
bench.calibrate_teacher(model, ops.normalize_pad(frames))
groups = param_groups_like_reference(model.student, 1e-4, 0.05)
teach = dict(zip((id(p) for p in model.student.parameters()), model.teacher.parameters()))
opt = FullModelGradientClippingAdamW(groups, lr=1e-4, clip_norm=0.01, ema_params=[teach[id(g["params"][0])] for g in groups])
mean, std = model.pixel_mean.flatten().cpu().numpy(), model.pixel_std.flatten().cpu().numpy()
prev = torch.cuda.memory_stats()
for i in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    images = ops.normalize_pad(frames, 32, mean, std); targets = TargetSet.from_list(masks, device=dev)
    opt.zero_grad(); out = model.forward_backward(images, targets); inv = opt.allreduce_grads(); opt.step(inv_scale=inv, ema_momentum=0.999)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = torch.cuda.memory_stats()
    print(f"it {i}: {dt*1e3:.1f} ms  device_alloc +{st['num_device_alloc']-prev['num_device_alloc']} free +{st['num_device_free']-prev['num_device_free']} retries +{st['num_alloc_retries']-prev['num_alloc_retries']} "
          f"reserved {st['reserved_bytes.all.current']/2**30:.1f} GiB allocated peak {st['allocated_bytes.all.peak']/2**30:.1f} inactive_split {st['inactive_split_bytes.all.current']/2**30:.1f}", flush=True)
    prev = st

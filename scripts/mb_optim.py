"""optimizer + EMA step (csrc/optim.hip) on the real student/teacher parameter set: time per step and achieved HBM
bandwidth against the 36 B/parameter (+4 B for the norm pass) the step has to move; torch's own AdamW + clip + EMA loop
on the same GPU beside it"""
import sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2d_amd.modeling import build_kd_model
from s2d_amd.optim import FullModelGradientClippingAdamW, param_groups_like_reference
dev = torch.device("cuda")
model = build_kd_model().to(dev)
groups = param_groups_like_reference(model.student, 1e-4, 0.05)
teach = dict(zip((id(p) for p in model.student.parameters()), model.teacher.parameters()))
opt = FullModelGradientClippingAdamW(groups, lr=1e-4, clip_norm=0.01, ema_params=[teach[id(g["params"][0])] for g in groups])
n = sum(p.numel() for p in opt._params)
opt.grad_arena.normal_()


def t(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k


dt = t(lambda: opt.step(inv_scale=1.0 / 1024, ema_momentum=0.999))
print(f"{len(opt._params)} tensors, {n/1e6:.1f} M parameters, {opt._nchunks} chunks")
print(f"norm + AdamW + EMA: {dt*1e3:.3f} ms/step = {n*40/dt/1e12:.2f} TB/s of 8 (40 B/param algorithmic)", flush=True)
dt2 = t(lambda: opt.step(ema_momentum=0.999, check_inf=False) if opt.clip_norm == 0 else None) if False else None
# torch's path on the same parameters (separate tensors so the two do not fight over .grad)
ps = [torch.nn.Parameter(p.detach().clone()) for p in opt._params]
ts = [p.detach().clone() for p in ps]
for p in ps: p.grad = torch.randn_like(p)
topt = torch.optim.AdamW([{"params": [p]} for p in ps], 1e-4)


def torch_step():
    torch.nn.utils.clip_grad_norm_(itertools.chain(*[x["params"] for x in topt.param_groups]), 0.01)
    topt.step()
    with torch.no_grad():
        for p, tt in zip(ps, ts):
            tt.data.mul_(0.999).add_((1 - 0.999) * p.detach().data)


print(f"torch.optim.AdamW (foreach) + clip_grad_norm_ + python EMA loop on the same GPU: {t(torch_step, 5)*1e3:.2f} ms/step", flush=True)

# split: device time (events around the two launches) vs host time per call
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
t0 = time.perf_counter(); opt.step(inv_scale=1.0 / 1024, ema_momentum=0.999); host = time.perf_counter() - t0
torch.cuda.synchronize()
e0.record(); opt.step(inv_scale=1.0 / 1024, ema_momentum=0.999); e1.record(); torch.cuda.synchronize()
print(f"host time of one step() call {host*1e3:.3f} ms; device time between events {e0.elapsed_time(e1):.3f} ms")
e0.record()
for _ in range(20): opt.step(inv_scale=1.0 / 1024, ema_momentum=0.999)
e1.record(); torch.cuda.synchronize()
dev_ms = e0.elapsed_time(e1) / 20
print(f"20 steps back to back: {dev_ms:.3f} ms/step on the device timeline = {n*40/dev_ms/1e9:.2f} TB/s = {n*40/dev_ms/1e9/8:.2f} of the 8 TB/s HBM peak")

"""GPU parity of the optimizer + EMA step (csrc/optim.hip) against the reference's own dependency:
torch.optim.AdamW + torch.nn.utils.clip_grad_norm_ (train_net_video.py:188-213) and the EMA loop of
engine/train_loop.py:754-764, run on CPU tensors with the same values."""
import itertools

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = [(1,), (3,), (257,), (64, 3, 7, 7), (65536 + 5,), (300, 701), (2, 65536), (100, 256)]


def make(seed, shapes=SHAPES):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g) * 0.1 for s in shapes]


def reference_step(params, teacher, grads, opt, clip, inv_scale, m):
    """grad_scaler.step(optimizer) with FullModelGradientClippingOptimizer + the EMA update, as the reference runs them"""
    for p, g in zip(params, grads):
        p.grad = None if g is None else (g * inv_scale)                 # GradScaler.unscale_
    found_inf = any(p.grad is not None and not torch.isfinite(p.grad).all() for p in params)
    if not found_inf:                                                   # GradScaler.step skips on inf/nan
        if clip > 0:
            allp = itertools.chain(*[x["params"] for x in opt.param_groups])
            torch.nn.utils.clip_grad_norm_(allp, clip)                  # train_net_video.py:198-199
        opt.step()
    if m is not None:
        with torch.no_grad():
            for p, t in zip(params, teacher):
                t.data.mul_(m).add_((1 - m) * p.detach().data)          # train_loop.py:762-763
    return found_inf


@pytest.mark.parametrize("clip,inv_scale,with_none", [(0.01, 1.0, False), (0.0, 1.0, False), (5.0, 1.0 / 1024, True)])
def test_adamw_clip_ema_vs_torch(clip, inv_scale, with_none):
    from s2d_amd.optim import FullModelGradientClippingAdamW
    init = make(1)
    ref_p = [torch.nn.Parameter(x.clone()) for x in init]
    ref_t = [x.clone() + 0.01 for x in init]
    groups = [{"params": [p], "lr": 1e-4 * (1 + (i % 3)), "weight_decay": [0.05, 0.0, 0.01][i % 3]} for i, p in enumerate(ref_p)]
    ref_opt = torch.optim.AdamW(groups, 1e-4)
    hip_p = [torch.nn.Parameter(x.clone().to(DEV)) for x in init]
    hip_t = [(x.clone() + 0.01).to(DEV) for x in init]
    hgroups = [{"params": [p], "lr": g["lr"], "weight_decay": g["weight_decay"]} for p, g in zip(hip_p, groups)]
    opt = FullModelGradientClippingAdamW(hgroups, lr=1e-4, clip_norm=clip, ema_params=hip_t)
    for step in range(4):
        grads = make(10 + step)
        grads = [g * (1.0 / inv_scale) for g in grads]                  # "scaled" gradients as backward would leave them
        if with_none:
            grads[2] = None                                             # a parameter that received no gradient
        m = 0.999 if step != 1 else None
        # lr schedule change on the way, as a scheduler would do through param_groups
        for gg, hg in zip(ref_opt.param_groups, opt.param_groups):
            gg["lr"] = hg["lr"] = gg["lr"] * 0.9
        reference_step(ref_p, ref_t, grads, ref_opt, clip, inv_scale, m)
        opt.zero_grad()
        for p, g in zip(hip_p, grads):
            if g is not None:
                p.grad.copy_(g.to(DEV))
        if with_none:
            # torch skips parameters whose grad is None (no decay, no moment update): emulate by a null grad pointer
            opt._t_ptrs[5 * 2 + 1] = 0
        opt.step(inv_scale=inv_scale, ema_momentum=m)
        if clip > 0:
            tot = torch.sqrt(sum((g.double() * inv_scale).pow(2).sum() for g in grads if g is not None))
            np.testing.assert_allclose(opt.grad_norm(), float(tot), rtol=1e-6)
        assert not opt.found_inf() or clip == 0
        for i, (a, b) in enumerate(zip(hip_p, ref_p)):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=2e-6, atol=1e-8, err_msg=f"param {i} step {step}")
        for i, (a, b) in enumerate(zip(hip_t, ref_t)):
            np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=2e-6, atol=1e-8, err_msg=f"teacher {i} step {step}")
    sd, rsd = opt.state_dict(), ref_opt.state_dict()
    for i in rsd["state"]:
        if with_none and i == 2:
            continue
        # moments: a few ulp of the largest operand (m + w (g - m) cancels), i.e. absolute in the tensor's scale
        for key in ("exp_avg", "exp_avg_sq"):
            want = rsd["state"][i][key].numpy()
            np.testing.assert_allclose(sd["state"][i][key].cpu().numpy(), want, rtol=2e-6, atol=4e-7 * np.abs(want).max())
        assert float(sd["state"][i]["step"]) == float(rsd["state"][i]["step"])


def test_inf_gradient_skips_step_but_not_ema():
    """GradScaler semantics (train_loop.py:722-723): an inf/nan gradient skips optimizer.step; the EMA still runs"""
    from s2d_amd.optim import FullModelGradientClippingAdamW
    init = make(2, [(1000,), (70000,)])
    hip_p = [torch.nn.Parameter(x.clone().to(DEV)) for x in init]
    hip_t = [torch.zeros_like(x).to(DEV) for x in init]
    opt = FullModelGradientClippingAdamW(hip_p, lr=1e-3, clip_norm=1.0, ema_params=hip_t)
    opt.zero_grad()
    hip_p[0].grad.normal_()
    hip_p[1].grad.normal_()
    hip_p[1].grad[65536 + 17] = float("inf")
    opt.step(inv_scale=0.5, ema_momentum=0.9)
    assert opt.found_inf()
    for p, x, t in zip(hip_p, init, hip_t):
        np.testing.assert_array_equal(p.detach().cpu().numpy(), x.numpy())                    # untouched
        np.testing.assert_allclose(t.cpu().numpy(), (0.1 * x).numpy(), rtol=1e-6, atol=1e-9)  # 0.9 * 0 + 0.1 * p
    assert float(opt.exp_avg.abs().max()) == 0.0
    hip_p[1].grad[65536 + 17] = float("nan")
    opt.step(inv_scale=0.5)
    assert opt.found_inf()
    hip_p[1].grad[65536 + 17] = 0.0
    opt.step(inv_scale=0.5)
    assert not opt.found_inf()
    assert float((hip_p[0].detach().cpu() - init[0]).abs().max()) > 0


def test_build_optimizer_on_model_and_state_dict_roundtrip():
    from types import SimpleNamespace as NS
    from s2d_amd.modeling import build_kd_model
    from s2d_amd.optim import build_optimizer
    model = build_kd_model(num_queries=8, num_frames=2, num_points=64, dec_layers=3).to(DEV)
    cfg = NS(SOLVER=NS(OPTIMIZER="ADAMW", BASE_LR=1e-4, WEIGHT_DECAY=0.05, WEIGHT_DECAY_NORM=0.0, WEIGHT_DECAY_EMBED=0.0,
                       BACKBONE_MULTIPLIER=0.1, CLIP_GRADIENTS=NS(ENABLED=True, CLIP_TYPE="full_model", CLIP_VALUE=0.01)))
    opt = build_optimizer(cfg, model)
    n_train = sum(1 for p in model.parameters() if p.requires_grad)
    assert len(opt.param_groups) == n_train and opt.clip_norm == 0.01
    before_t = [t.detach().clone() for t in model.teacher.parameters()]
    before_s = [p.detach().clone() for p in model.student.parameters()]
    opt.zero_grad()
    opt.grad_arena.normal_()
    opt.step(ema_momentum=0.5)
    torch.cuda.synchronize()
    changed = sum(float((a - b.detach()).abs().max()) > 0 for a, b in zip(before_s, model.student.parameters()))
    assert changed == len(before_s)
    for t0, t1, s1 in zip(before_t, model.teacher.parameters(), model.student.parameters()):
        np.testing.assert_allclose(t1.detach().cpu().numpy(), (0.5 * t0 + 0.5 * s1.detach()).cpu().numpy(), rtol=1e-6, atol=1e-8)
    sd = opt.state_dict()
    opt2 = build_optimizer(cfg, model)
    opt2.load_state_dict(sd)
    assert opt2._step == 1 and torch.equal(opt2.exp_avg, opt.exp_avg)

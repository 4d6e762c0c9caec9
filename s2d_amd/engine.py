"""The device side of `Trainer.run_step` (model_training/mask2former_video/engine/train_loop.py:690-770) for the meta-archs of
s2d_amd.modeling: forward + loss + backward, gradient exchange, optimizer step, EMA teacher update -- the three reference
statements `loss_dict = self.model(data)`, `self.grad_scaler.scale(losses).backward()`, `self.grad_scaler.step(self.optimizer)`
and the EMA loop, with the data loading, copy-paste, logging and scheduling left to the trainer."""
import math
import os

from .modeling.criterion import TargetSet
from .modeling.meta_arch import _gt_target_list


def run_step(model, optimizer, data, iteration=0, ema_momentum=None):
    """one optimizer step on `data` (the mapper's list of dicts): returns the weighted loss dict (0-dim device tensors).
    Gradient accumulation (SOLVER.ACCUM_ITER > 1, :730-746): gradients are scaled by 1 / accum_iter and the optimizer (and the
    EMA, :761) only steps on every accum_iter-th call."""
    images = model.preprocess(data) if hasattr(model, "preprocess") else None
    if images is None:
        raise TypeError("run_step needs a meta-architecture of s2d_amd.modeling")
    Hp, Wp = images.shape[1:3]
    targets = TargetSet.from_list(_gt_target_list(data, model.num_frames, Hp, Wp, model.device), device=model.device)
    accum = max(int(getattr(model, "accum_iter", 1)), 1)
    if iteration % accum == 0:
        optimizer.zero_grad()
    last = (iteration + 1) % accum == 0
    if last and hasattr(model, "student") and os.environ.get("S2D_OVERLAP_ALLREDUCE", "0") == "1":
        # opt-in: the exchange starts part by part while the backward still runs (DDP's bucketed overlap, engine/defaults.py:76-85);
        # under accumulation only the last micro-step exchanges, as DDP's no_sync() iterations do.  The default below is ONE all-reduce
        # of the arena behind the backward: the overlapped form has only ever run over gloo (no multi-GPU node was available to
        # this pipeline; tests/test_gpu_multi.py runs it over RCCL wherever two GPUs are visible).
        from .optim import OverlappedAllReduce, student_parts
        ex = getattr(optimizer, "_exchange", None)
        if ex is None:
            ex = optimizer._exchange = OverlappedAllReduce(optimizer, student_parts(model))
        losses = model.forward_backward(images, targets, loss_scale=1.0 / accum, grad_ready=ex.ready)
        optimizer.step(inv_scale=ex.finish(), ema_momentum=ema_momentum)
        return losses
    losses = model.forward_backward(images, targets, loss_scale=1.0 / accum)
    if last:
        optimizer.step(inv_scale=optimizer.allreduce_grads(), ema_momentum=ema_momentum)
    return losses

"""Optimizer + EMA step of the reference's training loop on two HIP launches (csrc/optim.hip).

Mirrors what the reference builds and calls per iteration:
  * `Trainer.build_optimizer` (model_training/train_net_video.py:134-215): one param group per parameter with
    lr / weight-decay overrides, `torch.optim.AdamW` wrapped in `FullModelGradientClippingOptimizer`
    (`clip_grad_norm_` over ALL parameters before `step`);
  * `grad_scaler.step(optimizer)` (engine/train_loop.py:709-726): unscale, skip the step on inf/nan gradients;
  * the EMA teacher update (engine/train_loop.py:754-764).

`FullModelGradientClippingAdamW` keeps torch's `param_groups` / `state_dict()` layout so the trainer's LR scheduler and
checkpointer see what they expect; parameters stay where torch allocated them.  Gradients live in one flat arena
(`p.grad` are views), which is also what the data-parallel all-reduce runs on (`allreduce_grads`): a few large RCCL
all-reduces sized for the xGMI rings instead of one per parameter.  There is no CPU fallback."""
import math

import torch

from . import ops
from ._lib import lib

CHUNK = 16384          # elements per workgroup (256 threads x 16 float4): ~2900 workgroups for the 44 M-parameter student


def _align4(n):
    return (n + 3) // 4 * 4


class FullModelGradientClippingAdamW:
    """torch.optim.AdamW semantics (single-tensor formula, amsgrad=False, maximize=False) + full-model gradient clipping
    + optional EMA of `ema_params` towards the parameters, in that order, per `step()`."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, clip_norm=0.0, ema_params=None):
        params = list(params)
        if params and not isinstance(params[0], dict):
            params = [{"params": params}]
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        self.param_groups = []
        for g in params:
            g = dict(g)
            g["params"] = list(g["params"])
            for k, v in self.defaults.items():
                g.setdefault(k, v)
            g.setdefault("initial_lr", g["lr"])
            self.param_groups.append(g)
        b0, e0 = self.param_groups[0]["betas"], self.param_groups[0]["eps"]
        if any(g["betas"] != b0 or g["eps"] != e0 for g in self.param_groups):
            raise NotImplementedError("per-group betas / eps (the reference sets only lr and weight_decay per group)")
        self.clip_norm = float(clip_norm)
        self._params = [p for g in self.param_groups for p in g["params"]]
        if not self._params:
            raise ValueError("optimizer got an empty parameter list")
        dev = self._params[0].device
        for p in self._params:
            ops._chk(p.data)
        self.device = dev
        self._step = 0
        sizes = [p.numel() for p in self._params]
        offs, tot = [], 0
        for n in sizes:
            offs.append(tot)
            tot += _align4(n)
        self._offs, self._total = offs, tot
        # flat arenas: gradients (p.grad become views), first and second moments
        self.grad_arena = torch.zeros((tot,), device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros((tot,), device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros((tot,), device=dev, dtype=torch.float32)
        for p, o, n in zip(self._params, offs, sizes):
            p.grad = self.grad_arena[o:o + n].view_as(p)
        self._ema = list(ema_params) if ema_params is not None else None
        if self._ema is not None:
            if len(self._ema) != len(self._params) or any(e.shape != p.shape for e, p in zip(self._ema, self._params)):
                raise ValueError("ema_params must pair one-to-one with the optimized parameters (student/teacher zip, "
                                 "engine/train_loop.py:762)")
        self._build_tables()
        self.normbuf = torch.zeros((4,), device=dev, dtype=torch.float32)      # total norm, clip coefficient, found_inf

    # ------------------------------------------------------------------ tables
    def _build_tables(self):
        dev = self.device
        ptrs, numel, ct, co = [], [], [], []
        for i, p in enumerate(self._params):
            o, n = self._offs[i], p.numel()
            ptrs += [p.data_ptr(), self.grad_arena.data_ptr() + 4 * o, self.exp_avg.data_ptr() + 4 * o,
                     self.exp_avg_sq.data_ptr() + 4 * o, self._ema[i].data_ptr() if self._ema is not None else 0]
            numel.append(n)
            for off in range(0, n, CHUNK):
                ct.append(i)
                co.append(off)
        self._ptr_list = ptrs
        self._t_ptrs = torch.tensor(ptrs, dtype=torch.int64).to(dev)
        self._t_numel = torch.tensor(numel, dtype=torch.int64).to(dev)
        self._t_ct = torch.tensor(ct, dtype=torch.int32).to(dev)
        self._t_co = torch.tensor(co, dtype=torch.int64).to(dev)
        self._nchunks = len(ct)
        self._partial = torch.zeros((max(self._nchunks, 1),), device=dev, dtype=torch.float64)
        self._hyper_host = None
        self._t_hyper = torch.zeros((len(self._params), 2), device=dev, dtype=torch.float64)
        self._sync_hyper()

    def _sync_hyper(self):
        """per-tensor (lr, weight_decay) -> device, only when a scheduler (or the trainer) changed them"""
        h = [(g["lr"], g["weight_decay"]) for g in self.param_groups for _ in g["params"]]
        if h != self._hyper_host:
            self._hyper_host = h
            self._t_hyper.copy_(torch.tensor(h, dtype=torch.float64).view(-1, 2), non_blocking=False)

    def _check_live(self):
        """the tables hold raw pointers: parameters / gradients must not have been re-allocated behind our back"""
        for i, p in enumerate(self._params):
            if p.data_ptr() != self._ptr_list[5 * i]:
                raise RuntimeError("a parameter was re-allocated after the optimizer was built (use in-place updates)")
            if p.grad is None or p.grad.data_ptr() != self._ptr_list[5 * i + 1]:
                raise RuntimeError("p.grad no longer aliases the gradient arena: use zero_grad() of this optimizer "
                                   "(set_to_none is not supported)")

    # ------------------------------------------------------------------ torch.optim.Optimizer surface
    def zero_grad(self, set_to_none=False):
        if set_to_none:
            raise NotImplementedError("gradients live in a fixed arena; zero_grad() fills it with zeros")
        self.grad_arena.zero_()

    @torch.no_grad()
    def step(self, inv_scale=1.0, ema_momentum=None, check_inf=None):
        """one optimizer step.  inv_scale: 1 / GradScaler scale (times 1 / world size if the arena holds a SUM over
        ranks); ema_momentum: m of the EMA update, None = no EMA this step; check_inf: evaluate the inf/nan flag and
        skip the update on the device if set (default: whenever clipping is on or inv_scale != 1).
        Returns nothing and never synchronises; `found_inf()` / `grad_norm()` read the device flags when asked."""
        if self._step % 64 == 0:          # 2 x data_ptr() per tensor costs more host time than the step takes on the device
            self._check_live()
        self._sync_hyper()
        self._step += 1
        g0 = self.param_groups[0]
        beta1, beta2 = g0["betas"]
        bc1 = 1 - beta1 ** self._step
        bc2_sqrt = (1 - beta2 ** self._step) ** 0.5
        need_norm = self.clip_norm > 0 or (check_inf if check_inf is not None else inv_scale != 1.0)
        st = ops._stream()
        if need_norm:
            lib().call("s2d_optim_grad_norm_f32", self._t_ptrs, self._t_numel, self._t_ct, self._t_co, self._nchunks, CHUNK,
                       float(inv_scale), float(self.clip_norm), self._partial, self.normbuf, st)
        ema = -1.0 if (ema_momentum is None or self._ema is None) else float(ema_momentum)
        lib().call("s2d_optim_adamw_ema_f32", self._t_ptrs, self._t_numel, self._t_hyper, self._t_ct, self._t_co, self._nchunks,
                   CHUNK, 1.0, float(beta1), float(beta2), float(g0["eps"]), float(bc1), float(bc2_sqrt), float(inv_scale), ema,
                   self.normbuf if need_norm else None, st)
        # the kernel wrote the parameters (and the EMA copies) through raw pointers: torch's version counters did not move, so tell the
        # library's weight caches (packed / pre-split / transposed copies keyed by ops.version_of) that their sources changed
        for p in self._params:
            ops.bump_version(p)
        if ema >= 0.0:
            for e in self._ema:
                ops.bump_version(e)

    def grad_norm(self):
        return float(self.normbuf[0])

    def found_inf(self):
        return bool(self.normbuf[2] != 0)

    def state_dict(self):
        """torch.optim layout: {'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...index lists...]}"""
        state, groups, i = {}, [], 0
        for g in self.param_groups:
            idx = []
            for p in g["params"]:
                o, n = self._offs[i], p.numel()
                state[i] = {"step": torch.tensor(float(self._step)), "exp_avg": self.exp_avg[o:o + n].view_as(p).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + n].view_as(p).clone()}
                idx.append(i)
                i += 1
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": idx})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        for i, p in enumerate(self._params):
            s = sd["state"].get(i)
            if s is None:
                continue
            o, n = self._offs[i], p.numel()
            self.exp_avg[o:o + n].copy_(s["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(s["exp_avg_sq"].reshape(-1))
            self._step = int(s["step"])
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            for k, v in sg.items():
                if k != "params":
                    g[k] = v

    # ------------------------------------------------------------------ data-parallel exchange
    def allreduce_grads(self, bucket_bytes=256 << 20):
        """SUM-all-reduce the gradient arena over the ranks in a few large buckets (RCCL ring over xGMI is per-link
        bandwidth-bound: large messages, few launches); returns the factor 1 / world to fold into `step(inv_scale=...)`
        so the mean costs no extra pass."""
        return allreduce_flat(self.grad_arena, bucket_bytes)


class OverlappedAllReduce:
    """The gradient exchange started bucket by bucket WHILE the backward still runs, as the reference's DistributedDataParallel does
    (engine/defaults.py:76-85: bucketed all-reduce fired from autograd hooks inside `losses.backward()`, train_loop.py:719).

    The explicit backward of `forward_backward` finishes the student's parts in a fixed order -- predictor, pixel decoder, trunk --
    and reports each (`grad_ready=self.ready`).  A part's parameters are a contiguous range of the gradient arena (parameters are
    laid out in `model.parameters()` order: trunk | pixel decoder | predictor), so each report becomes one asynchronous SUM
    all-reduce of that range: RCCL runs it on its own stream behind the kernels already enqueued, beside the backward of the parts
    still to come.  `finish()` makes the compute stream wait for all of them and returns 1 / world for `step(inv_scale=...)`.
    Backends that cannot work on device memory asynchronously (the gloo rehearsal) reduce the whole arena in `finish()`: same
    result, no overlap."""

    def __init__(self, optimizer, parts):
        """parts: {name: iterable of parameters}; every optimized parameter must belong to exactly one part"""
        import torch.distributed as dist
        self.opt = optimizer
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.async_ok = self.active and (dist.get_backend() != "gloo" or not optimizer.grad_arena.is_cuda)     # gloo: host tensors only
        index = {id(q): i for i, q in enumerate(optimizer._params)}
        self.ranges, seen = {}, 0
        for name, ps in parts.items():
            idx = sorted(index[id(q)] for q in ps if id(q) in index)
            if not idx:
                continue
            if idx != list(range(idx[0], idx[-1] + 1)):
                raise ValueError(f"the parameters of part {name!r} are not contiguous in the gradient arena")
            lo = optimizer._offs[idx[0]]
            hi = optimizer._offs[idx[-1] + 1] if idx[-1] + 1 < len(optimizer._offs) else optimizer._total
            self.ranges[name] = (lo, hi)
            seen += len(idx)
        if seen != len(optimizer._params):
            raise ValueError("the parts do not cover the optimized parameters exactly once")
        self.works, self.done = [], set()

    def ready(self, name):
        """every gradient of part `name` has been written (enqueued on the current stream)"""
        if name in self.done or name not in self.ranges:
            return
        self.done.add(name)
        if self.async_ok:
            import torch.distributed as dist
            lo, hi = self.ranges[name]
            self.works.append(dist.all_reduce(self.opt.grad_arena[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        import torch.distributed as dist
        if not self.active:
            return 1.0
        if not self.async_ok:
            self.done.clear()
            return allreduce_flat(self.opt.grad_arena)
        for name in self.ranges:            # parts nobody reported (a caller without the hook): reduce them now
            self.ready(name)
        for w in self.works:
            w.wait()
        self.works, self.done = [], set()
        return 1.0 / dist.get_world_size()


def student_parts(model):
    """the three parts of a KD / plain meta-architecture's student in the order its explicit backward finishes them"""
    student = model.student if hasattr(model, "student") else None
    if student is None:
        return {"predictor": list(model.sem_seg_head.predictor.parameters()), "pixel_decoder": list(model.sem_seg_head.pixel_decoder.parameters()),
                "backbone": list(model.backbone.parameters())}
    return {"predictor": list(student[1].predictor.parameters()), "pixel_decoder": list(student[1].pixel_decoder.parameters()),
            "backbone": list(student[0].parameters())}


def allreduce_flat(flat, bucket_bytes=256 << 20):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 1.0
    if flat.is_cuda and dist.get_backend() == "gloo":
        # rehearsal of the N > 1 control flow without RCCL (several ranks sharing one GPU, or CPU tests): stage through the host
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
        return 1.0 / dist.get_world_size()
    per = max(bucket_bytes // flat.element_size(), 1)
    works = [dist.all_reduce(flat[o:o + per], op=dist.ReduceOp.SUM, async_op=True) for o in range(0, flat.numel(), per)]
    for w in works:
        w.wait()
    return 1.0 / dist.get_world_size()


NORM_MODULE_TYPES = (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d, torch.nn.SyncBatchNorm, torch.nn.GroupNorm,
                     torch.nn.InstanceNorm1d, torch.nn.InstanceNorm2d, torch.nn.InstanceNorm3d, torch.nn.LayerNorm,
                     torch.nn.LocalResponseNorm)


def param_groups_like_reference(model, base_lr, weight_decay, weight_decay_norm=0.0, weight_decay_embed=0.0,
                                backbone_multiplier=0.1, base_lr_multiplier_names=(), base_lr_multiplier=1.0):
    """the per-parameter groups of Trainer.build_optimizer (train_net_video.py:134-186), literally: one group per trainable
    parameter; lr x BACKBONE_MULTIPLIER when 'backbone' occurs in the MODULE name; weight decay WEIGHT_DECAY_NORM for
    parameters owned by normalisation modules, WEIGHT_DECAY_EMBED for nn.Embedding; lr x BASE_LR_MULTIPLIER for modules
    listed by name."""
    groups, memo = [], set()
    for module_name, module in model.named_modules():
        for pname, value in module.named_parameters(recurse=False):
            if not value.requires_grad or value in memo:
                continue
            memo.add(value)
            hp = {"lr": base_lr, "weight_decay": weight_decay}
            if "backbone" in module_name:
                hp["lr"] = hp["lr"] * backbone_multiplier
            if "relative_position_bias_table" in pname or "absolute_pos_embed" in pname:
                hp["weight_decay"] = 0.0
            if isinstance(module, NORM_MODULE_TYPES):
                hp["weight_decay"] = weight_decay_norm
            if isinstance(module, torch.nn.Embedding):
                hp["weight_decay"] = weight_decay_embed
            if module_name in base_lr_multiplier_names:
                hp["lr"] *= base_lr_multiplier
            groups.append({"params": [value], **hp})
    return groups


def build_optimizer(cfg, model):
    """Trainer.build_optimizer (train_net_video.py:134-215) for SOLVER.OPTIMIZER == 'ADAMW'; the EMA pairs are the
    (student, teacher) parameters when the model has both (engine/train_loop.py:754-764)."""
    s = cfg.SOLVER
    if s.OPTIMIZER != "ADAMW":
        raise NotImplementedError(f"no optimizer type {s.OPTIMIZER} on the device path")
    groups = param_groups_like_reference(model, s.BASE_LR, s.WEIGHT_DECAY, s.WEIGHT_DECAY_NORM, s.WEIGHT_DECAY_EMBED,
                                         s.BACKBONE_MULTIPLIER, tuple(getattr(s, "BASE_LR_MULTIPLIER_NAMES", ())),
                                         getattr(s, "BASE_LR_MULTIPLIER", 1.0))
    cg = s.CLIP_GRADIENTS
    clip = cg.CLIP_VALUE if (cg.ENABLED and cg.CLIP_TYPE == "full_model" and cg.CLIP_VALUE > 0.0) else 0.0
    ema = None
    if hasattr(model, "student") and getattr(model, "teacher", None) is not None:
        opt_ids = {id(p) for g in groups for p in g["params"]}
        pairs = [(p, t) for p, t in zip(model.student.parameters(), model.teacher.parameters()) if id(p) in opt_ids]
        order = {id(p): t for p, t in pairs}
        ema = [order[id(p)] for g in groups for p in g["params"]] if len(order) == len(opt_ids) else None
    return FullModelGradientClippingAdamW(groups, lr=s.BASE_LR, clip_norm=clip, ema_params=ema)


def ema_momentum_schedule(it, m_start, m_end, m_end_iter, accum_iter=1):
    """engine/train_loop.py:767-769: cosine ramp of the EMA momentum"""
    return m_end - (m_end - m_start) * (math.cos(math.pi * (it * accum_iter) / (m_end_iter * accum_iter)) + 1) / 2

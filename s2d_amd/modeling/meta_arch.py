"""Meta-architectures, host side: KDVideoMaskFormer (model_training/mask2former_video/kd_video_maskformer_model.py:29-610)
and VideoMaskFormer (video_maskformer_model.py:24-378), plus MaskFormerHead (mask2former/modeling/meta_arch/
mask_former_head.py:18-132).  Same registry names, same `forward(batched_inputs) -> dict[str, 0-dim tensor]` contract
in training (keys of kd_video_maskformer_model.py:314-326), same attributes the trainer touches (.student, .teacher =
nn.Sequential(backbone, head) so checkpoint keys are student.0.* / student.1.*; .criterion.weight_dict; .accum_iter;
.device).  The forward is the measured hot path: normalise/pad -> student -> teacher -> GT criterion -> KD targets ->
KD criterion -> rename/weight, all enqueued on the current stream without a host synchronisation.
"""
import torch
from torch import nn

from .. import ops
from .backbone import ResNet50
from .criterion import TargetSet, VideoHungarianMatcher, VideoSetCriterion
from .pixel_decoder import MSDeformAttnPixelDecoder
from .postprocess import inference_video
from .video_decoder import VideoMultiScaleMaskedTransformerDecoder

try:  # plug into detectron2's registries when it is installed (the drop-in boundary)
    from detectron2.modeling import META_ARCH_REGISTRY, SEM_SEG_HEADS_REGISTRY  # type: ignore
except Exception:  # detectron2 absent (as in this image): local registries with the same interface
    class _Registry(dict):
        def register(self, obj=None):
            if obj is None:
                return lambda o: self.register(o)
            self[obj.__name__] = obj
            return obj

        def get(self, name):
            return self[name]

    META_ARCH_REGISTRY, SEM_SEG_HEADS_REGISTRY = _Registry(), _Registry()

SEM_SEG_HEADS_REGISTRY.register(MSDeformAttnPixelDecoder)


@SEM_SEG_HEADS_REGISTRY.register()
class MaskFormerHead(nn.Module):
    def __init__(self, pixel_decoder, transformer_predictor, num_classes=1):
        super().__init__()
        self.pixel_decoder, self.predictor, self.num_classes = pixel_decoder, transformer_predictor, num_classes

    @classmethod
    def from_config(cls, cfg, input_shape=None):  # mask_former_head.py:87-113
        if cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES != 1:
            raise NotImplementedError("the class-loss and matcher kernels implement the class-agnostic S2D path "
                                      "(SEM_SEG_HEAD.NUM_CLASSES: 1 in every shipped config; labels forced to 0, matcher.py:238-243)")
        return cls(MSDeformAttnPixelDecoder.from_config(cfg), VideoMultiScaleMaskedTransformerDecoder.from_config(
            cfg, cfg.MODEL.SEM_SEG_HEAD.CONVS_DIM, True), cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES)

    def forward(self, features, training=True, aux_masks=True):  # layers(), mask_former_head.py:118-132 ("multi_scale_pixel_decoder")
        mask_features, multi_scale = self.pixel_decoder.forward_features(features)
        return self.predictor(multi_scale, mask_features, training, aux_masks)


class _Net(nn.Sequential):
    """nn.Sequential(backbone, head): keeps the reference's state_dict keys (kd_video_maskformer_model.py:94-95)."""

    def forward(self, x, training=True, aux_masks=True):
        return self[1](self[0](x), training, aux_masks)


def set_amp_compute(model, enabled=True):
    """Opt-in AMP compute (SOLVER.AMP.ENABLED, engine/train_loop.py:709): the modules torch.autocast runs in fp16 in the reference --
    the R50 trunk, the video decoder's linear layers and the mask-logit einsum -- take single-pass fp16 MFMA arithmetic (operands
    rounded to fp16, f32 accumulation) in the forward / loss path; the pixel decoder and the criterion stay fp32-class, as the
    reference forces them to (msdeformattn.py:314, matcher.py:266-268).  Default off: the library computes in fp32-class
    arithmetic whatever the autocast state.  Returns the model."""
    for m in model.modules():
        if isinstance(m, (ResNet50, VideoMultiScaleMaskedTransformerDecoder)):
            m.amp = bool(enabled)
    return model


def _test_kwargs(mf, npred_name, eval_student=False):
    """MODEL.MASK_FORMER.TEST.* keys of the eval branch (kd_video_maskformer_model.py:153, 224-230;
    video_maskformer_model.py:112, 179-181); absent TEST node -> the constructors' defaults."""
    t = getattr(mf, "TEST", None)
    if t is None:
        return {}
    kw = {"use_nms": t.USE_NMS, "nms_threshold": t.NMS_THRESH, npred_name: t.NUM_PREDICTIONS}
    if eval_student:
        kw["eval_student"] = t.EVAL_STUDENT
    return kw


class _HipGradBridge(torch.autograd.Function):
    """Makes the HIP backward visible to autograd, so that the reference trainer's own statements work unchanged
    (engine/train_loop.py:709-726: `loss_dict = self.model(data)`, `losses = sum(loss_dict.values())`,
    `self.grad_scaler.scale(losses).backward()`), including DistributedDataParallel's gradient hooks
    (engine/defaults.py:76-85), which hang on the parameters' AccumulateGrad nodes.

    forward: identity on the stacked weighted losses; the student parameters are inputs only so that the graph reaches
    them.  backward: returns, per parameter, the staged d(sum of losses)/d(parameter) the HIP kernels computed during the
    forward call, times the incoming d(total)/d(loss).  The trainer differentiates a plain (scaled) sum, so every loss
    receives the same upstream factor; anything else is refused rather than answered wrongly."""

    @staticmethod
    def forward(ctx, losses, holder, *params):
        ctx.holder = holder
        return losses.clone()

    @staticmethod
    def backward(ctx, g):
        staged = ctx.holder.pop("grads", None)
        if staged is None:
            raise RuntimeError("the HIP gradients of this forward call were already consumed (backward twice, or a newer forward)")
        g0 = g[0]
        if not bool((g == g0).all()):
            raise NotImplementedError("the losses of the HIP training forward must be reduced with equal weights (the trainer "
                                      "sums them, train_loop.py:715); re-weight through criterion.weight_dict instead")
        return (None, None) + tuple(gr * g0 for gr in staged)


def _bridge_losses(model, params, run):
    """run() = one forward_backward leaving d(sum of weighted losses)/dp in p.grad: stage those gradients (whatever the
    caller had accumulated in .grad stays untouched) and return the loss dict as autograd-visible 0-dim tensors"""
    saved = [p.grad for p in params]
    for p in params:
        p.grad = None
    try:
        out = run()
        staged = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    finally:
        for p, g in zip(params, saved):
            p.grad = g
    keys = list(out)
    holder = {"grads": staged}
    model._grad_holder = holder                    # a newer forward drops the previous call's staged gradients
    stacked = _HipGradBridge.apply(torch.stack([out[k] for k in keys]), holder, *params)
    return {k: stacked[i] for i, k in enumerate(keys)}


def _frames_to_device(batched_inputs, device):
    frames = [f for video in batched_inputs for f in video["image"]]
    x = torch.stack([f if isinstance(f, torch.Tensor) else torch.as_tensor(f) for f in frames])
    return x.to(device=device, dtype=torch.uint8, non_blocking=True).contiguous()


def _gt_target_list(batched_inputs, num_frames, Hp, Wp, device):
    """prepare_targets (kd_video_maskformer_model.py:358-386): paste per-frame BitMasks into [N,T,Hp,Wp], drop
    instances whose ids are -1 in every frame.  Accepts detectron2 Instances (gt_masks.tensor / gt_ids) or plain
    dicts {'gt_masks': [N,h,w], 'gt_ids': [N]}."""
    out = []
    for video in batched_inputs:
        inst = video["instances"]
        get = (lambda o, k: o[k]) if isinstance(inst[0], dict) else (lambda o, k: getattr(o, k))
        n = len(get(inst[0], "gt_ids"))
        masks = torch.zeros((n, num_frames, Hp, Wp), dtype=torch.uint8, device=device)
        ids = []
        for t, fr in enumerate(inst):
            m = get(fr, "gt_masks")
            m = m.tensor if hasattr(m, "tensor") else torch.as_tensor(m)
            h, w = m.shape[-2:]
            masks[:, t, :h, :w] = (m != 0).to(device=device, dtype=torch.uint8)
            ids.append(torch.as_tensor(get(fr, "gt_ids"))[:, None])
        valid = (torch.cat(ids, 1) != -1).any(-1)
        out.append(masks[valid.to(device)])
    return out


@META_ARCH_REGISTRY.register()
class KDVideoMaskFormer(nn.Module):
    def __init__(self, *, student_backbone, student_sem_seg_head, teacher_backbone, teacher_sem_seg_head, criterion,
                 num_queries, num_frames, size_divisibility=32, pixel_mean=ops.PIXEL_MEAN, pixel_std=ops.PIXEL_STD,
                 num_predictions_distillation=100, score_threshold_distillation=0.75, accum_iter=1, eval_student=False,
                 use_nms=False, nms_threshold=0.75, num_predictions_inference=10, distillation_nms=False):
        super().__init__()
        self.student = _Net(student_backbone, student_sem_seg_head)
        self.teacher = _Net(teacher_backbone, teacher_sem_seg_head)
        for p in self.teacher.parameters():
            p.requires_grad = False
        self.criterion = criterion
        self.num_queries, self.num_frames, self.size_divisibility = num_queries, num_frames, size_divisibility
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean, dtype=torch.float32).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std, dtype=torch.float32).view(-1, 1, 1), False)
        self.num_predictions_distillation = num_predictions_distillation
        self.score_threshold_distillation = score_threshold_distillation
        self.accum_iter, self.eval_student = accum_iter, eval_student
        self.use_nms, self.nms_threshold, self.num_predictions_inference = use_nms, nms_threshold, num_predictions_inference
        self.distillation_nms = distillation_nms          # MODEL.MASK_FORMER.DISTILLATION_NMS (off in every shipped config)
        # Optional two-stream schedule of forward_losses (teacher forward / GT criterion on a second HIP stream): ~8 % faster
        # and bitwise identical to the one-stream schedule (bench.py re-checks that on every run; DESIGN.md section 5,
        # "Streams", has the history of why the default here stays one stream).
        self.overlap_criteria = False
        # Round 5: clips are independent units, and the two criteria (VALU / LDS-bound point sampling) contend with EACH OTHER on two streams
        # (measured: 19.3 ms side by side = the sum of the two alone) but not with the matrix-bound forwards.  pipeline_clips walks the batch clip by
        # clip: clip b's criteria run on a third stream beside clip b + 1's forwards.  Same arithmetic per clip; the batch-wide normalisers
        # (num_masks, the class loss's weight sum) are applied when the clips' losses are combined.  Off by default (the batch form is what the
        # parity tests with injected points drive); bench.py turns it on for the timed schedules.
        self.kd_compact = True           # forward_backward: pseudo-target planes cut to the count found (one small read-back)
        self.pipeline_clips = False
        self._crit_stream = None
        # The teacher's intermediate mask predictions feed only its own attention masks (no loss reads them): by default
        # they are evaluated at the attention masks' source pixels only; True computes the full maps like the reference.
        self.teacher_aux_masks = False
        self.overlap_teacher, self._side = False, None
        self._side_delay_cycles = 0                 # tests only: spin the side stream this many cycles before the GT criterion

    @classmethod
    def from_config(cls, cfg):  # kd_video_maskformer_model.py:130-231
        mf = cfg.MODEL.MASK_FORMER
        sb, tb = ResNet50(), ResNet50()
        sh, th = MaskFormerHead.from_config(cfg), MaskFormerHead.from_config(cfg)
        cw, dw, mw = mf.CLASS_WEIGHT, mf.DICE_WEIGHT, mf.MASK_WEIGHT
        kc, km, kd = mf.KD_CLASS_WEIGHT, mf.KD_MASK_WEIGHT, mf.KD_DICE_WEIGHT
        if any([cw > 0, mw > 0, dw > 0]):
            matcher = VideoHungarianMatcher(cw, mw, dw, mf.TRAIN_NUM_POINTS)
        else:
            matcher = VideoHungarianMatcher(kc, km, kd, mf.TRAIN_NUM_POINTS)
        wd = {"loss_ce": cw, "loss_mask": mw, "loss_dice": dw, "kd_loss_ce": kc, "kd_loss_mask": km, "kd_loss_dice": kd}
        if mf.DEEP_SUPERVISION:
            aux = {}
            for i in range(mf.DEC_LAYERS - 1):
                aux.update({k + f"_{i}": v for k, v in wd.items()})
            wd.update(aux)
        crit = VideoSetCriterion(sh.num_classes, matcher=matcher, weight_dict=wd, eos_coef=mf.NO_OBJECT_WEIGHT,
                                 losses=["labels", "masks"], num_points=mf.TRAIN_NUM_POINTS,
                                 oversample_ratio=mf.OVERSAMPLE_RATIO, importance_sample_ratio=mf.IMPORTANCE_SAMPLE_RATIO,
                                 loss_strategy=mf.LOSS_STRATEGY, distillation_loss_strategy=mf.DISTILLATION_LOSS_STRATEGY)
        return cls(student_backbone=sb, student_sem_seg_head=sh, teacher_backbone=tb, teacher_sem_seg_head=th,
                   criterion=crit, num_queries=mf.NUM_OBJECT_QUERIES, num_frames=cfg.INPUT.SAMPLING_FRAME_NUM,
                   size_divisibility=mf.SIZE_DIVISIBILITY, pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD,
                   num_predictions_distillation=mf.NUM_PREDICTIONS_DISTILLATION,
                   score_threshold_distillation=mf.SCORE_THRESHOLD_DISTILLATION, accum_iter=cfg.SOLVER.ACCUM_ITER,
                   distillation_nms=bool(getattr(mf, "DISTILLATION_NMS", False)),
                   **_test_kwargs(mf, "num_predictions_inference", eval_student=True))

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess(self, batched_inputs):
        frames = _frames_to_device(batched_inputs, self.device)
        return ops.normalize_pad(frames, self.size_divisibility, self.pixel_mean.flatten().cpu().numpy(),
                                 self.pixel_std.flatten().cpu().numpy())

    def _kd_nms(self, tgt, cnt, ne, kept=None):
        """the optional mask-NMS of prepare_distillation_targets (kd_video_maskformer_model.py:484-520): greedy over the
        pseudo targets of a clip, a later candidate dropped when its IoU with a kept one exceeds nms_threshold (all pseudo
        labels are the single class).  Pair counts on the device (bit-packed planes, one launch per clip), the greedy walk on
        the K x K integers on the host -- this branch synchronises, the default path does not.  Candidate order: ascending
        query index (the reference walks them in torch.topk(sorted=False) order, which is implementation-defined)."""
        from .postprocess import greedy_mask_nms
        counts = cnt.cpu().tolist()
        for b, k in enumerate(counts):
            if k < 2:
                continue
            inter = ops.mask_pair_counts(ops.pack_mask_bits(tgt[b, :k].contiguous())).cpu().numpy()
            keep = greedy_mask_nms(inter, [0] * k, self.nms_threshold)
            if len(keep) < k:
                sel = torch.as_tensor(keep, device=tgt.device, dtype=torch.long)
                tgt[b, :len(keep)] = tgt[b, sel]
                ne[b, :len(keep)] = ne[b, sel]
                if kept is not None:
                    kept[b, :len(keep)] = kept[b, sel]
                tgt[b, len(keep):k] = 0
                ne[b, len(keep):k] = 0
                cnt[b] = len(keep)
        return tgt, cnt, ne

    @torch.no_grad()
    def forward_losses(self, images, gt_targets: TargetSet, coords_gt=None, coords_kd=None, kd_nmax=None):
        """the device-side hot path from normalised frames to the weighted loss dict (no host sync)"""
        Hp, Wp = images.shape[1:3]
        kd_nmax = kd_nmax or min(self.num_predictions_distillation, self.num_queries)
        # The teacher forward (+ its pseudo-target selection) is independent of the student forward and the GT
        # criterion: it runs on a second HIP stream so the launch tails and the small decoder kernels of one network
        # fill the CUs the other leaves idle.  Every kernel is deterministic, so the schedule does not change results.
        if self.pipeline_clips and images.shape[0] // self.num_frames > 1 and not getattr(self, "keep_kd_targets", False):
            return self._forward_losses_clip_pipeline(images, gt_targets, kd_nmax, coords_gt, coords_kd)
        main = torch.cuda.current_stream()
        if self.overlap_teacher:
            if self._side is None:
                self._side = torch.cuda.Stream(device=images.device)
            side = self._side
            side.wait_stream(main)        # images are ready; all earlier main-stream readers of side-pool memory are done
        else:
            side = main
        with torch.cuda.stream(side):
            teacher = self.teacher(images, True, aux_masks=self.teacher_aux_masks)
            tgt, cnt, kept, ne = ops.kd_targets(teacher.class_logits[-1], teacher.mask_logits[-1], teacher.dims, Hp, Wp, kd_nmax,
                                                self.score_threshold_distillation, self.num_predictions_distillation)
            if self.distillation_nms:
                tgt, cnt, ne = self._kd_nms(tgt, cnt, ne, kept)
        student = self.student(images, True)
        if self.overlap_teacher and self.overlap_criteria:
            side.wait_stream(main)        # student outputs ready
            main.wait_stream(side)        # pseudo targets ready
            with torch.cuda.stream(side):
                losses = self.criterion(student, gt_targets, False, coords_gt)
            kd = self.criterion(student, TargetSet(tgt, cnt, ne), True, coords_kd)
            main.wait_stream(side)
        else:
            losses = self.criterion(student, gt_targets, False, coords_gt)
            main.wait_stream(side)
            kd = self.criterion(student, TargetSet(tgt, cnt, ne), True, coords_kd)
        for k, v in kd.items():
            losses[k.replace("loss_", "kd_loss_")] = v
        wd = self.criterion.weight_dict                                   # :319-325
        out = {k: v * wd[k] for k, v in losses.items() if k in wd}
        self.last = dict(student=student, teacher=teacher, kd_count=cnt, kd_kept=kept)
        if getattr(self, "keep_kd_targets", False):       # tests: the pseudo-target planes the KD matcher saw (1.5 GB at c4 otherwise freed)
            self.last["kd_targets"] = tgt
        return out

    @torch.no_grad()
    def _forward_losses_clip_pipeline(self, images, gt_targets: TargetSet, kd_nmax, coords_gt=None, coords_kd=None):
        """forward_losses clip by clip (kd_video_maskformer_model.py:263-326 on each clip; no op of the path crosses clips): student
        forward of clip b on the main stream, teacher forward + pseudo targets on the side stream, and the clip's two criteria on a third
        stream beside clip b + 1's forwards (with overlap_teacher / overlap_criteria off everything runs on the main stream, same arithmetic:
        the two schedules stay bitwise comparable).  Normalisers that the reference takes over the whole batch:
          * num_masks = clamp(sum of targets over the batch / world, 1) (criterion.py:404-409).  GT pass: the counts are host numbers, the
            clip's call gets world * n_b / N as its world size, which makes its own normaliser N / world.  KD pass: the counts live on
            the device, each clip is normalised by its own clamp(n_b / world, 1) and the sum is rescaled by clamp(sum n_b / world, 1);
          * loss_ce = weighted mean over all B * Q queries (criterion.py:227-251): a clip's mean times its weight sum
            n_b + (Q - n_b) * eos_coef, summed, over the batch's weight sum."""
        T = self.num_frames
        B = images.shape[0] // T
        Hp, Wp = images.shape[1:3]
        Q = self.num_queries
        dev = images.device
        main = torch.cuda.current_stream()
        two = self.overlap_teacher
        if two and self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        if two and self.overlap_criteria and self._crit_stream is None:
            self._crit_stream = torch.cuda.Stream(device=dev)
        side = self._side if two else main
        crit = self._crit_stream if (two and self.overlap_criteria) else main
        ws0 = self.criterion.world_size
        world = self.criterion._world()
        eos = float(self.criterion.eos_coef)

        def clip_coords(c, b, nmax, off):
            """injected points (parity tests) of clip b: matcher [NL,B,P,2] by clip; over / rand [NL, B * maxm * T, n, 2] are indexed by a
            row's rank among the KEPT rows of its layer in (clip, slot, frame) order (csrc/loss.hip coord_rows), so clip b's window of a
            layer starts at the number of rows the earlier clips kept in that layer"""
            if c is None:
                return None
            rows = min(Q, nmax) * T
            out = {}
            for k, v in c.items():
                if k == "matcher":
                    out[k] = v[:, b:b + 1].contiguous()
                else:
                    out[k] = torch.stack([v[l, off[l]:off[l] + rows] for l in range(v.shape[0])]).contiguous()
            return out

        def kept_rows():
            """rows the criterion call just made kept, per layer (tests with injected points only: one small read back)"""
            return ops.point_loss_kept_rows(self.criterion.last_ctx["point_loss"])[0]

        inj_gt, inj_kd = coords_gt is not None, coords_kd is not None
        off_gt = off_kd = None
        ns = list(gt_targets.host_counts) if gt_targets.host_counts is not None else gt_targets.count.cpu().tolist()
        n_tot = float(sum(ns))
        crit.wait_stream(main)            # everything enqueued before this call (the third stream's pool memory included) is ordered
        keep, parts = [], []
        try:                                  # the GT pass borrows the criterion's world size for its batch-wide normaliser
            for b in range(B):
                img = images[b * T:(b + 1) * T]
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    teacher = self.teacher(img, True, aux_masks=self.teacher_aux_masks)
                    tgt, cnt, kept, ne = ops.kd_targets(teacher.class_logits[-1], teacher.mask_logits[-1], teacher.dims, Hp, Wp, kd_nmax,
                                                        self.score_threshold_distillation, self.num_predictions_distillation)
                    if self.distillation_nms:
                        tgt, cnt, ne = self._kd_nms(tgt, cnt, ne, kept)
                student = self.student(img, True)
                crit.wait_stream(main); crit.wait_stream(side)
                gt_b = TargetSet(gt_targets.masks[b:b + 1], gt_targets.count[b:b + 1], gt_targets.nonempty[b:b + 1], [ns[b]])
                with torch.cuda.stream(crit):
                    NLp = student.class_logits.shape[0]
                    off_gt, off_kd = off_gt or [0] * NLp, off_kd or [0] * NLp
                    self.criterion.world_size = world * (ns[b] / n_tot) if (ns[b] > 0 and n_tot > 0) else world
                    lg = self.criterion(student, gt_b, False, clip_coords(coords_gt, b, gt_targets.masks.shape[1], off_gt), keep_ctx=inj_gt)
                    if inj_gt:
                        off_gt = [o + k for o, k in zip(off_gt, kept_rows())]
                    self.criterion.world_size = world
                    lk = self.criterion(student, TargetSet(tgt, cnt, ne), True, clip_coords(coords_kd, b, tgt.shape[1], off_kd), keep_ctx=inj_kd)
                    if inj_kd:
                        off_kd = [o + k for o, k in zip(off_kd, kept_rows())]
                    self.criterion.last_ctx = None
                    nk = self.criterion.last_indices[2][-1:].to(torch.float32)          # matched pairs of the clip's final layer (KD pass)
                keep.append((student, teacher, tgt, cnt, kept, ne))
                parts.append((lg, lk, cnt, nk, float(min(Q, ns[b]))))
        finally:
            self.criterion.world_size = ws0
        with torch.cuda.stream(crit):
            # combine (tiny device ops, on the criteria's stream)
            cnts = torch.stack([p_[2].reshape(()).to(torch.float32) for p_ in parts])
            f_own = torch.clamp(cnts / world, min=1.0)
            f_all = torch.clamp(cnts.sum() / world, min=1.0)
            den_gt = [p_[4] + (Q - p_[4]) * eos for p_ in parts]         # host numbers: no upload (a pageable copy would block the host here)
            den_gt = [d / sum(den_gt) for d in den_gt]
            nk_all = torch.cat([p_[3] for p_ in parts])
            den_kd = nk_all + (Q - nk_all) * eos
            losses = {}
            for k in parts[0][0]:
                v = torch.stack([p_[0][k] for p_ in parts])
                if k == "loss_ce":
                    acc = parts[0][0][k] * den_gt[0]
                    for b in range(1, B):
                        acc = acc + parts[b][0][k] * den_gt[b]
                    losses[k] = acc
                else:
                    losses[k] = v.sum()
            for k in parts[0][1]:
                v = torch.stack([p_[1][k] for p_ in parts])
                losses[k.replace("loss_", "kd_loss_")] = (v * den_kd).sum() / den_kd.sum() if k == "loss_ce" else (v * f_own).sum() / f_all
            wd = self.criterion.weight_dict
            out = {k: v * wd[k] for k, v in losses.items() if k in wd}
        main.wait_stream(crit); main.wait_stream(side)
        self.last = dict(student=[k_[0] for k_ in keep], teacher=[k_[1] for k_ in keep], kd_count=torch.cat([k_[3].reshape(1) for k_ in keep]),
                         kd_kept=[k_[4] for k_ in keep], pipeline=True)
        return out

    @torch.no_grad()
    def forward_backward(self, images, gt_targets: TargetSet, coords_gt=None, coords_kd=None, kd_nmax=None, loss_scale=1.0, grad_ready=None):
        """One training iteration's device work up to the optimizer (engine/train_loop.py:709-726: forward, sum of the
        weighted losses, backward): returns the weighted loss dict of forward_losses and leaves d(sum of losses)/d(parameter)
        in .grad of every student parameter (accumulating, like autograd).  Every gradient is computed by the HIP kernels of
        s2d_amd/backward.py through explicit tapes of the student's activations; the teacher and both matchers carry no
        gradient, attention masks and sampled points are constants, as in the reference.  loss_scale multiplies the
        gradients only (1 / ACCUM_ITER under gradient accumulation, train_loop.py:737-741).  grad_ready(name): called when every
        gradient of the student's "predictor", "pixel_decoder", "backbone" has been enqueued, in that order -- the hook the
        overlapped gradient all-reduce hangs on (optim.OverlappedAllReduce; DDP's bucket hooks in the reference)."""
        wd = self.criterion.weight_dict
        Hp, Wp = images.shape[1:3]
        kd_nmax = kd_nmax or self.num_queries
        backbone, head = self.student[0], self.student[1]
        tb, tp, td = [], [], []
        main = torch.cuda.current_stream(images.device)
        if self.overlap_teacher:                                   # the frozen teacher's forward fills the student's launch tails
            if self._side is None:
                self._side = torch.cuda.Stream(device=images.device)
            side = self._side
            side.wait_stream(main)
        else:
            side = main
        with torch.cuda.stream(side):
            teacher = self.teacher(images, True, aux_masks=self.teacher_aux_masks)
            tgt, cnt, kept, ne = ops.kd_targets(teacher.class_logits[-1], teacher.mask_logits[-1], teacher.dims, Hp, Wp, kd_nmax,
                                                self.score_threshold_distillation, self.num_predictions_distillation)
            if self.distillation_nms:
                tgt, cnt, ne = self._kd_nms(tgt, cnt, ne, kept)
        feats = backbone(images, tb)
        mf, ms = head.pixel_decoder.forward_features(feats, tp)
        student = head.predictor(ms, mf, True, True, td)
        if self.kd_compact and coords_kd is None:
            # The backward's buffers scale with the number of target SLOTS (gradient planes, their transposes, the contraction length of the
            # mask-feature / mask-embedding GEMMs): with Q = 100 slots and ~10 pseudo targets nine tenths of them are zero rows.  One 4-byte
            # read-back on the side stream, behind the student's launches (the host is ahead of the device here; the reference synchronises
            # at the same place: topk + boolean indexing, kd_video_maskformer_model.py:452-470), cuts the planes to the targets found.
            # The sampled points do not depend on the padding (csrc/loss.hip key_row): the losses are those of the uncut pass, bit for bit.
            with torch.cuda.stream(side):
                m4 = max(4, (int(cnt.max()) + 3) // 4 * 4)
                if m4 < tgt.shape[1]:
                    tgt, ne = tgt[:, :m4].contiguous(), ne[:, :m4].contiguous()
        # overlap_criteria (with overlap_teacher): the GT criterion and the backward of its point loss run on the side stream beside
        # the KD criterion's, as in forward_losses -- same host order of the calls, so the same seeds
        crit_side = side if (self.overlap_teacher and self.overlap_criteria) else main
        if crit_side is not main:
            side.wait_stream(main)        # student outputs ready
        main.wait_stream(side)            # pseudo targets ready
        with torch.cuda.stream(crit_side):
            if crit_side is not main and self._side_delay_cycles:      # tests: hold the side stream back (schedule-independence check)
                torch.cuda._sleep(int(self._side_delay_cycles))
            losses = self.criterion(student, gt_targets, False, coords_gt, keep_ctx=True)
            ctx_gt = self.criterion.last_ctx
        kd = self.criterion(student, TargetSet(tgt, cnt, ne), True, coords_kd, keep_ctx=True)
        ctx_kd = self.criterion.last_ctx
        for k, v in kd.items():
            losses[k.replace("loss_", "kd_loss_")] = v
        # ---- backward
        NL, B = student.class_logits.shape[:2]
        Q, T, hm, wm = student.dims
        d_cls = torch.zeros_like(student.class_logits)
        sources = []
        for ctx, pre, strm in ((ctx_gt, "", crit_side), (ctx_kd, "kd_", main)):
            w_mask, w_dice = wd.get(pre + "loss_mask", 0.0), wd.get(pre + "loss_dice", 0.0)
            for i in range(NL - 1):
                if wd.get(pre + f"loss_mask_{i}", w_mask) != w_mask or wd.get(pre + f"loss_dice_{i}", w_dice) != w_dice:
                    raise NotImplementedError("per-layer loss weights that differ between decoder layers")
            if w_mask != 0.0 or w_dice != 0.0:
                with torch.cuda.stream(strm):
                    rows = ops.point_loss_backward(ctx["point_loss"], w_mask * loss_scale, w_dice * loss_scale).view(NL, B, ctx["maxm"], T * hm * wm)
                sources.append((rows, ctx["idx_q"]))
        main.wait_stream(crit_side)
        # the weighted dict is formed on the main stream: only here, behind the wait, are the GT criterion's loss tensors (written on
        # the side stream) ordered before their reader
        out = {k: v * wd[k] for k, v in losses.items() if k in wd}
        for ctx, pre in ((ctx_gt, ""), (ctx_kd, "kd_")):
            w_ce = wd.get(pre + "loss_ce", 0.0)
            if w_ce != 0.0:
                d_cls[NL - 1] += ops.class_loss_backward(student.class_logits[NL - 1], ctx["idx_q"][(NL - 1) * B:].contiguous(),
                                                         ctx["n_match"][(NL - 1) * B:].contiguous(), w_ce * loss_scale, self.criterion.eos_coef)
        from .. import backward as Bk
        Bk.begin_deferred_acc()          # the ~350 "param.grad += g" of the walk below: one multi-tensor launch per part
        try:
            d_mf, d_mem = head.predictor.backward(td[0], d_cls, sources)
            Bk.flush_acc()
            if grad_ready is not None:
                grad_ready("predictor")
            grads = head.pixel_decoder.backward_features(tp[0], d_mf, d_mem)
            Bk.flush_acc()
            if grad_ready is not None:
                grad_ready("pixel_decoder")
            backbone.backward(tb, grads)
        finally:
            Bk.flush_acc(end=True)
        if grad_ready is not None:
            grad_ready("backbone")
        self.last = dict(student=student, teacher=teacher, kd_count=cnt, kd_kept=kept)
        # the activation tapes are garbage once the gradients exist; keeping them alive into the next iteration's forward
        # costs 45 GiB of peak memory at c4 (keep_tapes = True for tests that inspect them)
        self.last_tapes = (tb, tp, td) if getattr(self, "keep_tapes", False) else None
        return out

    def forward(self, batched_inputs):
        """kd_video_maskformer_model.py:233-356.  Training: the weighted loss dict (:314-326).  With autograd enabled the
        losses carry a graph to the student parameters (the backward runs on the HIP kernels inside this call and is handed
        to autograd by _HipGradBridge), so `sum(loss_dict.values()).backward()` fills .grad as the reference trainer and
        DistributedDataParallel expect; under torch.no_grad() only forward + loss run.  engine.run_step is the direct path
        (same kernels, no staging copy of the gradients)."""
        images = self.preprocess(batched_inputs)
        if not self.training:
            return self.inference(images, batched_inputs)
        Hp, Wp = images.shape[1:3]
        gt = TargetSet.from_list(_gt_target_list(batched_inputs, self.num_frames, Hp, Wp, self.device), device=self.device)
        params = [p for p in self.student.parameters() if p.requires_grad]
        if torch.is_grad_enabled() and params:
            return _bridge_losses(self, params, lambda: self.forward_backward(images, gt))
        return self.forward_losses(images, gt)

    @torch.no_grad()
    def inference(self, images, batched_inputs):
        """eval branch (kd_video_maskformer_model.py:327-356): the whole video as one clip through the teacher (or the
        student, TEST.EVAL_STUDENT), then inference_video (:530-610) on the device -> {image_size, pred_scores,
        pred_labels, pred_masks}."""
        net = self.student if self.eval_student else self.teacher
        return _inference(net, images, batched_inputs, self.num_predictions_inference, self.use_nms, self.nms_threshold)


def _inference(net, images, batched_inputs, num_predictions, use_nms, nms_threshold):
    out = net(images, False)
    video = batched_inputs[0]
    first = video["image"][0]
    image_size = tuple(int(v) for v in first.shape[-2:])            # size without padding (images.image_sizes[0], :349)
    height, width = video.get("height", image_size[0]), video.get("width", image_size[1])   # :351-352
    return inference_video(out.class_logits[-1][0], out.mask_logits[-1][0], (out.T, out.hm, out.wm), tuple(images.shape[1:3]),
                           image_size, (int(height), int(width)), num_predictions, use_nms, nms_threshold)


@META_ARCH_REGISTRY.register()
class VideoMaskFormer(nn.Module):
    """Non-KD variant (video_maskformer_model.py:189-265): one network, one criterion pass."""

    def __init__(self, *, backbone, sem_seg_head, criterion, num_queries, num_frames, size_divisibility=32,
                 pixel_mean=ops.PIXEL_MEAN, pixel_std=ops.PIXEL_STD, use_nms=False, nms_threshold=0.75, num_predictions=10):
        super().__init__()
        self.backbone, self.sem_seg_head, self.criterion = backbone, sem_seg_head, criterion
        self.num_queries, self.num_frames, self.size_divisibility = num_queries, num_frames, size_divisibility
        self.use_nms, self.nms_threshold, self.num_predictions = use_nms, nms_threshold, num_predictions
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean, dtype=torch.float32).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std, dtype=torch.float32).view(-1, 1, 1), False)

    @classmethod
    def from_config(cls, cfg):  # video_maskformer_model.py:97-187 (the sparse-class / entropy / DropLoss variants are not on the path)
        mf = cfg.MODEL.MASK_FORMER
        if getattr(mf, "SPARSE_CLASS_WEIGHT", 0.0) > 0.0 or getattr(mf, "MASK_DROPLOSS", False) or getattr(mf, "LABEL_DROPLOSS", False):
            raise NotImplementedError("SPARSE_CLASS_WEIGHT / MASK_DROPLOSS / LABEL_DROPLOSS criterion variants")
        head = MaskFormerHead.from_config(cfg)
        cw, dw, mw = mf.CLASS_WEIGHT, mf.DICE_WEIGHT, mf.MASK_WEIGHT
        matcher = VideoHungarianMatcher(0.0 if getattr(mf, "NO_CLASS_MATCH", False) else cw, mw, dw, mf.TRAIN_NUM_POINTS)
        wd = {"loss_ce": cw, "loss_mask": mw, "loss_dice": dw}
        if mf.DEEP_SUPERVISION:
            aux = {}
            for i in range(mf.DEC_LAYERS - 1):
                aux.update({k + f"_{i}": v for k, v in wd.items()})
            wd.update(aux)
        crit = VideoSetCriterion(head.num_classes, matcher=matcher, weight_dict=wd, eos_coef=mf.NO_OBJECT_WEIGHT,
                                 losses=["labels", "masks"], num_points=mf.TRAIN_NUM_POINTS, oversample_ratio=mf.OVERSAMPLE_RATIO,
                                 importance_sample_ratio=mf.IMPORTANCE_SAMPLE_RATIO, loss_strategy=mf.LOSS_STRATEGY)
        m = cls(backbone=ResNet50(), sem_seg_head=head, criterion=crit, num_queries=mf.NUM_OBJECT_QUERIES,
                num_frames=cfg.INPUT.SAMPLING_FRAME_NUM, size_divisibility=mf.SIZE_DIVISIBILITY, pixel_mean=cfg.MODEL.PIXEL_MEAN,
                pixel_std=cfg.MODEL.PIXEL_STD, **_test_kwargs(mf, "num_predictions"))
        m.accum_iter = cfg.SOLVER.ACCUM_ITER
        return m

    @property
    def device(self):
        return self.pixel_mean.device

    @torch.no_grad()
    def forward_losses(self, images, gt_targets, coords=None):
        out = self.sem_seg_head(self.backbone(images), True)
        losses = self.criterion(out, gt_targets, False, coords)
        wd = self.criterion.weight_dict
        self.last = dict(outputs=out)
        return {k: v * wd[k] for k, v in losses.items() if k in wd}

    @torch.no_grad()
    def forward_backward(self, images, gt_targets, coords=None, loss_scale=1.0):
        """forward + loss + backward of the single network (see KDVideoMaskFormer.forward_backward): returns the weighted loss
        dict and leaves the gradients in .grad of the backbone / head parameters"""
        wd = self.criterion.weight_dict
        head = self.sem_seg_head
        tb, tp, td = [], [], []
        feats = self.backbone(images, tb)
        mf, ms = head.pixel_decoder.forward_features(feats, tp)
        out = head.predictor(ms, mf, True, True, td)
        losses = self.criterion(out, gt_targets, False, coords, keep_ctx=True)
        ctx = self.criterion.last_ctx
        NL, B = out.class_logits.shape[:2]
        Q, T, hm, wm = out.dims
        w_mask, w_dice, w_ce = wd.get("loss_mask", 0.0), wd.get("loss_dice", 0.0), wd.get("loss_ce", 0.0)
        for i in range(NL - 1):
            if wd.get(f"loss_mask_{i}", w_mask) != w_mask or wd.get(f"loss_dice_{i}", w_dice) != w_dice:
                raise NotImplementedError("per-layer loss weights that differ between decoder layers")
        sources = []
        if w_mask != 0.0 or w_dice != 0.0:
            rows = ops.point_loss_backward(ctx["point_loss"], w_mask * loss_scale, w_dice * loss_scale).view(NL, B, ctx["maxm"], T * hm * wm)
            sources.append((rows, ctx["idx_q"]))
        d_cls = torch.zeros_like(out.class_logits)
        if w_ce != 0.0:
            d_cls[NL - 1] = ops.class_loss_backward(out.class_logits[NL - 1], ctx["idx_q"][(NL - 1) * B:].contiguous(),
                                                    ctx["n_match"][(NL - 1) * B:].contiguous(), w_ce * loss_scale, self.criterion.eos_coef)
        d_mf, d_mem = head.predictor.backward(td[0], d_cls, sources)
        self.backbone.backward(tb, head.pixel_decoder.backward_features(tp[0], d_mf, d_mem))
        self.last = dict(outputs=out)
        return {k: v * wd[k] for k, v in losses.items() if k in wd}

    def forward(self, batched_inputs):
        frames = _frames_to_device(batched_inputs, self.device)
        images = ops.normalize_pad(frames, self.size_divisibility, self.pixel_mean.flatten().cpu().numpy(),
                                   self.pixel_std.flatten().cpu().numpy())
        if not self.training:                                           # video_maskformer_model.py:241-264
            net = lambda x, training: self.sem_seg_head(self.backbone(x), training)   # noqa: E731
            with torch.no_grad():
                return _inference(net, images, batched_inputs, self.num_predictions, self.use_nms, self.nms_threshold)
        Hp, Wp = images.shape[1:3]
        gt = TargetSet.from_list(_gt_target_list(batched_inputs, self.num_frames, Hp, Wp, self.device), device=self.device)
        params = [p for p in list(self.backbone.parameters()) + list(self.sem_seg_head.parameters()) if p.requires_grad]
        if torch.is_grad_enabled() and params:                          # see KDVideoMaskFormer.forward
            return _bridge_losses(self, params, lambda: self.forward_backward(images, gt))
        return self.forward_losses(images, gt)


def build_kd_model(num_queries=100, num_frames=8, num_points=160000, weights=(0.0, 5.0, 5.0), kd_weights=(0.0, 5.0, 5.0),
                   dec_layers=10, seed=0, teacher_bias=None, dropout=0.0):
    """Construct KDVideoMaskFormer with the shipped hyper-parameters
    (configs/imagenet_video/ytvis2021_kd_video_mask2former_R50_cls_agnostic.yaml) without a yacs config."""
    torch.manual_seed(seed)

    def head():
        return MaskFormerHead(MSDeformAttnPixelDecoder(transformer_dropout=dropout),
                              VideoMultiScaleMaskedTransformerDecoder(num_queries=num_queries, num_frames=num_frames,
                                                                      dec_layers=dec_layers - 1))
    sb, sh = ResNet50(), head()
    tb, th = ResNet50(), head()
    cw, mw, dw = weights
    matcher = VideoHungarianMatcher(cw, mw, dw, num_points) if any(w > 0 for w in weights) else VideoHungarianMatcher(*kd_weights, num_points)
    wd = {"loss_ce": cw, "loss_mask": mw, "loss_dice": dw, "kd_loss_ce": kd_weights[0], "kd_loss_mask": kd_weights[1],
          "kd_loss_dice": kd_weights[2]}
    aux = {}
    for i in range(dec_layers - 1):
        aux.update({k + f"_{i}": v for k, v in wd.items()})
    wd.update(aux)
    crit = VideoSetCriterion(1, matcher=matcher, weight_dict=wd, eos_coef=0.1, losses=["labels", "masks"], num_points=num_points,
                             oversample_ratio=3.0, importance_sample_ratio=0.75, loss_strategy="masks-only",
                             distillation_loss_strategy="masks-only")
    model = KDVideoMaskFormer(student_backbone=sb, student_sem_seg_head=sh, teacher_backbone=tb, teacher_sem_seg_head=th,
                              criterion=crit, num_queries=num_queries, num_frames=num_frames)
    model.teacher.load_state_dict(model.student.state_dict())   # teacher starts as a copy of the student (EMA init)
    if teacher_bias is not None:
        with torch.no_grad():
            model.teacher[1].predictor.class_embed.bias.copy_(torch.tensor(teacher_bias))
    return model

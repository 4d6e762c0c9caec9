"""VideoHungarianMatcher / VideoSetCriterion, host side.  Mirror
model_training/mask2former_video/modeling/matcher.py:200-294 and criterion.py:163-443 (same constructor arguments,
same `matcher(outputs, targets) -> [(idx_q, idx_t)]` and `criterion(outputs, targets, distillation) -> dict` call
shapes), but run every prediction layer and clip of a pass in a handful of batched launches with no host sync.

`VideoSetCriterion.forward` implements the INTENDED behaviour of criterion.py:390-427: the shipped method raises
AttributeError (its loss_map names undefined loss_labels_drop / loss_masks_drop, :380-385); the intent is
losses = ["labels", "masks"] (kd_video_maskformer_model.py:186), labels skipped for aux layers (:421-422), aux
layers evaluated with the non-distillation DropLoss strategy (:423).
"""
import torch
from torch import nn

from .. import ops
from .video_decoder import MaskOutputs


_COUNT_CACHE = {}


class TargetSet:
    """Device-resident targets of one criterion pass: masks u8 [B,Nmax,T,H,W], count i32 [B] (device),
    nonempty i32 [B,Nmax,T].  `host_counts` is set when the counts are known on the host (ground truth)."""

    def __init__(self, masks, count, nonempty, host_counts=None):
        self.masks, self.count, self.nonempty, self.host_counts = masks, count, nonempty, host_counts

    @staticmethod
    def from_list(mask_list, Nmax=None, device="cuda"):
        """mask_list: per clip a [N,T,H,W] tensor (bool / uint8 / float 0-1), as kd_video_maskformer_model.py:358-386
        prepares them.  Padding to (Hp,Wp) must already be done."""
        B = len(mask_list)
        ns = [int(m.shape[0]) for m in mask_list]
        T, H, W = mask_list[0].shape[1:]
        Nmax = Nmax or max(max(ns), 1)
        masks = torch.empty((B, Nmax, T, H, W), device=device, dtype=torch.uint8)
        for b, m in enumerate(mask_list):
            if ns[b]:
                mb = m.to(device=device, non_blocking=True)
                if mb.dtype == torch.uint8:
                    torch.ne(mb, 0, out=masks[b, :ns[b]].view(torch.bool))    # one pass, no temporary
                else:
                    masks[b, :ns[b]] = (mb != 0).to(torch.uint8)
            if ns[b] < Nmax:
                masks[b, ns[b]:].zero_()
        key = (tuple(ns), str(device))
        count = _COUNT_CACHE.get(key)                       # instance counts repeat: no H2D copy / host sync on the hot path
        if count is None:
            if len(_COUNT_CACHE) > 4096:
                _COUNT_CACHE.clear()
            count = _COUNT_CACHE[key] = torch.tensor(ns, dtype=torch.int32, device=device)
        return TargetSet(masks, count, ops.target_nonempty(masks, count), ns)


def _to_mask_outputs(outputs):
    """accept the native MaskOutputs or the reference's dict layout ({'pred_logits','pred_masks'[,'aux_outputs']})"""
    if isinstance(outputs, MaskOutputs):
        return outputs
    layers = list(outputs.get("aux_outputs", [])) + [outputs]
    cls = torch.stack([o["pred_logits"].float() for o in layers]).contiguous()
    pm = torch.stack([o["pred_masks"].float() for o in layers])            # [NL,B,Q,T,h,w]
    NL, B, Q, T, h, w = pm.shape
    ldq = (Q + 3) // 4 * 4
    ml = torch.zeros((NL, B, T * h * w, ldq), device=pm.device, dtype=torch.float32)
    ml[..., :Q] = pm.permute(0, 1, 3, 4, 5, 2).reshape(NL, B, T * h * w, Q)
    return MaskOutputs(cls, ml, Q, T, h, w)


class VideoHungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_mask: float = 1, cost_dice: float = 1, num_points: int = 0):
        super().__init__()
        assert cost_class != 0 or cost_mask != 0 or cost_dice != 0, "all costs cant be 0"
        self.cost_class, self.cost_mask, self.cost_dice, self.num_points = cost_class, cost_mask, cost_dice, num_points
        self.seed = 0

    @torch.no_grad()
    def match_all(self, out: MaskOutputs, targets: TargetSet, coords=None):
        """all layers x clips at once -> (idx_q, idx_t [NL*B, maxm], n_match [NL*B]) int32 on the device"""
        self.seed += 1
        C = ops.matcher_cost(out.mask_logits, out.class_logits, targets.masks, targets.count, out.dims, self.num_points,
                             (self.cost_class, self.cost_mask, self.cost_dice), coords=coords, seed=self.seed)
        B = out.mask_logits.shape[1]
        self.last_cost = C                      # [NL*B, Q, Nmax] (tests compare it with the oracle's cost matrices)
        return ops.lsap(C, targets.count, B)

    @torch.no_grad()
    def forward(self, outputs, targets):
        """reference call shape (matcher.py:297-318): targets = list of {'labels','masks'}; returns CPU int64 pairs
        for the LAST layer of `outputs`."""
        out = _to_mask_outputs({k: v for k, v in outputs.items() if k != "aux_outputs"} if isinstance(outputs, dict) else outputs)
        ts = targets if isinstance(targets, TargetSet) else TargetSet.from_list([t["masks"] for t in targets], device=out.mask_logits.device)
        iq, it, nm = self.match_all(out, ts)
        B = out.mask_logits.shape[1]
        iq, it, nm = iq[-B:].cpu(), it[-B:].cpu(), nm[-B:].cpu()
        return [(iq[b, :nm[b]].to(torch.int64), it[b, :nm[b]].to(torch.int64)) for b in range(B)]


class VideoSetCriterion(nn.Module):
    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses, num_points, oversample_ratio,
                 importance_sample_ratio, loss_strategy, reweight_distillation_loss=False,
                 distillation_loss_strategy="masks-only", world_size=None):
        super().__init__()
        assert loss_strategy in ["masks-only", "full"]
        self.num_classes, self.matcher, self.weight_dict, self.eos_coef, self.losses = num_classes, matcher, weight_dict, eos_coef, losses
        empty_weight = torch.ones(num_classes + 1)
        empty_weight[-1] = eos_coef
        self.register_buffer("empty_weight", empty_weight)
        self.num_points, self.oversample_ratio, self.importance_sample_ratio = num_points, oversample_ratio, importance_sample_ratio
        self.loss_strategy, self.distillation_loss_strategy = loss_strategy, distillation_loss_strategy
        self.world_size = world_size            # None: detectron2's comm.get_world_size() at call time (criterion.py:409)
        self.seed = 0

    def _world(self):
        if self.world_size is not None:
            return float(self.world_size)
        import torch.distributed as dist
        return float(dist.get_world_size()) if dist.is_available() and dist.is_initialized() else 1.0

    @torch.no_grad()
    def forward(self, outputs, targets, distillation=False, coords=None, keep_ctx=False):
        """-> {'loss_ce','loss_mask','loss_dice','loss_mask_i','loss_dice_i' (i = 0..NL-2)} of 0-dim tensors.
        coords (parity mode) = dict(matcher=[NL,B,P,2], over=[NL,R,3P,2], rand=[NL,R,P/4,2])."""
        out = _to_mask_outputs(outputs)
        ts = targets if isinstance(targets, TargetSet) else TargetSet.from_list([t["masks"] for t in targets], device=out.mask_logits.device)
        NL, B = out.mask_logits.shape[:2]
        c = coords or {}
        # "indices" (tests only): a fixed assignment instead of the matcher's, e.g. to difference the loss at constant matching
        iq, it, nm = c["indices"] if "indices" in c else self.matcher.match_all(out, ts, c.get("matcher"))
        # DropLoss strategy: the last layer uses (distillation ? distillation_loss_strategy : loss_strategy); aux layers
        # always loss_strategy (criterion.py:307-308, :423).  Both are "masks-only" in every shipped config.
        drop_last = (self.distillation_loss_strategy if distillation else self.loss_strategy) == "masks-only"
        drop_aux = self.loss_strategy == "masks-only"
        self.seed += 1
        kw = dict(oversample=self.oversample_ratio, importance=self.importance_sample_ratio, coords_over=c.get("over"),
                  coords_rand=c.get("rand"), seed=self.seed, world_size=self._world())
        losses = {}
        self.last_ctx = None
        if drop_last == drop_aux:
            L = ops.point_loss(out.mask_logits, ts.masks, ts.count, ts.nonempty, iq, it, nm, out.dims, self.num_points,
                               drop_empty=drop_aux, keep=keep_ctx, **kw)
            if keep_ctx:                                   # what the backward needs: selection state + the assignment
                L, pl_ctx = L
                self.last_ctx = dict(point_loss=pl_ctx, idx_q=iq, n_match=nm, maxm=iq.shape[-1])
        else:  # mixed strategies: two launches over the same buffers
            if keep_ctx:
                raise NotImplementedError("backward with different DropLoss strategies for the last and the auxiliary layers")
            L = ops.point_loss(out.mask_logits, ts.masks, ts.count, ts.nonempty, iq, it, nm, out.dims, self.num_points,
                               drop_empty=drop_aux, **kw)
            L2 = ops.point_loss(out.mask_logits, ts.masks, ts.count, ts.nonempty, iq, it, nm, out.dims, self.num_points,
                                drop_empty=drop_last, **kw)
            L = torch.cat([L[:-1], L2[-1:]], 0)
        if "labels" in self.losses:
            losses["loss_ce"] = ops.class_loss(out.class_logits[NL - 1], iq[(NL - 1) * B:], nm[(NL - 1) * B:], self.eos_coef)
        if "masks" in self.losses:
            losses["loss_mask"], losses["loss_dice"] = L[NL - 1, 0], L[NL - 1, 1]
            for i in range(NL - 1):
                losses[f"loss_mask_{i}"], losses[f"loss_dice_{i}"] = L[i, 0], L[i, 1]
        self.last_indices = (iq, it, nm)
        return losses

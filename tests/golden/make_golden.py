#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/*.npz by running
the reference's own Python (loaded unmodified from /root/reference through
_ref_shim.py) on seeded synthetic inputs.  Container-only; re-run with

    python tests/golden/make_golden.py [g_name ...]

Fixtures hold data only: seeds, the inputs that cannot be regenerated from a
seed (recorded torch.rand draws), and the reference's outputs.  Network weights
are regenerated from (name, shape, seed) by s2d_amd.utils.seeded.
Each case cites the reference file:line it exercises.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _ref_shim as R  # noqa: E402
from s2d_amd.utils.seeded import seeded_state  # noqa: E402
from s2d_amd.utils import synth  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  ({os.path.getsize(path)/1024:.0f} KiB)")


def load_seeded(module, seed):
    sd = seeded_state([(k, tuple(v.shape)) for k, v in module.state_dict().items()], seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)


class RandRecorder:
    """Record (and seed) every torch.rand draw made inside the reference."""

    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.log = []

    def __enter__(self):
        self._orig = torch.rand

        def fn(*size, **kw):
            kw.pop("device", None)
            t = self._orig(*size, generator=self.g, **kw)
            self.log.append(t.clone())
            return t

        torch.rand = fn
        return self

    def __exit__(self, *a):
        torch.rand = self._orig


# ----------------------------------------------------------------------------
def g_msda():
    F_ = R.ref("mask2former.modeling.pixel_decoder.ops.functions.ms_deform_attn_func")
    core = F_.ms_deform_attn_core_pytorch
    # (a) the ops/test.py fixture (test.py:24-31, 35-47, 51-63), on CPU
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    w = torch.rand(N, Lq, M, L, P) + 1e-5
    w /= w.sum(-1, keepdim=True).sum(-2, keepdim=True)
    out64 = core(value.double(), shapes, loc.double(), w.double())
    out32 = core(value, shapes, loc, w)
    save("msda_optest", value=value, loc=loc, w=w, shapes=shapes, out64=out64, out32=out32)

    # (b) S2D geometry: M=8, D=32, L=3, P=4 (msdeformattn.py:232-239); locations
    # spill outside [0,1] to exercise the zero-padding corner tests (.cuh:61-83,293)
    N, M, D, L, P = 2, 8, 32, 3, 4
    shapes = torch.as_tensor([(2, 3), (4, 6), (8, 12)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    Lq = S
    g = torch.Generator().manual_seed(11)
    value = torch.randn(N, S, M, D, generator=g)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g) * 1.3 - 0.15
    w = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
    value.requires_grad_(True); loc.requires_grad_(True); w.requires_grad_(True)
    out = core(value, shapes, loc, w)
    go = torch.randn(out.shape, generator=g)
    gv, gl, gw = torch.autograd.grad(out, (value, loc, w), go)
    save("msda_core", value=value, loc=loc, w=w, shapes=shapes, out=out, grad_out=go,
         grad_value=gv, grad_loc=gl, grad_w=gw)

    # (c) the module (ms_deform_attn.py:82-125)
    Mod = R.ref("mask2former.modeling.pixel_decoder.ops.modules.ms_deform_attn").MSDeformAttn
    mod = Mod(256, 3, 8, 4)
    load_seeded(mod, 21)
    N = 2
    query = torch.from_numpy(synth.randn(21, 1, (N, S, 256)))
    src = torch.from_numpy(synth.randn(21, 2, (N, S, 256)))
    Enc = R.ref("mask2former.modeling.pixel_decoder.msdeformattn").MSDeformAttnTransformerEncoder
    ref_pts = Enc.get_reference_points(shapes, torch.ones(N, 3, 2), "cpu")
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    with torch.no_grad():
        out = mod(query, ref_pts, src, shapes, lsi, None)
    save("msda_module", seed=21, shapes=shapes, ref_pts=ref_pts, out=out)


def g_pe():
    pe2 = R.ref("mask2former.modeling.transformer_decoder.position_encoding").PositionEmbeddingSine(128, normalize=True)
    pe3 = R.ref("mask2former_video.modeling.transformer_decoder.position_encoding").PositionEmbeddingSine3D(128, normalize=True)
    save("pe", pe2=pe2(torch.zeros(1, 4, 5, 7)), pe3=pe3(torch.zeros(1, 3, 4, 4, 6)))


def feature_inputs(seed, BT, h4, w4):
    return {
        "res2": synth.randn(seed, 2, (BT, 256, h4, w4)),
        "res3": synth.randn(seed, 3, (BT, 512, h4 // 2, w4 // 2)),
        "res4": synth.randn(seed, 4, (BT, 1024, h4 // 4, w4 // 4)),
        "res5": synth.randn(seed, 5, (BT, 2048, h4 // 8, w4 // 8)),
    }


def build_pixel_decoder(seed):
    m = R.ref("mask2former.modeling.pixel_decoder.msdeformattn")
    SS = sys.modules["detectron2.layers"].ShapeSpec
    shp = {"res2": SS(channels=256, stride=4), "res3": SS(channels=512, stride=8),
           "res4": SS(channels=1024, stride=16), "res5": SS(channels=2048, stride=32)}
    pd = m.MSDeformAttnPixelDecoder(shp, transformer_dropout=0.0, transformer_nheads=8,
                                    transformer_dim_feedforward=1024, transformer_enc_layers=6,
                                    conv_dim=256, mask_dim=256, norm="GN",
                                    transformer_in_features=["res3", "res4", "res5"], common_stride=4)
    load_seeded(pd, seed)
    return pd


def g_pixel_decoder():
    # msdeformattn.py:314-358 (+ encoder :61-131, reference points :141-153)
    seed = 31
    pd = build_pixel_decoder(seed)
    feats = {k: torch.from_numpy(v) for k, v in feature_inputs(seed, 2, 16, 24).items()}
    with torch.no_grad():
        mf, enc0, ms = pd.forward_features(feats)
    save("pixel_decoder", seed=seed, BT=2, h4=16, w4=24, mask_features=mf, ms0=ms[0], ms1=ms[1], ms2=ms[2])


def build_video_decoder(seed, Q, T):
    v = R.ref("mask2former_video.modeling.transformer_decoder.video_mask2former_transformer_decoder")
    dec = v.VideoMultiScaleMaskedTransformerDecoder(
        256, True, num_classes=1, hidden_dim=256, num_queries=Q, nheads=8, dim_feedforward=2048,
        dec_layers=9, pre_norm=False, mask_dim=256, enforce_input_project=False, num_frames=T)
    load_seeded(dec, seed)
    dec.train()
    return dec


def decoder_inputs(seed, BT, h4, w4):
    ms = [synth.randn(seed, 10, (BT, 256, h4 // 8, w4 // 8)),
          synth.randn(seed, 11, (BT, 256, h4 // 4, w4 // 4)),
          synth.randn(seed, 12, (BT, 256, h4 // 2, w4 // 2))]
    mf = synth.randn(seed, 13, (BT, 256, h4, w4), 0.5)
    # second half of the frames (= last clip): one dominant constant channel, so about half
    # the queries get all-negative masks there, which exercises the "fully masked row ->
    # unmask" fix (video_...decoder.py:413); the first clip keeps mixed masks.
    mf[BT // 2:] *= 0.1
    mf[BT // 2:, 0] += 3.0
    return ms, mf


def g_video_decoder():
    # video_mask2former_transformer_decoder.py:374-467
    seed, B, T, Q, h4, w4 = 41, 2, 2, 16, 16, 24
    dec = build_video_decoder(seed, Q, T)
    ms, mf = decoder_inputs(seed, B * T, h4, w4)
    # count how often the all-masked fix fires (proves the fixture exercises it)
    fired = []
    v = R.ref("mask2former_video.modeling.transformer_decoder.video_mask2former_transformer_decoder")
    orig = v.VideoMultiScaleMaskedTransformerDecoder.forward_prediction_heads

    def spy(self, output, mask_features, attn_mask_target_size):
        a, b, am = orig(self, output, mask_features, attn_mask_target_size)
        fired.append(int((am.sum(-1) == am.shape[-1]).sum()))
        return a, b, am

    v.VideoMultiScaleMaskedTransformerDecoder.forward_prediction_heads = spy
    with torch.no_grad():
        out = dec([torch.from_numpy(x) for x in ms], torch.from_numpy(mf))
    v.VideoMultiScaleMaskedTransformerDecoder.forward_prediction_heads = orig
    logits = torch.stack([a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]])
    masks = torch.stack([a["pred_masks"] for a in out["aux_outputs"]] + [out["pred_masks"]])
    print("   all-masked rows per head call:", fired)
    assert sum(fired) > 0
    save("video_decoder", seed=seed, B=B, T=T, Q=Q, h4=h4, w4=w4, logits=logits, masks=masks,
         all_masked_rows=np.array(fired))


def make_targets(seed, tag, ns, T, H, W):
    tg = []
    for b, n in enumerate(ns):
        m, ids = synth.ellipse_targets(seed, tag + b, n, T, H, W)
        tg.append({"labels": torch.zeros(n, dtype=torch.int64), "masks": torch.from_numpy(m).float(),
                   "ids": torch.from_numpy(ids)})
    return tg


def g_matcher():
    # matcher.py:225-294 (+ batch_dice_loss :15-30, batch_sigmoid_ce_loss :38-62)
    mt = R.ref("mask2former_video.modeling.matcher")
    for name, (B, Q, T, h, w, H, W, ns, P, seed) in {
        "matcher_small": (2, 16, 2, 16, 24, 64, 96, [3, 5], 256, 51),
        "matcher_q100": (1, 100, 2, 30, 54, 120, 216, [10], 1024, 52),
        "matcher_empty": (2, 16, 2, 16, 24, 64, 96, [0, 4], 256, 53),
        "matcher_wide": (1, 12, 2, 16, 24, 64, 96, [20], 256, 54),  # more targets than queries
    }.items():
        logits = torch.from_numpy(synth.randn(seed, 1, (B, Q, 2)))
        masks = torch.from_numpy(synth.smooth_logits(seed, 2, (B, Q, T), (h, w)))
        tg = make_targets(seed, 100, ns, T, H, W)
        m = mt.VideoHungarianMatcher(cost_class=2.0, cost_mask=5.0, cost_dice=5.0, num_points=P)
        Cs = []
        orig = mt.linear_sum_assignment

        def rec(C):
            Cs.append(np.asarray(C).copy())
            return orig(C)

        mt.linear_sum_assignment = rec
        with RandRecorder(seed) as rr:
            idx = m({"pred_logits": logits, "pred_masks": masks}, tg)
        mt.linear_sum_assignment = orig
        arrs = dict(seed=seed, dims=np.array([B, Q, T, h, w, H, W, P]), ns=np.array(ns),
                    cost_weights=np.array([2.0, 5.0, 5.0]))
        for b in range(B):
            arrs[f"coords{b}"] = rr.log[b]
            arrs[f"C{b}"] = Cs[b]
            arrs[f"i{b}"] = idx[b][0]
            arrs[f"j{b}"] = idx[b][1]
        save(name, **arrs)


def g_loss():
    # criterion.py:227-251 (loss_labels), :292-356 (loss_masks), point_features.py:63-116
    cr = R.ref("mask2former_video.modeling.criterion")
    B, Q, T, h, w, H, W, P, seed = 2, 16, 2, 16, 24, 64, 96, 256, 61
    ns = [3, 5]
    logits = torch.from_numpy(synth.randn(seed, 1, (B, Q, 2)))
    masks = torch.from_numpy(synth.smooth_logits(seed, 2, (B, Q, T), (h, w)))
    tg = make_targets(seed, 100, ns, T, H, W)
    crit = cr.VideoSetCriterion(1, matcher=None, weight_dict={}, eos_coef=0.1, losses=["labels", "masks"],
                                num_points=P, oversample_ratio=3.0, importance_sample_ratio=0.75,
                                loss_strategy="masks-only", distillation_loss_strategy="masks-only")
    rng = np.random.default_rng(seed)
    indices = []
    for b, n in enumerate(ns):
        qi = np.sort(rng.choice(Q, n, replace=False))
        tj = rng.permutation(n)
        indices.append((torch.as_tensor(qi), torch.as_tensor(tj)))
    outputs = {"pred_logits": logits, "pred_masks": masks}
    num_masks = float(sum(ns))
    ll = crit.loss_labels(outputs, tg, indices, num_masks, False)
    with RandRecorder(seed) as rr:
        lm = crit.loss_masks(outputs, tg, indices, num_masks, False)
    # rows kept by DropLoss, in order (criterion.py:307-322)
    tm = torch.cat([t["masks"][i] for t, (_, i) in zip(tg, indices)]).flatten(0, 1)
    keep = np.array([i for i in range(tm.shape[0]) if tm[i].sum() != 0])
    arrs = dict(seed=seed, dims=np.array([B, Q, T, h, w, H, W, P]), ns=np.array(ns), num_masks=num_masks,
                loss_ce=ll["loss_ce"], loss_mask=lm["loss_mask"], loss_dice=lm["loss_dice"],
                coords_over=rr.log[0], coords_rand=rr.log[1], keep=keep)
    for b in range(B):
        arrs[f"i{b}"] = indices[b][0]
        arrs[f"j{b}"] = indices[b][1]
    save("loss", **arrs)
    # all-empty targets -> zeros (criterion.py:315-318)
    tg0 = [{"labels": t["labels"], "masks": torch.zeros_like(t["masks"])} for t in tg]
    lm0 = crit.loss_masks(outputs, tg0, indices, num_masks, False)
    assert float(lm0["loss_mask"]) == 0.0 and float(lm0["loss_dice"]) == 0.0


class _FakeSelf:
    pass


def g_kd_and_criterion():
    """End-to-end loss golden: the call order of criterion.py:390-427 (matcher ->
    loss_labels -> loss_masks -> per aux layer: matcher -> loss_masks), once with
    ground-truth targets and once with distillation targets built by the
    reference's own prepare_distillation_targets (kd_video_maskformer_model.py:
    418-528), then the renaming/weighting of :314-326.  VideoSetCriterion.forward
    itself raises AttributeError as shipped (criterion.py:380-385 references
    undefined loss_*_drop), so the harness walks the same order by hand."""
    cr = R.ref("mask2former_video.modeling.criterion")
    mt = R.ref("mask2former_video.modeling.matcher")
    kd = R.ref("mask2former_video.kd_video_maskformer_model")
    B, Q, T, h, w, H, W, P, seed = 2, 16, 2, 16, 24, 64, 96, 256, 71
    NL = 10
    ns = [3, 4]
    s_logits = torch.from_numpy(synth.randn(seed, 1, (NL, B, Q, 2)))
    s_masks = torch.from_numpy(synth.smooth_logits(seed, 2, (NL, B, Q, T), (h, w)))
    t_logits = torch.from_numpy(synth.randn(seed, 3, (B, Q, 2), 2.0))
    t_masks = torch.from_numpy(synth.smooth_logits(seed, 4, (B, Q, T), (h, w)))
    tg = make_targets(seed, 100, ns, T, H, W)

    fs = _FakeSelf()
    fs.teacher = [None, _FakeSelf()]
    fs.teacher[1].num_classes = 1
    fs.device = torch.device("cpu")
    fs.num_queries = Q
    fs.num_predictions_distillation = 100 if Q >= 100 else Q
    fs.num_frames = T
    images = _FakeSelf()
    images.tensor = torch.zeros(B * T, 3, H, W)
    kd_t = kd.KDVideoMaskFormer.prepare_distillation_targets(
        fs, {"pred_logits": t_logits, "pred_masks": t_masks}, images, None, nms=False, score_threshold=0.75)
    kd_save = {}
    for b in range(B):
        kd_save[f"kd_masks{b}"] = np.packbits(kd_t[b]["masks"].numpy().astype(np.uint8), axis=-1)
        kd_save[f"kd_n{b}"] = kd_t[b]["masks"].shape[0]
        # which queries were kept, in the reference's (implementation-defined, sorted=False) order
        sc = F.softmax(t_logits[b], -1)[:, 0]
        kept = [int(q) for q in range(Q) if sc[q] >= 0.75]
        # recover order by matching masks
        order = []
        up = F.interpolate(t_masks[b], size=(H, W), mode="bilinear", align_corners=False) > 0
        for k in range(kd_t[b]["masks"].shape[0]):
            for q in kept:
                if q not in order and torch.equal(up[q], kd_t[b]["masks"][k]):
                    order.append(q)
                    break
        kd_save[f"kd_order{b}"] = np.array(order, dtype=np.int64)
    print("   KD targets kept per clip:", [kd_save[f"kd_n{b}"] for b in range(B)])

    weight_dict = {"loss_ce": 2.0, "loss_mask": 5.0, "loss_dice": 5.0,
                   "kd_loss_ce": 0.0, "kd_loss_mask": 5.0, "kd_loss_dice": 5.0}
    aux = {}
    for i in range(NL - 1):
        aux.update({k + f"_{i}": v for k, v in weight_dict.items()})
    weight_dict.update(aux)
    matcher = mt.VideoHungarianMatcher(cost_class=2.0, cost_mask=5.0, cost_dice=5.0, num_points=P)
    crit = cr.VideoSetCriterion(1, matcher=matcher, weight_dict=weight_dict, eos_coef=0.1,
                                losses=["labels", "masks"], num_points=P, oversample_ratio=3.0,
                                importance_sample_ratio=0.75, loss_strategy="masks-only",
                                distillation_loss_strategy="masks-only")

    def run(targets, distillation, rr_seed):
        with RandRecorder(rr_seed) as rr:
            outputs = {"pred_logits": s_logits[-1], "pred_masks": s_masks[-1]}
            all_idx = []
            indices = matcher(outputs, targets)
            all_idx.append(indices)
            num_masks = max(float(sum(len(t["labels"]) for t in targets)), 1.0)
            losses = {}
            losses.update(crit.loss_labels(outputs, targets, indices, num_masks, distillation))
            losses.update(crit.loss_masks(outputs, targets, indices, num_masks, distillation))
            for i in range(NL - 1):
                auxo = {"pred_logits": s_logits[i], "pred_masks": s_masks[i]}
                indices = matcher(auxo, targets)
                all_idx.append(indices)
                ld = crit.loss_masks(auxo, targets, indices, num_masks, False)
                losses.update({k + f"_{i}": v for k, v in ld.items()})
        return losses, rr.log, all_idx

    losses, log_gt, idx_gt = run(tg, False, seed)
    kd_targets = [{"labels": t["labels"], "masks": t["masks"]} for t in kd_t]
    dl, log_kd, idx_kd = run(kd_targets, True, seed + 1)
    for k in list(dl.keys()):
        if k.startswith("loss_"):
            dl[k.replace("loss_", "kd_loss_")] = dl.pop(k)
    losses.update(dl)
    for k in list(losses.keys()):
        if k in weight_dict:
            losses[k] = losses[k] * weight_dict[k]
        else:
            losses.pop(k)
    arrs = dict(seed=seed, dims=np.array([B, Q, T, h, w, H, W, P, NL]), ns=np.array(ns))
    arrs.update(kd_save)
    for k, v in losses.items():
        arrs["L_" + k] = np.float32(v)
    for nm, log in (("gt", log_gt), ("kd", log_kd)):
        arrs[f"nrand_{nm}"] = len(log)
        for i, t in enumerate(log):
            arrs[f"rand_{nm}_{i}"] = t
    for nm, idxs in (("gt", idx_gt), ("kd", idx_kd)):
        for li, ind in enumerate(idxs):
            for b in range(B):
                arrs[f"idx_{nm}_{li}_{b}_i"] = ind[b][0]
                arrs[f"idx_{nm}_{li}_{b}_j"] = ind[b][1]
    save("criterion_kd", **arrs)


class _Inst:
    """Stand-in for detectron2 Instances/BitMasks: attribute bag with the members
    prepare_targets touches (kd_video_maskformer_model.py:358-386)."""

    def __init__(self, masks, ids, classes, image_size):
        self.gt_masks = _FakeSelf()
        self.gt_masks.tensor = masks
        self.gt_ids = ids
        self.gt_classes = classes
        self.image_size = image_size

    def to(self, device):
        return self

    def __len__(self):
        return self.gt_ids.shape[0]


def g_prepare_targets():
    kd = R.ref("mask2former_video.kd_video_maskformer_model")
    seed, T, H0, W0, Hp, Wp, n = 81, 3, 60, 90, 64, 96, 4
    m, ids = synth.ellipse_targets(seed, 1, n, T, H0, W0, sparse=0.6)
    ids[2, :] = -1
    m[2] = 0  # instance never present -> dropped (:376-380)
    fs = _FakeSelf()
    fs.num_frames = T
    fs.device = torch.device("cpu")
    images = _FakeSelf()
    images.tensor = torch.zeros(T, 3, Hp, Wp)
    inst = [_Inst(torch.from_numpy(m[:, t]).bool(), torch.from_numpy(ids[:, t]),
                  torch.zeros(n, dtype=torch.int64), (H0, W0)) for t in range(T)]
    out = kd.KDVideoMaskFormer.prepare_targets(fs, [{"instances": inst}], images)
    save("prepare_targets", seed=seed, dims=np.array([T, H0, W0, Hp, Wp, n]),
         masks=np.packbits(out[0]["masks"].numpy().astype(np.uint8), axis=-1),
         ids=out[0]["ids"], labels=out[0]["labels"], n_out=out[0]["masks"].shape[0])


def g_keymask():
    # cotracker_matching.py:176-209 (K4), :453-503 (K3), :640-662 (K5), :665-719 (K6)
    km = R.ref("cotracker_matching")
    seed, T, H, W, Np = 91, 6, 48, 64, 200
    rng = synth.rng_for(seed, 0)
    # id map: 3 moving ellipse objects, ids 1..3 (0 = background); T,H,W,1 int64
    m, _ = synth.ellipse_targets(seed, 1, 3, T, H, W, sparse=0.0, rmin=6, rmax=14)
    idmap = np.zeros((T, H, W, 1), np.int64)
    for o in range(3):
        idmap[..., 0][m[o] > 0] = o + 1
    # tracks: points inside object 1 + noise, some half-integers (round-half-even) and some out of bounds
    ys, xs = np.nonzero(m[0, 0])
    sel = rng.integers(0, len(ys), Np)
    base = np.stack([xs[sel], ys[sel]], -1).astype(np.float32)
    tracks = np.zeros((1, T, Np, 2), np.float32)
    for t in range(T):
        tracks[0, t] = base + rng.normal(0, 1.0, (Np, 2)).astype(np.float32) + np.float32(t * 0.7)
    tracks[0, :, :8] = np.round(tracks[0, :, :8]) + 0.5      # exact .5 -> half-to-even
    tracks[0, :, 8:12] = np.array([-3.0, 5.0])               # out of bounds (x<0)
    tracks[0, :, 12:14] = np.array([W - 0.5, H - 0.5])       # rounds to W / H-? -> edge cases
    tr = torch.from_numpy(tracks)
    track_masks = km.pred_tracks_to_binary_masks(tr, H, W, return_mask=False)
    # segm_mask at a different size than the id map -> exercises nearest resize (:687-689)
    H2, W2 = 36, 50
    segm = torch.zeros(H2, W2, dtype=torch.uint8)
    tr2 = tr * torch.tensor([W2 / W, H2 / H])
    ids_t = torch.from_numpy(idmap)
    glob = {}
    clus = {}
    km_get_overall = km.get_overall_maskid
    km_get_cluster = km.get_cluster_maskid
    km.get_overall_maskid = lambda *a: -1
    km.get_cluster_maskid = lambda *a: -1
    matches, allc = km.extract_mask_matches(segm, tr2, ids_t, 0, (1, 4), 25, glob, clus, 0, 0.5)
    matches_same, allc_same = km.extract_mask_matches(torch.zeros(H, W, dtype=torch.uint8), tr, ids_t, 0,
                                                      (0, T - 1), 25, glob, clus, 0, 0.5)
    km.get_overall_maskid = km_get_overall
    km.get_cluster_maskid = km_get_cluster

    def pack(lst):
        return np.array([[d["frame_id"], d["mask_id"], d["iou"]] for d in lst], np.float64).reshape(-1, 3)

    seg1 = km.get_segmentation_mask(ids_t, 2, 2)
    save("keymask", seed=seed, tracks=tracks, idmap=idmap.astype(np.int16), track_masks=track_masks,
         resized_dims=np.array([H2, W2]), allc_resized=pack(allc), matches_resized=pack(matches),
         allc_same=pack(allc_same), matches_same=pack(matches_same), segmask_f2_o2=seg1)


def g_grouping():
    """K7: identify_visibility_windows.py:108-231 and cotracker_matching.py:764-840 on synthetic curves / matches"""
    import json, tempfile
    iv = R.ref("identify_visibility_windows")
    km = R.ref("cotracker_matching")
    rng = synth.rng_for(95, 0)
    T, n_obj = 24, 3
    # per frame, n_obj masks; each mask's visibility curve = its object's occlusion schedule + noise
    sched = [(2, 14), (8, 23), (0, 23)]
    video_data, curves = [], []
    for f in range(T):
        data = []
        for o in range(n_obj):
            c = np.zeros(T, np.float32)
            c[sched[o][0]:sched[o][1] + 1] = 1.0
            c = np.clip(c * rng.uniform(0.6, 1.0, T) + rng.uniform(0, 0.15, T), 0, 1).astype(np.float32)
            data.append({"object_id": o + 1, "visibility": c.tolist()})
            curves.append(c)
        video_data.append({"frame_id": f, "data": data})
    with tempfile.TemporaryDirectory() as td:
        out = iv.get_visibility_windows_for_video({"video_data": video_data}, "ds", "train", "vid", td, 0.3)
    # match matrix: 3 groups of masks that matched each other + 2 unmatched rows
    n = 30
    grp = np.repeat(np.arange(3), 10)
    mm = (grp[:, None] == grp[None, :]) & (rng.random((n, n)) > 0.08)
    mm[7, :] = False; mm[19, :] = False
    matches_data = []
    lookup = [[{"frame_id": i // 3, "mask_id": i % 3 + 1, "overall_mask_id": i} for i in range(n)]]
    for i in range(n):
        matches_data.append({"cluster_id": 0, "overall_mask_id": i,
                             "matches": [{"overall_mask_id": int(j)} for j in np.nonzero(mm[i])[0]]})
    cids, vt = km.temporal_correspondance_clustering(matches_data, lookup, False)
    lab = {}
    for l, lst in vt[0]["overall_mask_ids_per_label"].items():
        for e in lst:
            lab[(e[0], e[1]) if isinstance(e, (tuple, list)) else str(e)] = l
    save("grouping", curves=np.stack(curves), clusters_json=np.frombuffer(json.dumps(out["clusters"]).encode(), np.uint8),
         match_matrix=mm.astype(np.float32), factor=vt[0]["visibility_to_temporal_factor"],
         groups_json=np.frombuffer(json.dumps({str(k): v for k, v in vt[0]["overall_mask_ids_per_label"].items()}).encode(), np.uint8))


def _infer_inputs(seed, Q, C, T, h, w, dup):
    """class logits [Q,C+1] and low-resolution mask logits [Q,T,h,w]: soft ellipses (logit = margin in px); the queries
    listed in `dup` repeat another query's mask with a small shift so that mask-NMS has something to suppress."""
    rng = synth.rng_for(seed, 7)
    cls = rng.normal(0, 2.0, (Q, C + 1)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    masks = np.empty((Q, T, h, w), np.float32)
    par = []
    for q in range(Q):
        par.append((rng.uniform(0.25 * h, 0.75 * h), rng.uniform(0.25 * w, 0.75 * w), rng.uniform(0.12 * h, 0.3 * h),
                    rng.uniform(0.12 * w, 0.3 * w), rng.uniform(-0.6, 0.6, (T, 2))))
    for q, src, sh in dup:
        cy, cx, ry, rx, dr = par[src]
        par[q] = (cy + sh, cx - sh, ry, rx, dr)
    for q in range(Q):
        cy, cx, ry, rx, dr = par[q]
        for t in range(T):
            d = np.sqrt(((yy - cy - dr[:t + 1, 0].sum()) / ry) ** 2 + ((xx - cx - dr[:t + 1, 1].sum()) / rx) ** 2)
            masks[q, t] = (1.0 - d) * 4.0 + rng.normal(0, 0.05, (h, w))
    return cls, masks


def g_inference():
    """Eval branch: the upsample of kd_video_maskformer_model.py:340-346 followed by inference_video (:530-610):
    softmax scores, sorted top-k over Q x classes, crop to the unpadded size, bilinear resize to the output size,
    > 0, optional greedy same-label mask-NMS."""
    kd = R.ref("mask2former_video.kd_video_maskformer_model")
    cases = {
        # name: (seed, Q, C, K, T, h, w, Hp, Wp, ih, iw, oh, ow, use_nms, thr, dup)
        "agn_nms": (91, 12, 1, 6, 3, 16, 24, 64, 96, 60, 90, 90, 135, True, 0.75, [(1, 0, 0.4), (5, 4, 0.3), (7, 4, 0.5)]),
        "agn_plain": (92, 12, 1, 6, 3, 16, 24, 64, 96, 60, 90, 60, 90, False, 0.75, []),
        "multi_nms": (93, 10, 3, 8, 2, 16, 24, 64, 96, 64, 96, 48, 70, True, 0.5, [(2, 0, 0.3), (3, 0, 0.6), (9, 8, 0.2)]),
        "down_nms": (94, 16, 1, 10, 4, 24, 40, 96, 160, 90, 157, 45, 80, True, 0.75, [(3, 2, 0.2), (11, 10, 0.4)]),
    }
    arrs = {}
    for name, (seed, Q, C, K, T, h, w, Hp, Wp, ih, iw, oh, ow, use_nms, thr, dup) in cases.items():
        cls, masks = _infer_inputs(seed, Q, C, T, h, w, dup)
        if name == "multi_nms":            # duplicated masks must also share their best label, or label-aware NMS never fires
            for q, src, _ in dup:
                cls[q] = cls[src]
                cls[q, -1] += np.float32(0.05) * (q + 1)          # same best label, strictly lower score (no ties)
        fs = _FakeSelf()
        fs.teacher = [None, _FakeSelf()]
        fs.teacher[1].num_classes = C
        fs.device = torch.device("cpu")
        fs.num_queries = Q
        fs.num_predictions_inference = K
        fs.use_nms, fs.nms_threshold = use_nms, thr
        up = F.interpolate(torch.from_numpy(masks), size=(Hp, Wp), mode="bilinear", align_corners=False)   # :341-346
        out = kd.KDVideoMaskFormer.inference_video(fs, torch.from_numpy(cls), up, (ih, iw), oh, ow)
        assert out["image_size"] == (oh, ow)
        n = len(out["pred_scores"])
        arrs[f"{name}_dims"] = np.array([seed, Q, C, K, T, h, w, Hp, Wp, ih, iw, oh, ow, int(use_nms)])
        arrs[f"{name}_thr"] = np.float32(thr)
        arrs[f"{name}_cls"] = cls
        arrs[f"{name}_masks"] = masks
        arrs[f"{name}_scores"] = np.array(out["pred_scores"], np.float32)
        arrs[f"{name}_labels"] = np.array(out["pred_labels"], np.int64)
        arrs[f"{name}_out"] = np.packbits(torch.stack(out["pred_masks"]).numpy().astype(np.uint8), axis=-1) if n else np.zeros((0,), np.uint8)
        print(f"    {name}: kept {n} of {K}")
    save("inference", **arrs)


def g_idmaps():
    """cotracker_matching.py:22-84 load_masks: ordered multi-colour PNG masks -> per-frame id maps (black 0, the other
    colours 1..n in lexicographic (R,G,B) order, numbered per frame).  The PNGs are written with PIL; OpenCV (absent) is
    represented by three name-only stand-ins with its documented semantics: imread -> BGR array, cvtColor = channel swap."""
    import tempfile
    from PIL import Image
    cv2 = sys.modules["cv2"]
    cv2.IMREAD_COLOR, cv2.COLOR_BGR2RGB, cv2.COLOR_RGB2BGR = 1, 4, 4
    cv2.imread = lambda p, flag=1: np.ascontiguousarray(np.array(Image.open(p).convert("RGB"))[..., ::-1])
    cv2.cvtColor = lambda img, code: np.ascontiguousarray(img[..., ::-1])
    km = R.ref("cotracker_matching")
    seed, T, H, W = 97, 5, 40, 52
    rng = synth.rng_for(seed, 0)
    palette = rng.integers(0, 256, (9, 3)).astype(np.uint8)
    palette[0] = (0, 0, 0)
    palette[1] = (0, 0, 1)          # nearly black, blue only
    palette[2] = (1, 0, 0)          # sorts after every (0, *, *) colour
    palette[3] = (0, 255, 255)
    palette[4] = (255, 255, 255)
    frames = np.zeros((T, H, W, 3), np.uint8)
    for t in range(T):
        lab = np.zeros((H, W), np.int64)
        use = rng.permutation(np.arange(1, 9))[:int(rng.integers(0, 8))] if t != 2 else np.array([], np.int64)   # frame 2: all black
        for c in use:
            cy, cx, ry, rx = rng.uniform(5, H - 5), rng.uniform(5, W - 5), rng.uniform(3, 10), rng.uniform(3, 12)
            yy, xx = np.mgrid[0:H, 0:W]
            lab[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1] = c
        frames[t] = palette[lab]
    with tempfile.TemporaryDirectory() as d:
        for t in range(T):
            Image.fromarray(frames[t]).save(os.path.join(d, f"frame{t:04d}.png"))
        open(os.path.join(d, "notes.txt"), "w").write("ignored")
        out = km.load_masks(d)
    assert out.shape == (T, H, W, 1) and out.dtype == torch.int64
    save("idmaps", seed=seed, frames=frames, ids=out.numpy().astype(np.int16), n_ids=np.array([int(out[t].max()) for t in range(T)]))


from formats_case import formats_case as _formats_case, tree as _tree  # noqa: E402


def g_formats():
    """On-disk formats either side of keymask discovery, written by the reference's own functions into temporary directories:
    keymask_utils.save_segmentation_masks (:70-126), cotracker_matching.save_temporal_group_masks (:402-431),
    annotations.write_annotation_for_video (:8-139), merge_ytvis_jsons.main (:24-96).  The fixture holds the resulting file
    lists, PNG pixels and JSON documents.  Third-party names absent from the image are stood in for by name: imageio (imported,
    unused on this path) and pycocotools.mask, whose encode / area / toBbox are served by the oracle's restatement of maskApi.c
    (so the RLE strings inside the annotation JSON are the oracle's: parity unpinned, as everywhere for that format)."""
    import json
    import tempfile
    import types
    from PIL import Image
    from oracle import oracle_np as O
    R.install()
    sys.modules.setdefault("imageio", types.ModuleType("imageio"))
    pm = types.ModuleType("pycocotools.mask")

    def encode(arr):                                                   # [H,W,F] Fortran uint8 -> list of RLE dicts
        return [dict(size=[arr.shape[0], arr.shape[1]], counts=O.rle_encode(np.ascontiguousarray(arr[..., f]))[0]["counts"]) for f in range(arr.shape[2])]
    pm.encode = encode
    pm.area = lambda r: O.rle_area_bbox(O.rle_decode(r))[0]
    pm.toBbox = lambda r: np.asarray(O.rle_area_bbox(O.rle_decode(r))[1], np.float64)
    pk = types.ModuleType("pycocotools"); pk.mask = pm
    sys.modules["pycocotools"], sys.modules["pycocotools.mask"] = pk, pm
    ku, cm, an, mg = R.ref("keymask_utils"), R.ref("cotracker_matching"), R.ref("annotations"), R.ref("merge_ytvis_jsons")
    c = _formats_case()
    T, H, W = c["T"], c["H"], c["W"]
    out = {}
    with tempfile.TemporaryDirectory() as d:
        lbls = torch.from_numpy(c["ids"])
        vdir = ku.save_segmentation_masks(torch.zeros(T, 3, H, W), torch.zeros(T, 3, H, W), lbls, {"visibility": c["visibility"]}, os.path.join(d, "masks"))
        files, arrays = _tree(os.path.join(d, "masks"))
        out["seg_files"], out["seg_video_dir"] = files, os.path.relpath(vdir, d)
        seg_arrays = arrays
        # cluster_masks for the grouping step: per cluster the list of {'frame_id','mask_id','mask'} (what load_cluster_masks builds)
        cluster_masks = []
        for cid in range(2):
            lst = []
            for rel, a in sorted(arrays.items()):
                m = __import__("re").match(rf"vid_0007/cluster_{cid}/cluster{cid}_frame(\d+)_mask(-?\d+)\.png", rel)
                if m:
                    lst.append({"frame_id": int(m.group(1)), "mask_id": int(m.group(2)), "mask": a})
            cluster_masks.append(lst)
        gpath = os.path.join(d, "masks", "vid_0007")
        os.makedirs(os.path.join(gpath, "cluster_0", "group_9"))        # a stale group directory must disappear
        cm.save_temporal_group_masks(c["groupings"], cluster_masks, gpath)
        files2, arrays2 = _tree(gpath)
        out["group_files"] = files2
        json.dump(c["one2x"], open(os.path.join(gpath, "video_one2x_data.json"), "w"))
        vpath = os.path.join(d, "frames", "vid_0007")
        os.makedirs(vpath)
        for t in range(T):
            Image.fromarray(np.zeros((H, W, 3), np.uint8)).save(os.path.join(vpath, f"{t:05d}.jpg"))
        an.write_annotation_for_video(vpath, gpath, os.path.join(d, "ann"), c["visibility"])
        out["annotation"] = json.load(open(os.path.join(d, "ann", "vid_0007.json")))
        src = os.path.join(d, "per_video")
        os.makedirs(src)
        for i, doc in enumerate(c["merge_inputs"]):
            json.dump(doc, open(os.path.join(src, f"video_{i:02d}.json"), "w"))
        for name, thr in (("merged_all", -1.0), ("merged_filtered", 0.5)):
            mg.main(src, os.path.join(d, name + ".json"), thr)
            out[name] = json.load(open(os.path.join(d, name + ".json")))
        np.savez_compressed(os.path.join(HERE, "formats_png.npz"), **{"seg/" + k: v for k, v in seg_arrays.items()},
                            **{"grp/" + k: v for k, v in arrays2.items()})
    with open(os.path.join(HERE, "formats.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("  wrote formats.json, formats_png.npz")


def _sampling_cases():
    """video annotation skeletons (only the ids per frame matter) for the frame-selection goldens"""
    rng = np.random.default_rng(17)
    cases = []
    for L in (12, 30, 7, 3):
        annos = []
        for t in range(L):
            present = [i for i in range(4) if rng.random() < (0.75 if L != 7 else 0.3)]
            annos.append([{"id": i} for i in present])
        cases.append((L, annos))
    return cases


_DM = []


def _load_dataset_mapper():
    """the reference's data_video/dataset_mapper.py, loaded where it lies; detectron2 / fvcore names it imports are stood in for by
    name only (none of them is touched by the functions the generators call)"""
    import types
    if _DM:
        return _DM[0]
    R.install()
    class _Aug:
        def _init(self, params=None):
            if params:
                for k, v in params.items():
                    if k != "self" and not k.startswith("_"):
                        setattr(self, k, v)

        def _rand_range(self, low=1.0, high=None, size=None):
            if high is None:
                low, high = 0, low
            return np.random.uniform(low, high, size)

    class _Tf:
        def __init__(self, *a, **k):
            self.args = a

        def _set_attributes(self, params=None):
            if params:
                for k, v in params.items():
                    if k != "self" and not k.startswith("_"):
                        setattr(self, k, v)

        @classmethod
        def register_type(cls, *a, **k):
            pass
    T = types.ModuleType("detectron2.data.transforms"); T.Augmentation = _Aug; T.Transform = _Tf; T.AugmentationList = list
    T.RandomCrop = T.RandomBrightness = T.RandomContrast = T.RandomSaturation = T.RandomRotation = T.ResizeShortestEdge = _Tf
    dd = sys.modules["detectron2.data"]; dd.transforms = T; dd.detection_utils = types.ModuleType("detectron2.data.detection_utils")
    sys.modules["detectron2.data.transforms"], sys.modules["detectron2.data.detection_utils"] = T, dd.detection_utils
    st = sys.modules["detectron2.structures"]; st.BoxMode = object
    ft = types.ModuleType("fvcore.transforms"); ftt = types.ModuleType("fvcore.transforms.transform")
    for n in ("BlendTransform", "TransformList", "HFlipTransform", "NoOpTransform", "VFlipTransform", "CropTransform", "Transform"):
        setattr(ft, n, _Tf); setattr(ftt, n, _Tf)
    ft.transform = ftt
    sys.modules["fvcore.transforms"], sys.modules["fvcore.transforms.transform"] = ft, ftt
    del sys.modules["mask2former_video.data_video.dataset_mapper"]          # the shim's name-only stand-in: load the real file here
    _DM.append(R.ref("mask2former_video.data_video.dataset_mapper"))
    return _DM[0]


def g_sampling():
    """dataset_mapper.py:223-289 dense_frame_selection / random_frame_selection and augmentation.py:51-75 (the size rule of
    ResizeShortestEdge) called on the reference's own classes.  detectron2 / fvcore names the two files import are stood in for
    by name only (none of them is touched by these methods)."""
    import json
    import random
    import types
    R.install()

    dm = _load_dataset_mapper()
    au = R.ref("mask2former_video.data_video.augmentation")
    out = {"dense": [], "random": [], "resize": []}
    for ci, (L, annos) in enumerate(_sampling_cases()):
        for n, fr, shuffle in ((3, 5, False), (2, 2, True), (5, 20, False)):
            self_ = types.SimpleNamespace(sampling_frame_num=n, sampling_frame_range=fr, sampling_frame_shuffle=shuffle)
            for seed in range(6):
                random.seed(seed); np.random.seed(seed)
                sel = dm.YTVISDatasetMapper.dense_frame_selection(self_, annos, L)
                out["dense"].append({"case": ci, "n": n, "range": fr, "shuffle": shuffle, "seed": seed, "sel": [int(v) for v in sel]})
                if L > 2 * fr or n - 1 <= L:
                    random.seed(seed); np.random.seed(seed)
                    try:
                        sel = dm.YTVISDatasetMapper.random_frame_selection(self_, L)
                        out["random"].append({"L": L, "n": n, "range": fr, "shuffle": shuffle, "seed": seed, "sel": [int(v) for v in sel]})
                    except ValueError:
                        pass
    for style, sizes in (("choice_by_clip", (360, 480)), ("range", (320, 640)), ("choice", (288, 320, 352))):
        for (h, w) in ((720, 1280), (480, 854), (1080, 608)):
            for max_size in (10 ** 9, 768):
                r = au.ResizeShortestEdge(sizes, max_size, style, clip_frame_cnt=3 if "by_clip" in style else 1)
                np.random.seed(7)
                got = []
                for _ in range(6):
                    tr = r.get_transform(np.zeros((h, w, 3), np.uint8))
                    got.append([int(tr.new_h), int(tr.new_w)])
                out["resize"].append({"style": style, "sizes": list(sizes), "hw": [h, w], "max_size": max_size, "new_hw": got})
    with open(os.path.join(HERE, "sampling.json"), "w") as f:
        json.dump(out, f)
    print("  wrote sampling.json:", {k: len(v) for k, v in out.items()})


from copy_paste_cases import COPY_PASTE_CASES, copy_paste_case  # noqa: E402,F401


def _d2_structures():
    """Stand-ins for detectron2.structures.{Boxes, BitMasks, Instances} (third party, absent): the documented behaviour of the few
    operations the reference's data-side functions call -- field dict + indexing + cat; .tensor, nonempty, get_bounding_boxes;
    scale, nonempty.  Test infrastructure of the golden generators only."""
    class Boxes:
        def __init__(self, t):
            self.tensor = t

        device = property(lambda self: self.tensor.device)

        def scale(self, sx, sy):
            self.tensor[:, 0::2] *= sx
            self.tensor[:, 1::2] *= sy

        def nonempty(self, threshold=0.0):
            box = self.tensor
            return ((box[:, 2] - box[:, 0]) > threshold) & ((box[:, 3] - box[:, 1]) > threshold)

        def __getitem__(self, i):
            t = self.tensor[i]
            return Boxes(t[None] if t.dim() == 1 else t)

        def __len__(self):
            return self.tensor.shape[0]

        def to(self, *a, **k):
            return Boxes(self.tensor.to(*a, **k))

        @staticmethod
        def cat(bs):
            return Boxes(torch.cat([b.tensor for b in bs], 0))

    class BitMasks:
        def __init__(self, t):
            self.tensor = torch.as_tensor(t).to(torch.bool)

        device = property(lambda self: self.tensor.device)

        def __getitem__(self, i):
            t = self.tensor[i]
            return BitMasks(t[None] if t.dim() == 2 else t)

        def __len__(self):
            return self.tensor.shape[0]

        def to(self, *a, **k):
            return BitMasks(self.tensor.to(*a, **k))

        def nonempty(self):
            return self.tensor.flatten(1).any(dim=1)

        def get_bounding_boxes(self):
            b = torch.zeros((self.tensor.shape[0], 4), dtype=torch.float32)
            xa, ya = self.tensor.any(1), self.tensor.any(2)
            for i in range(self.tensor.shape[0]):
                x, y = torch.where(xa[i])[0], torch.where(ya[i])[0]
                if len(x) and len(y):
                    b[i] = torch.as_tensor([x[0], y[0], x[-1] + 1, y[-1] + 1], dtype=torch.float32)
            return Boxes(b)

        @staticmethod
        def cat(ms):
            return BitMasks(torch.cat([m.tensor for m in ms], 0))

    class Instances:
        def __init__(self, image_size, **kw):
            object.__setattr__(self, "_image_size", image_size)
            object.__setattr__(self, "_fields", {})
            for k, v in kw.items():
                self.set(k, v)

        image_size = property(lambda self: self._image_size)

        def __setattr__(self, name, val):
            if name.startswith("_"):
                object.__setattr__(self, name, val)
            else:
                self.set(name, val)

        def __getattr__(self, name):
            if name == "_fields" or name not in self._fields:
                raise AttributeError(name)
            return self._fields[name]

        def set(self, name, value):
            self._fields[name] = value

        def has(self, name):
            return name in self._fields

        def get(self, name):
            return self._fields[name]

        def get_fields(self):
            return self._fields

        def to(self, *a, **k):
            r = Instances(self._image_size)
            for kk, v in self._fields.items():
                r.set(kk, v.to(*a, **k) if hasattr(v, "to") else v)
            return r

        def __getitem__(self, item):
            if isinstance(item, int):
                item = slice(item, None, len(self))
            r = Instances(self._image_size)
            for k, v in self._fields.items():
                r.set(k, v[item])
            return r

        def __len__(self):
            for v in self._fields.values():
                return len(v)
            raise NotImplementedError("Empty Instances does not support __len__!")

        @staticmethod
        def cat(lst):
            r = Instances(lst[0]._image_size)
            for k in lst[0]._fields:
                vs = [i.get(k) for i in lst]
                r.set(k, torch.cat(vs, 0) if isinstance(vs[0], torch.Tensor) else type(vs[0]).cat(vs))
            return r

    return Boxes, BitMasks, Instances


def g_copy_paste():
    """engine/train_loop.py:30-156 propagate_sparse_masks and :377-590 CustomSimpleTrainer.copy_and_paste, called as they lie.
    detectron2's structures are absent: Instances / BitMasks / Boxes are stood in for by small classes with the documented
    behaviour of the few operations the two functions use (field dict + indexing + cat; .tensor + get_bounding_boxes; scale) --
    the pixel results pinned here (composited frames, instance masks, ids, the fall-back decisions and the state of both RNG
    streams after the call) do not depend on anything else.  tests/golden/copy_paste.npz."""
    import copy
    import random
    import types
    R.install()

    Boxes, BitMasks, Instances = _d2_structures()
    for name, attrs in {"detectron2.utils.events": dict(get_event_storage=lambda: None), "detectron2.engine": dict(SimpleTrainer=object),
                        "detectron2.structures.instances": dict(Instances=Instances)}.items():
        m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m
    st = sys.modules["detectron2.structures"]; st.BitMasks = BitMasks; st.Boxes = Boxes; st.Instances = Instances
    sys.modules["detectron2.utils"].comm = sys.modules["detectron2.utils.comm"]
    m = types.ModuleType("mask2former_video.data_video.build"); m.get_detection_dataset_dicts = None; m.build_detection_train_loader = None
    sys.modules["mask2former_video.data_video.build"] = m
    sys.modules["mask2former_video.data_video.dataset_mapper"].YTVISDatasetMapper = object
    R._pkg("mask2former_video.engine", os.path.join(R.MT, "mask2former_video", "engine"))
    tl = R.ref("mask2former_video.engine.train_loop")

    def to_ref(clip, hw):
        inst = []
        for fr in clip["instances"]:
            i = Instances(tuple(hw))
            m = BitMasks(torch.from_numpy(fr["gt_masks"]))
            i.gt_masks = m
            i.gt_boxes = m.get_bounding_boxes()
            i.gt_classes = torch.from_numpy(fr["gt_classes"])
            i.gt_ids = torch.from_numpy(fr["gt_ids"])
            inst.append(i)
        return {"image": [torch.from_numpy(f.copy()) for f in clip["image"]], "instances": inst}

    out = {}
    for case in COPY_PASTE_CASES:
        src, tgt = copy_paste_case(case)
        c = case["cfg"]
        self_ = types.SimpleNamespace(cfg_COPY_PASTE_RATE=c["rate"], cfg_COPY_PASTE_RANDOM_NUM=c["random_num"], cfg_COPY_PASTE_MIN_RATIO=c["lo"],
                                      cfg_COPY_PASTE_MAX_RATIO=c["hi"], cfg_COPY_PASTE_DENSIFY_SPARSE=c["densify"], cfg_VISUALIZE_COPY_PASTE=False)
        random.seed(case["seed"]); np.random.seed(case["seed"])
        res = tl.CustomSimpleTrainer.copy_and_paste(self_, [to_ref(src, case["src_hw"])], [to_ref(tgt, case["hw"])])[0]
        n = case["name"]
        out[f"{n}.rng"] = np.array([random.random(), np.random.rand()])          # the state both streams are left in
        out[f"{n}.image"] = np.stack([f.numpy() for f in res["image"]])
        for t, fr in enumerate(res["instances"]):
            out[f"{n}.masks{t}"] = np.packbits(fr.gt_masks.tensor.numpy().astype(np.uint8), axis=-1)
            out[f"{n}.ids{t}"] = fr.gt_ids.numpy().astype(np.int64)
            out[f"{n}.classes{t}"] = fr.gt_classes.numpy().astype(np.int64)
        print("  ", n, "instances per frame", [len(fr) for fr in res["instances"]], "image changed", bool((out[f"{n}.image"] != np.stack(tgt["image"])).any()))
        # propagate_sparse_masks on its own, on the case's raw target clip
        random.seed(case["seed"] + 1000)
        pr = tl.propagate_sparse_masks(to_ref(tgt, case["hw"])["instances"], max_shift=2)
        out[f"{n}.prop_rng"] = np.array([random.random()])
        for t, fr in enumerate(pr):
            out[f"{n}.prop_masks{t}"] = np.packbits(fr.gt_masks.tensor.numpy().astype(np.uint8), axis=-1)
            out[f"{n}.prop_ids{t}"] = fr.gt_ids.numpy().astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "copy_paste.npz"), **out)



from make_golden_cases import assemble_cases  # noqa: E402,F401


def g_assemble():
    """dataset_mapper.py:29-56 `filter_empty_instances` -- the reference's own function in that file -- applied to the per-frame
    Instances of seeded clips (BitMasks / Boxes stand-ins as above), slots numbered as :297-303 number them.  Pins the gt_ids the
    mapper hands to the model (absent, crowd, empty-mask and degenerate-box instances become -1).  tests/golden/assemble.json."""
    import json
    Boxes, BitMasks, Instances = _d2_structures()
    dm = _load_dataset_mapper()
    out = []
    for (H, W), video, sel in assemble_cases():
        _ids = set()
        for f in sel:
            _ids.update([a["id"] for a in video[f]])
        ids = {_id: i for i, _id in enumerate(_ids)}
        frames = []
        for f in sel:
            masks = torch.zeros((len(ids), H, W), dtype=torch.bool)
            gt = [-1] * len(ids)
            for a in video[f]:
                if a.get("iscrowd", 0) == 0:
                    masks[ids[a["id"]]] = torch.from_numpy(a["mask"]); gt[ids[a["id"]]] = a["id"]
            inst = Instances((H, W))
            inst.gt_masks = BitMasks(masks)
            inst.gt_ids = torch.tensor(gt)
            inst.gt_boxes = inst.gt_masks.get_bounding_boxes()
            inst = dm.filter_empty_instances(inst)
            frames.append([int(v) for v in inst.gt_ids.tolist()])
        out.append({"slots": {str(k): v for k, v in ids.items()}, "gt_ids": frames})
    with open(os.path.join(HERE, "assemble.json"), "w") as f:
        json.dump(out, f)


def g_config():
    """the shipped KD training configuration as the trainer resolves it: configs/imagenet_video/
    ytvis2021_kd_video_mask2former_R50_cls_agnostic.yaml merged over its _BASE_ (yaml data, key -> value; python tuples
    become lists).  The from_config methods of the drop-in meta-architecture are tested on exactly these keys."""
    import json
    import yaml

    class L(yaml.SafeLoader):
        pass
    L.add_constructor("tag:yaml.org,2002:python/tuple", lambda l, n: list(l.construct_sequence(n)))
    d = os.path.join(R.ROOT if hasattr(R, "ROOT") else "/root/reference", "model_training", "configs", "imagenet_video")

    def fix(v):
        if isinstance(v, dict):
            return {k: fix(x) for k, x in v.items()}
        if isinstance(v, str) and v.startswith("(") and v.endswith(")"):          # yacs evaluates "(360, 480)" literals
            import ast
            return list(ast.literal_eval(v))
        return v

    def load(name):
        c = fix(yaml.load(open(os.path.join(d, name)), Loader=L))
        base = c.pop("_BASE_", None)
        if base:
            b = load(base)

            def merge(a, o):
                for k, v in o.items():
                    if isinstance(v, dict) and isinstance(a.get(k), dict):
                        merge(a[k], v)
                    else:
                        a[k] = v
                return a
            c = merge(b, c)
        return c
    cfg = load("ytvis2021_kd_video_mask2former_R50_cls_agnostic.yaml")
    with open(os.path.join(HERE, "kd_config.json"), "w") as f:
        json.dump(cfg, f, indent=1, sort_keys=True)


def main():
    assert R.available(), "/root/reference not present: goldens can only be generated in the build container"
    R.install()
    only = set(sys.argv[1:])
    for fn in (g_msda, g_pe, g_pixel_decoder, g_video_decoder, g_matcher, g_loss, g_kd_and_criterion,
               g_prepare_targets, g_keymask, g_grouping, g_inference, g_idmaps, g_config, g_formats, g_sampling, g_copy_paste, g_assemble):
        if only and fn.__name__ not in only:
            continue
        print(fn.__name__)
        fn()


if __name__ == "__main__":
    main()

"""Data-side device kernels (csrc/augment.hip): clip augmentation against the oracle's restatement (and
against torch's own bilinear resize for the pure-resize case), and the video copy-paste against the reference's arithmetic,
which is plain torch (engine/train_loop.py:461-548: F.interpolate(...).byte() / .bool(), canvas placement, alpha composite)."""
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from s2d_amd.utils import synth

pytestmark = pytest.mark.gpu


def _clip(seed, T, H, W, n):
    fr = synth.smooth_frames_u8(seed, 1, T, H, W)                       # [T,3,H,W]
    m, _ = synth.ellipse_targets(seed, 2, n, T, H, W, sparse=0.0)      # [n,T,H,W]
    return fr, (m > 0).astype(np.uint8)


def test_clip_augmentation_vs_oracle(oracle):
    from s2d_amd.data import ClipAugmentation, augment_clip
    T, H0, W0, n = 3, 120, 200, 4
    fr, m = _clip(21, T, H0, W0, n)
    aug = ClipAugmentation(min_size=(72, 96), random_flip="flip_by_clip", augmentations=("brightness", "contrast", "rotation"),
                           crop=("absolute_range", (90, 110)), num_frames=T)
    np.random.seed(5)
    for _ in range(3):
        P, hw = aug.sample(T, H0, W0)
        got_f, got_m = augment_clip(torch.from_numpy(fr).cuda(), torch.from_numpy(m).cuda(), P, hw)
        ref_f = oracle.aug_warp_frames(fr, P, hw)
        ref_m = oracle.aug_warp_masks(m, P, hw)
        d = np.abs(got_f.cpu().numpy().astype(np.int32) - ref_f.astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() < 1e-3                  # contrast mean: float32 vs float64 reduction order
        np.testing.assert_array_equal(got_m.cpu().numpy(), ref_m)
        assert got_f.shape == (T, 3) + tuple(hw) and got_m.shape == (n, T) + tuple(hw)
    # pure resize (+ flip): the same taps and weights as torch's bilinear resize of the clip
    aug = ClipAugmentation(min_size=(60,), sample_style="choice_by_clip", random_flip="none", num_frames=T)
    P, hw = aug.sample(T, H0, W0)
    got_f, got_m = augment_clip(torch.from_numpy(fr).cuda(), torch.from_numpy(m).cuda(), P, hw)
    want = torch.round(F.interpolate(torch.from_numpy(fr).float(), size=hw, mode="bilinear", align_corners=False)).to(torch.uint8)
    d = (got_f.cpu().int() - want.int()).abs()
    assert int(d.max()) <= 1 and float((d != 0).float().mean()) < 2e-3    # x.5 ties after float rounding
    np.testing.assert_array_equal(got_f.cpu().numpy(), oracle.aug_warp_frames(fr, P, hw))


def _golden_cases():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from copy_paste_cases import COPY_PASTE_CASES, copy_paste_case
    return COPY_PASTE_CASES, copy_paste_case


def _to_dev(clip):
    return {"image": [torch.from_numpy(f.copy()).cuda() for f in clip["image"]],
            "instances": [{"gt_masks": torch.from_numpy(fr["gt_masks"]).cuda(), "gt_ids": fr["gt_ids"].copy(), "gt_classes": fr["gt_classes"].copy()}
                          for fr in clip["instances"]]}


def test_copy_and_paste_vs_reference_goldens():
    """`CustomSimpleTrainer.copy_and_paste` (engine/train_loop.py:377-590) on the device against what the reference's own function
    returned for the same seeded clips and the same `random` / `numpy.random` seeds (tests/golden/copy_paste.npz): every frame's
    composite image, every instance mask, ids and classes, the fall-back decisions -- and the state both RNG streams are left in,
    i.e. the draws are consumed exactly as the reference consumes them (also on a rate miss and with no source instance).  Cases:
    plain paste, sparse targets, COPY_PASTE_RANDOM_NUM, cancel by overlap, rate miss, DENSIFY_SPARSE, no source / no target
    instances."""
    from tests.conftest import golden
    from s2d_amd.data import copy_and_paste
    g = golden("copy_paste")
    cases, make = _golden_cases()
    for case in cases:
        n, c = case["name"], case["cfg"]
        src, tgt = make(case)
        random.seed(case["seed"]); np.random.seed(case["seed"])
        out = copy_and_paste([_to_dev(src)], [_to_dev(tgt)], rate=c["rate"], random_num=c["random_num"], min_ratio=c["lo"], max_ratio=c["hi"],
                             densify_sparse=c["densify"])[0]
        np.testing.assert_array_equal(np.array([random.random(), np.random.rand()]), g[f"{n}.rng"], err_msg=f"{n}: RNG streams")
        img = np.stack([f.cpu().numpy() for f in out["image"]])
        d = np.abs(img.astype(np.int32) - g[f"{n}.image"].astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() < 1e-3, (n, int(d.max()), float((d != 0).mean()))    # .byte() of a float a hair from an integer
        W = img.shape[-1]
        for t, fr in enumerate(out["instances"]):
            want = np.unpackbits(g[f"{n}.masks{t}"], axis=-1)[..., :W].astype(bool)
            got = fr["gt_masks"].cpu().numpy().astype(bool)
            assert got.shape == want.shape, (n, t, got.shape, want.shape)
            np.testing.assert_array_equal(got, want, err_msg=f"{n}: masks of frame {t}")
            np.testing.assert_array_equal(np.asarray(fr["gt_ids"]), g[f"{n}.ids{t}"], err_msg=f"{n}: ids of frame {t}")
            np.testing.assert_array_equal(np.asarray(fr["gt_classes"]), g[f"{n}.classes{t}"], err_msg=f"{n}: classes of frame {t}")


def test_propagate_sparse_masks_vs_reference_goldens():
    """engine/train_loop.py:30-156 on the raw target clips of the same cases: filled masks (the +-2 px jitter draws included), id
    order, and the RNG state afterwards"""
    from tests.conftest import golden
    from s2d_amd.data import propagate_sparse_masks
    g = golden("copy_paste")
    cases, make = _golden_cases()
    for case in cases:
        n = case["name"]
        _, tgt = make(case)
        random.seed(case["seed"] + 1000)
        out = propagate_sparse_masks(_to_dev(tgt)["instances"], max_shift=2)
        np.testing.assert_array_equal(np.array([random.random()]), g[f"{n}.prop_rng"])
        W = tgt["image"][0].shape[-1]
        for t, fr in enumerate(out):
            want = np.unpackbits(g[f"{n}.prop_masks{t}"], axis=-1)[..., :W].astype(bool)
            np.testing.assert_array_equal(fr["gt_masks"].cpu().numpy().astype(bool), want, err_msg=f"{n}: frame {t}")
            np.testing.assert_array_equal(np.asarray(fr["gt_ids"]), g[f"{n}.prop_ids{t}"])


def test_copy_and_paste_clip_dense_form():
    """the dense-tensor convenience form: a paste that happens adds instances to every frame and leaves the pixels outside the
    pasted masks alone; a rate miss returns the inputs"""
    from s2d_amd.data import copy_and_paste_clip
    T, H, W = 3, 96, 144
    tf, tm = _clip(41, T, H, W, 2)
    sf, sm = _clip(42, 3, 80, 112, 2)
    args = [torch.from_numpy(a).cuda() for a in (sf, sm, tf, tm)]
    random.seed(1); np.random.seed(1)
    pasted = 0
    for _ in range(12):
        f, m, info = copy_and_paste_clip(*args, rate=1.0, min_ratio=0.3, max_ratio=0.6)
        if info["pasted"]:
            pasted += 1
            assert f.shape == args[2].shape and m.shape[1:] == args[3].shape[1:] and m.shape[0] >= 2
            changed = (f != args[2]).any(1)                            # [T,H,W]
            assert bool((changed & ~m[2:].any(0).bool()).sum() == 0)   # pixels outside the pasted masks are the target's
    assert pasted > 0
    f, m, info = copy_and_paste_clip(*args, rate=0.0)
    assert not info["pasted"] and torch.equal(f, args[2])

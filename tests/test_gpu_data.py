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


def _reference_copy_paste(src_frame, src_masks, tgt_frames, tgt_masks, pf, keep):
    """engine/train_loop.py:461-548 for one clip, the drawn numbers given: returns (frames [T,3,H,W], masks [N+K,T,H,W])"""
    T, _, H, W = tgt_frames.shape
    K, N = src_masks.shape[0], tgt_masks.shape[0]
    out_f, out_m = [], []
    for f in range(T):
        h_new, w_new, h_shift, w_shift = (int(v) for v in pf[f])
        img_new = F.interpolate(src_frame[None].float(), size=(h_new, w_new), mode="bilinear", align_corners=False).byte().squeeze(0)
        m_new = F.interpolate(src_masks[None].float(), size=(h_new, w_new), mode="bilinear", align_corners=False).bool().squeeze(0)
        masks_all = torch.zeros(K, H, W)
        image_all = torch.zeros_like(tgt_frames[f])
        image_all[:, h_shift:h_shift + h_new, w_shift:w_shift + w_new] += img_new
        masks_all[:, h_shift:h_shift + h_new, w_shift:w_shift + w_new] += m_new
        copied = masks_all.bool() & torch.from_numpy(keep.astype(bool))[:, None, None]
        alpha = copied.sum(0) > 0
        out_f.append((alpha * image_all.byte()) + (~alpha * tgt_frames[f]))
        out_m.append(torch.cat([(~alpha) * tgt_masks[:, f].bool(), copied]))
    return torch.stack(out_f), torch.stack(out_m, 1).to(torch.uint8)


def test_copy_paste_vs_reference_arithmetic():
    from s2d_amd._lib import lib
    T, H, W, N = 3, 96, 144, 3
    Hs, Ws, K = 80, 112, 2
    tf, tm = _clip(31, T, H, W, N)
    sf, sm = _clip(32, 1, Hs, Ws, K)
    pf = np.array([[70, 100, 5, 20], [96, 144, 0, 0], [48, 72, 40, 60]], np.int32)
    for keep in (np.array([1, 1], np.uint8), np.array([0, 1], np.uint8)):
        dev = "cuda"
        out_f = torch.empty((T, 3, H, W), device=dev, dtype=torch.uint8)
        out_m = torch.empty((N + K, T, H, W), device=dev, dtype=torch.uint8)
        lib().call("s2d_copy_paste_u8", torch.from_numpy(tf).to(dev), torch.from_numpy(tm).to(dev), N, T, H, W, torch.from_numpy(sf[0]).to(dev),
                   torch.from_numpy(sm[:, 0].copy()).to(dev), K, Hs, Ws, torch.from_numpy(pf).to(dev), torch.from_numpy(keep).to(dev), out_f, out_m,
                   torch.cuda.current_stream().cuda_stream)
        ref_f, ref_m = _reference_copy_paste(torch.from_numpy(sf[0]), torch.from_numpy(sm[:, 0].copy()), torch.from_numpy(tf), torch.from_numpy(tm), pf, keep)
        np.testing.assert_array_equal(out_m.cpu().numpy(), ref_m.numpy())
        d = (out_f.cpu().int() - ref_f.int()).abs()
        assert int(d.max()) <= 1 and float((d != 0).float().mean()) < 1e-3       # .byte() of a float a hair below / above an integer
    # the overlap integers behind the "copy covers half of a target" rule (:515-527)
    counts = torch.empty((K, N), device="cuda", dtype=torch.int32)
    area = torch.empty((N,), device="cuda", dtype=torch.int32)
    lib().call("s2d_copy_paste_overlap", torch.from_numpy(tm).cuda(), N, T, H, W, torch.from_numpy(sm[:, 0].copy()).cuda(), K, Hs, Ws,
               int(pf[0, 0]), int(pf[0, 1]), int(pf[0, 2]), int(pf[0, 3]), counts, area, torch.cuda.current_stream().cuda_stream)
    _, ref_m = _reference_copy_paste(torch.from_numpy(sf[0]), torch.from_numpy(sm[:, 0].copy()), torch.from_numpy(tf), torch.from_numpy(tm), pf,
                                     np.ones(K, np.uint8))
    x = ref_m[N:, 0].reshape(K, -1).float(); y = torch.from_numpy(tm[:, 0]).reshape(N, -1).float()
    np.testing.assert_array_equal(counts.cpu().numpy(), (x @ y.t()).numpy().astype(np.int32))
    np.testing.assert_array_equal(area.cpu().numpy(), y.sum(1).numpy().astype(np.int32))


def test_copy_and_paste_clip_end_to_end():
    """the driver with the reference's draws: a paste that happens adds K instances to every frame, keeps the untouched pixels,
    and cancels cleanly when a copy would cover half of a target"""
    from s2d_amd.data import copy_and_paste_clip
    T, H, W = 3, 96, 144
    tf, tm = _clip(41, T, H, W, 2)
    sf, sm = _clip(42, 2, 80, 112, 2)
    args = [torch.from_numpy(a).cuda() for a in (sf, sm, tf, tm)]
    random.seed(1); np.random.seed(1)
    seen = {"pasted": 0, "cancelled": 0}
    for _ in range(12):
        f, m, info = copy_and_paste_clip(*args, rate=1.0, min_ratio=0.3, max_ratio=0.6)
        if info["pasted"]:
            seen["pasted"] += 1
            assert f.shape == args[2].shape and m.shape[1:] == args[3].shape[1:] and m.shape[0] >= 2
            changed = (f != args[2]).any(1)                            # [T,H,W]
            pasted_area = m[-2:].any(0)
            assert bool((changed & ~pasted_area).sum() == 0)           # pixels outside the pasted masks are the target's
            assert bool((m[:-2].bool() & pasted_area[None]).sum() == 0)
        else:
            seen["cancelled"] += 1
            assert f is args[2] and m is args[3]
    assert seen["pasted"] > 0
    f, m, info = copy_and_paste_clip(*args, rate=0.0)
    assert f is args[2] and not info["pasted"]

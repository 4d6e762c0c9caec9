"""Clip augmentation on the device: the augmentation list of data_video/augmentation.py:116-168 (RandomCrop ->
ResizeShortestEdge -> RandomFlip -> brightness -> contrast -> rotation) drawn on the host per the reference's policy (size and
flip once per clip, the detectron2 transforms per frame) and applied to the T frames and all instance masks of a clip in one
resampling pass each (s2d_aug_warp_frames_u8 / s2d_aug_warp_masks_u8).  The transform classes themselves are detectron2 /
fvcore (absent from the reference tree): parameter ranges and composition follow their published semantics, parity unpinned."""
import math
import sys

import numpy as np
import torch

from .._lib import lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


class ClipAugmentation:
    """configuration of build_augmentation(cfg, is_train=True), augmentation.py:116-158"""

    def __init__(self, min_size=(360, 480), max_size=sys.maxsize, sample_style="choice_by_clip", random_flip="flip_by_clip",
                 augmentations=(), crop=None, num_frames=2):
        self.min_size = (min_size, min_size) if isinstance(min_size, int) else tuple(min_size)
        self.max_size, self.sample_style, self.random_flip = max_size, sample_style, random_flip
        self.augmentations, self.crop, self.num_frames = tuple(augmentations), crop, num_frames

    @classmethod
    def from_config(cls, cfg):
        crop = (cfg.INPUT.CROP.TYPE, tuple(cfg.INPUT.CROP.SIZE)) if cfg.INPUT.CROP.ENABLED else None
        return cls(cfg.INPUT.MIN_SIZE_TRAIN, getattr(cfg.INPUT, "MAX_SIZE_TRAIN", sys.maxsize), cfg.INPUT.MIN_SIZE_TRAIN_SAMPLING,
                   cfg.INPUT.RANDOM_FLIP, cfg.INPUT.AUGMENTATIONS, crop, cfg.INPUT.SAMPLING_FRAME_NUM)

    def sample(self, T, H0, W0, rng=np.random):
        """-> (params float32 [T,16] in the layout of s2d_aug_warp_frames_u8, (H1, W1))"""
        by_clip = "by_clip" in self.sample_style
        size = flip = None
        P = np.zeros((T, 16), np.float32)
        out_hw = None
        for t in range(T):
            # crop (T.RandomCrop, per frame): "absolute_range": h, w drawn from [min(size), max] capped by the image
            cx, cy, cw, ch = 0, 0, W0, H0
            if self.crop is not None:
                ctype, csize = self.crop
                if ctype == "absolute_range":
                    ch = int(rng.randint(min(H0, csize[0]), min(H0, csize[1]) + 1))
                    cw = int(rng.randint(min(W0, csize[0]), min(W0, csize[1]) + 1))
                elif ctype == "absolute":
                    ch, cw = min(csize[0], H0), min(csize[1], W0)
                else:                                       # "relative": fractions of the image
                    ch, cw = int(H0 * csize[0] + 0.5), int(W0 * csize[1] + 0.5)
                cy = int(rng.randint(H0 - ch + 1)); cx = int(rng.randint(W0 - cw + 1))
            if size is None or not by_clip:                 # ResizeShortestEdge.get_transform, :51-75
                if "range" in self.sample_style:
                    size = int(rng.randint(self.min_size[0], self.min_size[1] + 1))
                else:
                    size = int(rng.choice(self.min_size))
            scale = size * 1.0 / min(ch, cw)
            newh, neww = (size, scale * cw) if ch < cw else (scale * ch, size)
            if max(newh, neww) > self.max_size:
                s2 = self.max_size * 1.0 / max(newh, neww)
                newh, neww = newh * s2, neww * s2
            neww, newh = int(neww + 0.5), int(newh + 0.5)
            if out_hw is None:
                out_hw = (newh, neww)
            H1, W1 = out_hw                                  # one output size per clip (frames of a clip are stacked downstream)
            if self.random_flip != "none" and (flip is None or self.random_flip != "flip_by_clip"):
                flip = bool(rng.uniform() < 0.5)             # RandomFlip, :101-115
            bright = float(rng.uniform(0.9, 1.1)) if "brightness" in self.augmentations else 1.0
            contrast = float(rng.uniform(0.9, 1.1)) if "contrast" in self.augmentations else 1.0
            angle, cxr, cyr = 0.0, 0.5, 0.5
            if "rotation" in self.augmentations:             # T.RandomRotation([-15, 15], expand=False, center in [0.4, 0.6]^2)
                angle = float(rng.uniform(-15, 15))
                cxr, cyr = float(rng.uniform(0.4, 0.6)), float(rng.uniform(0.4, 0.6))
            # inverse map: output pixel -> (un-rotate about the centre) -> (un-flip) -> (un-resize) -> (+ crop origin)
            th = math.radians(angle)
            c, s = math.cos(th), math.sin(th)
            ox, oy = cxr * W1, cyr * H1
            R = np.array([[c, -s, ox - c * ox + s * oy], [s, c, oy - s * ox - c * oy], [0, 0, 1]])      # rotation by +angle about (ox, oy): the inverse of the image's
            Fm = np.array([[-1.0, 0, W1], [0, 1, 0], [0, 0, 1]]) if (flip and self.random_flip != "vertical") else (
                np.array([[1.0, 0, 0], [0, -1, H1], [0, 0, 1]]) if flip else np.eye(3))
            S = np.array([[cw / W1, 0, cx], [0, ch / H1, cy], [0, 0, 1]])
            A = S @ Fm @ R
            P[t, :6] = A[:2].reshape(-1)
            P[t, 6:10] = (cx, cy, cw, ch)
            P[t, 10], P[t, 11], P[t, 12] = bright, contrast, (-1.0 if contrast != 1.0 else 0.0)
        return P, out_hw


def augment_clip(frames_u8, masks_u8, params, out_hw):
    """frames u8 CUDA [T,3,H0,W0], masks u8 CUDA [N,T,H0,W0] (or None), params float32 [T,16] (host), out_hw = (H1, W1)
    -> (frames u8 [T,3,H1,W1], masks u8 [N,T,H1,W1] or None)"""
    T, _, H0, W0 = frames_u8.shape
    H1, W1 = out_hw
    p = torch.from_numpy(np.ascontiguousarray(params, np.float32)).to(frames_u8.device)
    out = torch.empty((T, 3, H1, W1), device=frames_u8.device, dtype=torch.uint8)
    lib().call("s2d_aug_warp_frames_u8", frames_u8.contiguous(), T, H0, W0, p, H1, W1, out, _stream())
    mo = None
    if masks_u8 is not None:
        N = masks_u8.shape[0]
        mo = torch.empty((N, T, H1, W1), device=frames_u8.device, dtype=torch.uint8)
        lib().call("s2d_aug_warp_masks_u8", masks_u8.contiguous(), N, T, H0, W0, p, H1, W1, mo, _stream())
    return out, mo

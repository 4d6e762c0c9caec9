"""Where does the device matcher cost differ from a float64 evaluation?  (diagnostic; GPU)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_np as O          # noqa: E402  (diagnostic script: oracle as the checker)
from s2d_amd import ops                    # noqa: E402
from s2d_amd.utils import synth            # noqa: E402
from tests.test_gpu_criterion import pixel_major, pad_targets, make_targets, _dev   # noqa: E402


def cost64(masks, tgt, coords):
    Q, N = masks.shape[0], tgt.shape[0]
    tm = O.point_sample(tgt, np.repeat(coords, N, 0)).reshape(N, -1).astype(np.float64)
    om = O.point_sample(masks.astype(np.float32), np.repeat(coords, Q, 0)).reshape(Q, -1).astype(np.float64)
    sp = np.maximum(om, 0) + np.log1p(np.exp(-np.abs(om)))
    cm = (sp.sum(-1)[:, None] - om @ tm.T) / om.shape[1]
    sg = 1.0 / (1.0 + np.exp(-om))
    cd = 1 - (2 * (sg @ tm.T) + 1) / (sg.sum(-1)[:, None] + tm.sum(-1)[None, :] + 1)
    return cm, cd, om, tm


for scale in (1.0, 4.0, 16.0):
    B, Q, T, h, w, P, ns = 1, 100, 2, 120, 216, 12544, [10]
    H, W = 4 * h, 4 * w
    seed = 77
    logits = synth.randn(seed, 1, (B, Q, 2))
    masks = (synth.smooth_logits(seed, 2, (B, Q, T), (h, w)) * scale).astype(np.float32)
    tg = make_targets(seed, 7, ns, T, H, W)
    tgt, cnt = pad_targets(tg, max(ns), T, H, W)
    coords = np.random.default_rng(seed).random((1, B, P, 2), dtype=np.float32)
    cm, cd, om, tm = cost64(masks[0], tg[0], coords[0, 0][None])
    for name, wts, ref in (("mask", (0.0, 1.0, 0.0), cm), ("dice", (0.0, 0.0, 1.0), cd)):
        C = ops.matcher_cost(_dev(pixel_major(masks)[None]), _dev(logits[None]), _dev(tgt), _dev(cnt), (Q, T, h, w), P, wts,
                             coords=_dev(coords)).cpu().numpy()[0][:, :ns[0]].astype(np.float64)
        e = np.abs(C - ref)
        print(f"scale {scale:5.1f} |x|max {np.abs(om).max():7.2f} cost_{name}: max|C| {np.abs(ref).max():9.4f}  max abs err {e.max():.3e}  rel-to-max {e.max()/np.abs(ref).max():.3e}  "
              f"mean err {e.mean():.3e}  worst (q,n) {np.unravel_index(e.argmax(), e.shape)}")
        if name == "mask":
            # row-wise (query) vs column-wise structure of the error: SP is per query, A per pair
            print("      err by query (max over n) top5:", np.sort(e.max(1))[-5:], " err spread over n for worst q:", e[e.max(1).argmax()])

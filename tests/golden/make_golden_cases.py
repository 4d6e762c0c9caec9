"""Seeded inputs shared by generators in make_golden.py and the tests that read their fixtures.  Data only."""
import numpy as np


def assemble_cases():
    """seeded per-frame annotations (ids, classes, masks incl. empty and one-pixel-wide ones, a crowd annotation) of small clips"""
    rng = np.random.default_rng(23)
    cases = []
    for H, W, T, idpool in ((24, 32, 4, [7, 3, 11, 2]), (16, 16, 3, [5, 100, 42]), (20, 28, 5, [9, 1, 4, 8, 6])):
        video = []
        for t in range(T + 2):
            annos = []
            for k, i in enumerate(idpool):
                if rng.random() < 0.7:
                    m = np.zeros((H, W), bool)
                    kind = rng.integers(4)
                    if kind == 0:
                        pass                                              # annotated, but the (transformed) mask is empty
                    elif kind == 1:
                        m[rng.integers(H), rng.integers(W)] = True        # a single pixel
                    else:
                        y0, x0 = rng.integers(H - 4), rng.integers(W - 4)
                        m[y0:y0 + rng.integers(1, 5), x0:x0 + rng.integers(1, 5)] = True
                    annos.append({"id": int(i), "category_id": int(k % 2), "iscrowd": int(rng.random() < 0.1), "mask": m})
            video.append(annos)
        sel = sorted(rng.choice(T + 2, T, replace=False).tolist())
        cases.append(((H, W), video, sel))
    return cases

"""Seeded synthetic inputs: clips, sparse ellipse targets,
smooth mask logits, point tracks.  Pure numpy so the golden generator, the
oracle, the tests and bench.py all see bit-identical inputs from a seed."""
import numpy as np


def rng_for(seed, tag=0):
    return np.random.Generator(np.random.PCG64([int(seed), int(tag)]))


def randn(seed, tag, shape, scale=1.0):
    return (rng_for(seed, tag).standard_normal(shape) * scale).astype(np.float32)


def smooth_field(rng, shape_hw, cells=4, amp=4.0):
    """Low-frequency random field: coarse noise bilinearly upsampled. float32 [h,w]."""
    h, w = shape_hw
    gh, gw = max(2, h // cells), max(2, w // cells)
    g = rng.standard_normal((gh, gw)) * amp
    ys = np.linspace(0, gh - 1, h)
    xs = np.linspace(0, gw - 1, w)
    y0 = np.floor(ys).astype(int).clip(0, gh - 2)
    x0 = np.floor(xs).astype(int).clip(0, gw - 2)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx).astype(np.float32)


def smooth_logits(seed, tag, lead_shape, hw, cells=4, amp=4.0):
    rng = rng_for(seed, tag)
    n = int(np.prod(lead_shape))
    out = np.stack([smooth_field(rng, hw, cells, amp) for _ in range(n)])
    return out.reshape(tuple(lead_shape) + tuple(hw))


def ellipse_targets(seed, tag, n_inst, T, H, W, sparse=0.5, rmin=None, rmax=None):
    """Moving axis-aligned ellipses; each instance present in a random `1-sparse`
    fraction of frames (others: empty mask, id -1).  Returns
    masks uint8 [n,T,H,W], ids int64 [n,T] (-1 = absent)."""
    rng = rng_for(seed, tag)
    rmin = rmin or max(2.0, min(H, W) / 30.0)
    rmax = rmax or max(rmin + 1.0, min(H, W) / 4.5)
    yy, xx = np.mgrid[0:H, 0:W]
    masks = np.zeros((n_inst, T, H, W), np.uint8)
    ids = np.full((n_inst, T), -1, np.int64)
    for i in range(n_inst):
        cy, cx = rng.uniform(0.2 * H, 0.8 * H), rng.uniform(0.2 * W, 0.8 * W)
        ry, rx = rng.uniform(rmin, rmax), rng.uniform(rmin, rmax)
        present = rng.random(T) >= sparse
        if not present.any():
            present[rng.integers(T)] = True
        for t in range(T):
            cy += rng.uniform(-8, 8) * H / 720.0
            cx += rng.uniform(-8, 8) * W / 720.0
            if present[t]:
                m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
                masks[i, t] = m
                if m.any():
                    ids[i, t] = i
    return masks, ids


def smooth_frames_u8(seed, tag, T, H, W):
    """uint8 RGB frames [T,3,H,W]: blurred uniform noise."""
    rng = rng_for(seed, tag)
    out = np.empty((T, 3, H, W), np.uint8)
    for t in range(T):
        for c in range(3):
            f = smooth_field(rng, (H, W), cells=9, amp=1.0)
            f = (f - f.min()) / max(f.max() - f.min(), 1e-6)
            out[t, c] = (f * 255.0).astype(np.uint8)
    return out

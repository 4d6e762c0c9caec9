"""COCO run-length encoding of device masks (csrc/rle.hip) -- what `pycocotools.mask.encode / area / toBbox` produce for
the evaluator (model_training/mask2former_video/data_video/ytvis_eval.py:345-350) and the keymask annotation writer
(keymask_ident/annotations.py:100-106) -- without moving the byte masks to the host: the device emits run boundaries,
two more launches turn them into run lengths and the LEB-like ASCII strings (`encode`); `runs_from_boundaries` /
`strings_from_runs` are the same two steps in numpy for callers that want the run lengths themselves.

pycocotools is a third-party dependency that is not in the reference tree: the format is restated from its published
algorithm (maskApi.c rleEncode / rleToString); parity is unpinned beyond round trips and hand-derived strings."""
import numpy as np
import torch

from . import ops
from ._lib import lib


def boundaries(masks):
    """masks u8/bool CUDA [F,H,W] -> (positions int64 numpy [total], frame_off int64 numpy [F+1], area int32 [F], bbox int32 [F,4])"""
    if masks.dtype == torch.bool:
        masks = masks.view(torch.uint8)
    ops._chk(masks, torch.uint8)
    F, H, W = masks.shape
    dev = masks.device
    col_off = torch.empty((F, W), device=dev, dtype=torch.int32)
    nb = torch.empty((F,), device=dev, dtype=torch.int32)
    area = torch.empty((F,), device=dev, dtype=torch.int32)
    bbox = torch.empty((F, 4), device=dev, dtype=torch.int32)
    lib().call("s2d_rle_count_u8", masks, F, H, W, col_off, nb, area, bbox, ops._stream())
    nb_h = nb.cpu().numpy().astype(np.int64)                         # the one sync: sizes the output
    frame_off = np.zeros(F + 1, np.int64)
    np.cumsum(nb_h, out=frame_off[1:])
    pos = torch.empty((max(int(frame_off[-1]), 1),), device=dev, dtype=torch.int32)
    lib().call("s2d_rle_positions_u8", masks, F, H, W, col_off, torch.from_numpy(frame_off[:-1].copy()).to(dev), pos, ops._stream())
    return pos.cpu().numpy()[:frame_off[-1]].astype(np.int64), frame_off, area.cpu().numpy(), bbox.cpu().numpy()


def runs_from_boundaries(pos, frame_off, hw):
    """-> (counts int64 [total + F], count_off int64 [F+1]): per frame [p0, p1-p0, ..., hw - p_last] ([hw] if no boundary)"""
    F = len(frame_off) - 1
    n = np.diff(frame_off)
    count_off = np.zeros(F + 1, np.int64)
    np.cumsum(n + 1, out=count_off[1:])
    ext = np.empty(int(frame_off[-1]) + 2 * F, np.int64)             # per frame: 0, positions..., hw
    starts = frame_off[:-1] + 2 * np.arange(F)
    ends = starts + n + 1
    keep = np.ones(ext.shape[0], bool)
    keep[starts] = False
    keep[ends] = False
    ext[keep] = pos
    ext[starts] = 0
    ext[ends] = hw
    d = np.diff(ext)
    valid = np.ones(d.shape[0], bool)
    valid[ends[:-1]] = False                                         # differences across frame joins
    return d[valid], count_off


def strings_from_runs(counts, count_off):
    """rleToString for every frame at once: 5 bits per char + continuation bit, chars 48.., counts[i>2] delta-coded vs i-2"""
    F = len(count_off) - 1
    idx = np.arange(counts.shape[0]) - np.repeat(count_off[:-1], np.diff(count_off))      # index within its frame
    x = counts.copy()
    sel = idx > 2
    x[sel] -= counts[np.nonzero(sel)[0] - 2]
    chars = np.zeros((x.shape[0], 13), np.uint8)
    used = np.zeros((x.shape[0], 13), bool)
    active = np.ones(x.shape[0], bool)
    for k in range(13):
        if not active.any():
            break
        c = (x & 0x1F).astype(np.int64)
        x = x >> 5                                                    # arithmetic shift (negative deltas)
        more = np.where((c & 0x10) != 0, x != -1, x != 0)
        c = np.where(more, c | 0x20, c) + 48
        chars[active, k] = c[active]
        used[active, k] = True
        active = active & more
    flat = chars[used]                                               # row-major: the chars of a count stay together
    per_count = used.sum(1)
    char_off = np.zeros(F + 1, np.int64)
    np.cumsum(np.add.reduceat(per_count, count_off[:-1]) if F else [], out=char_off[1:])
    buf = flat.tobytes()
    return [buf[char_off[f]:char_off[f + 1]] for f in range(F)]


def encode(masks):
    """masks CUDA [F,H,W] u8/bool -> list of {'size': [H, W], 'counts': bytes} (mask_util.encode of an [H,W,F] Fortran array),
    plus areas [F] and boxes [F,4] (x, y, w, h as float64, mask_util.toBbox).  Everything up to the final strings runs on
    the device; two small copies come back (the boundary counts, then the strings)."""
    if masks.dtype == torch.bool:
        masks = masks.view(torch.uint8)
    ops._chk(masks, torch.uint8)
    F, H, W = masks.shape
    if F == 0:
        return [], np.zeros((0,), np.int64), np.zeros((0, 4), np.float64)
    dev, st = masks.device, ops._stream()
    col_off = torch.empty((F, W), device=dev, dtype=torch.int32)
    nb = torch.empty((F,), device=dev, dtype=torch.int32)
    area = torch.empty((F,), device=dev, dtype=torch.int32)
    bbox = torch.empty((F, 4), device=dev, dtype=torch.int32)
    lib().call("s2d_rle_count_u8", masks, F, H, W, col_off, nb, area, bbox, st)
    frame_off = torch.zeros((F + 1,), device=dev, dtype=torch.int64)
    torch.cumsum(nb, 0, out=frame_off[1:])
    total = int(frame_off[-1])                                       # sync 1: sizes the outputs
    ncounts = total + F
    pos = torch.empty((max(total, 1),), device=dev, dtype=torch.int32)
    lib().call("s2d_rle_positions_u8", masks, F, H, W, col_off, frame_off, pos, st)
    ws = torch.empty((lib().call("s2d_rle_string_workspace_bytes", ncounts),), device=dev, dtype=torch.uint8)
    chars = torch.empty((7 * ncounts,), device=dev, dtype=torch.uint8)
    str_off = torch.empty((F + 1,), device=dev, dtype=torch.int64)
    lib().call("s2d_rle_strings_u8", pos, frame_off, F, H * W, ncounts, ws, ws.numel(), chars, str_off, st)
    so = str_off.cpu().numpy()                                       # sync 2
    buf = chars[:int(so[-1])].cpu().numpy().tobytes()
    return ([{"size": [H, W], "counts": buf[so[f]:so[f + 1]]} for f in range(F)], area.cpu().numpy().astype(np.int64),
            bbox.cpu().numpy().astype(np.float64))


def encode_video_predictions(pred_masks):
    """pred_masks: CUDA bool/u8 [K,T,H,W] (inference_video before its device->host copy) -> per instance a list of T RLE dicts
    with str counts, the `segmentations` field instances_to_coco_json_video builds (ytvis_eval.py:345-357)"""
    K, T, H, W = pred_masks.shape
    rles, _, _ = encode(pred_masks.reshape(K * T, H, W))
    for r in rles:
        r["counts"] = r["counts"].decode("utf-8")
    return [rles[k * T:(k + 1) * T] for k in range(K)]

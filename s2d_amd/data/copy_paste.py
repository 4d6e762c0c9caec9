"""Video copy-paste and sparse-mask densification of the trainer, for clips that live on the GPU: the behaviour of
model_training/mask2former_video/engine/train_loop.py:30-156 (`propagate_sparse_masks`) and :377-590
(`CustomSimpleTrainer.copy_and_paste`), pinned by tests/golden/copy_paste.npz (written by the reference's own two functions).
`propagate_sparse_masks` is a plan-then-launch design (host: id tables -> fill plan with the reference's RNG draw order; device: all
planes of the clip in one launch); `copy_and_paste` replays the reference loop's decisions on integer tables after one launch per frame.

What is mirrored, including what looks accidental in the reference but is what it ships:
  * the order and number of draws from `random` and `numpy.random` -- also on a rate miss or with no source instance, where the
    reference still draws the choice, the frame id and every frame's ratio and shifts (:420-441, :461-468);
  * `copied_instances.gt_masks` is reassigned to the pasted canvas at the end of every frame (:512-514), so the copied masks are
    resized and shifted CUMULATIVELY from frame to frame while the image patch is resized from the source frame each time;
  * a frame without target instances appends the running `copied_instances` object itself (:524-525): such frames all show the masks
    that object holds when the loop ends;
  * frame 0 decides: a copy covering >= 50 % of a target's area (or a target of zero area: 0/0 is not < 0.5) cancels the paste for
    every frame (:527-535); on later frames a zero-area target empties the copies from there on (nan is not < 2.0);
  * COPY_PASTE_DENSIFY_SPARSE with a paste due only densifies (:433-439); otherwise the result is densified at the end (:566-569);
    a clip whose frames end with different instance counts falls back to the untouched target (:573-580).

Clips are the mapper's dicts (data_video/dataset_mapper.py:306-404) with device tensors: {"image": T x uint8 [3,H,W], "instances":
T x {"gt_masks": bool [n,H,W], "gt_ids": int64 [n], "gt_classes": int64 [n]}} -- the form KDVideoMaskFormer.forward accepts.  The
resize / paste / composite of a frame is one launch (s2d_copy_paste_frame_u8); all T frames are enqueued before the single small
device-to-host copy of the integer overlap / area tables that the loop's decisions are replayed on.  Boxes are not produced (the
path never reads them)."""
import copy
import random

import numpy as np
import torch

from .._lib import lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _frame(masks, ids, classes):
    return {"gt_masks": masks, "gt_ids": ids, "gt_classes": classes}


def _ids(fr, key="gt_ids"):
    v = fr[key]
    return v.detach().cpu().numpy().astype(np.int64) if isinstance(v, torch.Tensor) else np.asarray(v, np.int64)


def _masks(fr, device=None):
    m = fr["gt_masks"]
    m = m.tensor if hasattr(m, "tensor") else torch.as_tensor(m)
    return m.to(device=device or m.device, dtype=torch.bool)


# ------------------------------------------------------------------------------------------------ propagate_sparse_masks (:30-156)
def _host_ids_classes(frames):
    """every frame's ids and classes as host int64 arrays; device-resident ones come over in ONE copy for the clip"""
    vals, on_dev = [], []
    for fr in frames:
        for key in ("gt_ids", "gt_classes"):
            v = fr[key]
            vals.append(v)
            on_dev.append(isinstance(v, torch.Tensor) and v.is_cuda)
    if any(on_dev):
        flat = torch.cat([v.reshape(-1).to(torch.int64) for v, d in zip(vals, on_dev) if d]).cpu().numpy()
        pos = 0
    out = []
    for v, d in zip(vals, on_dev):
        if d:
            out.append(flat[pos:pos + v.numel()].copy()); pos += v.numel()
        else:
            out.append(np.asarray(v.detach().numpy() if isinstance(v, torch.Tensor) else v, np.int64).reshape(-1))
    return out[0::2], out[1::2]


def _fill_plan(ids, classes, has_pixels, max_shift):
    """Which instance planes each frame gains, from the id tables alone.  An id missing from a frame is filled from its LATEST earlier
    sighting (original planes only: synthesised ones are never sources), ids are visited in order of FIRST sighting, and each fill
    draws its jitter as two `random.randint(-max_shift, max_shift)` calls, x then y -- the consumption order of the reference loop.
    -> per frame a list of (source frame, source slot, dx, dy, id, class or None)."""
    first_seen, latest = [], {}
    fills = []
    for t, (ids_t, cls_t) in enumerate(zip(ids, classes)):
        if has_pixels[t]:
            for slot, tid in enumerate(ids_t.tolist()):
                if tid not in latest:
                    first_seen.append(tid)
                latest[tid] = (t, slot, int(cls_t[slot]) if slot < len(cls_t) else None)
        here = set(ids_t.tolist())
        row = []
        for tid in first_seen:
            if tid in here:
                continue
            dx = random.randint(-max_shift, max_shift) if max_shift > 0 else 0
            dy = random.randint(-max_shift, max_shift) if max_shift > 0 else 0
            ft, fs, fc = latest[tid]
            row.append((ft, fs, dx, dy, tid, fc))
        fills.append(row)
    return fills


def propagate_sparse_masks(instances_per_frame, max_shift=2):
    """Densify a sparsely annotated clip (engine/train_loop.py:30-156): a frame that lacks an instance id seen earlier receives that
    id's most recent mask, shifted by a random jitter of up to max_shift pixels (out[y][x] = mask[y + dy][x + dx], zero outside).
    The host only reads the id tables and writes a plan (_fill_plan); every plane of every frame that changes -- kept ones first,
    filled ones behind them, as the reference concatenates -- is then produced by one launch (s2d_shift_planes_u8) into one
    buffer the returned frames are views of.  Returns new per-frame dicts; the inputs are not modified."""
    if not instances_per_frame:
        return instances_per_frame
    planes = [_masks(fr).contiguous() for fr in instances_per_frame]
    ids, classes = _host_ids_classes(instances_per_frame)
    fills = _fill_plan(ids, classes, [m.numel() > 0 for m in planes], max_shift)
    out = [_frame(m, i, c) for m, i, c in zip(planes, ids, classes)]
    changed = [t for t, row in enumerate(fills) if row]
    if not changed:
        return out
    H, W = planes[0].shape[-2:]
    dev = planes[0].device
    rows, spans = [], []
    for t in changed:
        keep = planes[t].shape[0] if planes[t].numel() > 0 else 0
        start = len(rows)
        rows += [(planes[t].data_ptr() + k * H * W, 0, 0) for k in range(keep)]
        rows += [(planes[ft].data_ptr() + fs * H * W, dx, dy) for ft, fs, dx, dy, _, _ in fills[t]]
        spans.append((start, len(rows)))
    table = np.zeros((len(rows),), dtype=[("src", np.uint64), ("dx", np.int32), ("dy", np.int32)])
    table["src"], table["dx"], table["dy"] = zip(*rows)
    plan = torch.from_numpy(table.view(np.int64).reshape(-1, 2)).to(dev)
    buf = torch.empty((len(rows), H, W), dtype=torch.bool, device=dev)
    lib().call("s2d_shift_planes_u8", plan, len(rows), int(H), int(W), buf, _stream())
    for t, (a, b) in zip(changed, spans):
        new_ids = np.asarray([f[4] for f in fills[t]], np.int64)
        new_cls = np.asarray([f[5] for f in fills[t] if f[5] is not None], np.int64)
        out[t] = _frame(buf[a:b], np.concatenate([ids[t], new_ids]), np.concatenate([classes[t], new_cls]))
    return out


# ------------------------------------------------------------------------------------------------ copy_and_paste (:377-590)
class _Copied:
    """the loop's `copied_instances`: a mutable object the reference both reassigns fields of and appends to its output list"""

    def __init__(self, masks, ids, classes):
        self.masks, self.ids, self.classes = masks, ids, classes

    def __len__(self):
        return len(self.ids)


def copy_and_paste(sources, targets, rate=1.0, random_num=False, min_ratio=0.5, max_ratio=1.0, densify_sparse=False):
    """`CustomSimpleTrainer.copy_and_paste(sources, targets)` with cfg_COPY_PASTE_RATE / _RANDOM_NUM / _MIN_RATIO / _MAX_RATIO /
    _DENSIFY_SPARSE as keyword arguments.  Returns the new target clips (a clip that was not pasted is returned as it came in)."""
    outs = []
    for source, target in zip(sources, targets):
        outs.append(_pair(source, target, rate, random_num, min_ratio, max_ratio, densify_sparse))
    return outs


def _pair(source, target, rate, random_num, lo, hi, densify):
    src_inst, src_img = source["instances"], source["image"]
    tgt_inst, tgt_img = target["instances"], target["image"]
    T = len(tgt_inst)
    dev = tgt_img[0].device
    n_src = len(_ids(src_inst[0]))
    if rate >= random.random() and n_src > 0:                                            # :420-428
        num_copy = (1 if n_src == 1 else int(np.random.randint(1, max(1, n_src)))) if random_num else n_src
    else:
        num_copy = 0
    if num_copy > 0 and densify:                                                         # :430-439: densify only
        new = dict(target)
        try:
            new["instances"] = propagate_sparse_masks(tgt_inst, max_shift=2)
        except Exception:
            pass
        return _same_counts_or(new, target)
    choice = np.random.choice(n_src, num_copy, replace=False)                            # :441
    frame_id = int(np.random.randint(1, max(1, len(src_inst)))) - 1                      # :443
    sm = _masks(src_inst[frame_id], dev)[torch.as_tensor(choice, dtype=torch.long, device=dev)]
    copied = _Copied(sm, _ids(src_inst[frame_id])[choice], _ids(src_inst[frame_id], "gt_classes")[choice])
    sf = src_img[frame_id].to(dev).contiguous()
    Hs, Ws = sf.shape[-2:]
    # every frame's draws first (they do not depend on the data), then all frames enqueued, then ONE copy of the integer tables
    geo = []
    for f in range(T):                                                                   # :461-468
        H, W = tgt_img[f].shape[-2:]
        ratio = random.uniform(lo, hi)
        w_new, h_new = int(ratio * W), int(ratio * H)
        w_shift = random.randint(0, max(0, W - w_new))
        h_shift = random.randint(0, max(0, H - h_new))
        geo.append((h_new, w_new, h_shift, w_shift))
    K = num_copy
    if K == 0:                                                                           # :475-488: every frame keeps the original
        return _finish(target, [t for t in tgt_img], [_frame(_masks(fr, dev), _ids(fr), _ids(fr, "gt_classes")) for fr in tgt_inst], target)
    cur = sm.to(torch.uint8).contiguous()
    canv, comp, tout, stats, ns = [], [], [], [], []
    for f in range(T):
        tm = _masks(tgt_inst[f], dev).to(torch.uint8).contiguous()
        N = tm.shape[0]
        H, W = tgt_img[f].shape[-2:]
        canvas = torch.empty((K, H, W), device=dev, dtype=torch.uint8)
        of = torch.empty_like(tgt_img[f]); ot = torch.empty_like(tm)
        st = torch.empty((max(K * N + 2 * N, 1),), device=dev, dtype=torch.int32)
        lib().call("s2d_copy_paste_frame_u8", sf, Hs, Ws, cur, K, cur.shape[1], cur.shape[2], tgt_img[f].contiguous(), tm, N, H, W, *geo[f], canvas, of,
                   ot, st, st[K * N:], st[K * N + N:], _stream())
        canv.append(canvas); comp.append(of); tout.append(ot); stats.append(st); ns.append(N)
        cur = canvas                                                                     # :512-514: the next frame transforms this canvas
    host = torch.cat(stats).cpu().numpy()                                                # the one synchronisation of the pair
    tabs, o = [], 0
    for f in range(T):
        N = ns[f]
        n = max(K * N + 2 * N, 1)
        tabs.append((host[o:o + K * N].reshape(K, N), host[o + K * N:o + K * N + N], host[o + K * N + N:o + K * N + 2 * N]))
        o += n
    # replay of the loop's decisions (:516-560) on the integer tables
    new_img, new_inst = [], []
    sum_keep = None
    for f in range(T):
        N = ns[f]
        orig = _frame(_masks(tgt_inst[f], dev), _ids(tgt_inst[f]), _ids(tgt_inst[f], "gt_classes"))
        if len(copied) == 0:                                                             # :475-488
            new_img.append(tgt_img[f]); new_inst.append(orig)
            continue
        copied.masks = canv[f].bool()                                                    # :512 (mutates the running object)
        if N == 0:                                                                       # :516-525: the object itself is appended
            new_img.append(comp[f]); new_inst.append(copied)
            continue
        inter, tarea, alive = tabs[f]
        with np.errstate(divide="ignore", invalid="ignore"):
            ioy = inter.astype(np.float32) / tarea.astype(np.float32)[None, :]           # :394-399 mode 'ioy', float32
        mx = np.where(np.isnan(ioy).any(1), np.float32("nan"), ioy.max(1))               # torch.max propagates nan
        if f == 0:
            keep = mx < 0.5
            sum_keep = int(keep.sum())
        else:
            if sum_keep is None:
                raise RuntimeError("frame 0 has no target instance but a later frame has: the reference reads `sum_keep` before assignment here")
            keep = mx < 2.0
        if sum_keep < len(keep):                                                         # :533-536
            new_img.append(tgt_img[f]); new_inst.append(orig)
            continue
        if keep.all():
            copied = _Copied(copied.masks, copied.ids, copied.classes)                   # :538 indexing makes a new object
            live = alive > 0
            sel = torch.as_tensor(np.nonzero(live)[0], dtype=torch.long, device=dev)
            masks = torch.cat([tout[f].bool()[sel], copied.masks], 0)
            new_img.append(comp[f])
            new_inst.append(_frame(masks, np.concatenate([orig["gt_ids"][live], copied.ids]), np.concatenate([orig["gt_classes"][live], copied.classes])))
        else:
            # nan row (a target of zero area on a later frame): every copy is dropped, alpha is empty, the frame keeps its image and
            # its targets of non-zero area; the following frames see an empty `copied_instances`
            assert not keep.any()
            copied = _Copied(copied.masks[:0], copied.ids[:0], copied.classes[:0])
            live = tarea > 0
            sel = torch.as_tensor(np.nonzero(live)[0], dtype=torch.long, device=dev)
            new_img.append(tgt_img[f])
            new_inst.append(_frame(orig["gt_masks"][sel], orig["gt_ids"][live], orig["gt_classes"][live]))
    frames = [fr if isinstance(fr, dict) else _frame(fr.masks, fr.ids, fr.classes) for fr in new_inst]   # aliases resolve to their final state
    return _finish(target, new_img, frames, target)


def _finish(target, images, frames, original):
    new = dict(target)
    new["image"] = images
    try:
        new["instances"] = propagate_sparse_masks(frames, max_shift=2)                   # :566-569
    except Exception:
        new["instances"] = frames
    return _same_counts_or(new, original)


def _same_counts_or(new, original):
    """:573-580: all frames of a clip must end with the same number of instances, else the untouched target is returned"""
    counts = {len(_ids(fr)) for fr in new["instances"]}
    return new if len(counts) == 1 else original


def copy_and_paste_clip(src_frames, src_masks, tgt_frames, tgt_masks, rate=1.0, random_num=False, min_ratio=0.8, max_ratio=1.0):
    """dense-tensor convenience form: src_frames u8 [Ts,3,Hs,Ws], src_masks u8 [Ks,Ts,Hs,Ws], tgt_frames u8 [T,3,H,W], tgt_masks u8
    [N,T,H,W] (every instance present in every frame) -> (frames [T,3,H,W], masks [N',T,H,W], info).  Runs copy_and_paste on the
    equivalent per-frame clips."""
    Ks, N, T = src_masks.shape[0], tgt_masks.shape[0], tgt_frames.shape[0]
    ar = lambda n, o=0: np.arange(n, dtype=np.int64) + o
    src = {"image": [f for f in src_frames], "instances": [_frame(src_masks[:, t].bool(), ar(Ks), np.zeros(Ks, np.int64)) for t in range(src_frames.shape[0])]}
    tgt = {"image": [f for f in tgt_frames], "instances": [_frame(tgt_masks[:, t].bool(), ar(N, 1000), np.zeros(N, np.int64)) for t in range(T)]}
    out = copy_and_paste([src], [tgt], rate, random_num, min_ratio, max_ratio)[0]
    if out is tgt:
        return tgt_frames, tgt_masks, {"pasted": False}
    frames = torch.stack(list(out["image"]))
    masks = torch.stack([fr["gt_masks"] for fr in out["instances"]], 1).to(torch.uint8)
    return frames, masks, {"pasted": bool((frames != tgt_frames).any()) or masks.shape[0] != N}

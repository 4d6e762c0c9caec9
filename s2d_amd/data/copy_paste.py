"""Video copy-paste of the trainer (model_training/mask2former_video/engine/train_loop.py:377-590) for one (source, target)
clip pair with the clips on the GPU: the random draws and the keep / fall-back rules are the reference's host logic, the
resize + composite of all T frames is one launch (s2d_copy_paste_u8), the overlap test one small launch and a K x N copy."""
import random

import numpy as np
import torch

from .._lib import lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def copy_and_paste_clip(src_frames, src_masks, tgt_frames, tgt_masks, rate=1.0, random_num=False, min_ratio=0.8, max_ratio=1.0):
    """src_frames u8 [Ts,3,Hs,Ws], src_masks u8 [Ks,Ts,Hs,Ws] (instances of the labelled clip), tgt_frames u8 [T,3,H,W],
    tgt_masks u8 [N,T,H,W] -> (frames [T,3,H,W], masks [N',T,H,W], info dict).  Follows :417-560 per target clip:
      * with probability `rate` (DATALOADER.COPY_PASTE_RATE) copy num_copy instances (all, or 1..Ks-1 when random_num) chosen
        without replacement from ONE random source frame (:420-441);
      * per target frame a fresh resize ratio in [min_ratio, max_ratio] and shift (:461-468); frame 0 decides which copies
        survive: a copy covering >= 50 % of some target instance's area cancels the paste for the whole clip (:515-532);
      * targets lose the pasted area, targets left empty in a frame are dropped there, and a clip whose frames end up with
        different instance counts falls back to the unmodified target (:549-560, :573-580).
    `propagate_sparse_masks` (densification of sparse annotations) is not part of this step."""
    Ts, _, Hs, Ws = src_frames.shape
    T, _, H, W = tgt_frames.shape
    Ks, N = src_masks.shape[0], tgt_masks.shape[0]
    info = {"pasted": False}
    if not (rate >= random.random() and Ks > 0):
        return tgt_frames, tgt_masks, info
    num_copy = (1 if Ks == 1 else int(np.random.randint(1, max(1, Ks)))) if random_num else Ks
    choice = np.random.choice(Ks, num_copy, replace=False)
    frame_id = int(np.random.randint(1, max(1, Ts))) - 1
    sm = src_masks[torch.as_tensor(choice, device=src_masks.device), frame_id].contiguous()      # [K,Hs,Ws]
    sf = src_frames[frame_id].contiguous()
    K = num_copy
    pf = np.zeros((T, 4), np.int32)
    for f in range(T):
        ratio = random.uniform(min_ratio, max_ratio)
        w_new, h_new = int(ratio * W), int(ratio * H)
        pf[f] = (h_new, w_new, random.randint(0, max(0, H - h_new)), random.randint(0, max(0, W - w_new)))
    dev = tgt_frames.device
    keep = np.ones(K, np.uint8)
    if N > 0:
        counts = torch.empty((K, N), device=dev, dtype=torch.int32)
        area = torch.empty((N,), device=dev, dtype=torch.int32)
        lib().call("s2d_copy_paste_overlap", tgt_masks.contiguous(), N, T, H, W, sm, K, Hs, Ws, int(pf[0, 0]), int(pf[0, 1]), int(pf[0, 2]),
                   int(pf[0, 3]), counts, area, _stream())
        c, a = counts.cpu().numpy().astype(np.float32), area.cpu().numpy().astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ioy = c / a[None, :]                                   # inter / target area, float32 as in the reference (0/0 = nan: not < 0.5)
        keep = (ioy.max(1) < 0.5).astype(np.uint8)
        if keep.sum() < K:                                         # :529-532 on frame 0 -> every frame keeps the original
            info["cancelled"] = "a copy covers half of a target instance"
            return tgt_frames, tgt_masks, info
    out_f = torch.empty_like(tgt_frames)
    out_m = torch.empty((N + K, T, H, W), device=dev, dtype=torch.uint8)
    lib().call("s2d_copy_paste_u8", tgt_frames.contiguous(), tgt_masks.contiguous(), N, T, H, W, sf, sm, K, Hs, Ws,
               torch.from_numpy(pf).to(dev), torch.from_numpy(keep).to(dev), out_f, out_m, _stream())
    if N > 0:
        alive = out_m[:N].flatten(2).any(-1).bool()                       # [N,T]: targets with area left, per frame (:549)
        per_frame = alive.sum(0)
        if int(per_frame.min()) != int(per_frame.max()) or not bool(alive.all(1).eq(alive.any(1)).all()):
            info["cancelled"] = "instance counts differ between frames"
            return tgt_frames, tgt_masks, info
        sel = torch.cat([alive[:, 0], torch.ones(K, dtype=torch.bool, device=dev)])
        out_m = out_m[sel]
    info.update(pasted=True, choice=choice.tolist(), frame_id=frame_id, paste_frames=pf.tolist())
    return out_f, out_m, info

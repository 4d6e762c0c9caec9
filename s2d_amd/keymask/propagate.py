"""Keymask propagate-and-match, host side.  Mirrors the function boundaries of
/root/reference/keymask_ident/cotracker_matching.py (pred_tracks_to_binary_masks :453-503,
compute_point_mask_intersection :640-662, extract_mask_matches :665-719) and cotracker_occlusions.py:359, on top of
the HIP kernels of csrc/keymask.hip.  One launch computes every (frame, object) point/mask ratio of a tracked mask;
the reference issues O(frames x objects) launches with two .item() syncs each."""
import numpy as np
import torch

from .. import ops
from .._lib import lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def pred_tracks_to_binary_masks(pred_tracks, height, width, return_mask=False):
    """pred_tracks [B,T,P,2] (x,y) float32 CUDA -> uint8 [B,T,H,W] (return_mask=False form only: the convex-hull fill
    of the True branch is cv2 visualisation code, not on the propagation path)."""
    assert not return_mask
    B, T, P, _ = pred_tracks.shape
    out = torch.empty((B, T, height, width), device=pred_tracks.device, dtype=torch.uint8)
    tr = pred_tracks.contiguous().float()
    for b in range(B):
        lib().call("s2d_tracks_to_masks_u8", tr[b], T, P, height, width, out[b], _stream())
    return out


class IdMap:
    """A video's pseudo-mask id map (T,H,W,1) int64 on the device + per-frame presence table (computed once)."""

    def __init__(self, all_video_masks, max_id=None):
        ids = all_video_masks[..., 0] if all_video_masks.dim() == 4 else all_video_masks
        self.ids = ids.to(device="cuda", dtype=torch.int64).contiguous()
        self.T, self.Hi, self.Wi = self.ids.shape
        self.max_id = int(max_id if max_id is not None else int(self.ids.max()))
        self.presence = torch.empty((self.T, self.max_id + 1), device="cuda", dtype=torch.uint8)
        lib().call("s2d_idmap_presence_u8", self.ids, self.T, self.Hi, self.Wi, self.max_id, self.presence, _stream())
        self._presence_host = None

    def frame_object_ids(self, t):
        """torch.sort(torch.unique(frame)[1:]) of :680-681 -- the smallest id present is dropped (assumed background)"""
        if self._presence_host is None:
            self._presence_host = self.presence.cpu().numpy()
        return np.nonzero(self._presence_host[t])[0][1:]


def point_id_counts(track_masks, idmap: IdMap):
    """track_masks u8 [T,H,W] -> (counts int32 [T,max_id+1], total int32 [T]) on the device"""
    T, H, W = track_masks.shape
    counts = torch.empty((T, idmap.max_id + 1), device="cuda", dtype=torch.int32)
    total = torch.empty((T,), device="cuda", dtype=torch.int32)
    lib().call("s2d_point_id_counts", track_masks, idmap.ids, T, H, W, idmap.Hi, idmap.Wi, idmap.max_id, counts, total, _stream())
    return counts, total


def compute_point_mask_intersection(pointmask, mask, grid_size=None):
    """single pair, reference signature (:640-662): #(points & mask) / #points as a python float"""
    pm = (pointmask != 0).to(device="cuda", dtype=torch.uint8).contiguous()[None]
    idm = IdMap((mask != 0).to(torch.int64)[None], max_id=1)
    c, t = point_id_counts(pm, idm)
    c, t = c.cpu().numpy(), t.cpu().numpy()
    return 0.0 if t[0] == 0 else int(c[0, 1]) / int(t[0])


def extract_mask_matches(segm_mask_hw, pred_tracks, idmap: IdMap, v_range, matching_threshold=0.5):
    """(:665-719) segm_mask_hw = (H,W) of the tracked cluster mask; pred_tracks [1,T,P,2]; returns (matches,
    all_comparisons) as lists of dicts {frame_id, mask_id, iou}."""
    H, W = segm_mask_hw
    tm = pred_tracks_to_binary_masks(pred_tracks, H, W)[0]
    assert tm.shape[0] == idmap.T
    counts, total = point_id_counts(tm, idmap)
    counts, total = counts.cpu().numpy(), total.cpu().numpy()      # ONE sync per tracked mask
    matches, allc = [], []
    for t in range(v_range[0], v_range[1] + 1):
        for oid in idmap.frame_object_ids(t):
            iou = 0.0 if total[t] == 0 else int(counts[t, oid]) / int(total[t])
            rec = {"frame_id": t, "mask_id": int(oid), "iou": iou}
            allc.append(rec)
            if iou > matching_threshold:
                matches.append(rec)
    return matches, allc


def color_masks_to_ids(rgb):
    """load_masks (cotracker_matching.py:22-84) after the PNG decode: rgb CUDA u8 [T,H,W,3] -> int64 [T,H,W,1]; per frame
    black -> 0, the other colours -> 1..n in sorted (R,G,B) order.  Raises if a frame has more than 4096 colours."""
    ops._chk(rgb, torch.uint8)
    T, H, W, C = rgb.shape
    if C != 3:
        raise ValueError("rgb must be [T,H,W,3]")
    dev = rgb.device
    ws = torch.empty((lib().call("s2d_color_ids_workspace_words", T),), device=dev, dtype=torch.int32)
    n_ids = torch.empty((T,), device=dev, dtype=torch.int32)
    ids = torch.empty((T, H, W, 1), device=dev, dtype=torch.int64)
    ovf = torch.empty((1,), device=dev, dtype=torch.int32)
    lib().call("s2d_color_masks_to_ids", rgb, T, H, W, ws, n_ids, ids, ovf, _stream())
    if int(ovf) != 0:
        raise RuntimeError("a frame holds more than 4096 distinct colours: not a colour-mask image")
    return ids


def visibility_curve(pred_visibility):
    """pred_visibility [1,T,P] bool -> [T] float32 (cotracker_occlusions.py:359)"""
    v = pred_visibility[0].to(device="cuda", dtype=torch.uint8).contiguous()
    T, P = v.shape
    out = torch.empty((T,), device="cuda", dtype=torch.float32)
    lib().call("s2d_visibility_curve_f32", v, T, P, out, _stream())
    return out


def local_correlation(fmap_nhwc, coords, support, r=3):
    """K1 (self-defined, parity unpinned): fmap [T,H,W,C], coords [T,Np,2], support [Np,(2r+1)^2,C] ->
    corr [T,Np,(2r+1)^2,(2r+1)^2]"""
    T, H, W, C = fmap_nhwc.shape
    Np = coords.shape[1]
    S = (2 * r + 1) ** 2
    out = torch.empty((T, Np, S, S), device=fmap_nhwc.device, dtype=torch.float32)
    lib().call("s2d_local_corr_f32", fmap_nhwc.contiguous(), coords.contiguous(), support.contiguous(), T, Np, H, W, C, r, out, _stream())
    return out

"""K7: the host-side grouping steps of keymask discovery (tiny 0/1 matrices; sklearn DBSCAN exactly as the reference
calls it).  Mirrors /root/reference/keymask_ident/identify_visibility_windows.py:108-231
(get_visibility_windows_for_video, get_visible_ranges :61-80, get_highly_visible_rows :82-96) and
cotracker_matching.py:764-840 (temporal_correspondance_clustering, crop_bool_tensor :722-762).  File I/O (JSON dumps)
and debug prints of the reference are left out: out of scope."""
import numpy as np
from sklearn.cluster import DBSCAN


def get_visible_ranges(maj_vote):
    """inclusive (start, end) index pairs of the runs of ones in a 0/1 vector (identify_visibility_windows.py:65-88):
    the rising and falling edges of the zero-padded vector, paired up"""
    v = np.asarray(maj_vote).astype(bool).reshape(-1)
    edges = np.flatnonzero(np.diff(np.concatenate(([0], v.view(np.int8), [0]))))
    return [(int(a), int(b) - 1) for a, b in zip(edges[0::2], edges[1::2])]


def get_highly_visible_rows(cluster_vis, runs, threshold=0.8):
    out = {}
    for (start, end) in runs:
        frac = cluster_vis[:, start:end + 1].sum(1) / np.float32(end - start + 1)
        out[(start, end)] = np.nonzero(frac > threshold)[0].tolist()
    return out


def visibility_windows(vis_curves, row_ids, visibility_threshold):
    """vis_curves float32 [N,T] (one visibility curve per (frame, mask), the K2 output); row_ids: list of
    {'frame_id','object_id'} per row.  Returns the reference's `clusters` list (identify_visibility_windows.py:113-215)."""
    x = np.asarray(vis_curves, np.float32)
    labels = DBSCAN(eps=0.2, min_samples=5, metric="hamming").fit(x > visibility_threshold).labels_
    vis_all = (x > visibility_threshold).astype(np.float32)
    mask = labels != -1
    vis, labs = vis_all[mask], labels[mask]
    rid = [row_ids[i] for i in range(len(row_ids)) if mask[i]]
    out = []
    for l in np.unique(labs):
        idxs = np.nonzero(labs == l)[0]
        cv = vis[idxs]
        n_i = cv.shape[0]
        maj = (cv.sum(0) > (n_i / 2)).astype(np.float32)
        ranges = get_visible_ranges(maj)
        winners = get_highly_visible_rows(cv, ranges, threshold=0.3)
        all_cand, all_vis = [], []
        for start_end, rows in winners.items():
            cands = []
            for row in rows:
                r = int(idxs[row])
                all_vis.append({"frame_id": rid[r]["frame_id"], "mask_id": rid[r]["object_id"]})
                if start_end[0] <= rid[r]["frame_id"] <= start_end[1]:
                    cands.append({"start_frame": start_end[0], "end_frame": start_end[1], "frame_id": rid[r]["frame_id"],
                                  "mask_id": rid[r]["object_id"]})
            all_cand.append({"range": start_end, "candidates": cands})
        out.append({"cluster_id": int(l), "cluster_size": n_i, "ranges": ranges, "all_candidates": all_cand,
                    "all_visible_masks": all_vis})
    return out


def crop_bool_tensor(a):
    rows, cols = np.any(a, 1), np.any(a, 0)
    if not rows.any() or not cols.any():
        return np.zeros((0, 0), bool), (0, 0)
    ri, ci = np.where(rows)[0], np.where(cols)[0]
    return a[ri[0]:ri[-1] + 1, ci[0]:ci[-1] + 1], (int(ri[0]), int(ci[0]))


def temporal_groups(match_matrix):
    """match_matrix float32 [n,n] 0/1 (row = tracked mask, col = matched mask, overall mask ids).  Returns
    (labels per cropped row, (row_offset, col_offset), factor) or None when the matrix is empty
    (cotracker_matching.py:786-818)."""
    m, off = crop_bool_tensor(np.asarray(match_matrix, np.float32))
    if m.shape[0] == 0 or m.shape[1] == 0:
        return None
    if m.shape[1] > 50:
        eps, min_samples = 0.05, 5
    elif m.shape[1] < 10:
        eps, min_samples = 0.1, 3
    else:
        eps, min_samples = 0.1, 5
    labels = DBSCAN(eps=eps, min_samples=min_samples, metric="hamming").fit(m).labels_.copy()
    labels[m.sum(1) == 0] = -1
    factor = len(set(labels[labels != -1].tolist()))
    return labels, off, factor

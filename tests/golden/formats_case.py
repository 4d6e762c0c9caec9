"""Synthetic keymask results shared by the golden generator (make_golden.py g_formats, which runs the reference's writers on
them) and tests/test_gpu_formats.py (which runs this repo's writers on them): inputs only, no reference code."""
import os

import numpy as np

from s2d_amd.utils import synth


def formats_case():
    """synthetic keymask results shared by the generator and the test: a 6-frame 20x28 id map with 3 objects, two visibility
    clusters with candidate lists, a grouping of the candidates, one2x scores"""
    T, H, W = 6, 20, 28
    rng = synth.rng_for(131, 0)
    ids = np.zeros((T, 1, H, W), np.int64)
    for t in range(T):
        for o, (cy, cx) in enumerate(((5 + t, 6 + t), (12, 20 - t), (15 - t, 8))):
            yy, xx = np.mgrid[0:H, 0:W]
            ids[t, 0][((yy - cy) / 3.5) ** 2 + ((xx - cx) / 4.5) ** 2 <= 1] = o + 1
    cand = lambda fr, mids: [{"frame_id": int(f), "mask_id": int(m)} for f in fr for m in mids]
    visibility = {"video_name": "vid_0007", "clusters": [
        {"cluster_id": 0, "ranges": [[0, 2], [4, 5]], "all_candidates": [{"range": [0, 2], "candidates": cand((0, 1, 2), (1, 2))},
                                                                         {"range": [4, 5], "candidates": cand((4, 5), (1,))}]},
        {"cluster_id": 1, "ranges": [[1, 4]], "all_candidates": [{"range": [1, 4], "candidates": cand((1, 3, 4), (3, -1))}]}]}
    groupings = [{"cluster_id": 0, "overall_mask_ids_per_label": {0: [(0, 1), (1, 1), (2, 1), (4, 1)], 1: [(0, 2), (2, 2), (3, 9)]}},
                 {"cluster_id": 1, "overall_mask_ids_per_label": {0: [(1, 3), (3, 3), (4, 3)]}}]
    one2x = {"cluster_0": {"group_0": {"avg_one2x": 0.123456}, "group_1": {"avg_one2x": 0.987}}, "cluster_1": {"group_0": {"avg_one2x": 0.5}}}
    merge_inputs = []
    for v in range(3):
        anns = [{"video_id": 1, "iscrowd": 0, "height": H, "width": W, "length": T, "segmentations": [None] * T, "bboxes": [None] * T,
                 "areas": [None] * T, "category_id": 7, "id": a + 1, "one2x": float(round(0.2 + 0.3 * a + 0.05 * v, 2)),
                 "visibility_ranges": [[0, T - 1]]} for a in range(2 + v % 2)]
        merge_inputs.append({"videos": [{"id": 1, "height": H, "width": W, "length": T, "file_names": [f"v{v}/{t:05d}.jpg" for t in range(T)]}],
                             "annotations": anns, "categories": [{"supercategory": "object", "id": 1, "name": "fg"}]})
    merge_inputs.append({"annotations": [], "categories": []})        # a file without a videos block is skipped
    return dict(T=T, H=H, W=W, ids=ids, visibility=visibility, groupings=groupings, one2x=one2x, merge_inputs=merge_inputs)


def tree(root):
    """{relative path: uint8 array} of every PNG under root + the sorted list of all relative paths"""
    from PIL import Image
    files, arrays = [], {}
    for d, _, fs in os.walk(root):
        for f in fs:
            rel = os.path.relpath(os.path.join(d, f), root)
            files.append(rel)
            if f.endswith(".png"):
                arrays[rel] = np.array(Image.open(os.path.join(d, f)))
    return sorted(files), arrays

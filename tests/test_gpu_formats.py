"""On-disk formats either side of keymask discovery: this repo's writers (s2d_amd/keymask/formats.py;
mask selection and run-length encoding on the device) against the file trees, PNG pixels and JSON documents the reference's
own writers produced on the same synthetic results (tests/golden/formats.json, formats_png.npz; generator make_golden.py
g_formats)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_mask_trees_annotation_json_and_merge_vs_reference(tmp_path):
    from PIL import Image
    from formats_case import formats_case, tree
    from s2d_amd.keymask import formats as F
    g = json.load(open(os.path.join(HERE, "golden", "formats.json")))
    png = np.load(os.path.join(HERE, "golden", "formats_png.npz"))
    c = formats_case()
    T, H, W = c["T"], c["H"], c["W"]
    d = str(tmp_path)
    # cluster mask tree (keymask_utils.py:70-126)
    vdir = F.save_segmentation_masks(None, None, torch.from_numpy(c["ids"]), {"visibility": c["visibility"]}, os.path.join(d, "masks"))
    assert os.path.relpath(vdir, d) == g["seg_video_dir"]
    files, arrays = tree(os.path.join(d, "masks"))
    assert files == g["seg_files"]
    for rel, a in arrays.items():
        np.testing.assert_array_equal(a, png["seg/" + rel], err_msg=rel)
    # temporal group tree (cotracker_matching.py:402-431), incl. removal of a stale group directory
    cluster_masks = []
    import re
    for cid in range(2):
        lst = []
        for rel, a in sorted(arrays.items()):
            m = re.match(rf"vid_0007/cluster_{cid}/cluster{cid}_frame(\d+)_mask(-?\d+)\.png", rel)
            if m:
                lst.append({"frame_id": int(m.group(1)), "mask_id": int(m.group(2)), "mask": a})
        cluster_masks.append(lst)
    gpath = os.path.join(d, "masks", "vid_0007")
    os.makedirs(os.path.join(gpath, "cluster_0", "group_9"))
    F.save_temporal_group_masks(c["groupings"], cluster_masks, gpath)
    files2, arrays2 = tree(gpath)
    assert files2 == g["group_files"]
    for rel, a in arrays2.items():
        np.testing.assert_array_equal(a, png["grp/" + rel], err_msg=rel)
    # per-video YTVIS JSON (annotations.py:8-139): every key, order of annotations, rounding, null frames; RLE strings / areas /
    # boxes from the device encoder equal the ones in the golden (the oracle's restatement of pycocotools: parity unpinned)
    json.dump(c["one2x"], open(os.path.join(gpath, "video_one2x_data.json"), "w"))
    vpath = os.path.join(d, "frames", "vid_0007")
    os.makedirs(vpath)
    for t in range(T):
        Image.fromarray(np.zeros((H, W, 3), np.uint8)).save(os.path.join(vpath, f"{t:05d}.jpg"))
    out = F.write_annotation_for_video(vpath, gpath, os.path.join(d, "ann"), c["visibility"])
    assert os.path.basename(out) == "vid_0007.json"
    assert json.load(open(out)) == g["annotation"]
    # dataset JSON (merge_ytvis_jsons.py:24-96), unfiltered and with the one2x filter
    src = os.path.join(d, "per_video")
    os.makedirs(src)
    for i, doc in enumerate(c["merge_inputs"]):
        json.dump(doc, open(os.path.join(src, f"video_{i:02d}.json"), "w"))
    for name, thr in (("merged_all", -1.0), ("merged_filtered", 0.5)):
        m = F.merge_ytvis_jsons(src, os.path.join(d, name + ".json"), thr)
        assert m == g[name] and json.load(open(os.path.join(d, name + ".json"))) == g[name]
    with pytest.raises(SystemExit):
        F.merge_ytvis_jsons(os.path.join(d, "ann", "nothing_here"), os.path.join(d, "x.json"))


def test_select_masks_vs_oracle_at_480p():
    """the batched (frame, object) mask selection at BASELINE config-3 size against numpy, incl. the all-objects id -1"""
    from s2d_amd.keymask import formats as F
    rng = np.random.default_rng(3)
    T, H, W = 32, 480, 854
    ids = rng.integers(0, 7, (T, H, W)).astype(np.int64)
    frames = rng.integers(0, T, 40)
    objs = rng.integers(-1, 7, 40)
    got = F.select_masks(torch.from_numpy(ids), frames, objs)
    for k in range(40):
        want = (ids[frames[k]] != 0) if objs[k] < 0 else (ids[frames[k]] == objs[k])
        np.testing.assert_array_equal(got[k], want.astype(np.uint8) * 255)
    with pytest.raises(IndexError):
        F.select_masks(torch.from_numpy(ids), [T], [1])

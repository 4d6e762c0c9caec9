"""On-disk formats either side of keymask discovery, with the reference's file names, directory
layout and JSON schema so that its downstream readers (`data_video/datasets/ytvis.py:259-388`, `convert_results_to_annotations.py`)
consume the output unchanged:

  * cluster mask trees     `<save_dir>/<video>/cluster_<c>/cluster<c>_frame<f>_mask<m>.png`
                           (keymask_ident/keymask_utils.py:70-126, save_segmentation_masks)
  * temporal group trees   `<path>/cluster_<c>/group_<g>/frame<f>_mask<m>.png`
                           (keymask_ident/cotracker_matching.py:402-431, save_temporal_group_masks)
  * per-video YTVIS JSON   COCO-RLE segmentations / boxes / areas per frame, one annotation per (cluster, group)
                           (keymask_ident/annotations.py:8-139, write_annotation_for_video)
  * dataset JSON           all per-video files merged, ids renumbered, optional one2x filter
                           (keymask_ident/merge_ytvis_jsons.py:24-96)

What runs on the device: the (frame, object) -> binary-mask selection for a whole video in one launch
(`s2d_idmap_select_masks_u8`) and the run-length encoding of every annotation frame of a video in one pass (`s2d_amd.rle`,
csrc/rle.hip).  PNG / JSON encoding and the directory walks are host I/O, as in the reference (PIL / json).  The RLE strings
follow pycocotools' published format; pycocotools is not in the reference tree, so that part is parity-unpinned
(s2d_amd/rle.py); everything else here is pinned by goldens produced by the reference's own functions
(tests/golden/make_golden.py g_formats)."""
import copy
import glob
import json
import os
import re
import shutil

import numpy as np
import torch
from PIL import Image

from .. import rle as _rle
from .._lib import lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def extract_visibility_data(visibility_data):
    """keymask_utils.py:19-34: per cluster the list of {'range', 'mask_candidates'} of its visibility windows"""
    per_cluster = [[{"range": c["range"], "mask_candidates": c["candidates"]} for c in cl["all_candidates"]]
                   for cl in visibility_data["clusters"]]
    return per_cluster, visibility_data["video_name"]


def select_masks(lbls, frames, objs):
    """lbls: the video's id map, (T,1,H,W) or (T,H,W[,1]) integer tensor; frames / objs: K (frame index, object id) pairs
    (object id -1 = every non-background id) -> uint8 numpy [K,H,W] with values {0,255}: get_segmentation_mask
    (keymask_utils.py:37-67) for all candidates in one launch and one device-to-host copy"""
    ids = lbls
    if ids.dim() == 4:
        ids = ids[:, 0] if ids.shape[1] == 1 else ids[..., 0]
    ids = ids.to(device="cuda", dtype=torch.int64).contiguous()
    T, H, W = ids.shape
    K = len(frames)
    if K == 0:
        return np.zeros((0, H, W), np.uint8)
    fr = np.asarray(frames, np.int64)
    if fr.min() < -T or fr.max() >= T:
        raise IndexError("frame index out of range")                  # what indexing the tensor raises in the reference
    fr = np.where(fr < 0, fr + T, fr).astype(np.int32)
    f_d = torch.from_numpy(fr).cuda()
    o_d = torch.from_numpy(np.asarray(objs, np.int32)).cuda()
    out = torch.empty((K, H, W), device="cuda", dtype=torch.uint8)
    lib().call("s2d_idmap_select_masks_u8", ids, T, H, W, f_d, o_d, K, out, _stream())
    return out.cpu().numpy()


def save_segmentation_masks(imgs, imgs_orig, lbls, meta, save_dir, debug=False):
    """keymask_utils.py:70-126, same signature and return value (the video's directory).  imgs / imgs_orig are only sliced
    for debug prints in the reference and are not read here."""
    per_cluster, video_name = extract_visibility_data(meta["visibility"])
    video_dir = os.path.join(save_dir, video_name)
    os.makedirs(video_dir, exist_ok=True)
    jobs = []
    for cid, cluster in enumerate(per_cluster):
        for window in cluster:
            for cand in window["mask_candidates"]:
                jobs.append((cid, int(cand["frame_id"]), int(cand["mask_id"])))
    masks = select_masks(lbls, [j[1] for j in jobs], [j[2] for j in jobs])
    for (cid, fid, mid), m in zip(jobs, masks):
        d = os.path.join(video_dir, f"cluster_{cid}")
        os.makedirs(d, exist_ok=True)
        Image.fromarray(m).save(os.path.join(d, f"cluster{cid}_frame{fid}_mask{mid}.png"))
    return video_dir


def save_temporal_group_masks(mask_groupings, cluster_masks, visibility_group_mask_path, idx_correction=0):
    """cotracker_matching.py:402-431: every cluster's group_* directories are rebuilt from the grouping result;
    cluster_masks[c] = list of {'frame_id', 'mask_id', 'mask' (uint8 HxW)}"""
    for grouping in mask_groupings:
        cid = grouping["cluster_id"]
        cdir = os.path.join(visibility_group_mask_path, f"cluster_{cid}")
        for old in glob.glob(os.path.join(cdir, "group_*")):
            shutil.rmtree(old)
        for gid, members in grouping["overall_mask_ids_per_label"].items():
            gdir = os.path.join(cdir, f"group_{gid}")
            os.makedirs(gdir, exist_ok=True)
            pool = cluster_masks[cid - idx_correction] if cid >= len(cluster_masks) else cluster_masks[cid]
            by_key = {}
            for m in pool:                                            # first match wins, as next(...) in the reference
                by_key.setdefault((m["frame_id"], m["mask_id"]), m)
            for fid, mid in members:
                hit = by_key.get((fid, mid))
                if hit is not None:
                    Image.fromarray(hit["mask"]).save(os.path.join(gdir, f"frame{fid}_mask{mid}.png"))


def write_annotation_for_video(video_path, cluster_masks_path, annotation_output_path, visibility_data):
    """annotations.py:8-139: one YTVIS-style JSON per video.  Every annotation frame of the video is run-length encoded on
    the device in one batch."""
    video_name = os.path.basename(video_path)
    video_files = sorted(f for f in os.listdir(video_path) if f.endswith((".jpg", ".png", ".jpeg")))
    if not video_files:
        return None
    with Image.open(os.path.join(video_path, video_files[0])) as img:
        width, height = img.size
    num_frames = len(video_files)
    video_data = {"license": 1, "coco_url": "", "height": height, "width": width, "length": num_frames,
                  "date_captured": "2019-04-11 00:55:41.903902",
                  "file_names": [os.path.join(video_name, f) for f in video_files], "flickr_url": "", "id": 1}
    cluster_dirs = sorted(d for d in os.listdir(cluster_masks_path)
                          if os.path.isdir(os.path.join(cluster_masks_path, d)) and d.startswith("cluster_")
                          and any(f.endswith(".png") for f in os.listdir(os.path.join(cluster_masks_path, d))))
    with open(os.path.join(cluster_masks_path, "video_one2x_data.json")) as f:
        one2x = json.load(f)
    pending, planes = [], []                                           # (annotation index, frame index) per mask plane
    annotations = []
    for cname in cluster_dirs:
        cdir = os.path.join(cluster_masks_path, cname)
        gdirs = sorted(d for d in os.listdir(cdir) if os.path.isdir(os.path.join(cdir, d)) and d.startswith("group_"))
        try:
            c_id = int(cname.replace("cluster_", ""))
            cv = next((c for c in visibility_data["clusters"] if c["cluster_id"] == c_id), None)
            ranges = cv["ranges"]
        except KeyError:
            ranges = [(-1, -1)]
        if cname not in one2x:
            continue
        for gname in gdirs:
            gdir = os.path.join(cdir, gname)
            ann = {"video_id": video_data["id"], "iscrowd": 0, "height": height, "width": width, "length": num_frames,
                   "segmentations": [None] * num_frames, "bboxes": [None] * num_frames, "areas": [None] * num_frames,
                   "category_id": 1, "id": len(annotations) + 1,
                   "one2x": round(float(one2x[cname][gname]["avg_one2x"]), 2), "visibility_ranges": ranges}
            for mf in (f for f in os.listdir(gdir) if f.endswith(".png")):
                m = re.search(r"frame(\d+)", mf)
                if not m or int(m.group(1)) >= num_frames:
                    continue
                arr = np.array(Image.open(os.path.join(gdir, mf)).convert("L")) > 0
                pending.append((len(annotations), int(m.group(1))))
                planes.append(arr)
            annotations.append(ann)
    # one device pass per mask size (all masks of a video share one size)
    by_shape = {}
    for i, p in enumerate(planes):
        by_shape.setdefault(p.shape, []).append(i)
    for shape, idxs in by_shape.items():
        batch = torch.from_numpy(np.stack([planes[i] for i in idxs]).astype(np.uint8)).cuda()
        rles, areas, boxes = _rle.encode(batch)
        for i, r, a, b in zip(idxs, rles, areas, boxes):
            ai, fi = pending[i]
            annotations[ai]["segmentations"][fi] = {"size": r["size"], "counts": r["counts"].decode("ascii")}
            annotations[ai]["areas"][fi] = int(a)
            annotations[ai]["bboxes"][fi] = [float(v) for v in b]
    data = {"videos": [video_data], "annotations": annotations, "categories": [{"supercategory": "object", "id": 1, "name": "fg"}]}
    os.makedirs(annotation_output_path, exist_ok=True)
    out = os.path.join(annotation_output_path, f"{video_name}.json")
    with open(out, "w") as f:
        json.dump(data, f)
    return out


def merge_ytvis_jsons(src_dir, out_file, one2x_threshold=-1.0):
    """merge_ytvis_jsons.py:24-96: per-video JSONs (sorted by file name) -> one dataset JSON; videos and annotations renumbered
    from 1, category forced to 1, annotations whose one2x exceeds a positive threshold dropped.  Returns the merged dict."""
    paths = sorted(glob.glob(os.path.join(os.path.abspath(src_dir), "*.json")))
    if not paths:
        raise SystemExit(f"No *.json files found in {os.path.abspath(src_dir)}")
    merged = {"info": "Merged YouTube-VOS style dataset",
              "licenses": {"url": "https://creativecommons.org/licenses/by/4.0/", "id": 1,
                           "name": "Creative Commons Attribution 4.0 License"},
              "videos": [], "categories": [{"supercategory": "object", "id": 1, "name": "fg"}], "annotations": []}
    vid = 1
    for p in paths:
        with open(p, "r", encoding="utf-8") as fh:
            data = json.load(fh)
        if not data.get("videos"):
            continue
        video = copy.deepcopy(data["videos"][0])
        video["id"] = vid
        merged["videos"].append(video)
        for ann in data.get("annotations", []):
            if one2x_threshold > 0 and ann.get("one2x", 0.0) > one2x_threshold:
                continue
            a = copy.deepcopy(ann)
            a["id"] = len(merged["annotations"]) + 1
            a["video_id"] = vid
            a["category_id"] = 1
            merged["annotations"].append(a)
        vid += 1
    os.makedirs(os.path.dirname(os.path.abspath(out_file)), exist_ok=True)
    with open(out_file, "w", encoding="utf-8") as fh:
        json.dump(merged, fh, indent=2)
    return merged

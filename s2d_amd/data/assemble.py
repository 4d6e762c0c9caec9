"""Per-frame instance assembly of the dataset mapper (model_training/mask2former_video/data_video/dataset_mapper.py:297-304,
:372-404 `gather_dataset_dict`, and `filter_empty_instances` :29-56) for annotations whose masks are already rasterised and
transformed on the device (the rasterisation / transformation itself is detectron2's `transform_instance_annotations` /
`annotations_to_instances`: third party, absent from the reference tree).

What the mapper does and this mirrors: the instance ids of the selected frames are collected into a Python `set` and numbered in
its ITERATION order (`ids[_id] = i`, :298-303); every frame gets one slot per id -- a dummy annotation (empty mask, id -1, class
`num_classes`) where the instance is not annotated, crowd annotations dropped (:376-389) -- and `filter_empty_instances` does not
remove anything: it sets `gt_ids` to -1 where the box or the mask is empty (:55).  So every frame of a clip has the same number of
slots, which is what KDVideoMaskFormer.prepare_targets relies on (kd_video_maskformer_model.py:358-386)."""
import numpy as np
import torch


def clip_id_slots(video_annos, selected_idx):
    """{annotation id: slot}: slots number the clip's distinct annotation ids in the ITERATION order of the Python set they were
    collected into frame by frame (dataset_mapper.py:297-303) -- that order, not sorted order, is the contract the goldens pin"""
    distinct = set()
    for f in selected_idx:
        distinct.update(a["id"] for a in video_annos[f])
    return {aid: slot for slot, aid in enumerate(distinct)}


def assemble_clip_instances(video_annos, selected_idx, image_shape, num_classes, device="cuda", box_threshold=1e-5):
    """video_annos[frame] = list of {"id", "category_id", "iscrowd" (optional), "mask": bool/u8 [H, W] tensor or array, already
    in the augmented frame's geometry}.  -> per selected frame {"gt_masks" bool [S, H, W], "gt_ids" int64 [S], "gt_classes" int64
    [S]} with S = number of distinct ids in the clip."""
    ids = clip_id_slots(video_annos, selected_idx)
    S = len(ids)
    H, W = image_shape
    out = []
    for frame_idx in selected_idx:
        masks = torch.zeros((S, H, W), dtype=torch.bool, device=device)
        gt_ids = np.full((S,), -1, np.int64)
        classes = np.full((S,), num_classes, np.int64)
        for anno in video_annos[frame_idx]:
            if anno.get("iscrowd", 0) != 0:
                continue
            idx = ids[anno["id"]]
            masks[idx] = torch.as_tensor(anno["mask"]).to(device=device, dtype=torch.bool)
            gt_ids[idx] = anno["id"]
            classes[idx] = anno["category_id"]
        if S:
            # filter_empty_instances (:29-56): box from the mask (BitMasks.get_bounding_boxes), non-empty iff wider and taller than
            # the threshold; mask non-empty iff any pixel -- for a box taken from the mask the two tests coincide
            cols, rows = masks.any(1), masks.any(2)
            xs = torch.arange(W, device=device)[None].expand(S, W)
            ys = torch.arange(H, device=device)[None].expand(S, H)
            x0 = torch.where(cols, xs, W).amin(1); x1 = torch.where(cols, xs + 1, 0).amax(1)
            y0 = torch.where(rows, ys, H).amin(1); y1 = torch.where(rows, ys + 1, 0).amax(1)
            keep = ((x1 - x0).float() > box_threshold) & ((y1 - y0).float() > box_threshold) & masks.flatten(1).any(1)
            gt_ids[~keep.cpu().numpy()] = -1
        out.append({"gt_masks": masks, "gt_ids": gt_ids, "gt_classes": classes})
    return out

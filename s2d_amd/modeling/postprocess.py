"""Eval-side step after the path: `inference_video` of both meta-architectures
(model_training/mask2former_video/kd_video_maskformer_model.py:530-610, video_maskformer_model.py:298-360) on the
device kernels of csrc/infer.hip.  The python side keeps only what the reference keeps on the host: the greedy loop over
K <= a few hundred candidates (on the K x K integer counts, not on the masks) and the final lists."""
import numpy as np
import torch

from .. import ops


def greedy_mask_nms(inter, labels, thr):
    """:552-583 on the pairwise counts: inter [K,K] int64 numpy (diagonal = areas), labels [K]; candidates are already in
    descending score order.  IoU = float32(sum(a & b)) / float32(sum(a | b)) as the reference forms it, 0 if the union is
    empty; a later same-label candidate is suppressed when IoU > thr."""
    K = inter.shape[0]
    area = np.diagonal(inter)
    alive = list(range(K))
    keep = []
    thr = np.float32(thr)
    while alive:
        cur = alive.pop(0)
        keep.append(cur)
        rem = []
        for o in alive:
            if labels[o] != labels[cur]:
                rem.append(o)
                continue
            i = np.float32(inter[cur, o])
            u = np.float32(area[cur] + area[o] - inter[cur, o])
            iou = i / u if u > 0 else np.float32(0.0)
            if iou <= thr:
                rem.append(o)
        alive = rem
    return keep


@torch.no_grad()
def inference_video(class_logits, mask_logits, dims, padded, img_size, out_size, num_predictions, use_nms=False,
                    nms_threshold=0.75, rle=False):
    """class_logits [Q,C+1]; mask_logits pixel-major [T*hm*wm, ldq] of ONE video; dims = (T, hm, wm); padded = network
    input size (Hp,Wp), img_size = size without padding, out_size = (height, width) of the original video.
    Returns the reference's dict: image_size, pred_scores (list of float), pred_labels (list of int), pred_masks (list of
    CPU bool tensors [T,H,W]).  rle=True: the masks stay on the device and `pred_masks` holds, per prediction, the list of T
    COCO RLE dicts that instances_to_coco_json_video (data_video/ytvis_eval.py:345-350) would build from them."""
    if class_logits.shape[0] == 0:                                   # :584-587
        return {"image_size": tuple(out_size), "pred_scores": [], "pred_labels": [], "pred_masks": []}
    scores, query, label = ops.infer_select(class_logits, num_predictions)
    masks, bits = ops.infer_masks(mask_logits, dims, padded, img_size, out_size, query, want_bits=use_nms)
    if use_nms:
        inter = ops.mask_pair_counts(bits).cpu().numpy()
        labels_h = label.cpu().numpy()
        keep = greedy_mask_nms(inter, labels_h, nms_threshold)
        sel = torch.as_tensor(keep, device=masks.device, dtype=torch.long)
        masks, scores, label = masks[sel], scores[sel], label[sel]
    if rle:
        from ..rle import encode_video_predictions
        return {"image_size": tuple(out_size), "pred_scores": scores.tolist(), "pred_labels": label.tolist(),
                "pred_masks": encode_video_predictions(masks), "pred_masks_format": "coco_rle"}
    masks_h = torch.empty(masks.shape, dtype=torch.uint8, pin_memory=True)   # pinned: the copy runs at PCIe rate, not pageable rate
    masks_h.copy_(masks, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    masks_h = masks_h.view(torch.bool)
    return {"image_size": tuple(out_size), "pred_scores": scores.tolist(), "pred_labels": label.tolist(),
            "pred_masks": [m for m in masks_h]}

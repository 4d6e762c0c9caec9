"""Seeded inputs of the copy-paste / propagate_sparse_masks fixture (tests/golden/copy_paste.npz): shared by the generator
(make_golden.py g_copy_paste, which feeds them to the reference's own functions) and by tests/test_gpu_data.py.  Data only."""
import numpy as np

from s2d_amd.utils import synth


def copy_paste_case(case):
    """seeded inputs of one copy-paste fixture case (shared by the generator and tests/test_gpu_data.py): a labelled source clip
    with Ks dense instances and an unlabelled-style target clip with sparse per-frame instances, as the mapper hands them to the
    trainer (data_video/dataset_mapper.py:306-404: per frame `image` u8 [3,H,W] and Instances with gt_masks / gt_ids / gt_classes)"""
    seed, T, (H, W), (Hs, Ws), Ks, N, sparse = case["seed"], case["T"], case["hw"], case["src_hw"], case["Ks"], case["N"], case["sparse"]
    src_f = synth.smooth_frames_u8(seed, 1, T, Hs, Ws)
    tgt_f = synth.smooth_frames_u8(seed, 2, T, H, W)
    sm, sids = synth.ellipse_targets(seed, 3, max(Ks, 1), T, Hs, Ws, sparse=0.0)
    tm, tids = synth.ellipse_targets(seed, 4, max(N, 1), T, H, W, sparse=sparse, rmax=case.get("rmax"))
    if Ks == 0:
        sm, sids = sm[:0], sids[:0]
    if N == 0:
        tm, tids = tm[:0], tids[:0]

    def frames(masks, ids, imgs, dense):
        inst = []
        for t in range(T):
            sel = np.arange(masks.shape[0]) if dense else np.nonzero(ids[:, t] >= 0)[0]
            inst.append({"gt_masks": masks[sel, t].astype(bool), "gt_ids": np.where(ids[sel, t] >= 0, ids[sel, t], sel).astype(np.int64) + case.get("id0", 0),
                         "gt_classes": np.zeros(len(sel), np.int64)})
        return {"image": [imgs[t] for t in range(T)], "instances": inst}
    src = frames(sm, sids, src_f, True)
    tgt = frames(tm, tids, tgt_f, False)
    for t in range(T):
        tgt["instances"][t]["gt_ids"] = tgt["instances"][t]["gt_ids"] + 100         # ids of the two clips do not collide
    return src, tgt


COPY_PASTE_CASES = [
    dict(name="paste", seed=11, T=3, hw=(40, 56), src_hw=(32, 48), Ks=3, N=2, sparse=0.0, rmax=5.0, cfg=dict(rate=1.0, random_num=False, lo=0.8, hi=1.0, densify=False)),
    dict(name="paste_sparse_targets", seed=12, T=4, hw=(40, 56), src_hw=(48, 40), Ks=2, N=3, sparse=0.5, rmax=5.0, cfg=dict(rate=1.0, random_num=False, lo=0.5, hi=0.9, densify=False)),
    dict(name="random_num", seed=13, T=3, hw=(36, 52), src_hw=(36, 52), Ks=4, N=1, sparse=0.0, rmax=4.0, cfg=dict(rate=1.0, random_num=True, lo=0.6, hi=1.0, densify=False)),
    dict(name="cancel_by_overlap", seed=14, T=3, hw=(40, 56), src_hw=(40, 56), Ks=4, N=3, sparse=0.0, rmax=4.0, cfg=dict(rate=1.0, random_num=False, lo=1.0, hi=1.0, densify=False)),
    dict(name="rate_miss", seed=15, T=3, hw=(40, 56), src_hw=(32, 48), Ks=3, N=2, sparse=0.3, cfg=dict(rate=0.0, random_num=False, lo=0.8, hi=1.0, densify=False)),
    dict(name="densify_only", seed=16, T=5, hw=(40, 56), src_hw=(32, 48), Ks=2, N=3, sparse=0.5, cfg=dict(rate=1.0, random_num=False, lo=0.8, hi=1.0, densify=True)),
    dict(name="no_source_instances", seed=17, T=3, hw=(40, 56), src_hw=(32, 48), Ks=0, N=2, sparse=0.0, cfg=dict(rate=1.0, random_num=False, lo=0.8, hi=1.0, densify=False)),
    dict(name="no_target_instances", seed=18, T=3, hw=(40, 56), src_hw=(32, 48), Ks=2, N=0, sparse=0.0, cfg=dict(rate=1.0, random_num=False, lo=0.7, hi=0.9, densify=False)),
]

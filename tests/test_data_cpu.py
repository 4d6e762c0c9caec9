"""Data-side host logic against goldens produced by the reference's own mapper / augmentation classes
(tests/golden/sampling.json, generator make_golden.py g_sampling): clip frame selection (dataset_mapper.py:223-289) with the
reference's RNG consumption, and the shortest-edge size rule (augmentation.py:51-75)."""
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _cases():
    rng = np.random.default_rng(17)                      # must stay identical to make_golden._sampling_cases
    cases = []
    for L in (12, 30, 7, 3):
        annos = []
        for t in range(L):
            present = [i for i in range(4) if rng.random() < (0.75 if L != 7 else 0.3)]
            annos.append([{"id": i} for i in present])
        cases.append((L, annos))
    return cases


def test_frame_selection_matches_the_reference_mapper():
    from s2d_amd.data import dense_frame_selection, random_frame_selection
    g = json.load(open(os.path.join(HERE, "golden", "sampling.json")))
    cases = _cases()
    assert len(g["dense"]) == 72 and len(g["random"]) > 40
    for e in g["dense"]:
        L, annos = cases[e["case"]]
        random.seed(e["seed"]); np.random.seed(e["seed"])
        sel = dense_frame_selection(annos, L, e["n"], e["range"], e["shuffle"])
        assert [int(v) for v in sel] == e["sel"], e
    for e in g["random"]:
        random.seed(e["seed"]); np.random.seed(e["seed"])
        sel = random_frame_selection(e["L"], e["n"], e["range"], e["shuffle"])
        assert [int(v) for v in sel] == e["sel"], e


def test_resize_shortest_edge_size_rule_matches_the_reference():
    from s2d_amd.data import ClipAugmentation
    g = json.load(open(os.path.join(HERE, "golden", "sampling.json")))
    for e in g["resize"]:
        aug = ClipAugmentation(min_size=e["sizes"], max_size=e["max_size"], sample_style=e["style"], random_flip="none", num_frames=3)
        np.random.seed(7)
        got = []
        if "by_clip" in e["style"]:
            for _ in range(2):                            # two clips of three frames: one draw each
                P, hw = aug.sample(3, e["hw"][0], e["hw"][1])
                got += [list(hw)] * 3
                # a pure resize: the map scales by H0 / H1, W0 / W1 and nothing else
                np.testing.assert_allclose(P[:, :6], np.tile([e["hw"][1] / hw[1], 0, 0, 0, e["hw"][0] / hw[0], 0], (3, 1)), rtol=1e-6)
        else:
            for _ in range(6):
                _, hw = aug.sample(1, e["hw"][0], e["hw"][1])
                got.append(list(hw))
        assert got == e["new_hw"], e


def test_augmentation_parameters_respect_the_policy():
    """size and flip once per clip, crop / photometric / rotation per frame; every crop lies inside the frame"""
    from s2d_amd.data import ClipAugmentation
    aug = ClipAugmentation(min_size=(360, 480), random_flip="flip_by_clip", augmentations=("brightness", "contrast", "rotation"),
                           crop=("absolute_range", (600, 720)), num_frames=5)
    np.random.seed(3)
    for _ in range(20):
        P, (H1, W1) = aug.sample(5, 720, 1280)
        assert P.shape == (5, 16) and min(H1, W1) in (360, 480)
        assert (P[:, 6] >= 0).all() and (P[:, 7] >= 0).all() and (P[:, 6] + P[:, 8] <= 1280).all() and (P[:, 7] + P[:, 9] <= 720).all()
        assert ((P[:, 8] >= 600) & (P[:, 8] <= 720) & (P[:, 9] >= 600) & (P[:, 9] <= 720)).all()
        assert ((P[:, 10] >= 0.9) & (P[:, 10] <= 1.1) & (P[:, 11] >= 0.9) & (P[:, 11] <= 1.1)).all()
        assert len(set(np.sign(P[:, 0] * np.cos(0.3)).tolist())) == 1          # one flip decision per clip (rotation <= 15 deg keeps a11's sign)


def test_instance_assembly_matches_the_reference_filter():
    """the mapper's per-frame slots (dataset_mapper.py:297-303: ids numbered in `set` iteration order; dummy slots for frames
    where an instance is not annotated; crowd annotations dropped) and `filter_empty_instances` (:29-56: gt_ids -> -1 for an
    empty mask or a degenerate box), against the reference's own filter function run on the same seeded clips
    (tests/golden/assemble.json, make_golden.py g_assemble)"""
    import torch
    from make_golden_cases import assemble_cases
    from s2d_amd.data import assemble_clip_instances, clip_id_slots
    g = json.load(open(os.path.join(HERE, "golden", "assemble.json")))
    for ((H, W), video, sel), want in zip(assemble_cases(), g):
        ids = clip_id_slots(video, sel)
        assert {str(k): v for k, v in ids.items()} == want["slots"]
        frames = assemble_clip_instances(video, sel, (H, W), num_classes=1, device="cpu")
        assert len(frames) == len(sel)
        for fr, ref_ids, f in zip(frames, want["gt_ids"], sel):
            assert fr["gt_ids"].tolist() == ref_ids
            assert fr["gt_masks"].shape == (len(ids), H, W) and fr["gt_masks"].dtype == torch.bool
            for a in video[f]:
                if a.get("iscrowd", 0) == 0:
                    assert torch.equal(fr["gt_masks"][ids[a["id"]]], torch.from_numpy(a["mask"]))
                    assert fr["gt_classes"][ids[a["id"]]] == a["category_id"]
            absent = [s for s in range(len(ids)) if s not in {ids[a["id"]] for a in video[f] if a.get("iscrowd", 0) == 0}]
            assert all(fr["gt_classes"][s] == 1 and not fr["gt_masks"][s].any() for s in absent)

"""K7 host grouping (sklearn DBSCAN on small 0/1 matrices) vs goldens produced by the reference's own functions."""
import json

import numpy as np

from tests.conftest import golden


def test_visibility_windows_golden():
    from s2d_amd.keymask.grouping import visibility_windows
    g = golden("grouping")
    curves = g["curves"]
    T, n_obj = curves.shape[1], 3
    row_ids = [{"frame_id": f, "object_id": o + 1} for f in range(T) for o in range(n_obj)]
    out = visibility_windows(curves, row_ids, 0.3)
    ref = json.loads(bytes(g["clusters_json"]).decode())
    assert len(out) == len(ref) > 0
    for a, b in zip(out, ref):
        assert a["cluster_id"] == b["cluster_id"] and a["cluster_size"] == b["cluster_size"]
        assert [list(r) for r in a["ranges"]] == b["ranges"]
        assert a["all_visible_masks"] == b["all_visible_masks"]
        assert [c["candidates"] for c in a["all_candidates"]] == [c["candidates"] for c in b["all_candidates"]]


def test_temporal_groups_golden():
    from s2d_amd.keymask.grouping import temporal_groups
    g = golden("grouping")
    labels, (ro, co), factor = temporal_groups(g["match_matrix"])
    assert factor == int(g["factor"])
    ref = json.loads(bytes(g["groups_json"]).decode())
    got = {}
    for i, l in enumerate(labels):
        if l != -1:
            got.setdefault(str(int(l)), []).append(i + ro)
    # reference stores (frame_id, mask_id) of each overall id: lookup used in the fixture is id -> (id // 3, id % 3 + 1)
    for l, lst in ref.items():
        ids = sorted(int(e[0]) * 3 + int(e[1]) - 1 for e in lst)
        assert sorted(got[l]) == ids
    assert temporal_groups(np.zeros((5, 5), np.float32)) is None

"""GPU parity of the device COCO-RLE encoder (csrc/rle.hip + s2d_amd/rle.py) vs the CPU restatement of pycocotools'
rleEncode / rleToString / area / toBbox (oracle_np.rle_*; third party absent -> parity unpinned, see tests/test_oracle.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("H,W,F", [(37, 29, 6), (48, 64, 5), (5, 4100, 2), (180, 316, 3), (1, 1, 2)])
def test_encode_vs_oracle(oracle, H, W, F):
    from s2d_amd import rle
    rng = np.random.default_rng(H * 1000 + W)
    frames = [(rng.random((H, W)) < pr).astype(np.uint8) for pr in np.linspace(0.0, 1.0, F)]
    if H > 20:
        frames[1][:] = 0
        frames[1][H // 4:H // 2, W // 5:W // 2] = 1                    # a blob: long runs, multi-char counts
    m = torch.from_numpy(np.stack(frames)).to(DEV)
    rles, area, bbox = rle.encode(m)
    for f, g in enumerate(frames):
        r, _ = oracle.rle_encode(g)
        assert rles[f]["size"] == [H, W]
        assert rles[f]["counts"] == r["counts"], f"frame {f}"
        a, bb = oracle.rle_area_bbox(g)
        assert int(area[f]) == a
        assert bbox[f].tolist() == bb
        np.testing.assert_array_equal(oracle.rle_decode(rles[f]), g)    # round trip through the wire format
    # bool input and the evaluator-shaped helper
    seg = rle.encode_video_predictions(m.view(torch.bool).reshape(1, F, H, W))
    assert len(seg) == 1 and len(seg[0]) == F and isinstance(seg[0][0]["counts"], str)
    assert seg[0][F - 1]["counts"].encode() == rles[F - 1]["counts"]


def test_inference_masks_encode_without_host_copy(oracle):
    """the eval step's masks encoded where they were made: RLE of ops.infer_masks output == RLE of the oracle's masks"""
    from s2d_amd import ops, rle
    from tests.conftest import golden
    from tests.test_oracle import infer_case
    from tests.test_gpu_infer import pixel_major
    c = infer_case(golden("inference"), "agn_plain")
    scores, query, label = ops.infer_select(torch.from_numpy(c["cls"]).to(DEV), c["K"])
    masks, _ = ops.infer_masks(pixel_major(c["masks"]), (c["T"], c["h"], c["w"]), (c["Hp"], c["Wp"]), (c["ih"], c["iw"]),
                               (c["oh"], c["ow"]), query)
    seg = rle.encode_video_predictions(masks)
    for k in range(c["K"]):
        for t in range(c["T"]):
            assert seg[k][t]["counts"].encode() == oracle.rle_encode(c["out"][k, t])[0]["counts"]


def test_inference_video_rle_output_golden(oracle):
    """inference_video(rle=True) on a golden case with NMS: scores / labels as the reference, masks as their RLE"""
    from s2d_amd.modeling.postprocess import inference_video
    from tests.conftest import golden
    from tests.test_oracle import infer_case
    from tests.test_gpu_infer import pixel_major
    c = infer_case(golden("inference"), "multi_nms")
    out = inference_video(torch.from_numpy(c["cls"]).to(DEV), pixel_major(c["masks"]), (c["T"], c["h"], c["w"]), (c["Hp"], c["Wp"]),
                          (c["ih"], c["iw"]), (c["oh"], c["ow"]), c["K"], c["nms"], c["thr"], rle=True)
    assert out["pred_labels"] == c["labels"].tolist() and len(out["pred_masks"]) == len(c["scores"])
    for k, inst in enumerate(out["pred_masks"]):
        for t, r in enumerate(inst):
            np.testing.assert_array_equal(oracle.rle_decode(r), c["out"][k, t])

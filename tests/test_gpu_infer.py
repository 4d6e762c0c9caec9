"""GPU parity of the eval-side step: inference_video on the HIP kernels of csrc/infer.hip vs the
golden vectors from the reference's own inference_video and vs the CPU oracle, through the C ABI."""
import numpy as np
import pytest
import torch

from tests.conftest import golden
from tests.test_oracle import INFER_CASES, infer_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def pixel_major(masks, ldq=None):
    """reference layout [Q,T,h,w] -> the decoder's pixel-major [T*h*w, ldq]"""
    Q, T, h, w = masks.shape
    ldq = ldq or (Q + 3) // 4 * 4
    out = np.full((T * h * w, ldq), np.nan, np.float32)          # padding columns must never be read
    out[:, :Q] = masks.reshape(Q, -1).T
    return torch.from_numpy(out).to(DEV)


@pytest.mark.parametrize("name", INFER_CASES)
def test_inference_video_golden(name):
    from s2d_amd.modeling.postprocess import inference_video
    c = infer_case(golden("inference"), name)
    out = inference_video(torch.from_numpy(c["cls"]).to(DEV), pixel_major(c["masks"]), (c["T"], c["h"], c["w"]), (c["Hp"], c["Wp"]),
                          (c["ih"], c["iw"]), (c["oh"], c["ow"]), c["K"], c["nms"], c["thr"])
    assert out["image_size"] == (c["oh"], c["ow"])
    assert isinstance(out["pred_scores"], list) and isinstance(out["pred_labels"], list)
    np.testing.assert_allclose(out["pred_scores"], c["scores"], rtol=1e-5)
    assert out["pred_labels"] == c["labels"].tolist()
    assert len(out["pred_masks"]) == len(c["scores"])
    for m, e in zip(out["pred_masks"], c["out"]):
        assert m.dtype == torch.bool and m.device.type == "cpu" and tuple(m.shape) == e.shape
        np.testing.assert_array_equal(m.numpy(), e)              # boolean masks bit-exact


@pytest.mark.parametrize("shape", [
    dict(Q=100, C=1, K=20, T=5, h=46, w=80, Hp=184, Wp=320, ih=180, iw=316, oh=360, ow=632),     # 2x up, ow % 4 == 0
    dict(Q=37, C=3, K=50, T=3, h=24, w=40, Hp=96, Wp=160, ih=90, iw=157, oh=67, ow=101),         # odd sizes, words straddle rows
    dict(Q=16, C=1, K=16, T=2, h=23, w=40, Hp=92, Wp=160, ih=92, iw=160, oh=92, ow=160),         # identity second stage
])
def test_select_masks_and_pair_counts_vs_oracle(oracle, shape):
    from s2d_amd import ops
    s = shape
    rng = np.random.default_rng(s["Q"])
    cls = rng.normal(0, 2, (s["Q"], s["C"] + 1)).astype(np.float32)
    lo = rng.normal(0, 1, (s["Q"], s["T"], s["h"] // 4 + 1, s["w"] // 4 + 1)).astype(np.float32)
    masks = oracle.resize_bilinear(lo, s["h"], s["w"]) * 3                                  # smooth fields with both signs
    r = oracle.inference_video(cls, masks, s["Hp"], s["Wp"], s["ih"], s["iw"], s["oh"], s["ow"], s["K"])
    scores, query, label = ops.infer_select(torch.from_numpy(cls).to(DEV), s["K"])
    np.testing.assert_allclose(scores.cpu().numpy(), r["all_scores"], rtol=1e-5)
    np.testing.assert_array_equal(query.cpu().numpy(), r["all_query"])
    np.testing.assert_array_equal(label.cpu().numpy(), r["all_labels"])
    m, bits = ops.infer_masks(pixel_major(masks), (s["T"], s["h"], s["w"]), (s["Hp"], s["Wp"]), (s["ih"], s["iw"]), (s["oh"], s["ow"]),
                              query, want_bits=True)
    m = m.cpu().numpy().astype(bool)
    ref = r["logits"] > 0
    sure = np.abs(r["logits"]) > 1e-5                                                         # sign of a value within rounding of 0
    assert (~sure).mean() < 1e-3
    np.testing.assert_array_equal(m[sure], ref[sure])
    # bit words = little-endian packing of the flat mask, tail bits zero
    N = s["T"] * s["oh"] * s["ow"]
    words = (N + 31) // 32
    flat = np.zeros((s["K"], words * 32), np.uint8)
    flat[:, :N] = m.reshape(s["K"], -1)
    exp_bits = np.packbits(flat, axis=-1, bitorder="little").view(np.uint32)
    np.testing.assert_array_equal(bits.cpu().numpy().view(np.uint32), exp_bits)
    inter = ops.mask_pair_counts(bits).cpu().numpy()
    mf = m.reshape(s["K"], -1).astype(np.int64)
    np.testing.assert_array_equal(inter, mf @ mf.T)                                           # integer counts exact


def test_model_eval_branch_vs_oracle(oracle):
    """KDVideoMaskFormer.forward in eval mode: whole video as one clip through the teacher, then inference_video"""
    from s2d_amd.modeling import build_kd_model
    from s2d_amd.utils import synth
    from tests.parity import seeded_load
    T, H0, W0, Q, NL, K = 3, 60, 90, 12, 4, 5
    model = build_kd_model(num_queries=Q, num_frames=2, num_points=64, dec_layers=NL)
    seeded_load(model.student, 7)
    pt = seeded_load(model.teacher, 8)
    model = model.to(DEV).eval()
    model.num_predictions_inference, model.use_nms, model.nms_threshold = K, True, 0.5
    frames = synth.smooth_frames_u8(9, 1, T, H0, W0)
    video = {"image": [torch.from_numpy(f) for f in frames], "height": 90, "width": 135}
    out = model([video])
    x = oracle.normalize_pad(frames)
    Hp, Wp = x.shape[-2:]
    feats = oracle.resnet50(pt, x, "0.")
    mf, ms = oracle.pixel_decoder(pt, feats, "1.pixel_decoder.")
    t_logits, t_masks = oracle.video_decoder(pt, ms, mf, T, "1.predictor.", n_layers=NL - 1)
    r = oracle.inference_video(t_logits[-1, 0], t_masks[-1, 0], Hp, Wp, H0, W0, 90, 135, K, True, 0.5)
    assert out["image_size"] == (90, 135)
    np.testing.assert_allclose(out["pred_scores"], r["scores"], rtol=1e-3)
    assert out["pred_labels"] == r["labels"].tolist()
    assert len(out["pred_masks"]) == len(r["keep"])
    lg = r["logits"][r["keep"]]
    sure = np.abs(lg) > 1e-3 * np.abs(lg).max()
    got = np.stack([m.numpy() for m in out["pred_masks"]])
    assert got.shape == lg.shape
    np.testing.assert_array_equal(got[sure], (lg > 0)[sure])
    assert sure.mean() > 0.98

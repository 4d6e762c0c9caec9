"""Size-independent properties of the hot path at BASELINE's full metric shapes (T=8, 720p, Q=100, P=160000,
2 clips): run-to-run determinism (bitwise), clip-order equivariance, valid assignments, finite losses."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import bench
    from s2d_amd import ops
    from s2d_amd.modeling import TargetSet, build_kd_model
    dev = torch.device("cuda:0")
    B, T, H0, W0, Q, P, N = bench.CONFIGS["c4"]
    model = build_kd_model(num_queries=Q, num_frames=T, num_points=P).to(dev)
    frames, masks = bench.synth_batch(0, B, T, H0, W0, N, dev)
    bench.calibrate_teacher(model, ops.normalize_pad(frames))
    return model, frames, masks, (B, T, Q, N)


def _run(model, frames, masks):
    from s2d_amd import ops
    from s2d_amd.modeling import TargetSet
    model.criterion.seed = 0
    model.criterion.matcher.seed = 0
    losses = model.forward_losses(ops.normalize_pad(frames), TargetSet.from_list(masks, device=frames.device))
    torch.cuda.synchronize()
    iq, it, nm = (x.cpu().numpy() for x in model.criterion.last_indices)
    return {k: float(v) for k, v in losses.items()}, (iq, it, nm), model.last["kd_count"].cpu().numpy()


def test_fullsize_properties(setup):
    model, frames, masks, (B, T, Q, N) = setup
    l1, (iq1, it1, nm1), kd1 = _run(model, frames, masks)
    assert len(l1) == 42 and all(np.isfinite(v) for v in l1.values())
    # every assignment is a valid partial matching: queries ascending and distinct, targets a permutation
    for p in range(iq1.shape[0]):
        k = nm1[p]
        assert k == min(Q, int(kd1[p % B]))
        q, t = iq1[p, :k], it1[p, :k]
        assert (np.diff(q) > 0).all() and sorted(t.tolist()) == list(range(k))
    # determinism: a second run with the same RNG seeds is bitwise identical (fixed-order reductions everywhere)
    l2, (iq2, it2, nm2), kd2 = _run(model, frames, masks)
    assert l1 == l2 and (iq1 == iq2).all() and (it1 == it2).all() and (kd1 == kd2).all()
    # clips are independent units: swapping the two clips swaps the per-clip KD counts (the path shards by clip)
    fr = frames.view(B, T, *frames.shape[1:]).flip(0).reshape(frames.shape).contiguous()
    l3, _, kd3 = _run(model, fr, masks[::-1])
    assert (kd3 == kd1[::-1]).all()
    # ... and the class loss (no point sampling involved) is unchanged up to summation order over clips
    np.testing.assert_allclose(l3["kd_loss_ce"], l1["kd_loss_ce"], rtol=1e-6, atol=1e-7)


def test_two_stream_schedule_bitwise(setup):
    """the optional two-stream schedule (teacher forward + GT criterion on a second HIP stream) changes no bit of the
    losses, the assignments or the teacher's pseudo targets -- repeated, since a violation would be a race"""
    model, frames, masks, _ = setup
    try:
        model.overlap_teacher = model.overlap_criteria = False
        ref = _run(model, frames, masks)
        model.overlap_teacher = model.overlap_criteria = True
        for _ in range(8):
            cur = _run(model, frames, masks)
            assert cur[0] == ref[0]
            assert all((a == b).all() for a, b in zip(cur[1], ref[1])) and (cur[2] == ref[2]).all()
    finally:
        model.overlap_teacher = model.overlap_criteria = False


def test_teacher_tap_gathered_masks_bitwise(setup):
    """evaluating the teacher's intermediate mask logits only at the attention masks' source pixels (aux_masks=False)
    leaves its final prediction -- all the KD pass reads -- bit-identical"""
    from s2d_amd import ops
    model, frames, _, _ = setup
    images = ops.normalize_pad(frames)
    full = model.teacher(images, True, aux_masks=True)
    thin = model.teacher(images, True, aux_masks=False)
    torch.cuda.synchronize()
    assert thin.mask_logits.shape[0] < full.mask_logits.shape[0]
    assert torch.equal(thin.mask_logits[-1], full.mask_logits[-1])
    assert torch.equal(thin.class_logits, full.class_logits)

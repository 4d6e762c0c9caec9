"""BASELINE configs[4]-shaped inputs on one GPU: a SEQUENCE of differently sized clips through one model instance (the shipped
INPUT block: MIN_SIZE_TRAIN (360, 480), random crop `absolute_range`, ytvis2021_kd_...yaml:108-111; SA-V / MOSE / VIPSeg sources
are 1080p / 480p / 720p), and the DistributedDataParallel-wrapped training call of the reference trainer
(engine/defaults.py:76-85 + engine/train_loop.py:709-726) through _HipGradBridge."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sequence_of_clip_sizes_one_model_vs_oracle(oracle):
    """small clips of different sizes and aspect ratios, back to back through ONE KDVideoMaskFormer, each against the oracle
    (logits 1e-3, KD target counts, all 42 losses 1e-3), then the first size again: it must reproduce its first result bit for
    bit -- nothing the instance caches (position encodings, packed weights, level shapes, tap indices, workspaces) may leak
    from one resolution into the next"""
    from s2d_amd.modeling import build_kd_model
    from tests.parity import run_model_case, seeded_load
    Q, T, P, NL = 16, 2, 256, 10
    model = build_kd_model(num_queries=Q, num_frames=T, num_points=P, weights=(2.0, 5.0, 5.0), dec_layers=NL)
    ps, pt = seeded_load(model.student, 3), seeded_load(model.teacher, 4)
    model = model.to("cuda:0")
    sizes = [(60, 90), (90, 60), (45, 120), (96, 128), (60, 90)]
    first = None
    for i, (H0, W0) in enumerate(sizes):
        model.criterion.seed = 0; model.criterion.matcher.seed = 0
        hip, ref = run_model_case(model, ps, pt, oracle, seed=3, B=2, T=T, H0=H0, W0=W0, Q=Q, P=P, ns=(3, 4), NL=NL)
        for k in ("s_logits", "s_masks"):
            b = ref[k].astype(np.float64)
            np.testing.assert_allclose(hip[k], b, rtol=1e-3, atol=1e-3 * np.abs(b).max(), err_msg=f"clip size {H0}x{W0}: {k}")
        assert hip["kd_counts"] == ref["kd_counts"]
        for k, v in ref["losses"].items():
            np.testing.assert_allclose(hip["losses"][k], float(v), rtol=1e-3, atol=1e-6, err_msg=f"clip size {H0}x{W0}: {k}")
        if i == 0:
            first = hip
    assert first["losses"] == hip["losses"] and np.array_equal(first["s_masks"], hip["s_masks"])


def _batch(seed, T, H0, W0, n):
    from s2d_amd.utils import synth
    frames = synth.smooth_frames_u8(seed, 10, T, H0, W0)
    m, ids = synth.ellipse_targets(seed, 100, n, T, H0, W0)
    inst = [{"gt_masks": torch.from_numpy(m[:, t]).bool(), "gt_ids": torch.from_numpy(ids[:, t]),
             "gt_classes": torch.zeros(n, dtype=torch.int64)} for t in range(T)]
    return [{"image": [torch.from_numpy(frames[t]) for t in range(T)], "instances": inst, "height": H0, "width": W0}]


def test_mixed_resolution_training_calls_incl_1080p():
    """the reference's own call, `loss_dict = model(batched_inputs)` in training mode, on a mixture of source resolutions -- a
    360-short-edge clip, a 480p crop, a 1080p-shaped clip (1080 x 1920 -> padded 1088 x 1920), then the first again -- through
    one instance: 42 finite losses each, valid assignments, and the repeated clip bit-identical to its first evaluation"""
    from s2d_amd.modeling import build_kd_model
    Q, T, P = 20, 2, 1024
    model = build_kd_model(num_queries=Q, num_frames=T, num_points=P, weights=(2.0, 5.0, 5.0), dropout=0.0).to("cuda:0").train()
    seen = {}
    for (H0, W0) in [(360, 640), (384, 512), (1080, 1920), (360, 640), (480, 854), (1080, 1920)]:
        model.criterion.seed = 0; model.criterion.matcher.seed = 0
        with torch.no_grad():
            losses = model(_batch(7, T, H0, W0, 5))
        torch.cuda.synchronize()
        vals = {k: float(v) for k, v in losses.items()}
        assert len(vals) == 42 and all(np.isfinite(v) for v in vals.values()), (H0, W0)
        st = model.last["student"]
        assert st.hm == (H0 + 31) // 32 * 32 // 4 and st.wm == (W0 + 31) // 32 * 32 // 4
        iq, it, nm = (x.cpu().numpy() for x in model.criterion.last_indices)
        kd = model.last["kd_count"].cpu().numpy()
        for p in range(iq.shape[0]):
            k = nm[p]
            assert k == min(Q, int(kd[p % 1]))
            assert (np.diff(iq[p, :k]) > 0).all() and sorted(it[p, :k].tolist()) == list(range(k))
        if (H0, W0) in seen:
            assert seen[(H0, W0)] == vals, f"{H0}x{W0} evaluated twice gives different losses"
        seen[(H0, W0)] = vals


def test_training_iteration_on_a_1080p_clip_gradients_finite_and_reproducible():
    """criterion.py:292-356 has no frame-size bound: a 1080p-shaped clip (padded 1088 x 1920, 2.09 M pixels -- its bit-packed target
    plane does not fit LDS, so the point loss keeps it in a per-workgroup scratch) goes through the whole training call,
    forward + loss + backward: every student gradient finite, two evaluations on the same seeds bit-identical, and the losses equal to
    the forward-only call's"""
    from s2d_amd import ops
    from s2d_amd.modeling import TargetSet, build_kd_model
    from s2d_amd.modeling.meta_arch import _gt_target_list
    Q, T, P = 20, 2, 1024
    model = build_kd_model(num_queries=Q, num_frames=T, num_points=P, weights=(2.0, 5.0, 5.0), dropout=0.0).to("cuda:0").train()
    batch = _batch(7, T, 1080, 1920, 5)
    params = [p for p in model.student.parameters()]

    def once():
        for p in params:
            p.grad = None
        model.criterion.seed = 0; model.criterion.matcher.seed = 0
        images = model.preprocess(batch)
        Hp, Wp = images.shape[1:3]
        gt = TargetSet.from_list(_gt_target_list(batch, model.num_frames, Hp, Wp, model.device), device=model.device)
        out = model.forward_backward(images, gt)
        torch.cuda.synchronize()
        return {k: float(v) for k, v in out.items()}, [p.grad.clone() for p in params]

    l1, g1 = once()
    l2, g2 = once()
    model.last_tapes = None
    assert len(l1) == 42 and l1 == l2 and all(np.isfinite(v) for v in l1.values())
    assert all(bool(torch.isfinite(g).all()) for g in g1) and sum(int(g.abs().max() > 0) for g in g1) > 300
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))
    model.criterion.seed = 0; model.criterion.matcher.seed = 0
    with torch.no_grad():
        fwd = {k: float(v) for k, v in model(batch).items()}
    for k, v in fwd.items():
        np.testing.assert_allclose(l1[k], v, rtol=1e-5, atol=1e-7, err_msg=k)


def test_ddp_wrapped_training_step_through_the_grad_bridge():
    """DistributedDataParallel(model) + `sum(loss_dict.values()).backward()` (engine/defaults.py:76-85, train_loop.py:709-726) on
    two ranks that share this GPU (gloo: RCCL refuses two ranks on one device): the all-reduced .grad of every student
    parameter equals the mean of the two ranks' single-rank gradients (tests/_ddp_bridge_worker.py prints the worst deviation)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29653", os.path.join(ROOT, "tests", "_ddp_bridge_worker.py")],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    ok = [l for l in r.stdout.splitlines() if l.startswith("DDP_BRIDGE_OK")]
    assert ok, r.stdout[-2000:]
    print(ok[0])

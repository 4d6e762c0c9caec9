"""The BENCHED configuration against the oracle at its own size: one clip of BASELINE configs[3] / bench.py's c4 (T = 8,
720 x 1280 -> 736 x 1280, Q = 100, P = 160 000, N = 10 sparse ground-truth instances) through VideoMaskFormer (student forward +
ground-truth VideoSetCriterion, injected points) vs oracle_np, plus the matcher cost kernel and the point loss at that size on
their own, and the KD pass's matcher on the device's own pseudo targets.  Every kernel that switches path by size (the 3 x 32 + 16
query tiles of the matcher at Q = 100, the point loss's LDS / global plane split, the radix select over 480 000 keys per row, the
cross-attention key splits at K = 117 760, 32-bit index arithmetic at M = 942 080 x K = 2 304) runs here jointly.

Reference lines: matcher.py:225-294, criterion.py:292-356, video_mask2former_transformer_decoder.py:374-467.
Bars: mask / class logits of the 10 heads 1e-3 of the logit scale, Hungarian indices of the GT pass BIT-EXACT (no tie rule),
the 21 losses 1e-3 relative."""
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

T, H0, W0, Q, P, N, NL = 8, 720, 1280, 100, 160000, 10, 10


def _threads(oracle):
    import bench
    n = bench._usable_cores()
    oracle.lib().orc_set_threads(n)
    return n


@pytest.fixture(scope="module")
def case(oracle):
    from threadpoolctl import threadpool_limits
    from s2d_amd import ops
    from s2d_amd.modeling import TargetSet, VideoMaskFormer, VideoSetCriterion, build_kd_model
    from tests.parity import make_case, make_coords, seeded_load
    dev = torch.device("cuda:0")
    seed = 7
    kd = build_kd_model(num_queries=Q, num_frames=T, num_points=P, dec_layers=NL)
    ps = seeded_load(kd.student, seed)
    kd = kd.to(dev)
    wd = {k: v for k, v in kd.criterion.weight_dict.items() if not k.startswith("kd_")}
    crit = VideoSetCriterion(1, matcher=kd.criterion.matcher, weight_dict=wd, eos_coef=0.1, losses=["labels", "masks"], num_points=P,
                             oversample_ratio=3.0, importance_sample_ratio=0.75, loss_strategy="masks-only")
    vm = VideoMaskFormer(backbone=kd.student[0], sem_seg_head=kd.student[1], criterion=crit, num_queries=Q, num_frames=T).to(dev)
    vm.train()
    frames, tg = make_case(seed, 1, T, H0, W0, Q, P, (N,))
    images = ops.normalize_pad(torch.from_numpy(frames).to(dev))
    Hp, Wp = images.shape[1:3]
    assert (Hp, Wp) == (736, 1280)
    m, ids = tg[0]
    pad = np.zeros((m.shape[0], T, Hp, Wp), np.uint8)
    pad[:, :, :H0, :W0] = m
    gts = [torch.from_numpy(pad[(ids != -1).any(-1)])]
    Ngt = gts[0].shape[0]
    assert Ngt == N
    cg = make_coords(seed + 10, NL, 1, Q, Ngt, T, P)
    cgd = {k: torch.from_numpy(v).to(dev) for k, v in cg.items()}
    losses = vm.forward_losses(images, TargetSet.from_list(gts, device=dev), cgd)
    torch.cuda.synchronize()
    st = vm.last["outputs"]
    assert st.dims == (Q, T, 184, 320)
    hip = dict(losses={k: float(v) for k, v in losses.items()}, s_logits=st.class_logits.cpu().numpy(),
               s_masks=[st.pred_masks(i).cpu().numpy() for i in range(NL)],
               idx=tuple(x.cpu().numpy() for x in crit.last_indices), cost=crit.matcher.last_cost.cpu().numpy())
    # the oracle, on every usable core (about 90 s on a GPU box's 16-core share)
    n = _threads(oracle)
    t0 = time.perf_counter()
    with threadpool_limits(limits=n):
        x = oracle.normalize_pad(frames)
        feats = oracle.resnet50(ps, x, "0.")
        mf, ms = oracle.pixel_decoder(ps, feats, "1.pixel_decoder.")
        o_logits, o_masks = oracle.video_decoder(ps, ms, mf, T, "1.predictor.", n_layers=NL - 1)
        t1 = time.perf_counter()
        tgt = [oracle.prepare_targets(m, ids, Hp, Wp)[0]]
        num_masks = max(float(tgt[0].shape[0]), 1.0)
        ref_losses, ref_idx, ref_cost = {}, {}, {}
        for layer in [NL - 1] + list(range(NL - 1)):
            C = oracle.matcher_cost(o_logits[layer][0], o_masks[layer][0], tgt[0], cg["matcher"][layer, 0][None], 0.0, 5.0, 5.0)
            idx = [oracle.lsap(C)]
            ref_cost[layer], ref_idx[layer] = C, idx[0]
            if layer == NL - 1:
                ref_losses["loss_ce"] = oracle.loss_labels(o_logits[layer], idx)
            lm, ld = oracle.loss_masks(o_masks[layer], tgt, idx, num_masks, cg["over"][layer], cg["rand"][layer], P=P)
            suf = "" if layer == NL - 1 else f"_{layer}"
            ref_losses["loss_mask" + suf], ref_losses["loss_dice" + suf] = lm, ld
    t2 = time.perf_counter()
    print(f"c4-size oracle on {n} threads: forward {t1 - t0:.1f} s, 10-layer GT criterion {t2 - t1:.1f} s")
    ref = dict(losses={k: np.float32(v * wd[k]) for k, v in ref_losses.items()}, s_logits=o_logits, s_masks=o_masks, idx=ref_idx, cost=ref_cost,
               tgt=tgt, raw=ref_losses)
    return dict(hip=hip, ref=ref, vm=vm, kd=kd, crit=crit, images=images, gts=gts, cg=cg, cgd=cgd, frames=frames, dev=dev, seed=seed, ps=ps)


def test_c4_student_logits_of_all_heads_vs_oracle(case):
    hip, ref = case["hip"], case["ref"]
    b = ref["s_logits"].astype(np.float64)
    np.testing.assert_allclose(hip["s_logits"], b, rtol=1e-3, atol=1e-3 * np.abs(b).max())
    worst = 0.0
    for layer in range(NL):
        b = ref["s_masks"][layer].astype(np.float64)
        sc = np.abs(b).max()
        d = float(np.abs(hip["s_masks"][layer] - b).max() / sc)
        worst = max(worst, d)
        assert d <= 1e-3, f"mask logits of head {layer}: {d:.3e} of the largest logit"
    print(f"c4-size mask logits vs oracle, worst head: {worst:.3e} of the largest logit")


def test_c4_gt_hungarian_indices_bit_exact_and_cost_matrices(case, oracle):
    from tests.test_gpu_e2e import _cost64
    hip, ref = case["hip"], case["ref"]
    iq, it, nm = hip["idx"]
    worst = 0.0
    for layer in range(NL):
        ri, rj = ref["idx"][layer]
        assert nm[layer] == len(ri) == N
        np.testing.assert_array_equal(iq[layer, :N], ri, err_msg=f"layer {layer}")
        np.testing.assert_array_equal(it[layer, :N], rj, err_msg=f"layer {layer}")
    # the matcher cost kernel at (Q=100, T=8, 184x320, P=160 000) in isolation: the device matrix vs a float64 evaluation of
    # matcher.py:236-287 on the DEVICE's own logits (three layers: the final head, the first, one of the middle)
    for layer in (NL - 1, 0, 4):
        Co, scale = _cost64(oracle, hip["s_logits"][layer][0], hip["s_masks"][layer][0], ref["tgt"][0], case["cg"]["matcher"][layer, 0][None], 0.0, 5.0, 5.0)
        Cd = hip["cost"][layer][:, :N].astype(np.float64)
        err = float(np.abs(Cd - Co).max() / scale)
        worst = max(worst, err)
        assert err <= 1e-5, f"cost matrix of layer {layer}: {err:.3e} of the largest cost term"
        # ... and against the oracle's own fp32 matrix (built from the ORACLE's logits: the 1e-3 logit tolerance passes through)
        e2 = float(np.abs(Cd - ref["cost"][layer]).max() / scale)
        assert e2 <= 1e-3, (layer, e2)
    print(f"c4-size device cost matrices vs float64 on the device's logits: worst {worst:.3e} of the largest cost term")


def test_c4_gt_losses_vs_oracle(case):
    hip, ref = case["hip"], case["ref"]
    assert sorted(hip["losses"]) == sorted(ref["losses"]) and len(hip["losses"]) == 21
    worst = 0.0
    for k, v in ref["losses"].items():
        np.testing.assert_allclose(hip["losses"][k], float(v), rtol=1e-3, atol=1e-6, err_msg=k)
        worst = max(worst, abs(hip["losses"][k] - float(v)) / max(abs(float(v)), 1e-6))
    print(f"c4-size 21 losses vs oracle: worst relative difference {worst:.3e}")


def test_c4_point_loss_op_on_oracle_logits(case, oracle):
    """criterion.py:292-356 at (R = 80 rows, 184 x 320 logit maps, 736 x 1280 targets, 3P = 480 000 oversampled points, exact
    top-120 000 by |logit|) in isolation: the device loss on the ORACLE's mask logits of one head, with the oracle's assignment,
    against the oracle's loss_masks -- no upstream tolerance involved, so the bar is summation order (1e-5)."""
    from s2d_amd.modeling import TargetSet
    ref, crit, dev = case["ref"], case["crit"], case["dev"]
    layer = NL - 1
    # the reference's own output layout ({'pred_logits' [B,Q,2], 'pred_masks' [B,Q,T,h,w]}, one head)
    out = {"pred_logits": torch.from_numpy(ref["s_logits"][layer]).to(dev), "pred_masks": torch.from_numpy(ref["s_masks"][layer]).to(dev)}
    cgd = {k: v[layer:layer + 1].contiguous() for k, v in case["cgd"].items()}
    losses = crit(out, TargetSet.from_list(case["gts"], device=dev), False, cgd)
    torch.cuda.synchronize()
    iq, it, nm = (x.cpu().numpy() for x in crit.last_indices)
    ri, rj = ref["idx"][layer]
    np.testing.assert_array_equal(iq[0, :N], ri)
    np.testing.assert_array_equal(it[0, :N], rj)
    for k in ("loss_mask", "loss_dice", "loss_ce"):
        np.testing.assert_allclose(float(losses[k]), float(ref["raw"][k]), rtol=1e-5, atol=1e-7, err_msg=k)


def test_c4_kd_pass_matcher_on_device_pseudo_targets(case, oracle):
    """the KD pass at the benched size: teacher forward (seeded, class bias calibrated to ~10 pseudo targets as bench.py does),
    prepare_distillation_targets, the KD matcher on them -- cost matrix of every layer vs float64 on the device's own inputs (1e-5),
    assignment = scipy's optimum of that matrix (or a proven tie on it, tests/test_gpu_e2e._same_assignment)"""
    from s2d_amd.modeling import TargetSet
    from tests.parity import make_coords, seeded_load
    from tests.test_gpu_e2e import TIES, _cost64, _same_assignment
    import bench
    kd, dev, images = case["kd"], case["dev"], case["images"]
    seeded_load(kd.teacher, case["seed"] + 1)
    bench.calibrate_teacher(kd, images)
    ck = make_coords(case["seed"] + 11, NL, 1, Q, Q, T, P)
    ckd = {k: torch.from_numpy(v).to(dev) for k, v in ck.items() if k == "matcher"}
    ck = None
    # over / rand of the KD pass: R = up to Q*T rows of 480 000 points would be 3 GB per layer: let the device generator draw them
    kd.keep_kd_targets = True
    kd.criterion.seed = 0
    losses = kd.forward_losses(images, TargetSet.from_list(case["gts"], device=dev), case["cgd"], ckd, kd_nmax=Q)
    torch.cuda.synchronize()
    assert all(np.isfinite(float(v)) for v in losses.values()) and len(losses) == 42
    for k, v in case["hip"]["losses"].items():          # the GT half of the KD model's step == the plain meta-arch's, bit for bit
        assert float(losses[k]) == v, k
    nk = int(kd.last["kd_count"][0])
    assert 3 <= nk <= 30
    tgt = kd.last["kd_targets"][0, :nk].cpu().numpy()
    iq, it, nm = (x.cpu().numpy() for x in kd.criterion.last_indices)
    Cdev = kd.criterion.matcher.last_cost.cpu().numpy()
    st = kd.last["student"]
    n0 = len(TIES)
    worst = 0.0
    for layer in (NL - 1, 0, 5):
        Co, scale = _cost64(oracle, st.class_logits[layer, 0].cpu().numpy(), st.pred_masks(layer)[0].cpu().numpy(), tgt,
                            ckd["matcher"][layer, 0][None].cpu().numpy(), 0.0, 5.0, 5.0)
        Cd = Cdev[layer][:, :nk].astype(np.float64)
        err = float(np.abs(Cd - Co).max() / scale)
        worst = max(worst, err)
        assert err <= 1e-5, (layer, err)
        assert nm[layer] == nk
        oi, oj = oracle.lsap(Co.astype(np.float32))
        _same_assignment(iq[layer, :nk], it[layer, :nk], oi, oj, Co)
    print(f"c4-size KD pass: {nk} pseudo targets, cost matrices worst {worst:.3e} of the largest term, proven ties: {TIES[n0:]}")

"""End-to-end GPU parity: the whole KD forward + distillation loss (R50 -> pixel decoder -> video decoder ->
GT criterion -> KD targets -> KD criterion) on HIP vs the CPU oracle, same seeded weights, same injected points."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


TIES = []
COST_ERR = []


def _same_assignment(iq, it, ri, rj, C):
    """Hungarian indices bit-exact -- except a PROVEN tie: when the two assignments differ, the oracle's own float64 cost of
    ours must equal its optimum to within fp32 cost rounding (2e-6 of the largest entry per matched pair).  Random-init
    teachers emit near-duplicate masks for different queries, whose columns of C then differ by less than the rounding of
    either implementation's sums; such a case is recorded in TIES, not hidden."""
    if np.array_equal(iq, ri) and np.array_equal(it, rj):
        return
    C = C.astype(np.float64)
    ours, best = C[iq, it].sum(), C[ri, rj].sum()
    assert sorted(it.tolist()) == sorted(rj.tolist()) and len(set(iq.tolist())) == len(iq)
    assert abs(ours - best) <= 2e-6 * np.abs(C).max() * len(ri), (ours, best, iq, it, ri, rj)
    TIES.append((float(ours - best), int((it != rj).sum() + (iq != ri).sum())))


def _cost64(oracle, logits, masks, tgt, coords, wc, wm, wd):
    """matcher.py:236-287 on the oracle's fp32 point samples (point_features.py:19-42) with the contractions and the activation
    sums in float64: the value both fp32 implementations (the reference's einsum, the device's split-fp16 MFMA + double
    reduction) approximate"""
    Q, N, P = masks.shape[0], tgt.shape[0], coords.shape[1]
    tm = oracle.point_sample(tgt, np.repeat(coords, N, 0)).reshape(N, -1).astype(np.float64)
    om = oracle.point_sample(masks.astype(np.float32), np.repeat(coords, Q, 0)).reshape(Q, -1).astype(np.float64)
    sp = np.maximum(om, 0) + np.log1p(np.exp(-np.abs(om)))
    cost_mask = (sp.sum(-1)[:, None] - om @ tm.T) / om.shape[1]
    sg = 1.0 / (1.0 + np.exp(-om))
    cost_dice = 1 - (2 * (sg @ tm.T) + 1) / (sg.sum(-1)[:, None] + tm.sum(-1)[None, :] + 1)
    l = logits.astype(np.float32)
    e = np.exp(l - l.max(-1, keepdims=True))
    cost_class = -np.repeat((e / e.sum(-1, keepdims=True))[:, :1].astype(np.float64), N, axis=1)
    # second value: the magnitude of the cost's terms (the sum itself can cancel to ~0: a random-init layer whose class, mask and
    # dice terms nearly offset has max|C| = 0.37 from terms of size 5 -- errors are measured against the terms)
    return wm * cost_mask + wc * cost_class + wd * cost_dice, float((abs(wm) * np.abs(cost_mask) + abs(wc) * np.abs(cost_class) + abs(wd) * np.abs(cost_dice)).max())


def _check(hip, ref, B, NL, oracle=None):
    # mask logits and class logits of all 10 prediction heads: 1e-3 relative (north star)
    for k in ("s_logits", "s_masks"):
        b = ref[k].astype(np.float64)
        np.testing.assert_allclose(hip[k], b, rtol=1e-3, atol=1e-3 * np.abs(b).max())
    assert hip["kd_counts"] == ref["kd_counts"]
    iq, it, nm = (x.cpu().numpy() for x in hip["model"].criterion.last_indices)   # KD pass ran last
    order = [NL - 1] + list(range(NL - 1))
    Cdev = hip["model"].criterion.matcher.last_cost.cpu().numpy()                  # KD pass: [NL*B, Q, Nmax]
    for li, layer in enumerate(order):
        for b in range(B):
            ri, rj = ref["idx_kd"][li][b]
            prob = layer * B + b
            assert nm[prob] == len(ri)
            if oracle is not None and len(ri):
                # The device COST MATRIX against the oracle's matcher evaluated on the device's own inputs (its student logits of
                # this layer, its pseudo-target planes, the same injected points): the matcher kernel in isolation, at this
                # test's full size, the oracle's contractions in float64.  1e-5 of the largest cost term.  (The end-to-end oracle's matrix, ref["cost_kd"], is built from
                # the oracle network's logits and pseudo masks, which differ from the device's by up to the 1e-3 logit
                # tolerance and by the few pseudo-mask pixels whose teacher logit is ~0.)
                Co, scale = _cost64(oracle, hip["s_logits"][layer][b], hip["s_masks"][layer][b], hip["kd_targets"][b],
                                    hip["coords_kd"]["matcher"][layer, b][None], *hip["matcher_weights"])
                Cd = Cdev[prob][:, :Co.shape[1]].astype(np.float64)
                err = float(np.abs(Cd - Co).max() / scale)
                COST_ERR.append(err)
                assert err <= 1e-5, f"KD cost matrix, layer {layer}, clip {b}: {err:.3e} of the largest cost term"
                # ... and the device assignment is an optimum of that matrix: scipy's on the oracle's matrix, or a tie shown on it
                oi, oj = oracle.lsap(Co.astype(np.float32))
                _same_assignment(iq[prob, :len(ri)], it[prob, :len(ri)], oi, oj, Co)
            _same_assignment(iq[prob, :len(ri)], it[prob, :len(ri)], ri, rj, ref["cost_kd"][li][b])
    assert sorted(hip["losses"]) == sorted(ref["losses"])
    for k, v in ref["losses"].items():
        np.testing.assert_allclose(hip["losses"][k], float(v), rtol=1e-3, atol=1e-6, err_msg=k)


def test_kd_forward_loss_small(oracle):
    from tests.parity import run_case
    hip, ref = run_case(oracle, seed=3, B=2, T=2, H0=60, W0=90, Q=16, P=256, ns=(3, 4))
    assert len(hip["losses"]) == 42
    _check(hip, ref, 2, 10, oracle)


def test_kd_forward_loss_config1_plumbing(oracle):
    """BASELINE config 1 shape: 1 x 256 x 256 frame, 10 queries, forward + matcher (+ losses)"""
    from tests.parity import run_case
    hip, ref = run_case(oracle, seed=4, B=1, T=1, H0=256, W0=256, Q=10, P=1024, ns=(3,))
    _check(hip, ref, 1, 10, oracle)


def test_config2_480p_two_frames_q100_both_meta_archs(oracle):
    """BASELINE configs[1]: one 2-frame 480p clip (480x854 -> 480x864), R50 Mask2Former-Video, 100 queries, P = 12 544,
    forward + VideoSetCriterion -- for KDVideoMaskFormer (student + teacher, GT + KD pass) and for the plain VideoMaskFormer
    the config's metric is quoted on (video_maskformer_model.py:224-241: one network, one criterion pass), HIP vs the oracle
    with injected points: logits 1e-3, Hungarian indices bit-exact, losses 1e-3"""
    import torch
    from s2d_amd.modeling import TargetSet, VideoMaskFormer, VideoSetCriterion
    from tests.parity import run_case
    hip, ref = run_case(oracle, seed=5, B=1, T=2, H0=480, W0=854, Q=100, P=12544, ns=(10,), kd_want=10)
    assert 5 <= hip["kd_counts"][0] <= 15
    assert hip["s_masks"].shape[-2:] == (120, 216)
    _check(hip, ref, 1, 10, oracle)
    kd = hip["model"]
    images, gts, cg = hip["inputs"]
    wd = {k: v for k, v in kd.criterion.weight_dict.items() if not k.startswith("kd_")}
    crit = VideoSetCriterion(1, matcher=kd.criterion.matcher, weight_dict=wd, eos_coef=0.1, losses=["labels", "masks"], num_points=12544,
                             oversample_ratio=3.0, importance_sample_ratio=0.75, loss_strategy="masks-only")
    vm = VideoMaskFormer(backbone=kd.student[0], sem_seg_head=kd.student[1], criterion=crit, num_queries=100, num_frames=2).to(images.device)
    vm.train()
    losses = vm.forward_losses(images, TargetSet.from_list(gts, device=images.device), cg)
    torch.cuda.synchronize()
    assert sorted(losses) == sorted(k for k in ref["losses"] if not k.startswith("kd_")) and len(losses) == 21
    for k, v in losses.items():
        np.testing.assert_allclose(float(v), float(ref["losses"][k]), rtol=1e-3, atol=1e-6, err_msg=k)
    iq, it, nm = (x.cpu().numpy() for x in crit.last_indices)
    order = [9] + list(range(9))
    for li, layer in enumerate(order):
        ri, rj = ref["idx_gt"][li][0]
        assert nm[layer] == len(ri)
        np.testing.assert_array_equal(iq[layer, :len(ri)], ri)          # ground-truth targets are distinct objects: no ties
        np.testing.assert_array_equal(it[layer, :len(ri)], rj)
    print("config 2: proven near-tie assignments (cost difference, differing entries):", TIES)
    print("largest relative difference of a device KD cost matrix from the oracle's:", max(COST_ERR))


def test_amp_compute_mode_vs_oracle_with_fp16_operands(oracle):
    """Opt-in AMP compute (the reference trains under `with autocast():`, engine/train_loop.py:709): the R50 trunk, the video
    decoder's linear layers and the mask-logit einsum take single-pass fp16 MFMA arithmetic (operands rounded to fp16, f32
    accumulate); pixel decoder, matcher and losses stay fp32-class.  Against the oracle with the same operands rounded to fp16 at
    the same layers -- and the mode really changes the numbers (it differs from the fp32-class forward by far more).

    What two AMP implementations can agree on: an operand whose f32 value differs in its last bits between the two (another
    summation order upstream) can round to the neighbouring fp16 number, a 5e-4 relative step in that operand, and a mask logit
    that crosses 0 by it flips a bit of the next layer's attention mask, which moves that query's row of the layer by up to a
    few 1e-2 of the logit scale (scripts/diag_amp_err.py prints the per-layer statistics: medians 1e-4 .. 5e-4, isolated layers
    with 1e-2 maxima, the layers after them back at 5e-4).  So the bounds are per layer: median 1e-3 of the scale -- the agreement
    of the arithmetic -- and maximum 5e-2 -- a flipped bit, not a wrong layer; losses 2e-2."""
    import torch
    from s2d_amd import ops
    from tests.parity import run_case
    hip, ref = run_case(oracle, seed=3, B=2, T=2, H0=60, W0=90, Q=16, P=256, ns=(3, 4), amp=True)
    # "A flipped bit, not a wrong layer", checked instead of asserted in prose: the attention-mask bits the DEVICE formed from its own
    # mask logits (the kernel the decoder ran, s2d_attn_mask_bits) against the bits the oracle formed from its logits, per head.
    # flipped[L][b, q]: query q of clip b meets a different mask in decoder layer L.  A clip is "clean up to head L" when no query of
    # it met a different mask in any layer < L (a flipped row reaches the clip's other queries through self-attention, so the unit
    # of the argument is the clip): every logit of a clean clip must agree to 1e-3 -- the agreement of the arithmetic -- and every
    # element outside 1e-3 must sit in a clip that a flip reached.
    st = hip["model"].last["student"]
    Q, T, hm, wm = st.dims
    NLh, B = st.mask_logits.shape[:2]
    Hp, Wp = hm * 4, wm * 4
    sizes = [(Hp // 32, Wp // 32), (Hp // 16, Wp // 16), (Hp // 8, Wp // 8)]
    flipped = []
    for L in range(NLh - 1):
        hl, wl = sizes[L % 3]
        bits, _ = ops.attn_mask_bits(st.mask_logits[L].contiguous(), B, Q, T, hm, wm, hl, wl)
        bits = bits.cpu().numpy().astype(np.uint32)                                               # [B, K, 4]
        dev_bits = np.stack([(bits[:, :, q // 32] >> np.uint32(q % 32)) & np.uint32(1) for q in range(Q)], 1).astype(bool)   # [B, Q, K]
        ref_bits = (oracle.resize_bilinear(ref["s_masks"][L], hl, wl) < 0).reshape(B, Q, T * hl * wl)
        flipped.append((dev_bits != ref_bits).any(-1))
    reached = np.zeros((NLh, B), bool)                           # reached[L, b]: some layer < L of clip b ran with a different mask
    for L in range(1, NLh):
        reached[L] = reached[L - 1] | flipped[L - 1].any(-1)
    n_out = n_out_in_flipped_rows = 0
    for k in ("s_logits", "s_masks"):
        a, b = hip[k].astype(np.float64), ref[k].astype(np.float64)
        sc = np.abs(b).max()
        for layer in range(a.shape[0]):
            d = np.abs(a[layer] - b[layer]) / sc
            assert np.median(d) < 1e-3 and d.max() < 5e-2, (k, layer, float(np.median(d)), float(d.max()))
            for clip in range(B):
                if not reached[layer, clip]:
                    # three fp16 steps (3 x 4.9e-4 of an operand): the pixel decoder in front of the AMP layers is fp32-CLASS, not bitwise
                    # the oracle's (1e-6: split-fp16 products; since round 5 the encoder FFN's residual is hi + lo / 2048 of its input too),
                    # and an operand that differs in its last bits can round to the neighbouring fp16 number in either implementation
                    assert d[clip].max() < 1.5e-3, (k, layer, clip, float(d[clip].max()), "no attention-mask bit differs up to here")
                else:
                    rows = np.zeros(Q, bool)
                    for L in range(layer):
                        rows |= flipped[L][clip]
                    big = d[clip].reshape(Q, -1) > 1.5e-3 if k == "s_masks" else d[clip] > 1.5e-3
                    n_out += int(big.sum()); n_out_in_flipped_rows += int(big[rows].sum())
    print(f"AMP: heads with a flipped attention-mask row per clip: {[int(f.any(-1).sum()) for f in flipped]}; elements outside 1e-3: {n_out}, "
          f"of which in a query row whose own mask flipped: {n_out_in_flipped_rows}")
    assert hip["kd_counts"] == ref["kd_counts"]
    for k, v in ref["losses"].items():
        np.testing.assert_allclose(hip["losses"][k], float(v), rtol=2e-2, atol=1e-6, err_msg=k)
    full, _ = run_case(None, seed=3, B=2, T=2, H0=60, W0=90, Q=16, P=256, ns=(3, 4))
    d = np.abs(full["s_masks"] - hip["s_masks"]).max() / np.abs(full["s_masks"]).max()
    print(f"AMP vs fp32-class mask logits: {d:.3e} of the largest logit")
    assert d > 1e-4


@pytest.mark.parametrize("shape", [(1000, 256, 256), (4097, 100, 256), (300, 2048, 256), (200, 256, 2048), (77, 64, 196)])
def test_amp_gemm_kernel_vs_fp16_rounded_reference(shape):
    """s2d_gemm_nt_amp_f32 == (fp16-rounded A) . (fp16-rounded B)^T accumulated in f32 (+ epilogue), to f32 summation order"""
    import torch
    from s2d_amd import ops
    M, N, K = shape
    g = torch.Generator().manual_seed(M)
    A = torch.randn((M, K), generator=g).cuda(); B = torch.randn((N, K), generator=g).mul_(K ** -0.5).cuda()
    bias = torch.randn((N,), generator=g).cuda(); R = torch.randn((M, N), generator=g).cuda()
    with ops.amp_fp16(True):
        y = ops.gemm_nt(A, B, bias=bias, res=R if N % 4 == 0 else None, relu=True)
    ref = A.half().double() @ B.half().double().t() + bias.double()
    if N % 4 == 0:
        ref = ref + R.double()
    ref = torch.relu(ref)
    np.testing.assert_allclose(y.cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=2e-5)
    y3 = ops.gemm_nt(A, B, bias=bias, res=R if N % 4 == 0 else None, relu=True)          # outside the scope: fp32-class, a different number
    assert (y3 - y).abs().max() > 1e-4


def test_amp_conv_kernel_vs_fp16_rounded_reference():
    import torch
    from s2d_amd import ops
    g = torch.Generator().manual_seed(9)
    for (N, H, W, Cin, Cout, k, stride, pad) in [(2, 20, 28, 64, 64, 3, 1, 1), (1, 33, 47, 4, 64, 7, 2, 3), (2, 16, 24, 128, 256, 1, 2, 0), (1, 9, 11, 256, 96, 3, 2, 1)]:
        x = torch.randn((N, H, W, Cin), generator=g).cuda()
        w = torch.randn((Cout, k, k, Cin), generator=g).mul_((k * k * Cin) ** -0.5).cuda()
        sc = (torch.rand((Cout,), generator=g) + 0.5).cuda(); sh = torch.randn((Cout,), generator=g).cuda()
        with ops.amp_fp16(True):
            y = ops.conv2d_nhwc(x, w, stride, pad, scale=sc, bias=sh, relu=True)
        ref = torch.nn.functional.conv2d(x.half().double().permute(0, 3, 1, 2), w.half().double().permute(0, 3, 1, 2), None, stride, pad)
        ref = torch.relu(ref * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]).permute(0, 2, 3, 1)
        np.testing.assert_allclose(y.cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=2e-5, err_msg=str((N, H, W, Cin, Cout, k, stride, pad)))


def test_clip_pipeline_equals_the_batch_form_and_is_schedule_independent():
    """KDVideoMaskFormer.pipeline_clips (round 5: clip b's criteria beside clip b + 1's forwards) against the batch form on the same injected points:
    every one of the 42 losses to 1e-5 relative (the batch-wide normalisers -- num_masks of both passes, the class loss's weight sum -- are applied
    when the clips are combined; only the order of the sums over clips differs), and its three-stream schedule bitwise equal to itself on one stream"""
    import torch
    from s2d_amd.modeling import TargetSet
    from tests.parity import run_case
    hip, _ = run_case(None, seed=6, B=2, T=2, H0=60, W0=90, Q=16, P=256, ns=(3, 4))
    model = hip["model"]
    images, gts, cg = hip["inputs"]
    ck = {k: torch.from_numpy(v).to(images.device) for k, v in hip["coords_kd"].items()}
    model.keep_kd_targets = False
    ts = lambda: TargetSet.from_list(gts, device=images.device)
    Q = 16
    model.pipeline_clips = False
    model.overlap_teacher = model.overlap_criteria = False
    ref = {k: float(v) for k, v in model.forward_losses(images, ts(), cg, ck, kd_nmax=Q).items()}
    for k, v in ref.items():
        assert v == hip["losses"][k], k                     # the batch form is what run_case ran
    model.pipeline_clips = True
    one = {k: float(v) for k, v in model.forward_losses(images, ts(), cg, ck, kd_nmax=Q).items()}
    model.overlap_teacher = model.overlap_criteria = True
    outs = []
    for _ in range(4):
        outs.append({k: float(v) for k, v in model.forward_losses(images, ts(), cg, ck, kd_nmax=Q).items()})
    torch.cuda.synchronize()
    model.pipeline_clips = False
    model.overlap_teacher = model.overlap_criteria = False
    assert sorted(one) == sorted(ref) and len(one) == 42
    for k, v in ref.items():
        np.testing.assert_allclose(one[k], v, rtol=1e-5, atol=1e-7, err_msg=k)
    assert all(o == one for o in outs)

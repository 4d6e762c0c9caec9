"""GPU parity: device matcher (cost + LSAP), KD targets and point losses against the reference goldens."""
import numpy as np
import pytest
import torch

from s2d_amd.utils import synth
from tests.conftest import golden

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def pixel_major(masks, ldq=None):
    """[..., B, Q, T, h, w] query-major (reference layout) -> [..., B, T*h*w, ldq]"""
    *lead, Q, T, h, w = masks.shape
    ldq = ldq or (Q + 3) // 4 * 4
    out = np.zeros(tuple(lead) + (T * h * w, ldq), np.float32)
    out[..., :Q] = np.moveaxis(masks.reshape(tuple(lead) + (Q, T * h * w)), -2, -1)
    return out


def pad_targets(tg, Nmax, T, H, W):
    B = len(tg)
    out = np.zeros((B, Nmax, T, H, W), np.uint8)
    cnt = np.zeros(B, np.int32)
    for b, t in enumerate(tg):
        out[b, :t.shape[0]] = t
        cnt[b] = t.shape[0]
    return out, cnt


def make_targets(seed, tag, ns, T, H, W):
    return [synth.ellipse_targets(seed, tag + b, n, T, H, W)[0] for b, n in enumerate(ns)]


@pytest.mark.parametrize("name", ["matcher_small", "matcher_q100", "matcher_empty", "matcher_wide"])
def test_matcher_golden(name):
    from s2d_amd import ops
    g = golden(name)
    seed = int(g["seed"])
    B, Q, T, h, w, H, W, P = (int(v) for v in g["dims"])
    ns = [int(v) for v in g["ns"]]
    logits = synth.randn(seed, 1, (B, Q, 2))
    masks = synth.smooth_logits(seed, 2, (B, Q, T), (h, w))
    tg = make_targets(seed, 100, ns, T, H, W)
    Nmax = max(max(ns), 1)
    tgt, cnt = pad_targets(tg, Nmax, T, H, W)
    coords = np.stack([g[f"coords{b}"][0] for b in range(B)])[None]          # [NL=1,B,P,2]
    C = ops.matcher_cost(_dev(pixel_major(masks)[None]), _dev(logits[None]), _dev(tgt), _dev(cnt), (Q, T, h, w), P,
                         tuple(g["cost_weights"]), coords=_dev(coords))
    iq, it, nm = ops.lsap(C, _dev(cnt), B)
    C, iq, it, nm = C.cpu().numpy(), iq.cpu().numpy(), it.cpu().numpy(), nm.cpu().numpy()
    for b in range(B):
        n = ns[b]
        ref = g[f"C{b}"]
        if n:
            np.testing.assert_allclose(C[b][:, :n], ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())
        k = min(Q, n)
        assert nm[b] == k
        np.testing.assert_array_equal(iq[b, :k], g[f"i{b}"])      # Hungarian indices: bit-exact
        np.testing.assert_array_equal(it[b, :k], g[f"j{b}"])


def test_lsap_vs_scipy():
    """device LSAP == scipy on random / tie-heavy / constant matrices (scipy is the reference's solver)"""
    from scipy.optimize import linear_sum_assignment
    from s2d_amd import ops
    rng = np.random.default_rng(1)
    for Q, N in [(100, 10), (100, 100), (16, 5), (12, 20), (100, 1), (100, 128)]:
        mats = []
        for trial in range(6):
            C = rng.standard_normal((Q, N)).astype(np.float32)
            if trial == 3:
                C = np.round(C * 2) / 2
            if trial == 4:
                C[:] = 1.0
            if trial == 5:
                C = np.round(C)
            mats.append(C)
        Cd = _dev(np.stack(mats))
        cnt = _dev(np.full(len(mats), N, np.int32))
        iq, it, nm = (x.cpu().numpy() for x in ops.lsap(Cd, cnt, len(mats)))
        for k, C in enumerate(mats):
            ra, rb = linear_sum_assignment(C)
            assert nm[k] == len(ra)
            np.testing.assert_array_equal(iq[k, :len(ra)], ra)
            np.testing.assert_array_equal(it[k, :len(ra)], rb)


def _losses_from_golden(g, tg_list, s_logits, s_masks, rand_key, idx_key, B, Q, T, h, w, H, W, P, NL, Nmax=None):
    """run the device criterion for one pass with the golden's recorded torch.rand draws; returns dict, indices"""
    from s2d_amd import ops
    Nmax = Nmax or max(max(t.shape[0] for t in tg_list), 1)
    tgt, cnt = pad_targets(tg_list, Nmax, T, H, W)
    # the recorded draws, in the reference's call order: per layer call: B matcher draws, then (if any kept row)
    # coords_over + coords_rand.  Reference layer order: last layer first, then aux 0..NL-2.
    draws = [g[f"rand_{rand_key}_{i}"] for i in range(int(g[f"nrand_{rand_key}"]))]
    order = [NL - 1] + list(range(NL - 1))
    maxm = min(Q, Nmax)
    rows_l = B * maxm * T
    n_over, n_unc = int(P * 3.0), int(0.75 * P)
    n_rand = P - n_unc
    mcoords = np.zeros((NL, B, P, 2), np.float32)
    cover = np.zeros((NL, rows_l, n_over, 2), np.float32)
    crand = np.zeros((NL, rows_l, n_rand, 2), np.float32)
    it = iter(draws)
    for layer in order:
        for b in range(B):
            mcoords[layer, b] = next(it)[0]
        # does this layer call have kept rows?  known from the reference indices + targets
        kept = 0
        for b in range(B):
            tj = g[f"idx_{idx_key}_{order.index(layer)}_{b}_j"]
            for j in tj:
                kept += int(sum(tg_list[b][j, t].any() for t in range(T)))
        if kept:
            co, cr = next(it), next(it)
            assert co.shape[0] == kept
            cover[layer, :kept] = co
            crand[layer, :kept] = cr
    ml = _dev(pixel_major(s_masks))
    tgt_d, cnt_d = _dev(tgt), _dev(cnt)
    C = ops.matcher_cost(ml, _dev(s_logits), tgt_d, cnt_d, (Q, T, h, w), P, (2.0, 5.0, 5.0), coords=_dev(mcoords))
    iq, itt, nm = ops.lsap(C, cnt_d, B)
    ne = ops.target_nonempty(tgt_d, cnt_d)
    losses = ops.point_loss(ml, tgt_d, cnt_d, ne, iq, itt, nm, (Q, T, h, w), P, coords_over=_dev(cover), coords_rand=_dev(crand))
    ce = ops.class_loss(_dev(s_logits[NL - 1]), iq[(NL - 1) * B:], nm[(NL - 1) * B:])
    return losses.cpu().numpy(), float(ce), iq.cpu().numpy(), itt.cpu().numpy(), nm.cpu().numpy(), order


def test_full_criterion_and_kd_targets_golden():
    """the reference's whole loss path (GT pass + KD pass, 10 layers, 2 clips): indices bit-exact, 42 losses <= 1e-3"""
    from s2d_amd import ops
    g = golden("criterion_kd")
    seed = int(g["seed"])
    B, Q, T, h, w, H, W, P, NL = (int(v) for v in g["dims"])
    ns = [int(v) for v in g["ns"]]
    s_logits = synth.randn(seed, 1, (NL, B, Q, 2))
    s_masks = synth.smooth_logits(seed, 2, (NL, B, Q, T), (h, w))
    t_logits = synth.randn(seed, 3, (B, Q, 2), 2.0)
    t_masks = synth.smooth_logits(seed, 4, (B, Q, T), (h, w))
    tg = make_targets(seed, 100, ns, T, H, W)

    # --- KD targets on the device (kd_video_maskformer_model.py:436-468)
    tgt, cnt, kept, ne = ops.kd_targets(_dev(t_logits), _dev(pixel_major(t_masks)), (Q, T, h, w), H, W, Nmax=Q)
    tgt, cnt, kept, ne = tgt.cpu().numpy(), cnt.cpu().numpy(), kept.cpu().numpy(), ne.cpu().numpy()
    kd_ref_order = []
    for b in range(B):
        ref_order = g[f"kd_order{b}"]
        assert cnt[b] == int(g[f"kd_n{b}"])
        assert sorted(kept[b, :cnt[b]].tolist()) == sorted(ref_order.tolist())
        ref_masks = np.unpackbits(g[f"kd_masks{b}"], axis=-1)[..., :W]
        pos = {int(q): k for k, q in enumerate(kept[b, :cnt[b]])}
        pr = np.array([pos[int(q)] for q in ref_order])
        np.testing.assert_array_equal(tgt[b][pr], ref_masks)                       # binary KD targets: bit-exact
        np.testing.assert_array_equal(ne[b][pr], ref_masks.reshape(len(pr), T, -1).any(-1))
        kd_ref_order.append(tgt[b][pr])   # feed the criterion in the reference's target order -> indices comparable

    out = {}
    for key, tgl, pref in (("gt", tg, "loss_"), ("kd", kd_ref_order, "kd_loss_")):
        losses, ce, iq, itt, nm, order = _losses_from_golden(g, tgl, s_logits, s_masks, key, key, B, Q, T, h, w, H, W, P, NL)
        for li, layer in enumerate(order):
            for b in range(B):
                ri, rj = g[f"idx_{key}_{li}_{b}_i"], g[f"idx_{key}_{li}_{b}_j"]
                prob = layer * B + b
                assert nm[prob] == len(ri)
                np.testing.assert_array_equal(iq[prob, :len(ri)], ri)
                np.testing.assert_array_equal(itt[prob, :len(ri)], rj)
        wts = {"ce": 2.0 if key == "gt" else 0.0, "mask": 5.0, "dice": 5.0}
        out[pref + "ce"] = ce * wts["ce"]
        for layer in range(NL):
            suf = "" if layer == NL - 1 else f"_{layer}"
            out[pref + "mask" + suf] = losses[layer, 0] * wts["mask"]
            out[pref + "dice" + suf] = losses[layer, 1] * wts["dice"]
    ref_keys = sorted(k[2:] for k in g.files if k.startswith("L_"))
    assert sorted(out) == ref_keys
    for k in ref_keys:
        np.testing.assert_allclose(out[k], float(g["L_" + k]), rtol=1e-3, atol=1e-6)   # north-star tolerance: 1e-3 relative


def test_loss_golden_and_droploss_empty():
    from s2d_amd import ops
    g = golden("loss")
    seed = int(g["seed"])
    B, Q, T, h, w, H, W, P = (int(v) for v in g["dims"])
    ns = [int(v) for v in g["ns"]]
    logits = synth.randn(seed, 1, (B, Q, 2))
    masks = synth.smooth_logits(seed, 2, (B, Q, T), (h, w))
    tg = make_targets(seed, 100, ns, T, H, W)
    Nmax = max(ns)
    tgt, cnt = pad_targets(tg, Nmax, T, H, W)
    maxm = min(Q, Nmax)
    iq = np.zeros((B, maxm), np.int32); it = np.zeros((B, maxm), np.int32); nm = np.zeros(B, np.int32)
    for b in range(B):
        k = len(g[f"i{b}"])
        iq[b, :k], it[b, :k], nm[b] = g[f"i{b}"], g[f"j{b}"], k
    kept = len(g["keep"])
    rows_l = B * maxm * T
    cover = np.zeros((1, rows_l, 3 * P, 2), np.float32); cover[0, :kept] = g["coords_over"]
    crand = np.zeros((1, rows_l, P - int(0.75 * P), 2), np.float32); crand[0, :kept] = g["coords_rand"]
    tgt_d, cnt_d = _dev(tgt), _dev(cnt)
    ne = ops.target_nonempty(tgt_d, cnt_d)
    ml = _dev(pixel_major(masks)[None])
    L = ops.point_loss(ml, tgt_d, cnt_d, ne, _dev(iq), _dev(it), _dev(nm), (Q, T, h, w), P, coords_over=_dev(cover),
                       coords_rand=_dev(crand)).cpu().numpy()
    np.testing.assert_allclose(L[0, 0], float(g["loss_mask"]), rtol=1e-3)
    np.testing.assert_allclose(L[0, 1], float(g["loss_dice"]), rtol=1e-3)
    ce = float(ops.class_loss(_dev(logits), _dev(iq), _dev(nm)))
    np.testing.assert_allclose(ce, float(g["loss_ce"]), rtol=1e-4)
    # all-empty targets -> both losses exactly 0 (criterion.py:315-318)
    z = _dev(np.zeros_like(tgt))
    ne0 = ops.target_nonempty(z, cnt_d)
    L0 = ops.point_loss(ml, z, cnt_d, ne0, _dev(iq), _dev(it), _dev(nm), (Q, T, h, w), P).cpu().numpy()
    assert (L0 == 0).all()
    # device RNG mode: finite, and close to the injected-coordinate value (same estimator, other points)
    Lr = ops.point_loss(ml, tgt_d, cnt_d, ne, _dev(iq), _dev(it), _dev(nm), (Q, T, h, w), P, seed=7).cpu().numpy()
    assert np.isfinite(Lr).all() and abs(Lr[0, 1] - L[0, 1]) < 0.2 * abs(L[0, 1]) + 0.05


@pytest.mark.parametrize("B,Q,T,h,w,P,ns,dense", [
    (1, 37, 1, 13, 21, 77, [3], False),            # odd sizes, sparse points: every batch takes the direct-gather path
    (2, 128, 2, 24, 40, 4096, [5, 33], True),      # Q = 128, one clip on the N > 32 kernel, dense points: LDS-staged rows
    (2, 100, 3, 16, 28, 1500, [1, 17], True),      # tail batch (P % 32 != 0)
])
def test_matcher_cost_vs_oracle_odd_shapes(oracle, B, Q, T, h, w, P, ns, dense):
    """device cost matrix vs the CPU restatement on shapes the goldens do not cover, including points on the image border"""
    from s2d_amd import ops
    H, W = 4 * h, 4 * w
    seed = 40 + Q
    logits = synth.randn(seed, 1, (B, Q, 2))
    masks = synth.smooth_logits(seed, 2, (B, Q, T), (h, w))
    tg = make_targets(seed, 7, ns, T, H, W)
    Nmax = max(ns)
    tgt, cnt = pad_targets(tg, Nmax, T, H, W)
    rng = np.random.default_rng(seed)
    coords = rng.random((1, B, P, 2), dtype=np.float32)
    coords[0, :, :8] = np.array([[0, 0], [1, 1], [0, 1], [1, 0], [0.5, 0], [0, 0.5], [1, 0.5], [0.5, 1]], np.float32)
    wts = (2.0, 5.0, 5.0)
    C = ops.matcher_cost(_dev(pixel_major(masks)[None]), _dev(logits[None]), _dev(tgt), _dev(cnt), (Q, T, h, w), P, wts,
                         coords=_dev(coords)).cpu().numpy()
    for b in range(B):
        ref = oracle.matcher_cost(logits[b], masks[b], tg[b], coords[0, b][None], *wts)
        np.testing.assert_allclose(C[b][:, :ns[b]], ref, rtol=3e-5, atol=3e-5 * np.abs(ref).max())


def test_matcher_16_row_tile_kernel_opt_in(monkeypatch):
    """S2D_MATCHER_Q16=1 (16-query MFMA tiles, ceil(Q / 16) waves): same cost matrices as the default 32-row kernel to f32 summation
    order, same assignments -- Q = 100 (7 waves), Q = 16 (4 waves, the target side's minimum), one and two target tiles"""
    import torch
    from s2d_amd import ops
    dev = torch.device("cuda")
    for (NL, B, Q, T, hm, wm, N, P) in [(3, 2, 100, 2, 40, 56, 10, 4096), (2, 1, 16, 1, 24, 32, 20, 1000), (2, 2, 37, 2, 24, 40, 3, 777)]:
        g = torch.Generator(device=dev).manual_seed(Q)
        ml = (torch.randn((NL, B, T * hm * wm, Q), generator=g, device=dev) * 4).contiguous()
        cls = torch.randn((NL, B, Q, 2), generator=g, device=dev)
        tgt = (torch.rand((B, N, T, hm * 4, wm * 4), generator=g, device=dev) < 0.3).to(torch.uint8)
        cnt = torch.full((B,), N, dtype=torch.int32, device=dev)
        out = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("S2D_MATCHER_Q16", mode)
            C = ops.matcher_cost(ml, cls, tgt, cnt, (Q, T, hm, wm), P, (2.0, 5.0, 5.0), seed=5)
            out[mode] = (C, ops.lsap(C, cnt, B))
        a, b = out["0"][0], out["1"][0]
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())
        for x, y in zip(out["0"][1], out["1"][1]):
            assert torch.equal(x, y)


def test_matcher_three_32_row_tiles_plus_one_16_row_tile(monkeypatch):
    """96 < Q <= 112 (the shipped Q = 100) runs three 32-query tiles + one 16-query tile whose wave also builds the tap tables
    (matcher_cost_f16_mix_kernel); S2D_MATCHER_MIX=0 is the four-tile kernel: same cost matrices to f32 summation order, same
    assignments -- one and two target tiles, a logit row stride > Q, a point count that leaves a partial batch, injected sparse points
    (direct-gather batches), Q at both ends of the range"""
    import torch
    from s2d_amd import ops
    dev = torch.device("cuda")
    for (NL, B, Q, ldq, T, hm, wm, N, P) in [(3, 2, 100, 100, 2, 40, 56, 10, 4096), (2, 1, 97, 104, 1, 24, 32, 20, 1000),
                                            (2, 2, 112, 128, 2, 24, 40, 3, 777), (1, 1, 100, 100, 1, 46, 80, 32, 50)]:
        g = torch.Generator(device=dev).manual_seed(Q + P)
        ml = torch.randn((NL, B, T * hm * wm, ldq), generator=g, device=dev) * 4       # columns >= Q: the GEMM's padding, never used
        cls = torch.randn((NL, B, Q, 2), generator=g, device=dev)
        tgt = (torch.rand((B, N, T, hm * 4, wm * 4), generator=g, device=dev) < 0.3).to(torch.uint8)
        cnt = torch.full((B,), N, dtype=torch.int32, device=dev)
        out = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("S2D_MATCHER_MIX", mode)
            C = ops.matcher_cost(ml, cls, tgt, cnt, (Q, T, hm, wm), P, (2.0, 5.0, 5.0), seed=5)
            out[mode] = (C, ops.lsap(C, cnt, B))
        a, b = out["0"][0], out["1"][0]
        assert torch.isfinite(b).all()
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())
        for x, y in zip(out["0"][1], out["1"][1]):
            assert torch.equal(x, y)


@pytest.mark.parametrize("P,H,W", [(250, 64, 96), (256, 64, 96), (1000, 32, 128)])
def test_point_loss_vs_oracle_both_paths(oracle, P, H, W):
    """point loss vs the CPU restatement where the sampled logits cannot be kept (3P or P/4 not a multiple of 4: the
    recompute kernels) and where they can (the streamed kernels), same call"""
    from s2d_amd import ops
    B, Q, T, h, w = 2, 16, 2, H // 4, W // 4
    ns = [3, 2]
    seed = 90 + P
    masks = synth.smooth_logits(seed, 2, (B, Q, T), (h, w))
    tg = make_targets(seed, 100, ns, T, H, W)
    Nmax = max(ns)
    tgt, cnt = pad_targets(tg, Nmax, T, H, W)
    maxm = min(Q, Nmax)
    rng = np.random.default_rng(seed)
    iq = np.zeros((B, maxm), np.int32); it = np.zeros((B, maxm), np.int32); nm = np.array(ns, np.int32)
    indices = []
    for b in range(B):
        qi = np.sort(rng.choice(Q, ns[b], replace=False)); tj = rng.permutation(ns[b])
        iq[b, :ns[b]], it[b, :ns[b]] = qi, tj
        indices.append((qi, tj))
    # kept rows in the reference's order: clip-major, pair-major, frame-minor, empty (pair, frame) planes dropped
    rows = [(b, s, t) for b in range(B) for s in range(ns[b]) for t in range(T)]
    kept = [r for r in rows if tg[r[0]][indices[r[0]][1][r[1]], r[2]].any()]
    R = len(kept)
    n_rand = P - int(0.75 * P)
    cov = rng.random((R, 3 * P, 2), dtype=np.float32); crd = rng.random((R, n_rand, 2), dtype=np.float32)
    rows_l = B * maxm * T
    cover = np.zeros((1, rows_l, 3 * P, 2), np.float32); cover[0, :R] = cov
    crand = np.zeros((1, rows_l, n_rand, 2), np.float32); crand[0, :R] = crd
    tgt_d, cnt_d = _dev(tgt), _dev(cnt)
    ne = ops.target_nonempty(tgt_d, cnt_d)
    L = ops.point_loss(_dev(pixel_major(masks)[None]), tgt_d, cnt_d, ne, _dev(iq), _dev(it), _dev(nm), (Q, T, h, w), P,
                       coords_over=_dev(cover), coords_rand=_dev(crand)).cpu().numpy()
    # the reference flattens (pair, frame) into rows: [R,1,h,w] maps and [R,1,H,W] planes
    m_rows = [masks[b][None, indices[b][0]].transpose(1, 2, 0, 3, 4).reshape(-1, 1, 1, h, w)[:, 0] for b in range(B)]
    t_rows = [tg[b][indices[b][1]].reshape(-1, 1, H, W) for b in range(B)]
    src = np.concatenate(m_rows, 0); tt = np.concatenate(t_rows, 0)
    keep = np.array([i for i in range(tt.shape[0]) if tt[i].any()])
    assert len(keep) == R
    num_masks = max(float(sum(ns)), 1.0)
    # criterion.py:292-356 evaluated with the oracle's primitives on the flattened rows
    pl = oracle.point_sample(src[keep], cov)[:, 0]
    idx = np.argsort(np.abs(pl), axis=1, kind="stable")[:, :int(0.75 * P)]
    coords = np.concatenate([np.take_along_axis(cov, idx[..., None], 1), crd], 1)
    labels = oracle.point_sample(tt[keep], coords)[:, 0]
    lg = oracle.point_sample(src[keep], coords)[:, 0]
    bce = np.maximum(lg, 0) - lg * labels + np.log1p(np.exp(-np.abs(lg)))
    ref_mask = bce.mean(1).sum() / num_masks
    sg = oracle.sigmoid(lg)
    ref_dice = (1 - (2 * (sg * labels).sum(-1) + 1) / (sg.sum(-1) + labels.sum(-1) + 1)).sum() / num_masks
    np.testing.assert_allclose(L[0, 0], ref_mask, rtol=1e-3)
    np.testing.assert_allclose(L[0, 1], ref_dice, rtol=1e-3)


def test_rng_mode_points_are_iid_uniform_in_law():
    """Timing mode generates the 3P oversampled points of a row per map part (an exact multinomial split over the parts' v bands,
    then uniform inside each band) instead of testing every point against every part.  That is the law of i.i.d. uniform points
    (point_features.py:89-93 draws torch.rand): checked on the points the device functions emit (s2d_point_loss_rng_points) --
    every point inside its own part's band, band counts within binomial fluctuations and varying from row to row, u and v
    uniform (Kolmogorov-Smirnov), u and v uncorrelated, rows independent -- at the shipped geometry (184 x 320 map -> 2 parts) and a
    1080p-shaped one (272 x 480 -> 5 parts)."""
    from scipy import stats
    from s2d_amd._lib import lib
    for hm, wm, n_over, nrows in ((184, 320, 480000, 6), (272, 480, 60000, 8), (16, 24, 768, 4)):
        uv = torch.empty((nrows, n_over, 2), device="cuda", dtype=torch.float32)
        bounds = torch.zeros((nrows, 9), device="cuda", dtype=torch.int32)
        scratch = torch.empty((nrows + 1,), device="cuda", dtype=torch.int32)
        lib().call("s2d_point_loss_rng_points", 0xC0FFEE + hm, hm, wm, 0, nrows, n_over, uv, bounds, scratch, torch.cuda.current_stream().cuda_stream)
        uvh, bh = uv.cpu().numpy().astype(np.float64), bounds.cpu().numpy()
        rpp = min(124 * 1024 // ((wm + 8) * 4) - 3, hm)
        nparts = -(-hm // rpp)
        assert (uvh >= 0).all() and (uvh < 1).all()
        counts = []
        for r in range(nrows):
            b = bh[r, :nparts + 1]
            assert b[0] == 0 and b[-1] == n_over and (np.diff(b) >= 0).all()
            y0 = np.floor(uvh[r, :, 1] * hm - 0.5).astype(int)
            for j in range(nparts):
                ya = -1 if j == 0 else j * rpp
                yb = min((j + 1) * rpp, hm)
                own = y0[b[j]:b[j + 1]]
                # a point generated for part j lies in part j's band (floor(y) one past the band's top edge only by float rounding)
                assert (own >= ya).all() and (own <= yb).all() and (own == yb).mean() < 1e-4
                pj = ((hm - 0.5 if j == nparts - 1 else (j + 1) * rpp) - (-0.5 if j == 0 else j * rpp)) / hm
                n = b[j + 1] - b[j]
                assert abs(n - n_over * pj) < 5.5 * np.sqrt(n_over * pj * (1 - pj)) + 1, (hm, r, j, n, n_over * pj)
            counts.append(tuple(np.diff(b)))
            if n_over >= 10000:
                for c in (0, 1):
                    assert stats.kstest(uvh[r, :, c], "uniform").pvalue > 1e-4, (hm, r, c)
                assert abs(np.corrcoef(uvh[r, :, 0], uvh[r, :, 1])[0, 1]) < 5 / np.sqrt(n_over)
        if n_over >= 10000:
            assert len(set(counts)) > 1                       # the split is random per row, not the expectation rounded
            assert abs(np.corrcoef(np.sort(uvh[0, :, 0]), np.sort(uvh[1, :, 0]))[0, 1]) > 0.999   # same law ...
            assert not np.array_equal(uvh[0], uvh[1])         # ... different draws


def test_rng_mode_point_loss_does_not_depend_on_target_padding():
    """the sampled points of a matched (pair, frame) are keyed by (layer, clip, slot, frame) with the slot stride fixed at Q (csrc/loss.hip
    key_row), not at min(Q, Nmax): the same targets in planes padded to 3, 8 and Q slots give the same losses bit for bit in generator mode"""
    from s2d_amd import ops
    B, Q, T, H, W, P = 2, 16, 2, 64, 96, 512
    h, w = H // 4, W // 4
    ns = [3, 2]
    seed = 4711
    masks = synth.smooth_logits(seed, 2, (B, Q, T), (h, w))
    tg = make_targets(seed, 100, ns, T, H, W)
    ml = _dev(np.stack([pixel_major(masks), pixel_major((masks * 0.5).astype(np.float32))]))      # two layers
    rng = np.random.default_rng(seed)
    pairs = [(np.sort(rng.choice(Q, ns[b], replace=False)), rng.permutation(ns[b])) for b in range(B)]
    out = []
    for Nmax in (3, 8, Q):
        tgt, cnt = pad_targets(tg, Nmax, T, H, W)
        maxm = min(Q, Nmax)
        iq = np.zeros((2 * B, maxm), np.int32); it = np.zeros((2 * B, maxm), np.int32); nm = np.array(ns * 2, np.int32)
        for l in range(2):
            for b in range(B):
                iq[l * B + b, :ns[b]], it[l * B + b, :ns[b]] = pairs[b]
        tgt_d, cnt_d = _dev(tgt), _dev(cnt)
        ne = ops.target_nonempty(tgt_d, cnt_d)
        L = ops.point_loss(ml, tgt_d, cnt_d, ne, _dev(iq), _dev(it), _dev(nm), (Q, T, h, w), P, seed=99)
        out.append(L.cpu().numpy())
    assert np.isfinite(out[0]).all() and (out[0] > 0).all()
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])

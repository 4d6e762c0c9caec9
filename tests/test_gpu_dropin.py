"""The drop-in boundary on the GPU: fused counter-based dropout (the three nn.Dropout sites of the encoder
layers, msdeformattn.py:101-125), the autograd-visible training forward the reference trainer drives
(engine/train_loop.py:709-726), the `MultiScaleDeformableAttention` module / `MSDeformAttnFunction` under the reference's
call signatures (ops/functions/ms_deform_attn_func.py:32-49), the model built from the literal keys of the shipped
ytvis2021_kd_video_mask2former_R50_cls_agnostic.yaml (tests/golden/kd_config.json), and prepare_targets (A8)."""
import json
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from tests.test_oracle import golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle_np
    return oracle_np


# ------------------------------------------------------------------------------------------------ dropout
@pytest.mark.parametrize("M,N,p,seed,site", [(1000, 256, 0.3, 0x1234567890ABCDEF, 0), (777, 1024, 0.3, 42, 1), (333, 8, 0.1, 2 ** 63 + 5, 2),
                                              (64, 256, 0.0, 7, 0)])
def test_dropout_mask_vs_oracle(oracle, M, N, p, seed, site):
    """the device mask is bit-for-bit the oracle's restatement (Philox4x32-10 pinned by Random123's known answers)"""
    from s2d_amd import ops
    m = ops.dropout(torch.ones((M, N), device=DEV), p, seed, site).cpu().numpy()
    np.testing.assert_array_equal(m, oracle.dropout_multipliers(M, N, p, seed, site))


def test_dropout_statistics():
    """p = 0.3 (the shipped MODEL.MASK_FORMER.DROPOUT): keep rate, scale, and independence between rows, columns, sites and
    calls, at the size of one encoder-layer activation"""
    from s2d_amd import ops
    M, N, p = 19320 * 2, 256, 0.3
    ones = torch.ones((M, N), device=DEV)
    a = ops.dropout(ones, p, 1111, 0)
    b = ops.dropout(ones, p, 1111, 1)                  # another site of the same call
    c = ops.dropout(ones, p, 2222, 0)                  # another call
    n = M * N
    q = 1.0 - round(p * 256) / 256.0                    # the realised keep probability: p quantised to 1 / 256 (csrc/dropout.h)
    sd = (q * (1 - q) / n) ** 0.5
    for m in (a, b, c):
        keep = (m > 0).double().mean().item()
        assert abs(keep - q) < 5 * sd, (keep, q)
        vals = torch.unique(m)
        assert vals.numel() == 2 and vals[0] == 0 and abs(vals[1].item() - 1 / q) < 1e-6
        assert abs(m.double().mean().item() - 1.0) < 5 * sd / q          # E[x * mask / P(keep)] = x: unbiased for the realised keep probability
    ka, kb, kc = (a > 0).double(), (b > 0).double(), (c > 0).double()
    for x, y in ((ka, kb), (ka, kc), (ka[:, :-1], ka[:, 1:]), (ka[:-1], ka[1:]), (ka[:, ::16][:, :-1], ka[:, ::16][:, 1:])):
        cov = ((x - q) * (y - q)).mean().item()
        assert abs(cov) < 5 * q * (1 - q) / x.numel() ** 0.5, cov
    rows, cols = ka.mean(1), ka.mean(0)                # no dead / always-on rows or columns
    assert (rows - q).abs().max() < 6 * (q * (1 - q) / N) ** 0.5 and (cols - q).abs().max() < 6 * (q * (1 - q) / M) ** 0.5


@pytest.mark.parametrize("M,N,K,relu,res,static", [(1000, 256, 256, False, True, True), (5000, 1024, 256, True, False, True),
                                                    (777, 256, 1024, False, True, True), (130, 64, 96, True, True, False)])
def test_gemm_dropout_epilogue_bitwise(M, N, K, relu, res, static):
    """act(dropout(A W^T + b) + res) in the GEMM epilogue == the same GEMM, then the mask applied elementwise (the three
    encoder-layer shapes: output_proj, linear1 + ReLU, linear2)"""
    from s2d_amd import ops
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn((M, K), generator=g).to(DEV)
    W = torch.randn((N, K), generator=g).mul_(K ** -0.5).to(DEV)
    if static:
        W = torch.nn.Parameter(W)
    b = torch.randn((N,), generator=g).to(DEV)
    R = torch.randn((M, N), generator=g).to(DEV) if res else None
    p, seed, site = 0.3, 987654321, 2
    y = ops.gemm_nt(A, W, bias=b, res=R, relu=relu, dropout=(p, seed, site))
    ref = ops.gemm_nt(A, W, bias=b) * ops.dropout(torch.ones((M, N), device=DEV), p, seed, site)
    if res:
        ref = ref + R
    if relu:
        ref = torch.relu(ref)
    assert torch.equal(y, ref)
    assert torch.equal(ops.gemm_nt(A, W, bias=b, res=R, relu=relu, dropout=(0.0, seed, site)), ops.gemm_nt(A, W, bias=b, res=R, relu=relu))


def test_encoder_layer_eval_mode_is_identity_dropout():
    """nn.Dropout is the identity in eval mode: a p = 0.3 layer in eval() equals the p = 0 layer bit for bit"""
    from s2d_amd.modeling.pixel_decoder import MSDeformAttnTransformerEncoderLayer
    from s2d_amd import ops
    torch.manual_seed(1)
    a = MSDeformAttnTransformerEncoderLayer(dropout=0.3).to(DEV).eval()
    b = MSDeformAttnTransformerEncoderLayer(dropout=0.0).to(DEV).eval()
    b.load_state_dict(a.state_dict())
    shapes = [(4, 6), (8, 12), (16, 24)]
    S = sum(h * w for h, w in shapes)
    src = torch.randn((2, S, 256), device=DEV)
    pos = torch.randn((S, 256), device=DEV)
    shp = torch.tensor(shapes, dtype=torch.int64)
    assert torch.equal(a(src, pos, shp), b(src, pos, shp))
    a.train()
    y1, y2 = a(src, pos, shp), a(src, pos, shp)                       # training mode: fresh masks per call
    assert not torch.equal(y1, y2) and not torch.equal(y1, b(src, pos, shp))


# ------------------------------------------------------------------------------------------------ MSDA module / function
def test_msda_compat_module_and_function(oracle):
    """the reference's FFI names and call signatures (ops/src/vision.cpp:18-21; ms_deform_attn_func.py:32-49), values vs the
    goldens the reference's own core produced, and the extension's error conventions (ms_deform_attn_cuda.cu:33-57)"""
    import s2d_amd.compat as compat
    compat.install()
    import MultiScaleDeformableAttention as MSDA                       # the reference's import name
    from s2d_amd.compat.ms_deform_attn_func import MSDeformAttnFunction
    g = golden("msda_core")
    dev = torch.device(DEV)
    shapes = torch.as_tensor(g["shapes"], dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    v, loc, w = (torch.from_numpy(g[k]).to(dev) for k in ("value", "loc", "w"))
    out = MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, w, 128)
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-5, atol=1e-5)
    gv, gl, gw = MSDA.ms_deform_attn_backward(v, shapes, lsi, loc, w, torch.from_numpy(g["grad_out"]).to(dev), 128)
    vr, lr, wr = v.clone().requires_grad_(True), loc.clone().requires_grad_(True), w.clone().requires_grad_(True)
    o2 = MSDeformAttnFunction.apply(vr, shapes, lsi, lr, wr, 128)
    assert torch.equal(o2, out)
    (o2 * torch.from_numpy(g["grad_out"]).to(dev)).sum().backward()
    for got, got2, name in ((gv, vr.grad, "grad_value"), (gl, lr.grad, "grad_loc"), (gw, wr.grad, "grad_w")):
        ref = g[name]
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
        np.testing.assert_allclose(got2.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())   # (grad_value: float atomics, not bitwise)
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_forward(v.permute(0, 1, 3, 2), shapes, lsi, loc, w, 128)      # not contiguous
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_forward(v.cpu(), shapes, lsi, loc, w, 128)                    # not on the device
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_forward(torch.cat([v, v[:1]]), shapes, lsi, torch.cat([loc, loc[:1]]), torch.cat([w, w[:1]]), 2)   # 3 % 2


def test_msda_compat_float64_instantiation_forward_and_gradcheck():
    """the extension's float64 instantiation (ms_deform_attn_cuda.cu:69, :137), as the reference's ops/test.py exercises it: forward
    against a float64 grid_sample evaluation of the same sum (ms_deform_attn_func.py:52-72) and torch.autograd.gradcheck of
    MSDeformAttnFunction in double, channels not a multiple of 32, a level with out-of-map samples"""
    import s2d_amd.compat as compat
    compat.install()
    import MultiScaleDeformableAttention as MSDA
    from s2d_amd.compat.ms_deform_attn_func import MSDeformAttnFunction
    dev = torch.device(DEV)
    g = torch.Generator(device=dev).manual_seed(11)
    N, M, D, Lq, L, P = 2, 3, 10, 5, 2, 3
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    value = torch.rand((N, S, M, D), device=dev, dtype=torch.float64, generator=g) * 0.01
    loc = torch.rand((N, Lq, M, L, P, 2), device=dev, dtype=torch.float64, generator=g) * 1.2 - 0.1       # some samples outside the maps
    w = torch.rand((N, Lq, M, L, P), device=dev, dtype=torch.float64, generator=g) + 1e-5
    w = w / w.sum((-1, -2), keepdim=True)
    out = MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, w, 2)
    assert out.dtype == torch.float64 and out.shape == (N, Lq, M * D)
    ref = torch.zeros((N, Lq, M, D), device=dev, dtype=torch.float64)
    for l, (H, W) in enumerate(shapes.tolist()):
        vl = value[:, int(lsi[l]):int(lsi[l]) + H * W].permute(0, 2, 3, 1).reshape(N * M, D, H, W)
        grid = (2 * loc[:, :, :, l] - 1).permute(0, 2, 1, 3, 4).reshape(N * M, Lq, P, 2)
        smp = torch.nn.functional.grid_sample(vl, grid, mode="bilinear", padding_mode="zeros", align_corners=False)      # [N*M, D, Lq, P]
        ref += (smp.view(N, M, D, Lq, P) * w[:, :, :, l].permute(0, 2, 1, 3)[:, :, None]).sum(-1).permute(0, 3, 1, 2)
    assert float((out.view(N, Lq, M, D) - ref).abs().max()) < 1e-14
    vr, lr, wr = value.clone().requires_grad_(True), loc.clone().requires_grad_(True), w.clone().requires_grad_(True)
    assert torch.autograd.gradcheck(lambda a, b, c: MSDeformAttnFunction.apply(a, shapes, lsi, b, c, 2), (vr, lr, wr), eps=1e-6, atol=1e-7,
                                    rtol=1e-5, nondet_tol=1e-12)
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_forward(value, shapes, lsi, loc.float(), w, 2)                 # mixed precisions


def test_msda_compat_fresh_shape_tensors_of_different_pyramids(oracle):
    """The reference builds `spatial_shapes` / `level_start_index` fresh on every forward (msdeformattn.py:82-83:
    torch.as_tensor(list) -> cat / cumsum) and frees them afterwards, so the caching allocator hands the same small blocks to the
    next clip's tensors.  With MIN_SIZE_TRAIN (360, 480) + random crops consecutive clips have different pyramids of possibly the
    SAME total length S -- a host-side cache keyed by (pointer, version, shape) then serves the previous clip's shapes.  The shapes
    are read on the device now: a sequence of pyramids (equal S, transposed, smaller, larger) through the reference's calling
    pattern, forward and backward, each against the oracle."""
    import s2d_amd.compat as compat
    compat.install()
    import MultiScaleDeformableAttention as MSDA
    from s2d_amd.compat.ms_deform_attn_func import MSDeformAttnFunction
    dev = torch.device(DEV)
    N, M, D, P, Lq = 2, 8, 32, 4, 50
    pyramids = [[(6, 8), (12, 16), (24, 32)], [(8, 6), (16, 12), (32, 24)],      # same S, transposed levels
                [(4, 12), (8, 24), (16, 48)],                                    # same S again, other aspect ratio
                [(3, 5), (6, 10), (12, 20)], [(12, 23), (23, 45), (45, 90)], [(6, 8), (12, 16), (24, 32)]]
    seen_ptrs = set()
    for it, shapes_l in enumerate(pyramids):
        g = torch.Generator().manual_seed(100 + it)
        L = len(shapes_l)
        S = sum(h * w for h, w in shapes_l)
        value = torch.randn((N, S, M, D), generator=g)
        loc = torch.rand((N, Lq, M, L, P, 2), generator=g) * 1.2 - 0.1         # some samples outside the maps
        w = torch.softmax(torch.randn((N, Lq, M, L * P), generator=g), -1).view(N, Lq, M, L, P)
        go = torch.randn((N, Lq, M * D), generator=g)
        # the reference's construction, verbatim in effect: fresh device tensors, version 0, freed after the call
        spatial_shapes = torch.as_tensor(shapes_l, dtype=torch.long, device=dev)
        level_start_index = torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
        seen_ptrs.add(spatial_shapes.data_ptr())
        vr, lr, wr = (t.to(dev).requires_grad_(True) for t in (value, loc, w))
        out = MSDeformAttnFunction.apply(vr, spatial_shapes, level_start_index, lr, wr, 128)
        (out * go.to(dev)).sum().backward()
        o2 = MSDA.ms_deform_attn_forward(vr.detach(), spatial_shapes, level_start_index, lr.detach(), wr.detach(), 128)
        assert torch.equal(o2, out)
        shp = np.asarray(shapes_l, np.int64)
        lsi = np.concatenate([[0], np.cumsum(shp[:, 0] * shp[:, 1])[:-1]]).astype(np.int64)
        ref = oracle.msda_core(value.numpy(), shp, lsi, loc.numpy(), w.numpy())
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-5, atol=1e-5, err_msg=f"pyramid {it}")
        rv, rl, rw = oracle.msda_core_backward(value.numpy(), shp, lsi, loc.numpy(), w.numpy(), go.numpy())
        for got, want, name in ((vr.grad, rv, "grad_value"), (lr.grad, rl, "grad_loc"), (wr.grad, rw, "grad_attn")):
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * np.abs(want).max(), err_msg=f"pyramid {it} {name}")
        del spatial_shapes, level_start_index, out, o2, vr, lr, wr
    # (informational) the allocator did recycle the index tensors' block at least once -- the situation the old cache got wrong
    print("distinct spatial_shapes addresses over", len(pyramids), "calls:", len(seen_ptrs))


def test_msda_dev_shapes_rejected_on_device():
    """shapes that do not describe `value` cannot fail the call without a sync: the device-shape form writes zeros and raises
    the geometry record's flag (s2d_msda_dev_status); nothing is indexed by the bad numbers"""
    from s2d_amd import ops
    dev = torch.device(DEV)
    N, M, D, P, Lq, L = 1, 8, 32, 4, 20, 2
    S = 6 * 8 + 3 * 4
    value = torch.randn((N, S, M, D), device=dev)
    loc = torch.rand((N, Lq, M, L, P, 2), device=dev)
    w = torch.softmax(torch.randn((N, Lq, M, L * P), device=dev), -1).view(N, Lq, M, L, P)
    go = torch.randn((N, Lq, M * D), device=dev)
    good = torch.tensor([[6, 8], [3, 4]], dtype=torch.long, device=dev)
    lsi = torch.tensor([0, 48], dtype=torch.long, device=dev)
    out, ws = ops.msda_forward_dev(value, good, lsi, loc, w, want_ws=True)
    assert ops.msda_dev_status(ws) == 0 and out.abs().sum() > 0
    for bad in ([[6, 8], [30, 40]], [[0, 8], [3, 4]], [[-6, 8], [3, 4]]):
        shp = torch.tensor(bad, dtype=torch.long, device=dev)
        out, ws = ops.msda_forward_dev(value, shp, lsi, loc, w, want_ws=True)
        assert ops.msda_dev_status(ws) == 1 and torch.count_nonzero(out) == 0
        gv, gl, gw, ws = ops.msda_backward_dev(value, shp, lsi, loc, w, go, want_ws=True)
        assert ops.msda_dev_status(ws) == 1
        assert torch.count_nonzero(gv) == 0 and torch.count_nonzero(gl) == 0 and torch.count_nonzero(gw) == 0
    # overlapping levels: each level alone fits `value`, together they describe more pixels (and more bilinear cells than the
    # backward's workspace and sort keys are sized for) than it has -> the rejected path, not an out-of-range sort
    S2 = 6 * 8
    value2 = torch.randn((N, S2, M, D), device=dev)
    for shp, st in (([[6, 8], [6, 8]], [0, 0]), ([[6, 8], [4, 8]], [0, 16])):
        shp, st = torch.tensor(shp, dtype=torch.long, device=dev), torch.tensor(st, dtype=torch.long, device=dev)
        out, ws = ops.msda_forward_dev(value2, shp, st, loc, w, want_ws=True)
        assert ops.msda_dev_status(ws) == 1 and torch.count_nonzero(out) == 0
        gv, gl, gw, ws = ops.msda_backward_dev(value2, shp, st, loc, w, go, want_ws=True)
        assert ops.msda_dev_status(ws) == 1
        assert torch.count_nonzero(gv) == 0 and torch.count_nonzero(gl) == 0 and torch.count_nonzero(gw) == 0
    with pytest.raises(RuntimeError):
        ops.msda_forward_dev(value, good.cpu(), lsi, loc, w)                  # index tensors must be on the device
    with pytest.raises(RuntimeError):
        ops.msda_forward_dev(value, good.int(), lsi, loc, w)                  # and int64, as the extension reads them


def test_msda_compat_module_fails_loudly_on_bad_shapes():
    """the drop-in module must not train on zeros.  A call with shapes that do not describe `value` executes (zeros),
    sets the pinned error word, and the module raises at the next call / at check() -- without a per-call sync"""
    import s2d_amd.compat as compat
    compat.install()
    import MultiScaleDeformableAttention as MSDA
    dev = torch.device(DEV)
    N, M, D, P, Lq, L = 1, 8, 32, 4, 20, 2
    S = 6 * 8 + 3 * 4
    value = torch.randn((N, S, M, D), device=dev)
    loc = torch.rand((N, Lq, M, L, P, 2), device=dev)
    w = torch.softmax(torch.randn((N, Lq, M, L * P), device=dev), -1).view(N, Lq, M, L, P)
    good = torch.tensor([[6, 8], [3, 4]], dtype=torch.long, device=dev)
    bad = torch.tensor([[6, 8], [30, 40]], dtype=torch.long, device=dev)
    lsi = torch.tensor([0, 48], dtype=torch.long, device=dev)
    MSDA.reset()                                                    # (earlier tests feed the *_dev entry points bad shapes on purpose)
    MSDA.check()                                                    # clean
    out = MSDA.ms_deform_attn_forward(value, good, lsi, loc, w, 64)
    MSDA.check()
    assert out.abs().sum() > 0
    MSDA.ms_deform_attn_forward(value, bad, lsi, loc, w, 64)        # enqueued; nothing can be known on the host yet
    with pytest.raises(RuntimeError):
        MSDA.check()
    MSDA.check()                                                    # the flag was consumed by the raise
    MSDA.ms_deform_attn_forward(value, bad, lsi, loc, w, 64)
    torch.cuda.synchronize()                                        # (the trainer's own sync, e.g. _write_metrics)
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_forward(value, good, lsi, loc, w, 64)   # the next call raises on entry, before it launches anything
    go = torch.randn((N, Lq, M * D), device=dev)
    MSDA.ms_deform_attn_backward(value, bad, lsi, loc, w, go, 64)
    with pytest.raises(RuntimeError):
        MSDA.check()
    # the float64 instantiation (ops/test.py gradchecks in double) reports through the same word: a rejected level must not pass as zeros
    v64, l64, w64 = value.double(), loc.double(), w.double()
    MSDA.check()
    assert MSDA.ms_deform_attn_forward(v64, good, lsi, l64, w64, 64).abs().sum() > 0
    MSDA.check()
    MSDA.ms_deform_attn_forward(v64, bad, lsi, l64, w64, 64)
    with pytest.raises(RuntimeError):
        MSDA.check()
    MSDA.ms_deform_attn_backward(v64, bad, lsi, l64, w64, go.double(), 64)
    with pytest.raises(RuntimeError):
        MSDA.check()
    MSDA.check()


@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
def test_gemm_dropout_in_the_other_dense_modes(mode):
    """training-mode encoder layers under --dense bf16x3 / f32: the mask epilogue exists in the split-fp16 kernels only, the other
    modes apply the same mask as a separate pass (same values as that mode's GEMM followed by the mask)"""
    from s2d_amd import ops
    from s2d_amd.modeling.pixel_decoder import MSDeformAttnTransformerEncoderLayer
    ops.set_dense_mode(mode)
    try:
        g = torch.Generator().manual_seed(5)
        M, N, K = 500, 256, 256
        A = torch.randn((M, K), generator=g).to(DEV)
        W = torch.nn.Parameter(torch.randn((N, K), generator=g).mul_(K ** -0.5).to(DEV))
        b = torch.randn((N,), generator=g).to(DEV)
        R = torch.randn((M, N), generator=g).to(DEV)
        p, seed, site = 0.3, 1234, 1
        y = ops.gemm_nt(A, W, bias=b, res=R, relu=True, dropout=(p, seed, site))
        ref = torch.relu(ops.gemm_nt(A, W, bias=b) * ops.dropout(torch.ones((M, N), device=DEV), p, seed, site) + R)
        assert torch.equal(y, ref)
        torch.manual_seed(2)
        layer = MSDeformAttnTransformerEncoderLayer(dropout=0.3).to(DEV).train()
        shapes = [(4, 6), (8, 12), (16, 24)]
        S = sum(h * w for h, w in shapes)
        out = layer(torch.randn((2, S, 256), device=DEV), torch.randn((S, 256), device=DEV), torch.tensor(shapes, dtype=torch.int64))
        assert torch.isfinite(out).all()
    finally:
        ops.set_dense_mode("f16x3")


# ------------------------------------------------------------------------------------------------ A8 prepare_targets
def test_prepare_targets_product_path_vs_reference_golden():
    """_gt_target_list + TargetSet.from_list (the product's prepare_targets, kd_video_maskformer_model.py:358-386) against the
    reference's own output (tests/golden/prepare_targets.npz): padded bit planes, dropped never-present instance, counts,
    and the DropLoss predicate"""
    from s2d_amd.modeling import TargetSet
    from s2d_amd.modeling.meta_arch import _gt_target_list
    from s2d_amd.utils import synth
    g = golden("prepare_targets")
    T, H0, W0, Hp, Wp, n = (int(v) for v in g["dims"])
    m, ids = synth.ellipse_targets(int(g["seed"]), 1, n, T, H0, W0, sparse=0.6)
    ids[2, :] = -1
    m[2] = 0
    inst = [{"gt_masks": torch.from_numpy(m[:, t]).bool(), "gt_ids": torch.from_numpy(ids[:, t]),
             "gt_classes": torch.zeros(n, dtype=torch.int64)} for t in range(T)]
    lst = _gt_target_list([{"instances": inst}], T, Hp, Wp, torch.device(DEV))
    want = np.unpackbits(g["masks"], axis=-1)[..., :Wp]
    assert lst[0].shape[0] == int(g["n_out"])
    np.testing.assert_array_equal(lst[0].cpu().numpy(), want)
    ts = TargetSet.from_list(lst, device=DEV)
    assert ts.count.cpu().tolist() == [int(g["n_out"])] and ts.host_counts == [int(g["n_out"])]
    np.testing.assert_array_equal(ts.masks[0, :int(g["n_out"])].cpu().numpy(), want)
    np.testing.assert_array_equal(ts.nonempty[0].cpu().numpy() != 0, want.reshape(want.shape[0], T, -1).any(-1))
    # detectron2-style Instances objects (attribute access, BitMasks with .tensor) take the same path
    objs = [NS(gt_masks=NS(tensor=i["gt_masks"]), gt_ids=i["gt_ids"], gt_classes=i["gt_classes"]) for i in inst]
    lst2 = _gt_target_list([{"instances": objs}], T, Hp, Wp, torch.device(DEV))
    assert torch.equal(lst2[0], lst[0])


# ------------------------------------------------------------------------------------------------ shipped config + trainer statements
def _ns(d):
    return NS(**{k: _ns(v) if isinstance(v, dict) else v for k, v in d.items()})


def _shipped_cfg():
    return _ns(json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kd_config.json"))))


def _clip_batch(T, H0, W0, n, seed, B=1):
    from s2d_amd.utils import synth
    data = []
    for b in range(B):
        fr = synth.smooth_frames_u8(seed + b, 1, T, H0, W0)                         # [T,3,H0,W0]
        m, ids = synth.ellipse_targets(seed + b, 2, n, T, H0, W0, sparse=0.3)
        inst = [{"gt_masks": torch.from_numpy(m[:, t]).bool(), "gt_ids": torch.from_numpy(ids[:, t]),
                 "gt_classes": torch.zeros(n, dtype=torch.int64)} for t in range(T)]
        data.append({"image": [torch.from_numpy(fr[t]) for t in range(T)], "instances": inst, "height": H0, "width": W0})
    return data


def test_shipped_config_trains_with_the_reference_trainer_statements():
    """KDVideoMaskFormer built by the registry from the literal keys of the shipped KD yaml (DROPOUT 0.3, 160 000 points,
    3 frames, masks-only DropLoss) and driven by the reference trainer's own statements (train_loop.py:709-726):
        loss_dict = model(data); losses = sum(loss_dict.values()); grad_scaler.scale(losses).backward(); optimizer.step()
    with a stock torch optimizer.  The gradients autograd delivers are the HIP backward's."""
    from s2d_amd import ops
    from s2d_amd.modeling.meta_arch import META_ARCH_REGISTRY
    cfg = _shipped_cfg()
    assert cfg.MODEL.MASK_FORMER.DROPOUT == 0.3 and cfg.MODEL.META_ARCHITECTURE == "KDVideoMaskFormer"
    torch.manual_seed(0)
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE).from_config(cfg).to(DEV)
    model.teacher.load_state_dict(model.student.state_dict())
    with torch.no_grad():
        model.teacher[1].predictor.class_embed.bias.copy_(torch.tensor([2.0, -2.0]))       # some queries pass the 0.75 KD threshold
    model.train()
    T = cfg.INPUT.SAMPLING_FRAME_NUM
    data = _clip_batch(T, 64, 96, 3, seed=5, B=2)
    students = [p for p in model.student.parameters()]

    def seeded():
        torch.manual_seed(123); ops._DROP_CALLS[0] = 0
        model.criterion.seed = 0; model.criterion.matcher.seed = 0

    # the reference trainer's statements
    opt = torch.optim.AdamW(students, lr=1e-4)
    scale = 1024.0                                                             # a GradScaler's loss scale
    seeded()
    loss_dict = model(data)
    assert len(loss_dict) == 42 and all(v.dim() == 0 and v.requires_grad for v in loss_dict.values())
    losses = sum(loss_dict.values())
    (losses * scale).backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in students)
    assert all(p.grad is None for p in model.teacher.parameters())
    got = [p.grad.clone() for p in students]
    # the direct path with the same seeds: forward_backward leaves d(sum)/dp in .grad
    for p in students:
        p.grad = None
    seeded()
    from s2d_amd.modeling import TargetSet
    from s2d_amd.modeling.meta_arch import _gt_target_list
    images = model.preprocess(data)
    gt = TargetSet.from_list(_gt_target_list(data, T, images.shape[1], images.shape[2], model.device), device=model.device)
    direct = model.forward_backward(images, gt)
    assert all(float(direct[k]) == float(loss_dict[k]) for k in direct)
    for p, g in zip(students, got):                                            # (float atomics in two backward kernels: not bitwise run to run)
        ref = p.grad * scale
        assert float((g - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-12
    # an optimizer step moves the student; accumulation into existing .grad works like autograd's
    before = [p.detach().clone() for p in students[:8]]
    for p, g in zip(students, got):
        p.grad = g / scale
    opt.step()
    assert any(not torch.equal(a, b) for a, b in zip(before, students[:8]))
    seeded()
    keep = [p.grad.clone() for p in students[:8]]
    sum(model(data).values()).backward()
    for p, k in zip(students[:8], keep):
        assert not torch.equal(p.grad, k)                                      # accumulated on top of the stale .grad
    # unequal upstream weights are refused, not answered wrongly
    ld = model(data)
    with pytest.raises(NotImplementedError):
        (ld["loss_mask"] * 2 + ld["loss_dice"]).backward()
    # no graph, no backward work under no_grad; eval mode returns the inference dict (dropout inactive)
    with torch.no_grad():
        assert not any(v.requires_grad for v in model(data).values())
    model.eval()
    out = model(data[:1])
    assert set(out) >= {"image_size", "pred_scores", "pred_labels", "pred_masks"}


def test_distillation_nms_branch_vs_oracle(oracle):
    """MODEL.MASK_FORMER.DISTILLATION_NMS (kd_video_maskformer_model.py:484-520): duplicated / heavily overlapping pseudo
    targets are dropped greedily at IoU > NMS_THRESH, the survivors compacted in order, counts and DropLoss flags following;
    the keep set equals the oracle's mask_nms (pinned by the reference's inference goldens)"""
    from s2d_amd import ops
    from s2d_amd.modeling import build_kd_model
    T, H, W = 2, 32, 64
    yy, xx = np.mgrid[0:H, 0:W]
    disc = lambda cy, cx, r: ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r)
    planes = np.stack([np.stack([disc(16, 20, 9)] * T), np.stack([disc(16, 20, 9)] * T), np.stack([disc(16, 44, 8)] * T),
                       np.stack([disc(16, 21, 9)] * T), np.stack([disc(10, 44, 3)] * T)]).astype(np.uint8)       # 1 duplicates 0, 3 overlaps 0
    tgt = torch.zeros((2, 6, T, H, W), dtype=torch.uint8, device=DEV)
    tgt[0, :5] = torch.from_numpy(planes).to(DEV)
    tgt[1, :2] = torch.from_numpy(planes[[2, 4]]).to(DEV)
    cnt = torch.tensor([5, 2], dtype=torch.int32, device=DEV)
    ne = ops.target_nonempty(tgt, cnt)
    kept = torch.arange(6, dtype=torch.int32, device=DEV).repeat(2, 1).contiguous()
    model = build_kd_model(num_queries=8, num_frames=T, num_points=64)
    model.distillation_nms, model.nms_threshold = True, 0.75
    t2, c2, n2 = model._kd_nms(tgt.clone(), cnt.clone(), ne.clone(), kept)
    want = oracle.mask_nms(planes.astype(bool), np.zeros(5, np.int64), 0.75)
    assert list(want) == [0, 2, 4] and c2.cpu().tolist() == [3, 2]
    np.testing.assert_array_equal(t2[0, :3].cpu().numpy(), planes[[0, 2, 4]])
    assert int(t2[0, 3:].sum()) == 0 and int(n2[0, 3:].sum()) == 0 and kept[0, :3].cpu().tolist() == [0, 2, 4]
    np.testing.assert_array_equal(t2[1].cpu().numpy(), tgt[1].cpu().numpy())
    bits = ops.pack_mask_bits(torch.from_numpy(planes).to(DEV))
    inter = ops.mask_pair_counts(bits).cpu().numpy()
    flat = planes.reshape(5, -1).astype(np.int64)
    np.testing.assert_array_equal(inter, flat @ flat.T)


def test_amp_trainer_statements_autocast_and_gradscaler():
    """SOLVER.AMP.ENABLED True (every shipped config): the reference trainer wraps the forward in autocast and drives a
    GradScaler (engine/train_loop.py:709-726).  The HIP path computes in fp32 whatever the autocast state (the parity target),
    the losses come back fp32, and the scaler's loss scale flows through the bridge: unscaled gradients equal the unscaled run's"""
    from s2d_amd import ops
    from s2d_amd.modeling import build_kd_model
    torch.manual_seed(0)
    model = build_kd_model(num_queries=16, num_frames=2, num_points=256, weights=(2.0, 5.0, 5.0), dropout=0.0).to(DEV)
    with torch.no_grad():
        model.teacher[1].predictor.class_embed.bias.copy_(torch.tensor([2.0, -2.0]))
    model.train()
    data = _clip_batch(2, 64, 96, 3, seed=9, B=1)
    params = [p for p in model.student.parameters()]

    def grads(amp):
        for p in params:
            p.grad = None
        model.criterion.seed = 0; model.criterion.matcher.seed = 0
        scaler = torch.amp.GradScaler("cuda", enabled=amp, init_scale=2.0 ** 14)
        opt = torch.optim.SGD(params, lr=0.0)
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            loss_dict = model(data)
            losses = sum(loss_dict.values())
        assert losses.dtype == torch.float32
        scaler.scale(losses).backward()
        scaler.unscale_(opt)                                           # what scaler.step does first; also its inf check
        scaler.step(opt); scaler.update()
        return float(losses), [p.grad.clone() for p in params]

    l0, g0 = grads(False)
    l1, g1 = grads(True)
    assert l0 == l1
    for a, b in zip(g0, g1):
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-12

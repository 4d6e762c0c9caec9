"""GPU parity: MSDeformAttn kernels and the bandwidth-bound glue kernels vs goldens / the CPU oracle."""
import numpy as np
import pytest
import torch

from s2d_amd.utils import synth
from tests.conftest import golden

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def close(a, b, rtol):
    b = np.asarray(b, np.float64)
    np.testing.assert_allclose(np.asarray(a, np.float64), b, rtol=rtol, atol=rtol * max(np.abs(b).max(), 1e-30))


def test_msda_optest_fixture(oracle):
    """the reference's only known-answer test (ops/test.py:24-31,51-63), D=2 -> scalar path"""
    from s2d_amd import ops
    g = golden("msda_optest")
    lsi = oracle.level_start_index(g["shapes"])
    out = ops.msda_forward(_dev(g["value"]), g["shapes"], lsi, _dev(g["loc"]), _dev(g["w"])).cpu().numpy()
    np.testing.assert_allclose(out, g["out32"], rtol=1e-2, atol=1e-3)     # the reference's own tolerance
    np.testing.assert_allclose(out, g["out64"], rtol=1e-5, atol=1e-7)


def test_msda_forward_backward_golden(oracle):
    from s2d_amd import ops
    g = golden("msda_core")
    lsi = oracle.level_start_index(g["shapes"])
    v, lo, w = _dev(g["value"]), _dev(g["loc"]), _dev(g["w"])
    out = ops.msda_forward(v, g["shapes"], lsi, lo, w).cpu().numpy()
    close(out, g["out"], 1e-5)
    gv, gl, gw = ops.msda_backward(v, g["shapes"], lsi, lo, w, _dev(g["grad_out"]))
    close(gv.cpu().numpy(), g["grad_value"], 1e-4)
    close(gl.cpu().numpy(), g["grad_loc"], 1e-4)
    close(gw.cpu().numpy(), g["grad_w"], 1e-4)


@pytest.mark.parametrize("shapes,N", [([(6, 10), (12, 20), (23, 40)], 2), ([(23, 40), (46, 80), (92, 160)], 1)],
                         ids=["quarter_size_pyramid", "720p_pyramid"])
def test_msda_fused_vs_oracle_720p_shapes(oracle, shapes, N):
    """fused softmax+loc+gather against the oracle's module arithmetic: a quarter-size pyramid (two frames) and the real pyramid of
    a 736 x 1280 frame -- (23,40), (46,80), (92,160), S = 19 320 queries per frame, the benched geometry"""
    from s2d_amd import ops
    S = sum(h * w for h, w in shapes)
    M, D, L, P = 8, 32, 3, 4
    value = synth.randn(7, 1, (N, S, M * D))
    off = synth.randn(7, 2, (N, S, M, L, P, 2), 2.0)
    lg = synth.randn(7, 3, (N, S, M, L * P))
    oa = np.concatenate([off.reshape(N, S, -1), lg.reshape(N, S, -1)], -1)
    ref_pts = oracle.reference_points(shapes)
    norm = np.array([[w_, h_] for (h_, w_) in shapes], np.float32)
    loc = ref_pts[None, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    aw = oracle.softmax(lg, -1).reshape(N, S, M, L, P)
    ref = oracle.msda_core(value.reshape(N, S, M, D), np.array(shapes), oracle.level_start_index(shapes),
                           loc.astype(np.float32), aw.astype(np.float32))
    out = ops.msda_fused_forward(_dev(value), np.array(shapes), _dev(oa)).cpu().numpy()
    close(out, ref, 1e-5)
    # value / offsets+logits as column slices of one [N,S,288+256] buffer (the merged projection's output)
    both = _dev(np.concatenate([oa, value], -1))
    out1 = ops.msda_fused_forward(both[..., oa.shape[-1]:], np.array(shapes), both[..., :oa.shape[-1]]).cpu().numpy()
    assert np.array_equal(out1, out)
    # and the drop-in form on the same data
    out2 = ops.msda_forward(_dev(value.reshape(N, S, M, D)), np.array(shapes), oracle.level_start_index(shapes),
                            _dev(loc.astype(np.float32)), _dev(aw.astype(np.float32))).cpu().numpy()
    close(out2, ref, 1e-5)


def test_msda_windowed_kernel_opt_in_matches_the_gather_kernel(oracle, monkeypatch):
    """the opt-in LDS-windowed fused kernel (S2D_MSDA_WIN=1) against the oracle and bit for bit against the default gather kernel:
    offsets inside the windows (branch-free path), scattered beyond them (general path), border patches overhanging a level whose
    extent is not a multiple of 8, and column-slice operands"""
    import torch
    from s2d_amd import ops
    for shapes, scale in (([(6, 10), (12, 20), (23, 40)], 2.0), ([(5, 9), (11, 19), (22, 37)], 0.4), ([(6, 10), (12, 20), (23, 40)], 8.0)):
        S = sum(h * w for h, w in shapes)
        N, M, D, L, P = 2, 8, 32, 3, 4
        value = synth.randn(11, 1, (N, S, M * D))
        off = synth.randn(11, 2, (N, S, M, L, P, 2), scale)
        lg = synth.randn(11, 3, (N, S, M, L * P))
        oa = np.concatenate([off.reshape(N, S, -1), lg.reshape(N, S, -1)], -1)
        both = _dev(np.concatenate([oa, value], -1))
        v, o = both[..., oa.shape[-1]:], both[..., :oa.shape[-1]]
        monkeypatch.setenv("S2D_MSDA_WIN", "0")
        base = ops.msda_fused_forward(v, np.array(shapes), o)
        monkeypatch.setenv("S2D_MSDA_WIN", "1")
        win = ops.msda_fused_forward(v, np.array(shapes), o)
        assert torch.equal(win, base)
        ref_pts = oracle.reference_points(shapes)
        norm = np.array([[w_, h_] for (h_, w_) in shapes], np.float32)
        loc = ref_pts[None, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
        aw = oracle.softmax(lg, -1).reshape(N, S, M, L, P)
        ref = oracle.msda_core(value.reshape(N, S, M, D), np.array(shapes), oracle.level_start_index(shapes), loc.astype(np.float32), aw.astype(np.float32))
        close(win.cpu().numpy(), ref, 1e-5)


def test_msda_head_per_workgroup_kernel_matches_the_patch_kernel(oracle, monkeypatch):
    """the opt-in fused kernel with one head per workgroup and the coarsest level's plane of that head in LDS (S2D_MSDA_HEAD=1)
    bit for bit against the default 4 x 4-patch gather kernel and against the oracle: level extents that are /
    are not multiples of the 16 x 8 query patch, the coarsest level first / last / in the middle, offsets within a pixel and far
    outside the maps (zero-padding corners), column-slice operands, one frame and several"""
    import torch
    from s2d_amd import ops
    cases = (([(6, 10), (12, 20), (23, 40)], 2.0, 2), ([(23, 40), (46, 80), (92, 160)], 3.0, 1), ([(22, 37), (5, 9), (11, 19)], 0.4, 3),
             ([(8, 16), (16, 32), (3, 5)], 30.0, 2))
    for shapes, scale, N in cases:
        S = sum(h * w for h, w in shapes)
        M, D, L, P = 8, 32, 3, 4
        value = synth.randn(21, 1, (N, S, M * D))
        off = synth.randn(21, 2, (N, S, M, L, P, 2), scale)
        lg = synth.randn(21, 3, (N, S, M, L * P))
        oa = np.concatenate([off.reshape(N, S, -1), lg.reshape(N, S, -1)], -1)
        both = _dev(np.concatenate([oa, value], -1))
        v, o = both[..., oa.shape[-1]:], both[..., :oa.shape[-1]]
        monkeypatch.setenv("S2D_MSDA_HEAD", "0")
        base = ops.msda_fused_forward(v, np.array(shapes), o)
        monkeypatch.setenv("S2D_MSDA_HEAD", "1")
        head = ops.msda_fused_forward(v, np.array(shapes), o)
        assert torch.equal(head, base)
        ref_pts = oracle.reference_points(shapes)
        norm = np.array([[w_, h_] for (h_, w_) in shapes], np.float32)
        loc = ref_pts[None, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
        aw = oracle.softmax(lg, -1).reshape(N, S, M, L, P)
        ref = oracle.msda_core(value.reshape(N, S, M, D), np.array(shapes), oracle.level_start_index(shapes), loc.astype(np.float32), aw.astype(np.float32))
        close(head.cpu().numpy(), ref, 1e-5)


def test_normalize_pad_maxpool(oracle):
    from s2d_amd import ops
    fr = synth.smooth_frames_u8(3, 1, 2, 45, 70)
    ref = oracle.normalize_pad(fr)                       # [F,3,Hp,Wp]
    out = ops.normalize_pad(_dev(fr)).cpu().numpy()
    assert out.shape == (2, 64, 96, 4)
    close(out[..., :3].transpose(0, 3, 1, 2), ref, 1e-6)
    assert (out[..., 3] == 0).all()
    x = synth.randn(3, 2, (2, 64, 17, 23))
    y = ops.maxpool3x3s2(_dev(x.transpose(0, 2, 3, 1))).cpu().numpy().transpose(0, 3, 1, 2)
    np.testing.assert_array_equal(y, oracle.max_pool_3x3_s2_p1(x))


def test_groupnorm_layernorm_add_pe(oracle):
    from s2d_amd import ops
    x = synth.randn(4, 1, (2, 256, 12, 20)) * 3 + 1
    ga, be = synth.randn(4, 2, (256,)) * 0.1 + 1, synth.randn(4, 3, (256,)) * 0.1
    ref = oracle.group_norm(x, 32, ga, be)
    xd = _dev(x.transpose(0, 2, 3, 1))
    y = ops.groupnorm_nhwc(xd, 32, _dev(ga), _dev(be)).cpu().numpy().transpose(0, 3, 1, 2)
    close(y, ref, 1e-5)
    up = synth.randn(4, 4, (2, 256, 6, 10))
    ref2 = np.maximum(ref + oracle.resize_bilinear(up, 12, 20), 0)
    y2 = ops.groupnorm_nhwc(xd, 32, _dev(ga), _dev(be), up=_dev(up.transpose(0, 2, 3, 1)), relu=True)
    close(y2.cpu().numpy().transpose(0, 3, 1, 2), ref2, 1e-5)
    # layernorm(x + res)
    a, r = synth.randn(4, 5, (3, 50, 256)), synth.randn(4, 6, (3, 50, 256))
    close(ops.layernorm(_dev(a), _dev(ga), _dev(be), res=_dev(r)).cpu().numpy(), oracle.layer_norm(a + r, ga, be), 1e-5)
    # broadcast add
    b = synth.randn(4, 7, (50, 256))
    close(ops.add_bcast(_dev(a), _dev(b)).cpu().numpy(), a + b[None], 1e-7)
    # position encodings vs the reference goldens
    g = golden("pe")
    pe2 = ops.pe_sine(0, 5, 7).cpu().numpy().reshape(5, 7, 256).transpose(2, 0, 1)
    np.testing.assert_allclose(pe2, g["pe2"][0], rtol=0, atol=2e-5)
    pe3 = ops.pe_sine(3, 4, 6).cpu().numpy().reshape(3, 4, 6, 256).transpose(0, 3, 1, 2)
    np.testing.assert_allclose(pe3, g["pe3"][0], rtol=0, atol=2e-5)


def test_msda_backward_sorted_is_reproducible_and_matches_the_atomic_form():
    """the training step's MSDeformAttn backward (sampling graph inverted by a stable sort, grad_value rows gathered) against
    the reference-style scatter with float atomics on a 3-level pyramid with offsets that leave the maps: same gradients
    (the two differ only in summation order), and the sorted form is bitwise identical run to run"""
    import torch
    from s2d_amd import ops
    shapes = np.array([(12, 20), (23, 40), (46, 80)])
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    N, M, D, L, P = 3, 8, 32, 3, 4
    g = torch.Generator().manual_seed(11)
    value = torch.randn((N, S, M, D), generator=g).cuda()
    loc = (torch.rand((N, S, M, L, P, 2), generator=g) * 1.3 - 0.15).cuda()          # some samples outside [0, 1]
    loc[0, :50] = 0.5                                                               # a hot cell: hundreds of samples on one pixel
    aw = torch.softmax(torch.randn((N, S, M, L * P), generator=g), -1).view(N, S, M, L, P).cuda()
    go = torch.randn((N, S, M * D), generator=g).cuda()
    lsi = np.concatenate([[0], np.cumsum(shapes[:, 0] * shapes[:, 1])[:-1]])
    a = ops.msda_backward(value, shapes, lsi, loc, aw, go)
    b = ops.msda_backward(value, shapes, lsi, loc, aw, go)
    c = ops.msda_backward(value, shapes, lsi, loc, aw, go, atomics=True)
    for x, y, z, name in zip(a, b, c, ("grad_value", "grad_loc", "grad_attn")):
        assert torch.equal(x, y), name
        scale = float(z.abs().max())
        assert float((x - z).abs().max()) <= 2e-5 * scale, name

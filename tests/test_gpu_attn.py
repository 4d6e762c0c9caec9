"""GPU parity: attention-mask bits and the streamed masked cross-attention vs the CPU oracle."""
import math

import numpy as np
import pytest
import torch

from s2d_amd.utils import synth

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _ref_attn(o, q, k, v, mask, H=8):
    """q [B,Q,C], k/v [B,K,C], mask bool [B,Q,K] True = masked; float64 softmax attention per head."""
    B, Q, C = q.shape
    d = C // H
    out = np.zeros((B, Q, C))
    for b in range(B):
        for h in range(H):
            s = (q[b, :, h * d:(h + 1) * d].astype(np.float64) / math.sqrt(d)) @ k[b, :, h * d:(h + 1) * d].astype(np.float64).T
            if mask is not None:
                s = np.where(mask[b], -np.inf, s)
            s = s - s.max(-1, keepdims=True)
            a = np.exp(s)
            a /= a.sum(-1, keepdims=True)
            out[b, :, h * d:(h + 1) * d] = a @ v[b, :, h * d:(h + 1) * d].astype(np.float64)
    return out


@pytest.mark.parametrize("B,Q,T,hm,wm,hl,wl", [(2, 100, 2, 16, 24, 4, 6), (1, 16, 3, 16, 24, 8, 12), (2, 100, 2, 32, 40, 16, 20),
                                                 (1, 37, 1, 8, 8, 8, 8)])
def test_mask_and_cross_attention(oracle, B, Q, T, hm, wm, hl, wl):
    from s2d_amd import ops
    C, H = 256, 8
    ldq = (Q + 3) // 4 * 4
    ml = synth.smooth_logits(9, 1, (B, Q, T), (hm, wm))                 # [B,Q,T,hm,wm] query-major (reference layout)
    ml[:, 1] = -np.abs(ml[:, 1]) - 0.1                                   # query 1: masked everywhere -> the :413 fix
    if Q > 5:
        ml[0, 5] = np.abs(ml[0, 5]) + 0.1                                # never masked
    pm = np.zeros((B, T * hm * wm, ldq), np.float32)
    pm[:, :, :Q] = ml.transpose(0, 2, 3, 4, 1).reshape(B, T * hm * wm, Q)
    bits, unm = ops.attn_mask_bits(_dev(pm), B, Q, T, hm, wm, hl, wl)
    K = T * hl * wl
    ref_mask = (oracle.resize_bilinear(ml, hl, wl) < 0).reshape(B, Q, K)  # video_...decoder.py:460-463
    bw = bits.cpu().numpy().view(np.uint32)
    got = np.zeros((B, Q, K), bool)
    for qq in range(Q):
        got[:, qq, :] = (bw[:, :, qq // 32] >> (qq % 32)) & 1
    np.testing.assert_array_equal(got, ref_mask)
    um = unm.cpu().numpy().view(np.uint32)
    has = ~ref_mask.all(-1)
    for qq in range(Q):
        np.testing.assert_array_equal(((um[:, qq // 32] >> (qq % 32)) & 1).astype(bool), has[:, qq])
    assert not has[:, 1].any()

    q = synth.randn(9, 2, (B, Q, C))
    k = synth.randn(9, 3, (B, K, C))
    v = synth.randn(9, 4, (B, K, C))
    fixed = ref_mask.copy()
    fixed[fixed.all(-1)] = False
    ref = _ref_attn(oracle, q, k, v, fixed)
    out = ops.masked_attn(_dev(q), _dev(k), _dev(v), bits, unm).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    # unmasked (self-attention form)
    out2 = ops.masked_attn(_dev(q), _dev(q), _dev(q)).cpu().numpy()
    ref2 = _ref_attn(oracle, q, q, q, None)
    np.testing.assert_allclose(out2, ref2, rtol=1e-4, atol=1e-4 * np.abs(ref2).max())


def test_cross_attention_long_keys(oracle):
    """K large enough for 32 key splits; spiky scores exercise the online-softmax rescale"""
    from s2d_amd import ops
    B, Q, K, C = 1, 100, 20000, 256
    q = synth.randn(10, 1, (B, Q, C))
    k = synth.randn(10, 2, (B, K, C))
    v = synth.randn(10, 3, (B, K, C))
    k[0, 17000] = q[0, 3] * 4.0        # one key dominates query 3 late in the stream
    ref = _ref_attn(oracle, q, k, v, None)
    out = ops.masked_attn(_dev(q), _dev(k), _dev(v)).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


def test_attn_mask_bits_compact_matches_full():
    """the tap-index table reproduces the kernel's bilinear source rule: logits gathered at those pixels give the same bits"""
    from s2d_amd import ops
    B, Q, T, hm, wm = 2, 100, 3, 46, 80
    ml = torch.randn((B, T * hm * wm, Q), device="cuda")
    for (hl, wl) in [(6, 10), (12, 20), (23, 40), (7, 13)]:
        idx = ops.attn_mask_tap_index(T, hm, wm, hl, wl, "cuda")
        b0, u0 = ops.attn_mask_bits(ml, B, Q, T, hm, wm, hl, wl)
        b1, u1 = ops.attn_mask_bits(ml.index_select(1, idx).contiguous(), B, Q, T, hm, wm, hl, wl, compact=True)
        assert torch.equal(b0, b1) and torch.equal(u0, u1)


def test_masked_attn_reads_column_slices_in_place():
    """k / v as column slices of a wider projection output (row stride > C) == the same values copied out, bitwise"""
    import torch
    from s2d_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    B, Q, K, C = 2, 100, 1500, 256
    q = torch.randn((B, Q, C), device="cuda", generator=g)
    wide_k = torch.randn((B, K, 3 * C), device="cuda", generator=g)
    wide_v = torch.randn((B, K, 2 * C), device="cuda", generator=g)
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (B, K, 4), device="cuda", dtype=torch.int32, generator=g)
    unm = torch.full((B, 4), -1, device="cuda", dtype=torch.int32)
    ks, vs = wide_k[..., C:2 * C], wide_v[..., C:]
    a = ops.masked_attn(q, ks, vs, bits, unm)
    b = ops.masked_attn(q, ks.contiguous(), vs.contiguous(), bits, unm)
    assert torch.equal(a, b)

"""The encoder layer's feed-forward block in one launch (csrc/ffn.hip) against the CPU oracle and against the two-launch form.
Reference: mask2former/modeling/pixel_decoder/msdeformattn.py:116-131 (forward_ffn, norm1 / norm2 around it)."""
import numpy as np
import pytest
import torch

from s2d_amd.utils import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _params(F, seed):
    W1 = synth.randn(seed, 1, (F, 256)) * 0.06
    b1 = synth.randn(seed, 2, (F,)) * 0.1
    W2 = synth.randn(seed, 3, (256, F)) * 0.03
    b2 = synth.randn(seed, 4, (256,)) * 0.1
    g1, be1 = synth.randn(seed, 5, (256,)) * 0.2 + 1, synth.randn(seed, 6, (256,)) * 0.1
    g2, be2 = synth.randn(seed, 7, (256,)) * 0.2 + 1, synth.randn(seed, 8, (256,)) * 0.1
    return W1, b1, W2, b2, g1, be1, g2, be2


def _oracle_ffn(oracle, x, W1, b1, W2, b2, ln1, ln2, drop):
    x = x.astype(np.float64)
    xn = oracle.layer_norm(x, ln1[0].astype(np.float64), ln1[1].astype(np.float64)) if ln1 else x
    hid = np.maximum(xn @ W1.astype(np.float64).T + b1, 0)
    out = None
    if drop:
        p, seed, sh, so = drop
        hid = hid * oracle.dropout_multipliers(x.shape[0], W1.shape[0], p, seed, sh)
    out = hid @ W2.astype(np.float64).T + b2
    if drop:
        out = out * oracle.dropout_multipliers(x.shape[0], 256, p, seed, so)
    y = xn + out
    if ln2:
        y = oracle.layer_norm(y, ln2[0].astype(np.float64), ln2[1].astype(np.float64))
    return y, xn


@pytest.mark.parametrize("M,F", [(128, 1024), (200, 1024), (1, 64), (129, 32), (1000, 2048), (5000, 1024)])
@pytest.mark.parametrize("ln", ["none", "ln2", "ln1ln2"])
@pytest.mark.parametrize("p", [0.0, 0.3])
def test_ffn_fused_vs_oracle(oracle, M, F, ln, p):
    """float64 oracle; tolerance 2e-5 of the output scale (split-fp16 x3 through two contractions + LayerNorm), masks exact"""
    from s2d_amd import ops
    W1, b1, W2, b2, g1, be1, g2, be2 = _params(F, 11)
    x = synth.randn(12, 1, (M, 256)) * 1.5
    ln1 = (g1, be1) if ln == "ln1ln2" else None
    ln2 = (g2, be2) if ln != "none" else None
    drop = (p, 0x1234567887654321, 1, 2) if p > 0 else None
    ref, refn = _oracle_ffn(oracle, x, W1, b1, W2, b2, ln1, ln2, drop)
    W1d, W2d = torch.nn.Parameter(_dev(W1)), torch.nn.Parameter(_dev(W2))
    out = ops.ffn_fused(_dev(x), W1d, _dev(b1), W2d, _dev(b2), ln1=tuple(map(_dev, ln1)) if ln1 else None,
                        ln2=tuple(map(_dev, ln2)) if ln2 else None, dropout=drop, want_xn=ln1 is not None)
    if ln1 is not None:
        out, xn = out
        np.testing.assert_allclose(xn.cpu().numpy(), refn, rtol=0, atol=3e-6 * np.abs(refn).max())
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=2e-5 * np.abs(ref).max())


@pytest.mark.parametrize("M", [309, 4096])
def test_ffn_fused_vs_two_launch_form(M):
    """same masks, same arithmetic class: the fused launch against gemm_nt(dropout) x 2 + layernorm (the taped training path)"""
    from s2d_amd import ops
    W1, b1, W2, b2, g1, be1, g2, be2 = (_dev(a) for a in _params(1024, 21))
    W1, W2 = torch.nn.Parameter(W1), torch.nn.Parameter(W2)
    x = _dev(synth.randn(22, 1, (M, 256)))
    p, seed = 0.3, 987654321
    s1 = ops.layernorm(x, g1, be1)
    h = ops.gemm_nt(s1, W1, bias=b1, relu=True, dropout=(p, seed, 1))
    ref = ops.layernorm(ops.gemm_nt(h, W2, bias=b2, res=s1, dropout=(p, seed, 2)), g2, be2)
    out, xn = ops.ffn_fused(x, W1, b1, W2, b2, ln1=(g1, be1), ln2=(g2, be2), dropout=(p, seed, 1, 2), want_xn=True)
    assert torch.allclose(xn, s1, rtol=0, atol=2e-6 * float(s1.abs().max()))
    assert torch.allclose(out, ref, rtol=0, atol=1e-5 * float(ref.abs().max()))
    # a changed weight is re-packed (parameter version)
    with torch.no_grad():
        W1.mul_(2.0)
    out2 = ops.ffn_fused(x, W1, b1, W2, b2, ln1=(g1, be1), ln2=(g2, be2), dropout=(p, seed, 1, 2))
    h2 = ops.gemm_nt(s1, W1, bias=b1, relu=True, dropout=(p, seed, 1))
    ref2 = ops.layernorm(ops.gemm_nt(h2, W2, bias=b2, res=s1, dropout=(p, seed, 2)), g2, be2)
    assert torch.allclose(out2, ref2, rtol=0, atol=1e-5 * float(ref2.abs().max()))
    # run to run: same bits
    assert torch.equal(out2, ops.ffn_fused(x, W1, b1, W2, b2, ln1=(g1, be1), ln2=(g2, be2), dropout=(p, seed, 1, 2)))


@pytest.mark.parametrize("M,S,p", [(200, 50, 0.0), (1000, 333, 0.3), (4097, 4097, 0.3), (128, 7, 0.0), (40000, 2500, 0.3)])
def test_ffn_fused_with_next_layer_projection(oracle, M, S, p):
    """the launch that also applies the next encoder layer's merged projection (544 columns: 288 offsets / logits with the
    row-periodic pos term, 256 value columns with their bias) to its output rows: against the float64 oracle of
    [sampling_offsets | attention_weights | value_proj](LN2(...)) (ms_deform_attn.py:98-104), and the output itself unchanged"""
    from s2d_amd import ops
    F = 1024
    W1, b1, W2, b2, g1, be1, g2, be2 = _params(F, 31)
    x = synth.randn(32, 1, (M, 256)) * 1.5
    Wp = synth.randn(33, 1, (544, 256)) * 0.05
    bp = np.concatenate([np.zeros(288, np.float32), synth.randn(33, 2, (256,)) * 0.1]).astype(np.float32)
    pos = synth.randn(33, 3, (S, 288)) * 0.5
    drop = (p, 0x0123456789ABCDEF, 1, 2) if p > 0 else None
    ref, _ = _oracle_ffn(oracle, x, W1, b1, W2, b2, (g1, be1), (g2, be2), drop)
    refp = ref @ Wp.astype(np.float64).T + bp
    refp[:, :288] += pos[np.arange(M) % S]
    W1d, W2d, Wpd = torch.nn.Parameter(_dev(W1)), torch.nn.Parameter(_dev(W2)), ops.mark_static(_dev(Wp))
    args = (_dev(x), W1d, _dev(b1), W2d, _dev(b2))
    kw = dict(ln1=(_dev(g1), _dev(be1)), ln2=(_dev(g2), _dev(be2)), dropout=drop)
    y, out = ops.ffn_fused(*args, post=(Wpd, _dev(bp), _dev(pos)), **kw)
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=0, atol=2e-5 * np.abs(ref).max())
    np.testing.assert_allclose(out.cpu().numpy(), refp, rtol=0, atol=2e-5 * np.abs(refp).max())
    assert torch.equal(y, ops.ffn_fused(*args, **kw))                      # the output does not depend on the extra phase
    y2, out2 = ops.ffn_fused(*args, post=(Wpd, _dev(bp), _dev(pos)), **kw)
    assert torch.equal(out, out2) and torch.equal(y, y2)                    # run to run: same bits
    # against the launch it replaces (same arithmetic class)
    two = ops.gemm_nt(y, Wpd, bias=_dev(bp), res=_dev(pos), res_rows=S, res_cols=288)
    assert torch.allclose(out, two, rtol=0, atol=3e-6 * float(two.abs().max()))


@pytest.mark.parametrize("M,S,p,post", [(200, 50, 0.0, False), (1000, 333, 0.3, True), (4097, 4097, 0.3, False), (128, 7, 0.0, True),
                                        (1, 1, 0.3, True), (40000, 2500, 0.3, True)])
def test_ffn_fused_with_attention_output_projection(oracle, M, S, p, post):
    """the launch that starts from the deformable attention's sampled values: x1 = res + dropout1(output_proj(samp)) (mask site 0;
    ms_deform_attn.py:124, msdeformattn.py:124-125), then norm1, the FFN, norm2 (and optionally the next layer's projection) --
    against the float64 oracle of the whole chain and against the launches it replaces"""
    from s2d_amd import ops
    F = 1024
    W1, b1, W2, b2, g1, be1, g2, be2 = _params(F, 41)
    samp = synth.randn(42, 1, (M, 256)) * 1.2
    res = synth.randn(42, 2, (M, 256)) * 1.5
    Wo = synth.randn(43, 1, (256, 256)) * 0.07
    bo = synth.randn(43, 2, (256,)) * 0.1
    seed = 0x0FEDCBA987654321
    drop = (p, seed, 1, 2) if p > 0 else None
    a = samp.astype(np.float64) @ Wo.astype(np.float64).T + bo
    if p > 0:
        a = a * oracle.dropout_multipliers(M, 256, p, seed, 0)
    x1 = res + a
    ref, refn = _oracle_ffn(oracle, x1, W1, b1, W2, b2, (g1, be1), (g2, be2), drop)
    W1d, W2d, Wod = torch.nn.Parameter(_dev(W1)), torch.nn.Parameter(_dev(W2)), torch.nn.Parameter(_dev(Wo))
    kw = dict(ln1=(_dev(g1), _dev(be1)), ln2=(_dev(g2), _dev(be2)), dropout=drop)
    wargs = (W1d, _dev(b1), W2d, _dev(b2))
    pre = (Wod, _dev(bo), _dev(res), 0)
    pk = {}
    if post:
        Wp = synth.randn(44, 1, (544, 256)) * 0.05
        bp = np.concatenate([np.zeros(288, np.float32), synth.randn(44, 2, (256,)) * 0.1]).astype(np.float32)
        pos = synth.randn(44, 3, (S, 288)) * 0.5
        pk = dict(post=(ops.mark_static(_dev(Wp)), _dev(bp), _dev(pos)))
    out = ops.ffn_fused(_dev(samp), *wargs, pre=pre, want_xn=True, **pk, **kw)
    y, xn = out[0], out[1]
    np.testing.assert_allclose(xn.cpu().numpy(), refn, rtol=0, atol=5e-6 * np.abs(refn).max())
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=0, atol=2e-5 * np.abs(ref).max())
    if post:
        refp = ref @ Wp.astype(np.float64).T + bp
        refp[:, :288] += pos[np.arange(M) % S]
        np.testing.assert_allclose(out[2].cpu().numpy(), refp, rtol=0, atol=2e-5 * np.abs(refp).max())
    # the launches it replaces: the projection GEMM with dropout1 + residual in its epilogue, then the launch without the phase
    x1d = ops.gemm_nt(_dev(samp), Wod, bias=_dev(bo), res=_dev(res), dropout=(p, seed, 0) if p > 0 else None)
    two = ops.ffn_fused(x1d, *wargs, **pk, **kw)
    two_y = two[0] if post else two
    assert torch.allclose(y, two_y, rtol=0, atol=5e-6 * float(two_y.abs().max()))
    again = ops.ffn_fused(_dev(samp), *wargs, pre=pre, want_xn=True, **pk, **kw)
    assert all(torch.equal(u, v) for u, v in zip(out, again))              # run to run: same bits


def test_encoder_forward_with_and_without_fused_projection():
    """the pixel decoder's encoder stack, forward-only path: fused FFN launches that carry the next layer's projection against the
    stack with every projection as its own launch and against the fully unfused stack (same dropout masks: same seeds)"""
    from s2d_amd import ops
    from s2d_amd.modeling.pixel_decoder import MSDeformAttnTransformerEncoderLayer
    torch.manual_seed(3)
    layers = [MSDeformAttnTransformerEncoderLayer(dropout=0.3).to(DEV).train() for _ in range(3)]
    for l in layers:
        for q in l.parameters():
            if q.dim() > 1:
                torch.nn.init.xavier_uniform_(q)
        torch.nn.init.normal_(l.self_attn.sampling_offsets.weight, std=0.02)
    shapes = [(6, 10), (12, 20), (24, 40)]
    S = sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(4)
    src0 = torch.randn((2, S, 256), generator=g).to(DEV)
    pos = torch.randn((S, 256), generator=g).to(DEV)
    shp = torch.tensor(shapes, dtype=torch.int64)

    def run(fuse_ffn, fuse_next, fuse_pre=False):
        ops._DROP_CALLS[0] = 1000                      # same Philox keys in every arrangement
        src, both = src0, None
        for i, l in enumerate(layers):
            l.fuse_ffn, l.fuse_next, l.fuse_pre = fuse_ffn, fuse_next, fuse_pre
            nxt = layers[i + 1].self_attn if (fuse_ffn and fuse_next and i + 1 < len(layers)) else None
            out = l(src, pos, shp, None, both=both, nxt=nxt)
            src, both = out if nxt is not None else (out, None)
        return src

    a, b, c = run(True, True), run(True, False), run(False, False)
    sc = float(c.abs().max())
    assert float((a - b).abs().max()) < 5e-6 * sc and float((a - c).abs().max()) < 2e-5 * sc
    d, e = run(True, True, True), run(True, False, True)          # ... and with the attention's output projection in the launch as well
    assert float((d - a).abs().max()) < 1e-5 * sc and float((e - a).abs().max()) < 1e-5 * sc
    for l in layers:
        l.fuse_ffn = l.fuse_next = l.fuse_pre = True


def test_ffn_fused_rejects_unsupported_sizes():
    from s2d_amd._lib import lib
    assert lib().call("s2d_ffn_pack_words", 128, 1024, 0, 0) == -1
    assert lib().call("s2d_ffn_pack_words", 256, 1000, 0, 0) == -1
    assert lib().call("s2d_ffn_pack_words", 256, 4096, 0, 0) == -1
    assert lib().call("s2d_ffn_pack_words", 256, 1024, 100, 0) == -1
    assert lib().call("s2d_ffn_pack_words", 256, 1024, 0, 2) == -1
    assert lib().call("s2d_ffn_pack_words", 256, 1024, 0, 0) == 32 * 16384
    assert lib().call("s2d_ffn_pack_words", 256, 1024, 544, 0) == 32 * 16384 + 17 * 8192
    assert lib().call("s2d_ffn_pack_words", 256, 1024, 544, 1) == 32 * 16384 + 25 * 8192

"""
oracle/oracle_np.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy (+ a little plain C, oracle/s2d_oracle.c) restatement of the S2D hot path:
the per-clip R50 Mask2Former-Video forward and the VideoHungarianMatcher /
VideoSetCriterion distillation loss, plus the keymask propagation steps.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; nothing under s2d_amd/ does.

Parity status: every stage below except the R50 backbone and the CoTracker
correlation (K1) is PINNED against golden vectors produced by the reference's
own Python (tests/golden/make_golden.py; checked in tests/test_oracle.py).
R50 lives in detectron2 (unpinned git HEAD, model_training/requirements.txt:1,
not under /root/reference) and K1 in co-tracker (requirements.txt:2): for those
two the oracle restates the published architecture/algorithm and is
"parity unpinned"; see DESIGN.md.

Layouts follow the reference (NCHW feature maps, [N,S,C] token lists).
Paths cited are relative to /root/reference/model_training unless noted.
"""
import ctypes
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libs2d_oracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = ctypes.CDLL(path)
        _LIB.orc_point_mask_iou.restype = ctypes.c_double
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# --------------------------------------------------------------------------- basic layers
# AMP: restate the modules torch.autocast runs in fp16 under the reference trainer (engine/train_loop.py:709 `with autocast():`,
# SOLVER.AMP.ENABLED True) with both operands of every linear / convolution / einsum rounded to fp16 (nearest-even, as `.half()`)
# and f32 accumulation -- the R50 trunk, the video decoder's linear layers, the mask-logit einsum.  The pixel decoder and the
# matcher run in fp32 in the reference (msdeformattn.py:314, matcher.py:266-268).  `AMP = True` arms it; resnet50() and
# video_decoder() switch the rounding on for their own extent.
AMP = False
_AMP_ON = [False]


def _r16(a):
    return a.astype(np.float16).astype(np.float32) if _AMP_ON[0] else a


class _amp_scope:
    def __enter__(self):
        self.prev = _AMP_ON[0]
        _AMP_ON[0] = bool(AMP)

    def __exit__(self, *a):
        _AMP_ON[0] = self.prev


def linear(x, w, b=None):
    y = _r16(x) @ _r16(w).T
    return y if b is None else y + b


def relu(x):
    return np.maximum(x, 0)


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * w + b


def group_norm(x, G, w, b, eps=1e-5):
    N, C, H, W = x.shape
    xg = x.reshape(N, G, -1).astype(np.float64)
    mu = xg.mean(-1, keepdims=True)
    var = xg.var(-1, keepdims=True)
    y = ((xg - mu) / np.sqrt(var + eps)).reshape(N, C, H, W).astype(x.dtype)
    return y * w[None, :, None, None] + b[None, :, None, None]


def conv2d(x, w, b=None, stride=1, pad=0):
    """x [N,C,H,W], w [O,C,kh,kw] -> [N,O,Ho,Wo]; im2col + one matmul."""
    N, C, H, W = x.shape
    O, _, kh, kw = w.shape
    dt_in = x.dtype
    x, w = _r16(x), _r16(w)
    if pad:
        x = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    if kh == 1 and kw == 1:
        xs = x[:, :, ::stride, ::stride][:, :, :Ho, :Wo]
        y = np.einsum("nchw,oc->nohw", xs, w[:, :, 0, 0], optimize=True)
    else:
        s = x.strides
        win = np.lib.stride_tricks.as_strided(
            x, (N, C, kh, kw, Ho, Wo), (s[0], s[1], s[2], s[3], s[2] * stride, s[3] * stride), writeable=False)
        cols = win.reshape(N, C * kh * kw, Ho * Wo)
        y = (w.reshape(O, -1) @ cols).reshape(N, O, Ho, Wo)
    if b is not None:
        y = y + b[None, :, None, None]
    return y.astype(dt_in)


def max_pool_3x3_s2_p1(x):
    N, C, H, W = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)), constant_values=-np.inf)
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    s = xp.strides
    win = np.lib.stride_tricks.as_strided(xp, (N, C, Ho, Wo, 3, 3), (s[0], s[1], s[2] * 2, s[3] * 2, s[2], s[3]),
                                          writeable=False)
    return win.max(axis=(4, 5))


def resize_bilinear(x, OH, OW):
    """F.interpolate(x, size=(OH,OW), mode='bilinear', align_corners=False); x [...,H,W] float32."""
    lead = x.shape[:-2]
    H, W = x.shape[-2:]
    xi = _c(x.reshape(-1, H, W), np.float32)
    out = np.empty((xi.shape[0], OH, OW), np.float32)
    lib().orc_resize_bilinear_f32(_p(xi), xi.shape[0], H, W, OH, OW, _p(out))
    return out.reshape(lead + (OH, OW))


def point_sample(inp, coords):
    """point_features.py:19-42. inp [R,C,H,W] (float32 or uint8), coords [R,P,2] -> [R,C,P] float32."""
    R, C, H, W = inp.shape
    P = coords.shape[1]
    co = _c(coords, np.float32)
    out = np.empty((R, C, P), np.float32)
    if inp.dtype == np.uint8:
        a = _c(inp, np.uint8)
        lib().orc_point_sample_u8(_p(a), _p(co), R, C, H, W, P, _p(out))
    else:
        a = _c(inp, np.float32)
        lib().orc_point_sample_f32(_p(a), _p(co), R, C, H, W, P, _p(out))
    return out


# --------------------------------------------------------------------------- positional encodings
def pe_sine_2d(H, W, num_pos_feats=128, temperature=10000.0):
    """mask2former/modeling/transformer_decoder/position_encoding.py:29-52 (normalize=True, no mask).
    Returns [2*num_pos_feats, H, W] float32."""
    f32 = np.float32
    scale = f32(2 * math.pi)
    y_embed = np.arange(1, H + 1, dtype=f32)[:, None].repeat(W, 1)
    x_embed = np.arange(1, W + 1, dtype=f32)[None, :].repeat(H, 0)
    eps = f32(1e-6)
    y_embed = y_embed / (y_embed[-1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, -1:] + eps) * scale
    dim_t = np.arange(num_pos_feats, dtype=f32)
    dim_t = (f32(temperature) ** (2 * np.floor(dim_t / 2) / f32(num_pos_feats))).astype(f32)
    px = x_embed[:, :, None] / dim_t
    py = y_embed[:, :, None] / dim_t
    px = np.stack((np.sin(px[:, :, 0::2]), np.cos(px[:, :, 1::2])), axis=3).reshape(H, W, -1)
    py = np.stack((np.sin(py[:, :, 0::2]), np.cos(py[:, :, 1::2])), axis=3).reshape(H, W, -1)
    return np.concatenate((py, px), axis=2).transpose(2, 0, 1).astype(f32)


def pe_sine_3d(T, H, W, num_pos_feats=128, temperature=10000.0):
    """mask2former_video/modeling/transformer_decoder/position_encoding.py:29-57. -> [T, 2F, H, W]."""
    f32 = np.float32
    scale = f32(2 * math.pi)
    eps = f32(1e-6)
    z = np.arange(1, T + 1, dtype=f32)[:, None, None] * np.ones((1, H, W), f32)
    y = np.arange(1, H + 1, dtype=f32)[None, :, None] * np.ones((T, 1, W), f32)
    x = np.arange(1, W + 1, dtype=f32)[None, None, :] * np.ones((T, H, 1), f32)
    z = z / (z[-1:, :, :] + eps) * scale
    y = y / (y[:, -1:, :] + eps) * scale
    x = x / (x[:, :, -1:] + eps) * scale
    dim_t = np.arange(num_pos_feats, dtype=f32)
    dim_t = (f32(temperature) ** (2 * np.floor(dim_t / 2) / f32(num_pos_feats))).astype(f32)
    dim_tz = np.arange(num_pos_feats * 2, dtype=f32)
    dim_tz = (f32(temperature) ** (2 * np.floor(dim_tz / 2) / f32(num_pos_feats * 2))).astype(f32)
    px = x[..., None] / dim_t
    py = y[..., None] / dim_t
    pz = z[..., None] / dim_tz
    sc = lambda p: np.stack((np.sin(p[..., 0::2]), np.cos(p[..., 1::2])), axis=4).reshape(T, H, W, -1)
    pos = np.concatenate((sc(py), sc(px)), axis=3) + sc(pz)
    return pos.transpose(0, 3, 1, 2).astype(f32)


# --------------------------------------------------------------------------- MSDeformAttn
def msda_core(value, shapes, lsi, loc, w):
    """ops/functions/ms_deform_attn_func.py:52-72 (== cuda/ms_deform_im2col_cuda.cuh:242-304)."""
    N, S, M, D = value.shape
    Lq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    dt = np.float64 if value.dtype == np.float64 else np.float32
    v, lo, ww = _c(value, dt), _c(loc, dt), _c(w, dt)
    sh, ls = _c(shapes, np.int64), _c(lsi, np.int64)
    out = np.empty((N, Lq, M * D), dt)
    fn = lib().orc_msda_forward_f64 if dt == np.float64 else lib().orc_msda_forward_f32
    fn(_p(v), _p(sh), _p(ls), _p(lo), _p(ww), N, S, M, D, L, Lq, P, _p(out))
    return out


def msda_core_backward(value, shapes, lsi, loc, w, grad_out):
    N, S, M, D = value.shape
    Lq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    v, lo, ww, go = (_c(a, np.float32) for a in (value, loc, w, grad_out))
    sh, ls = _c(shapes, np.int64), _c(lsi, np.int64)
    gv, gl, gw = np.zeros_like(v), np.zeros_like(lo), np.zeros_like(ww)
    lib().orc_msda_backward_f32(_p(v), _p(sh), _p(ls), _p(lo), _p(ww), _p(go), N, S, M, D, L, Lq, P,
                                _p(gv), _p(gl), _p(gw))
    return gv, gl, gw


def level_start_index(shapes):
    sizes = [int(h) * int(w) for h, w in shapes]
    return np.array([0] + list(np.cumsum(sizes)[:-1]), np.int64)


def reference_points(shapes):
    """msdeformattn.py:141-153 with valid_ratios == 1 -> [S, L, 2] (x,y); each query's own centre for all levels."""
    pts = []
    for (H, W) in shapes:
        ry = (np.linspace(0.5, H - 0.5, H, dtype=np.float32) / np.float32(H))[:, None].repeat(W, 1)
        rx = (np.linspace(0.5, W - 0.5, W, dtype=np.float32) / np.float32(W))[None, :].repeat(H, 0)
        pts.append(np.stack((rx.reshape(-1), ry.reshape(-1)), -1))
    ref = np.concatenate(pts, 0)
    return np.repeat(ref[:, None, :], len(shapes), axis=1)


def msda_module(p, pre, query, ref_pts, src, shapes, M=8, Pn=4):
    """ops/modules/ms_deform_attn.py:82-125. query/src [N,S,C]; ref_pts [S,L,2] or [N,S,L,2]."""
    N, S, C = src.shape
    L = len(shapes)
    D = C // M
    value = linear(src, p[pre + "value_proj.weight"], p[pre + "value_proj.bias"]).reshape(N, S, M, D)
    off = linear(query, p[pre + "sampling_offsets.weight"], p[pre + "sampling_offsets.bias"]).reshape(N, -1, M, L, Pn, 2)
    aw = linear(query, p[pre + "attention_weights.weight"], p[pre + "attention_weights.bias"]).reshape(N, -1, M, L * Pn)
    aw = softmax(aw, -1).reshape(N, -1, M, L, Pn)
    norm = np.array([[w_, h_] for (h_, w_) in shapes], np.float32)  # (W_l, H_l)  :106
    rp = ref_pts if ref_pts.ndim == 4 else ref_pts[None]
    loc = rp[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    out = msda_core(value.astype(np.float32), np.array(shapes), level_start_index(shapes),
                    loc.astype(np.float32), aw.astype(np.float32))
    return linear(out, p[pre + "output_proj.weight"], p[pre + "output_proj.bias"])


def encoder_layer(p, pre, src, pos, ref_pts, shapes):
    """msdeformattn.py:116-131 (post-norm; dropout = 0 in parity runs)."""
    src2 = msda_module(p, pre + "self_attn.", src + pos, ref_pts, src, shapes)
    src = layer_norm(src + src2, p[pre + "norm1.weight"], p[pre + "norm1.bias"])
    ff = linear(relu(linear(src, p[pre + "linear1.weight"], p[pre + "linear1.bias"])),
                p[pre + "linear2.weight"], p[pre + "linear2.bias"])
    return layer_norm(src + ff, p[pre + "norm2.weight"], p[pre + "norm2.bias"])


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123
    reference implementation's constants).  Pinned by Random123's published known-answer vectors (tests/test_oracle.py).
    Arguments: uint32 arrays (broadcastable); returns four uint32 arrays."""
    c = [np.asarray(x, np.uint64) for x in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = np.uint64(k0), np.uint64(k1)
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = c
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return tuple(x.astype(np.uint32) for x in (c0, c1, c2, c3))


def dropout_thresh(p):
    """T = round(p * 256) clamped to [0, 255]: an element is dropped iff its 8 random bits < T (s2d_amd/csrc/dropout.h)"""
    return max(0, min(int(np.float32(p) * np.float32(256.0) + np.float32(0.5)), 255))


def dropout_scale(p):
    """256 / (256 - T) = 1 / P(keep): the multiplier of a kept element (float32, as the kernels form it)"""
    return np.float32(256.0) / np.float32(256 - dropout_thresh(p))


def dropout_multipliers(M, N, p, seed, site):
    """the [M, N] float32 multipliers (0 or 256 / (256 - T)) of the library's counter-based dropout (s2d_amd/csrc/dropout.h):
    block (row, col // 16) draws Philox4x32-10(counter = (row, col // 16, site, 0), key = seed); element col % 16 = e takes
    byte (e & 3) of word (e >> 2) and is kept iff those 8 bits >= T = round(p * 256).  Semantics of the sites: nn.Dropout
    at mask2former/modeling/pixel_decoder/msdeformattn.py:101-125 (x * mask / P(keep)) with p quantised to T / 256; the mask
    stream itself is this library's own definition (torch's is an implementation detail of its CUDA kernels), so dropout
    parity is distributional plus exact agreement with this restatement."""
    assert N % 8 == 0
    thresh = dropout_thresh(p)
    if thresh == 0:
        return np.ones((M, N), np.float32)
    nb = (N + 15) // 16
    rows = np.arange(M, dtype=np.uint32)[:, None]
    cb = np.arange(nb, dtype=np.uint32)[None, :]
    r = philox4x32_10(rows, cb, np.uint32(site), np.uint32(0), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    bits = np.empty((M, nb, 16), np.uint32)
    for e in range(16):
        bits[..., e] = (r[e >> 2] >> np.uint32(8 * (e & 3))) & np.uint32(0xFF)
    return np.where(bits.reshape(M, nb * 16)[:, :N] >= thresh, dropout_scale(p), np.float32(0.0)).astype(np.float32)


def pixel_decoder(p, feats, pre="", n_layers=6):
    """MSDeformAttnPixelDecoder.forward_features, msdeformattn.py:314-358.
    feats: dict res2..res5 [BT,C,h,w].  Returns mask_features [BT,256,h2,w2], [ms0, ms1, ms2]."""
    srcs, poss, shapes = [], [], []
    for idx, f in enumerate(("res5", "res4", "res3")):
        x = feats[f].astype(np.float32)
        y = conv2d(x, p[f"{pre}input_proj.{idx}.0.weight"], p[f"{pre}input_proj.{idx}.0.bias"])
        y = group_norm(y, 32, p[f"{pre}input_proj.{idx}.1.weight"], p[f"{pre}input_proj.{idx}.1.bias"])
        BT, C, h, w = y.shape
        shapes.append((h, w))
        srcs.append(y.reshape(BT, C, h * w).transpose(0, 2, 1))
        pe = pe_sine_2d(h, w).reshape(C, h * w).T
        poss.append(pe[None] + p[f"{pre}transformer.level_embed"][idx][None, None, :])  # :75
    src = np.concatenate(srcs, 1)
    pos = np.concatenate(poss, 1)
    ref = reference_points(shapes)
    out = src
    for l in range(n_layers):
        out = encoder_layer(p, f"{pre}transformer.encoder.layers.{l}.", out, pos, ref, shapes)
    lsi = level_start_index(shapes)
    outs = []
    for i, (h, w) in enumerate(shapes):
        z = out[:, lsi[i]:lsi[i] + h * w]
        outs.append(z.transpose(0, 2, 1).reshape(z.shape[0], -1, h, w))
    # one extra FPN level on res2 (:343-351)
    x = feats["res2"].astype(np.float32)
    cur = conv2d(x, p[f"{pre}adapter_1.weight"])
    cur = group_norm(cur, 32, p[f"{pre}adapter_1.norm.weight"], p[f"{pre}adapter_1.norm.bias"])
    y = cur + resize_bilinear(outs[-1], cur.shape[2], cur.shape[3])
    y = conv2d(y, p[f"{pre}layer_1.weight"], None, 1, 1)
    y = relu(group_norm(y, 32, p[f"{pre}layer_1.norm.weight"], p[f"{pre}layer_1.norm.bias"]))
    outs.append(y)
    mf = conv2d(outs[-1], p[f"{pre}mask_features.weight"], p[f"{pre}mask_features.bias"])
    return mf, outs[:3]


# --------------------------------------------------------------------------- video decoder
def mha(p, pre, q, k, v, mask=None, H=8):
    """nn.MultiheadAttention forward (seq-first): q [Lq,B,C], k/v [Lk,B,C]; mask bool [B*H,Lq,Lk] True=masked."""
    C = q.shape[-1]
    W, bi = p[pre + "in_proj_weight"], p[pre + "in_proj_bias"]
    qq = linear(q, W[:C], bi[:C])
    kk = linear(k, W[C:2 * C], bi[C:2 * C])
    vv = linear(v, W[2 * C:], bi[2 * C:])
    Lq, B, _ = qq.shape
    Lk = kk.shape[0]
    d = C // H
    qh = qq.reshape(Lq, B * H, d).transpose(1, 0, 2)
    kh = kk.reshape(Lk, B * H, d).transpose(1, 0, 2)
    vh = vv.reshape(Lk, B * H, d).transpose(1, 0, 2)
    s = (qh * np.float32(1.0 / math.sqrt(d))) @ kh.transpose(0, 2, 1)
    if mask is not None:
        s = np.where(mask, -np.inf, s)
    a = softmax(s, -1)
    o = (a @ vh).transpose(1, 0, 2).reshape(Lq, B, C)
    return linear(o, p[pre + "out_proj.weight"], p[pre + "out_proj.bias"])


def mlp3(p, pre, x):
    x = relu(linear(x, p[pre + "layers.0.weight"], p[pre + "layers.0.bias"]))
    x = relu(linear(x, p[pre + "layers.1.weight"], p[pre + "layers.1.bias"]))
    return linear(x, p[pre + "layers.2.weight"], p[pre + "layers.2.bias"])


def prediction_heads(p, pre, output, mask_features, target_hw, nheads=8):
    """video_mask2former_transformer_decoder.py:448-467. output [Q,B,C]; mask_features [B,T,C,h,w]."""
    d = layer_norm(output, p[pre + "decoder_norm.weight"], p[pre + "decoder_norm.bias"]).transpose(1, 0, 2)
    cls = linear(d, p[pre + "class_embed.weight"], p[pre + "class_embed.bias"])
    emb = mlp3(p, pre + "mask_embed.", d)
    masks = np.einsum("bqc,btchw->bqthw", _r16(emb), _r16(mask_features), optimize=True).astype(np.float32)
    B, Q, T = masks.shape[:3]
    rs = resize_bilinear(masks, target_hw[0], target_hw[1])
    # sigmoid(x) < 0.5  <=>  x < 0 (sigmoid is monotone, sigmoid(0) == 0.5 exactly)
    am = (rs < 0).reshape(B, Q, T * target_hw[0] * target_hw[1])
    am = np.repeat(am[:, None], nheads, axis=1).reshape(B * nheads, Q, -1)
    return cls, masks, am


def video_decoder(p, ms_feats, mask_features, T, pre="", n_layers=9, nheads=8):
    """VideoMultiScaleMaskedTransformerDecoder.forward, video_mask2former_transformer_decoder.py:374-446
    (training mode: bs = BT // T).  Returns logits [n_layers+1,B,Q,K+1], masks [n_layers+1,B,Q,T,h,w]."""
    with _amp_scope():
        return _video_decoder(p, ms_feats, mask_features, T, pre, n_layers, nheads)


def _video_decoder(p, ms_feats, mask_features, T, pre="", n_layers=9, nheads=8):
    BT, C, hm, wm = mask_features.shape
    B = BT // T
    mf = mask_features.reshape(B, T, C, hm, wm)
    src, pos, sizes = [], [], []
    for i in range(3):
        x = ms_feats[i]
        h, w = x.shape[-2:]
        sizes.append((h, w))
        pe = pe_sine_3d(T, h, w)                                   # [T,C,h,w]
        pe = np.broadcast_to(pe[None], (B, T, C, h, w)).reshape(B, T, C, h * w)
        s = x.reshape(BT, C, h * w) + p[pre + "level_embed.weight"][i][None, :, None]
        s = s.reshape(B, T, C, h * w)
        pos.append(pe.transpose(1, 3, 0, 2).reshape(T * h * w, B, C))  # (T*hw) x B x C, t-major (:394-397)
        src.append(s.transpose(1, 3, 0, 2).reshape(T * h * w, B, C))
    qe = np.repeat(p[pre + "query_embed.weight"][:, None, :], B, axis=1)
    out = np.repeat(p[pre + "query_feat.weight"][:, None, :], B, axis=1)
    logits, masks = [], []
    c, m, am = prediction_heads(p, pre, out, mf, sizes[0], nheads)
    logits.append(c); masks.append(m)
    for i in range(n_layers):
        lvl = i % 3
        am = am.copy()
        am[am.sum(-1) == am.shape[-1]] = False                      # :413
        ca = f"{pre}transformer_cross_attention_layers.{i}."
        t2 = mha(p, ca + "multihead_attn.", out + qe, src[lvl] + pos[lvl], src[lvl], am, nheads)
        out = layer_norm(out + t2, p[ca + "norm.weight"], p[ca + "norm.bias"])
        sa = f"{pre}transformer_self_attention_layers.{i}."
        t2 = mha(p, sa + "self_attn.", out + qe, out + qe, out, None, nheads)
        out = layer_norm(out + t2, p[sa + "norm.weight"], p[sa + "norm.bias"])
        ff = f"{pre}transformer_ffn_layers.{i}."
        t2 = linear(relu(linear(out, p[ff + "linear1.weight"], p[ff + "linear1.bias"])),
                    p[ff + "linear2.weight"], p[ff + "linear2.bias"])
        out = layer_norm(out + t2, p[ff + "norm.weight"], p[ff + "norm.bias"])
        c, m, am = prediction_heads(p, pre, out, mf, sizes[(i + 1) % 3], nheads)
        logits.append(c); masks.append(m)
    return np.stack(logits), np.stack(masks)


# --------------------------------------------------------------------------- R50 backbone (parity unpinned)
def frozen_bn(x, p, pre, eps=1e-5):
    scale = p[pre + "weight"] / np.sqrt(p[pre + "running_var"] + eps)
    shift = p[pre + "bias"] - p[pre + "running_mean"] * scale
    return x * scale[None, :, None, None] + shift[None, :, None, None]


R50_STAGES = (("res2", 3, 64, 256, 1), ("res3", 4, 128, 512, 2), ("res4", 6, 256, 1024, 2), ("res5", 3, 512, 2048, 2))


def resnet50(p, x, pre=""):
    """detectron2 build_resnet_backbone, R-50, STRIDE_IN_1X1 False, FrozenBN (d2, not in the reference tree;
    call sites kd_video_maskformer_model.py:132,135; cfg configs/imagenet_video/Base-YouTubeVIS-...yaml:2-16).
    x [N,3,H,W] normalised.  Returns dict res2..res5."""
    with _amp_scope():
        return _resnet50(p, x, pre)


def _resnet50(p, x, pre=""):
    y = relu(frozen_bn(conv2d(x, p[pre + "stem.conv1.weight"], None, 2, 3), p, pre + "stem.conv1.norm."))
    y = max_pool_3x3_s2_p1(y)
    outs = {}
    for name, nblk, mid, outc, stride in R50_STAGES:
        for b in range(nblk):
            bp = f"{pre}{name}.{b}."
            s = stride if b == 0 else 1
            if (bp + "shortcut.weight") in p:
                sc = frozen_bn(conv2d(y, p[bp + "shortcut.weight"], None, s, 0), p, bp + "shortcut.norm.")
            else:
                sc = y
            z = relu(frozen_bn(conv2d(y, p[bp + "conv1.weight"], None, 1, 0), p, bp + "conv1.norm."))
            z = relu(frozen_bn(conv2d(z, p[bp + "conv2.weight"], None, s, 1), p, bp + "conv2.norm."))
            z = frozen_bn(conv2d(z, p[bp + "conv3.weight"], None, 1, 0), p, bp + "conv3.norm.")
            y = relu(z + sc)
        outs[name] = y
    return outs


def r50_param_shapes(pre=""):
    """Parameter names/shapes of detectron2's R-50 (public checkpoint layout)."""
    out = [(pre + "stem.conv1.weight", (64, 3, 7, 7))]
    out += [(pre + f"stem.conv1.norm.{k}", (64,)) for k in ("weight", "bias", "running_mean", "running_var")]
    inc = 64
    for name, nblk, mid, outc, stride in R50_STAGES:
        for b in range(nblk):
            bp = f"{pre}{name}.{b}."
            if b == 0:
                out.append((bp + "shortcut.weight", (outc, inc, 1, 1)))
                out += [(bp + f"shortcut.norm.{k}", (outc,)) for k in ("weight", "bias", "running_mean", "running_var")]
            for cn, shp in (("conv1", (mid, inc, 1, 1)), ("conv2", (mid, mid, 3, 3)), ("conv3", (outc, mid, 1, 1))):
                out.append((bp + cn + ".weight", shp))
                out += [(bp + f"{cn}.norm.{k}", (shp[0],)) for k in ("weight", "bias", "running_mean", "running_var")]
            inc = outc
    return out


# --------------------------------------------------------------------------- input / targets
PIXEL_MEAN = np.array([123.675, 116.28, 103.53], np.float32)
PIXEL_STD = np.array([58.395, 57.12, 57.375], np.float32)


def normalize_pad(frames_u8, div=32):
    """kd_video_maskformer_model.py:263-269 + ImageList.from_tensors (zero pad bottom/right to /div)."""
    N, _, H, W = frames_u8.shape
    Hp, Wp = (H + div - 1) // div * div, (W + div - 1) // div * div
    x = (frames_u8.astype(np.float32) - PIXEL_MEAN[None, :, None, None]) / PIXEL_STD[None, :, None, None]
    out = np.zeros((N, 3, Hp, Wp), np.float32)
    out[:, :, :H, :W] = x
    return out


def prepare_targets(masks_u8, ids, Hp, Wp):
    """kd_video_maskformer_model.py:358-386 for one clip. masks [N,T,H0,W0] u8, ids [N,T] (-1 absent).
    Returns (masks [N',T,Hp,Wp] u8, ids [N',T], labels [N'])."""
    N, T, H0, W0 = masks_u8.shape
    out = np.zeros((N, T, Hp, Wp), np.uint8)
    out[:, :, :H0, :W0] = masks_u8
    valid = (ids != -1).any(-1)
    return out[valid], ids[valid], np.zeros(int(valid.sum()), np.int64)


def kd_targets(t_logits, t_masks, Hp, Wp, K=100, thr=0.75):
    """prepare_distillation_targets, kd_video_maskformer_model.py:436-526, one clip, nms off.
    t_logits [Q,2], t_masks [Q,T,h,w].  Kept queries are returned in ascending query order (the reference's
    topk(sorted=False) order is implementation-defined; every downstream quantity is order-invariant up to
    the permutation of target indices).  Returns (masks [K',T,Hp,Wp] u8, kept query ids)."""
    sc = softmax(t_logits.astype(np.float32), -1)[:, 0]
    Q = sc.shape[0]
    k = min(K, Q)
    order = np.argsort(-sc, kind="stable")[:k]
    kept = np.sort(order[sc[order] >= np.float32(thr)])
    up = resize_bilinear(t_masks[kept].astype(np.float32), Hp, Wp)
    return (up > 0).astype(np.uint8), kept


def mask_nms(masks, labels, thr):
    """greedy same-label mask-NMS in score order, kd_video_maskformer_model.py:552-583.  masks [K,...] bool sorted by
    score; IoU is formed in float32 from integer counts as the reference does (.float() / .float()), union 0 -> IoU 0."""
    flat = masks.reshape(masks.shape[0], -1)
    idx = list(range(flat.shape[0]))
    keep = []
    while idx:
        cur = idx.pop(0)
        keep.append(cur)
        rem = []
        for o in idx:
            if labels[o] != labels[cur]:
                rem.append(o)
                continue
            inter = np.float32(np.count_nonzero(flat[cur] & flat[o]))
            union = np.float32(np.count_nonzero(flat[cur] | flat[o]))
            iou = inter / union if union > 0 else np.float32(0.0)
            if iou <= np.float32(thr):
                rem.append(o)
        idx = rem
    return keep


def inference_video(cls_logits, masks_lowres, Hp, Wp, ih, iw, oh, ow, K, use_nms=False, thr=0.75):
    """eval branch of KDVideoMaskFormer.forward (kd_video_maskformer_model.py:340-356) + inference_video (:530-610).
    cls_logits [Q,C+1], masks_lowres [Q,T,h,w] -> dict(scores [n] f32, labels [n] i64, masks [n,T,oh,ow] bool,
    query [n] i64 (the query each prediction came from), logits [K,T,oh,ow] f32 before NMS, keep list).
    Ties in the sorted top-k resolve to the lower flat index."""
    Q, C1 = cls_logits.shape
    C = C1 - 1
    sc = softmax(cls_logits.astype(np.float32), -1)[:, :-1].reshape(-1)
    order = np.argsort(-sc, kind="stable")[:K]
    scores, labels, qidx = sc[order], (order % C).astype(np.int64), order // C
    up = resize_bilinear(masks_lowres[qidx].astype(np.float32), Hp, Wp)[:, :, :ih, :iw]       # :341-346, :545
    lg = resize_bilinear(np.ascontiguousarray(up), oh, ow)                                    # :546-548
    m = lg > 0
    keep = mask_nms(m, labels, thr) if use_nms else list(range(len(order)))
    return dict(scores=scores[keep], labels=labels[keep], masks=m[keep], query=qidx[keep], logits=lg, keep=keep,
                all_scores=scores, all_labels=labels, all_query=qidx)


# --------------------------------------------------------------------------- matcher
def softplus(x):
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def matcher_cost(logits, masks, tgt_masks, coords, w_class, w_mask, w_dice):
    """VideoHungarianMatcher.memory_efficient_forward, matcher.py:236-287, one clip.
    logits [Q,K+1], masks [Q,T,h,w], tgt_masks [N,T,H,W] (u8 or float), coords [1,P,2] -> C [Q,N] float32."""
    Q = masks.shape[0]
    N = tgt_masks.shape[0]
    P = coords.shape[1]
    prob = softmax(logits.astype(np.float32), -1)
    cost_class = -np.repeat(prob[:, :1], N, axis=1)                                   # labels forced 0 (:238-243)
    tm = point_sample(tgt_masks, np.repeat(coords, N, 0)).reshape(N, -1) if N else np.zeros((0, masks.shape[1] * P), np.float32)
    om = point_sample(masks.astype(np.float32), np.repeat(coords, Q, 0)).reshape(Q, -1)
    hw = om.shape[1]
    pos, neg = softplus(-om), softplus(om)                                            # BCE vs ones / zeros (:51-56)
    cost_mask = (pos @ tm.T + neg @ (1 - tm).T) / np.float32(hw)
    sg = sigmoid(om)
    num = 2 * (sg @ tm.T)
    den = sg.sum(-1)[:, None] + tm.sum(-1)[None, :]
    cost_dice = 1 - (num + 1) / (den + 1)
    C = np.float32(w_mask) * cost_mask + np.float32(w_class) * cost_class + np.float32(w_dice) * cost_dice
    return C.astype(np.float32)


def lsap(C):
    """scipy.optimize.linear_sum_assignment as called at matcher.py:289 (C float32 -> double)."""
    C = _c(C, np.float64)
    nr, nc = C.shape
    k = min(nr, nc)
    a, b = np.empty(k, np.int64), np.empty(k, np.int64)
    rc = lib().orc_lsap_f64(_p(C), nr, nc, _p(a), _p(b))
    if rc != 0:
        raise ValueError("cost matrix is infeasible" if rc == -1 else "matrix contains invalid numeric entries")
    return a, b


def matcher(logits, masks, targets, coords_list, w_class, w_mask, w_dice):
    """list over clips of (idx_q, idx_t) int64."""
    out = []
    for b in range(len(targets)):
        C = matcher_cost(logits[b], masks[b], targets[b], coords_list[b], w_class, w_mask, w_dice)
        out.append(lsap(C))
    return out


# --------------------------------------------------------------------------- criterion
def loss_labels(logits, indices, eos_coef=0.1):
    """criterion.py:227-251. logits [B,Q,2]; matched -> class 0, rest -> class 1 (= num_classes)."""
    B, Q, K = logits.shape
    tgt = np.full((B, Q), K - 1, np.int64)
    for b, (qi, _) in enumerate(indices):
        tgt[b, qi] = 0
    wts = np.ones(K, np.float32); wts[-1] = eos_coef
    lg = logits.astype(np.float32)
    lse = np.log(np.exp(lg - lg.max(-1, keepdims=True)).sum(-1)) + lg.max(-1)
    nll = lse - np.take_along_axis(lg, tgt[..., None], -1)[..., 0]
    w = wts[tgt]
    return np.float32((nll * w).sum() / w.sum())


def loss_masks(masks, targets, indices, num_masks, coords_over=None, coords_rand=None, P=None, rng=None,
               drop=True):
    """criterion.py:292-356 + point_features.py:63-116. masks [B,Q,T,h,w]; targets list of [N,T,H,W] u8.
    coords_over [R,3P,2], coords_rand [R,P/4,2] are the injected torch.rand draws (R = kept rows)."""
    src = np.concatenate([masks[b][qi] for b, (qi, _) in enumerate(indices)], 0)
    tgt = np.concatenate([targets[b][tj] for b, (_, tj) in enumerate(indices)], 0)
    src = src.reshape((-1, 1) + src.shape[2:]).astype(np.float32)
    tgt = tgt.reshape((-1, 1) + tgt.shape[2:])
    if drop:                                                           # temporal DropLoss (:307-322)
        keep = np.array([i for i in range(tgt.shape[0]) if tgt[i].any()], np.int64)
        if keep.size == 0:
            return np.float32(0), np.float32(0)
        src, tgt = src[keep], tgt[keep]
    R = src.shape[0]
    if callable(coords_over):                                          # lazily drawn, like the reference
        coords_over = coords_over()
        coords_rand = coords_rand()
    if coords_over is not None and coords_over.shape[0] > R:         # padded injection buffers: first R slots
        coords_over, coords_rand = coords_over[:R], coords_rand[:R]
    if coords_over is None:
        coords_over = rng.random((R, 3 * P, 2), dtype=np.float32)
        coords_rand = rng.random((R, P - int(0.75 * P), 2), dtype=np.float32)
    n_unc = coords_over.shape[1] // 3 * 3 // 4 if P is None else int(0.75 * P)
    pl = point_sample(src, coords_over)[:, 0]                          # [R,3P]
    unc = -np.abs(pl)
    idx = np.argsort(-unc, axis=1, kind="stable")[:, :n_unc]           # topk(k) by value (:102)
    sel = np.take_along_axis(coords_over, idx[..., None], 1)
    coords = np.concatenate([sel, coords_rand], 1) if coords_rand.shape[1] > 0 else sel
    labels = point_sample(tgt, coords)[:, 0]
    lg = point_sample(src, coords)[:, 0]
    bce = np.maximum(lg, 0) - lg * labels + np.log1p(np.exp(-np.abs(lg)))
    loss_mask = bce.mean(1).sum() / num_masks
    sg = sigmoid(lg)
    num = 2 * (sg * labels).sum(-1)
    den = sg.sum(-1) + labels.sum(-1)
    loss_dice = (1 - (num + 1) / (den + 1)).sum() / num_masks
    return np.float32(loss_mask), np.float32(loss_dice)


def criterion(logits_all, masks_all, targets, rand_iter, P, weights=(2.0, 5.0, 5.0), world_size=1):
    """The intended call order of VideoSetCriterion.forward (criterion.py:390-427): matcher -> loss_labels ->
    loss_masks on the last layer, then per aux layer matcher -> loss_masks.  logits_all [NL,B,Q,2],
    masks_all [NL,B,Q,T,h,w]; targets list of [N,T,H,W] u8; rand_iter yields the torch.rand draws in order.
    Returns (dict of unweighted losses, list of indices per layer call)."""
    NL, B = logits_all.shape[:2]
    wc, wm, wd = weights
    num_masks = max(float(sum(t.shape[0] for t in targets)) / world_size, 1.0)
    losses, all_idx = {}, []

    def one(layer):
        coords = [next(rand_iter) for _ in range(B)]
        idx = matcher(logits_all[layer], masks_all[layer], targets, coords, wc, wm, wd)
        all_idx.append(idx)
        return idx

    def masks_loss(layer, idx):
        draw = lambda: next(rand_iter)
        return loss_masks(masks_all[layer], targets, idx, num_masks, draw, draw)

    idx = one(NL - 1)
    losses["loss_ce"] = loss_labels(logits_all[NL - 1], idx)
    losses["loss_mask"], losses["loss_dice"] = masks_loss(NL - 1, idx)
    for i in range(NL - 1):
        idx = one(i)
        lm, ld = masks_loss(i, idx)
        losses[f"loss_mask_{i}"], losses[f"loss_dice_{i}"] = lm, ld
    return losses, all_idx


def kd_forward_losses(s_logits, s_masks, gt_targets, kd_tgts, rand_gt, rand_kd, P, weight_dict,
                      matcher_weights=(2.0, 5.0, 5.0)):
    """kd_video_maskformer_model.py:300-326: GT criterion pass, KD criterion pass, rename, weight, drop."""
    losses, idx_gt = criterion(s_logits, s_masks, gt_targets, rand_gt, P, matcher_weights)
    dl, idx_kd = criterion(s_logits, s_masks, kd_tgts, rand_kd, P, matcher_weights)
    for k, v in dl.items():
        losses[k.replace("loss_", "kd_loss_")] = v
    out = {k: np.float32(v * weight_dict[k]) for k, v in losses.items() if k in weight_dict}
    return out, idx_gt, idx_kd


# --------------------------------------------------------------------------- keymask (paths rel. /root/reference/keymask_ident)
def color_masks_to_ids(frames):
    """load_masks, cotracker_matching.py:49-72 (after the PNG decode): frames u8 [T,H,W,3] RGB -> int64 [T,H,W,1]; per frame
    black 0, the other unique colours 1..n in sorted (R,G,B) order (tuple order == order of the packed 24-bit key)."""
    T, H, W, _ = frames.shape
    out = np.zeros((T, H, W, 1), np.int64)
    for t in range(T):
        key = (frames[t, ..., 0].astype(np.int64) << 16) | (frames[t, ..., 1].astype(np.int64) << 8) | frames[t, ..., 2]
        uniq = np.unique(key)
        uniq = uniq[uniq != 0]
        out[t, ..., 0] = np.where(key == 0, 0, np.searchsorted(uniq, key) + 1)
    return out


def tracks_to_masks(tracks, H, W):
    """pred_tracks_to_binary_masks(return_mask=False), cotracker_matching.py:453-503. tracks [T,Np,2] -> [T,H,W] u8."""
    tr = _c(tracks, np.float32)
    T, Np = tr.shape[:2]
    out = np.empty((T, H, W), np.uint8)
    lib().orc_tracks_to_masks(_p(tr), T, Np, H, W, _p(out))
    return out


def point_mask_iou(ids_frame, oid, pm):
    """get_segmentation_mask :176-209 + nearest resize :687-689 + compute_point_mask_intersection :640-662."""
    ids = _c(ids_frame, np.int64)
    pmc = _c(pm, np.uint8)
    return lib().orc_point_mask_iou(_p(ids), ids.shape[0], ids.shape[1], ctypes.c_int64(int(oid)), _p(pmc),
                                    pmc.shape[0], pmc.shape[1])


def extract_mask_matches(tracks, idmap, H, W, v_range, thr=0.5):
    """cotracker_matching.py:665-719.  tracks [T,Np,2] at (H,W) scale, idmap [T,Hi,Wi] int.
    Returns (matches, all_comparisons) as arrays of rows (frame_id, mask_id, iou)."""
    tm = tracks_to_masks(tracks, H, W)
    allc, matches = [], []
    for fid in range(v_range[0], v_range[1] + 1):
        oids = np.unique(idmap[fid])
        for oid in oids[1:]:  # the reference drops the smallest id, assumed background (:680-681)
            iou = point_mask_iou(idmap[fid], oid, tm[fid])
            allc.append((fid, int(oid), iou))
            if iou > thr:
                matches.append((fid, int(oid), iou))
    return np.array(matches, np.float64).reshape(-1, 3), np.array(allc, np.float64).reshape(-1, 3)


def visibility_curve(vis):
    """cotracker_occlusions.py:359: mean over points of pred_visibility.float(). vis [T,Np] bool -> [T]."""
    return vis.astype(np.float32).mean(1)


def local_correlation(fmap, coords, support, r=3):
    """K1, SELF-DEFINED (co-tracker is a third-party dependency absent from /root/reference, requirements.txt:2;
    parity unpinned): local 4-D correlation.  fmap [T,H,W,C], coords [T,Np,2] (x,y px), support [Np,S,C], S=(2r+1)^2.
    corr[t,n,i,j] = <bilinear(fmap[t], coords[t,n] + offset_i), support[n,j]>, zero padding."""
    T, H, W, C = fmap.shape
    Np = coords.shape[1]
    k = 2 * r + 1
    dy, dx = np.meshgrid(np.arange(-r, r + 1), np.arange(-r, r + 1), indexing="ij")
    out = np.zeros((T, Np, k * k, k * k), np.float32)
    fp = np.pad(fmap.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
    for t in range(T):
        x = coords[t, :, 0, None].astype(np.float32) + dx.reshape(1, -1).astype(np.float32)
        y = coords[t, :, 1, None].astype(np.float32) + dy.reshape(1, -1).astype(np.float32)
        x0, y0 = np.floor(x), np.floor(y)
        fx, fy = (x - x0).astype(np.float64), (y - y0).astype(np.float64)
        nb = np.zeros((Np, k * k, C))
        for ky in (0, 1):
            for kx in (0, 1):
                xx, yy = x0.astype(int) + kx, y0.astype(int) + ky
                ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
                w = (fx if kx else 1 - fx) * (fy if ky else 1 - fy)
                v = fp[t, np.clip(yy, -1, H) + 1, np.clip(xx, -1, W) + 1]
                nb += np.where(ok[..., None], v, 0.0) * w[..., None]
        out[t] = np.einsum("nic,njc->nij", nb, support.astype(np.float64))
    return out


# --------------------------------------------------------------------------- COCO RLE (pycocotools, third party: restated, unpinned)
def rle_encode(mask):
    """maskApi.c rleEncode + rleToString for one [H,W] mask (what mask_util.encode returns for np.array(mask[:, :, None],
    order="F"): ytvis_eval.py:347, keymask_ident/annotations.py:100).  Pure-python loops: small cases only."""
    H, W = mask.shape
    flat = (np.asarray(mask) != 0).T.reshape(-1)          # column-major
    cnts, p, c = [], False, 0
    for v in flat:
        if v != p:
            cnts.append(c)
            c, p = 0, v
        c += 1
    cnts.append(c)
    s = bytearray()
    for i, x in enumerate(cnts):
        x = int(x)
        if i > 2:
            x -= int(cnts[i - 2])
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            s.append(ch + 48)
    return {"size": [H, W], "counts": bytes(s)}, cnts


def rle_decode(rle):
    """rleFrString + rleDecode: the inverse, used for round-trip checks"""
    H, W = rle["size"]
    s = rle["counts"] if isinstance(rle["counts"], (bytes, bytearray)) else rle["counts"].encode()
    cnts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            ch = s[p] - 48
            x |= (ch & 0x1F) << (5 * k)
            more = bool(ch & 0x20)
            p += 1
            k += 1
            if not more and (ch & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    flat = np.zeros(H * W, np.uint8)
    pos, v = 0, 0
    for c in cnts:
        flat[pos:pos + c] = v
        pos += c
        v ^= 1
    return flat.reshape(W, H).T


def rle_area_bbox(mask):
    """mask_util.area / toBbox of a binary mask: set pixels; tight [x, y, w, h], zeros when empty"""
    m = np.asarray(mask) != 0
    if not m.any():
        return 0, [0.0, 0.0, 0.0, 0.0]
    ys, xs = np.nonzero(m)
    return int(m.sum()), [float(xs.min()), float(ys.min()), float(xs.max() - xs.min() + 1), float(ys.max() - ys.min() + 1)]


# --------------------------------------------------------------------------- data-side step
def aug_warp_frames(frames, params, out_hw):
    """restatement of s2d_aug_warp_frames_u8 (the single-pass composition of the mapper's augmentation list,
    data_video/augmentation.py:116-168 over detectron2 / fvcore transforms: parity unpinned, see csrc/augment.hip).
    frames u8 [T,3,H0,W0], params f32 [T,16] -> u8 [T,3,H1,W1].  float32 arithmetic in the kernel's operation order."""
    f32 = np.float32
    T, _, H0, W0 = frames.shape
    H1, W1 = out_hw
    out = np.zeros((T, 3, H1, W1), np.uint8)
    px = (np.arange(W1, dtype=f32) + f32(0.5))[None, :]
    py = (np.arange(H1, dtype=f32) + f32(0.5))[:, None]
    for t in range(T):
        a11, a12, a13, a21, a22, a23, cx, cy, cw, ch, bright, contrast, cmean = (f32(v) for v in params[t, :13])
        sx = a11 * px + a12 * py + a13
        sy = a21 * px + a22 * py + a23
        inside = (sx >= cx) & (sy >= cy) & (sx < cx + cw) & (sy < cy + ch)
        fx, fy = sx - f32(0.5), sy - f32(0.5)
        x0f, y0f = np.floor(fx), np.floor(fy)
        lx, ly = fx - x0f, fy - y0f
        xl, xh, yl, yh = int(cx), int(cx + cw) - 1, int(cy), int(cy + ch) - 1
        x0 = np.clip(x0f.astype(np.int64), xl, xh); x1 = np.clip(x0f.astype(np.int64) + 1, xl, xh)
        y0 = np.clip(y0f.astype(np.int64), yl, yh); y1 = np.clip(y0f.astype(np.int64) + 1, yl, yh)
        if cmean < 0:
            crop = frames[t, :, yl:yh + 1, xl:xh + 1].astype(np.float64)
            cmean = f32(f32(crop.sum() / crop.size) * bright)
        one = f32(1.0)
        for c in range(3):
            p = frames[t, c].astype(f32)
            v = (one - ly) * ((one - lx) * p[y0, x0] + lx * p[y0, x1]) + ly * ((one - lx) * p[y1, x0] + lx * p[y1, x1])
            q = np.rint(v).astype(f32)
            if bright != 1:
                q = np.trunc(np.clip(bright * q, f32(0), f32(255)))
            if contrast != 1:
                q = np.trunc(np.clip((one - contrast) * cmean + contrast * q, f32(0), f32(255)))
            out[t, c] = np.where(inside, q, 0).astype(np.uint8)
    return out


def aug_warp_masks(masks, params, out_hw):
    """masks u8 [N,T,H0,W0] -> u8 [N,T,H1,W1] (0/1): nearest source pixel under the same per-frame maps"""
    f32 = np.float32
    N, T, H0, W0 = masks.shape
    H1, W1 = out_hw
    out = np.zeros((N, T, H1, W1), np.uint8)
    px = (np.arange(W1, dtype=f32) + f32(0.5))[None, :]
    py = (np.arange(H1, dtype=f32) + f32(0.5))[:, None]
    for t in range(T):
        a11, a12, a13, a21, a22, a23, cx, cy, cw, ch = (f32(v) for v in params[t, :10])
        sx = a11 * px + a12 * py + a13
        sy = a21 * px + a22 * py + a23
        inside = (sx >= cx) & (sy >= cy) & (sx < cx + cw) & (sy < cy + ch)
        iy = np.clip(np.floor(sy).astype(np.int64), 0, H0 - 1); ix = np.clip(np.floor(sx).astype(np.int64), 0, W0 - 1)
        for n in range(N):
            out[n, t] = np.where(inside, masks[n, t][iy, ix] != 0, 0)
    return out

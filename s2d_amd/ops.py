"""Torch-tensor front ends of the C ABI (include/s2d_hip.h).  Tensors are plumbing only: device memory,
the current stream, and the caching allocator for outputs.  Every function launches hand-written HIP."""
import weakref

import torch

from ._lib import lib


PROFILE = None   # bench.py sets this to a list: (start_event, end_event, algorithmic flops) per dense launch


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _Ev:
    """HIP event without the default flags' system-scope fence (s2d_prof_event_*): cheap enough to bracket every dense launch"""
    __slots__ = ("h",)

    def __init__(self):
        self.h = lib().call("s2d_prof_event_create")
        if not self.h:
            raise RuntimeError("hipEventCreateWithFlags failed")

    def record(self):
        lib().call("s2d_prof_event_record", self.h, _stream())

    def elapsed_time(self, other):
        import ctypes
        ms = ctypes.c_double(0.0)
        lib().call("s2d_prof_event_elapsed", self.h, other.h, ctypes.addressof(ms))
        return ms.value

    def __del__(self):
        try:
            lib().call("s2d_prof_event_destroy", self.h)
        except Exception:
            pass


class _Timed:
    """HIP-event bracket around one dense-contraction launch on the current stream (bench.py roofline)."""

    def __init__(self, flops, tag=None):
        self.flops = flops
        self.tag = tag

    def __enter__(self):
        if PROFILE is not None:
            self.s, self.e = _Ev(), _Ev()
            self.s.record()

    def __exit__(self, *a):
        if PROFILE is not None:
            self.e.record()
            PROFILE.append((self.s, self.e, self.flops, self.tag))


def _chk(t, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda or not t.is_contiguous() or t.dtype != dtype:
        raise RuntimeError(f"s2d op needs a contiguous {dtype} CUDA tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")


_MODE = "f16x3"
_SPLIT = {}      # (data_ptr, shape, row stride) -> (split image, version of the owning tensor, the owning tensor)


def version_of(t):
    """content version of a tensor for this library's caches: torch's in-place counter of the owning tensor + the count of rewrites
    torch cannot see (the multi-tensor optimizer kernel updates parameters and EMA copies through raw pointers: bump_version)"""
    base = t._base if t._base is not None else t
    return base._version + getattr(base, "_s2d_version", 0)


def bump_version(t):
    """declare that t's storage was rewritten behind torch's back (a kernel of this library wrote through its raw pointer)"""
    base = t._base if t._base is not None else t
    base._s2d_version = getattr(base, "_s2d_version", 0) + 1


def repack(old, new):
    """A module's cached packed weight copy after its sources changed: when the previous copy has the same shape it is overwritten
    IN PLACE (torch's counter moves, so the caches hanging off its address -- split images, transposed / flipped copies -- refresh
    into their existing buffers); otherwise `new` becomes the copy.  A fresh tensor per optimizer step would leave a dead generation of
    every such cache behind per step."""
    if old is not None and old.shape == new.shape and old.device == new.device and old.dtype == new.dtype:
        old.copy_(new)
        return old
    new = new.contiguous()
    if new._base is not None:
        new = new.clone()
    return mark_static(new)


def mark_static(t):
    """Declare a tensor a static weight (a packed / concatenated copy of parameters that its module caches): dense launches
    reading it as the B operand may then use a cached pre-split fp16 image instead of splitting it in every launch."""
    t._s2d_static = True
    return t


def clear_weight_cache():
    _SPLIT.clear()
    _FFN_PACK.clear()


_SWEEP = [256]


def _sweep_split_cache():
    """drop the images whose owning tensor is gone (called when a new key enters; amortised)"""
    if len(_SPLIT) < _SWEEP[0]:
        return
    for k in [k for k, e in _SPLIT.items() if e[2]() is None]:
        del _SPLIT[k]
    _SWEEP[0] = max(256, 2 * len(_SPLIT))


def _static_split(B, N, K, ldb):
    """cached fp16 hi/lo image of a static weight matrix, or None (dynamic tensor / other dense mode)"""
    if _MODE == "f32" or N * ((K + 31) // 32) * 128 > 0xFFFFFF00:
        return None
    base = B._base if B._base is not None else B
    if not (isinstance(base, torch.nn.Parameter) or getattr(base, "_s2d_static", False) or getattr(B, "_s2d_static", False)):
        return None
    key = (B.data_ptr(), N, K, ldb, _MODE)      # the image is fp16 hi / scaled lo or bf16 hi / lo, by the mode in force
    ent = _SPLIT.get(key)
    ver = base._version + getattr(base, "_s2d_version", 0)     # _s2d_version: buffers this library rewrites in place (backward.py)
    if ent is None or ent[1] != ver or ent[2]() is not base:
        # a stale image of the same tensor is overwritten in place (same size): steady-state training allocates nothing here
        same = ent is not None and ent[2]() is base
        img = ent[0] if same else torch.empty((lib().call("s2d_split_weights_words", N, K),), device=B.device, dtype=torch.int32)
        lib().call("s2d_split_weights_f16", B, N, K, ldb, img, _stream())
        if not same:
            _sweep_split_cache()
        # a WEAK reference to the owner: a packed copy its module has replaced (every optimizer step re-packs) dies, and its image
        # with it at the next sweep; an address recycled under the same key fails the identity test above and is split again
        _SPLIT[key] = ent = (img, ver, weakref.ref(base))
    return ent[0]


def set_dense_mode(mode):
    """"f16x3" (default: split-fp16 x3 on the f16 MFMA, ~3e-7 relative), "bf16x3" (split-bf16 x3, ~5e-6, no range
    limit) or "f32" (fp32-input MFMA, exact f32 FMA chain)."""
    global _MODE
    lib().call("s2d_set_dense_mode", {"f32": 0, "bf16x3": 1, "f16x3": 2}[mode])
    _MODE = mode


_AMP = [0]


class amp_fp16:
    """`with ops.amp_fp16(enabled):` -- the dense launches inside take fp16-operand / f32-accumulate / f32-output arithmetic
    (autocast-like; one MFMA pass: s2d_gemm_nt_amp_f32 / s2d_conv2d_nhwc_amp_f32) instead of the fp32-class split.  The modules the
    reference runs under autocast wrap their forward in it when the model's `amp_compute` is set (engine/train_loop.py:709); the
    pixel decoder and the criterion never do (msdeformattn.py:314, matcher.py:266-268 force fp32).  Forward / loss only: the
    gradient kernels stay fp32-class."""

    def __init__(self, enabled=True):
        self.enabled = bool(enabled)

    def __enter__(self):
        _AMP[0] += 1 if self.enabled else 0

    def __exit__(self, *a):
        _AMP[0] -= 1 if self.enabled else 0


def amp_active():
    return _AMP[0] > 0 and _MODE == "f16x3"


def gemm_nt(A, B, scale=None, bias=None, res=None, relu=False, out=None, res_rows=0, res_cols=0, dropout=None):
    """out[..., M, N] = act(A[..., M, K] @ B[(...), N, K]^T * scale + bias + res).
    A may be 2-D or batched 3-D; B 2-D (shared) or 3-D (per batch).  res_rows > 0: res is [res_rows, ldr] and row r of the
    output receives res[r % res_rows]; res_cols > 0: only the first res_cols columns receive it (ldr = res.shape[-1]).
    dropout = (p, seed, site[, row0]): act(dropout_p(A @ B^T + bias) + res) with the counter-based mask of csrc/dropout.h fused into
    the epilogue (2-D operands, N % 8 == 0)."""
    for t in (A, B, scale, bias, res, out):
        _chk(t)
    if dropout is not None and dropout[0] > 0.0:
        p, seed, site = dropout[:3]
        row0 = dropout[3] if len(dropout) > 3 else 0
        assert A.dim() == 2 and B.dim() == 2 and scale is None and not res_rows and not res_cols
        M, K = A.shape
        N = B.shape[0]
        if out is None:
            out = torch.empty((M, N), device=A.device, dtype=torch.float32)
        if _MODE != "f16x3" or N % 8 or out.shape[-1] % 8 or (res is not None and res.shape[-1] % 8):
            # the mask epilogue exists in the split-fp16 kernels only (s2d_gemm_nt_dropout_f32 rejects the other dense modes and
            # rows that are not whole 8-column mask blocks): same mask, same order of operations, as separate passes
            assert N % 8 == 0 and out.shape[-1] == N, "dropout masks are generated per 8-column block"
            gemm_nt(A, B, bias=bias, out=out)
            dropout_apply(out, p, seed, site, row0, out=out)      # (`dropout` is this function's argument)
            if res is not None:
                out.add_(res)
            if relu:
                out.relu_()
            return out
        with _Timed(2.0 * M * N * K, ("gemm", 1, M, N, K, 4.0 * (M * K + N * K + M * N * (2 if res is not None else 1)))):
            lib().call("s2d_gemm_nt_dropout_f32", A, B, out, M, N, K, K, K, out.shape[-1], bias, res, res.shape[-1] if res is not None else N,
                       int(relu), _static_split(B, N, K, K), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(site), int(row0), _stream())
        return out
    batched = A.dim() == 3
    bs = A.shape[0] if batched else 1
    M, K = A.shape[-2:]
    N = B.shape[-2]
    assert B.shape[-1] == K
    if out is None:
        out = torch.empty((bs, M, N) if batched else (M, N), device=A.device, dtype=torch.float32)
    ldc = out.shape[-1]          # out may be wider than N (padded row stride)
    assert ldc >= N and out.shape[-2] == M
    sA = M * K if batched else 0
    sB = N * K if B.dim() == 3 else 0
    sC = M * ldc
    Bs = _static_split(B, N, K, K) if B.dim() == 2 and not amp_active() else None
    with _Timed(2.0 * bs * M * N * K, ("gemm", bs, M, N, K, 4.0 * bs * (M * K + N * K * (1 if B.dim() == 3 else 1.0 / bs) + M * N * (2 if res is not None else 1)))):
        ldr = res.shape[-1] if res is not None else N
        assert res is None or (ldr >= (res_cols or N) and res.shape[-2] == (res_rows or M))
        if amp_active():
            lib().call("s2d_gemm_nt_amp_f32", A, B, out, M, N, K, K, K, ldc, bs, sA, sB, sC, scale, bias, res, ldr,
                       res.shape[-2] * ldr if res is not None and res.dim() == 3 else 0, res_rows, res_cols, int(relu), _stream())
            return out
        lib().call("s2d_gemm_nt_f32", A, B, out, M, N, K, K, K, ldc, bs, sA, sB, sC, scale, bias, res, ldr,
                   res.shape[-2] * ldr if res is not None and res.dim() == 3 else 0, res_rows, res_cols, int(relu), Bs, _stream())
    return out


def gemm_nt_gate(A, B, gate, gate_scale=1.0, res=None, scale=None):
    """(A @ B^T * scale + res) * gate_scale where gate > 0, else 0 (gate [M, N], scale [N]): a dgrad GEMM with the ReLU / dropout
    gate (and the folded-BatchNorm factor) of the layer it differentiates in its epilogue"""
    for t in (A, B, gate, res, scale):
        _chk(t)
    M, K = A.shape
    N = B.shape[0]
    assert gate.shape == (M, N) and gate.stride(-1) == 1 and (res is None or res.shape == (M, N))
    out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    with _Timed(2.0 * M * N * K, ("gemm", 1, M, N, K, 4.0 * (M * K + N * K + M * N * (3 if res is not None else 2)))):
        lib().call("s2d_gemm_nt_gate_f32", A, B, out, M, N, K, K, K, N, scale, res, N, gate, gate.stride(0), float(gate_scale),
                   _static_split(B, N, K, K), _stream())
    return out


def gate_fusable(N):
    """the gate epilogue exists in the split-fp16 kernels' 16-B row epilogues"""
    return _MODE == "f16x3" and N % 4 == 0


def split_rows(A):
    """fp16 hi/lo row image of a [M, K] fp32 matrix (the layout gemm_nt_presplit reads; one HBM pass)"""
    _chk(A)
    M, K = A.shape
    img = torch.empty((lib().call("s2d_split_weights_words", M, K),), device=A.device, dtype=torch.int32)
    lib().call("s2d_split_weights_f16", A, M, K, K, img, _stream())
    return img


def gemm_nt_presplit(A_split, M, K, B, bias=None, res=None, relu=False, out=None, res_rows=0, res_cols=0):
    """gemm_nt with A handed over as split_rows(A) (or written in that layout by its producer); B must be a static weight"""
    for t in (B, bias, res, out):
        _chk(t)
    N = B.shape[0]
    Bs = _static_split(B, N, K, K)
    assert Bs is not None and B.shape[1] == K
    if out is None:
        out = torch.empty((M, N), device=B.device, dtype=torch.float32)
    with _Timed(2.0 * M * N * K, ("gemm", 1, M, N, K, 4.0 * (M * K + N * K + M * N * (2 if res is not None else 1)))):
        lib().call("s2d_gemm_nt_presplit_f32", A_split, B, out, M, N, K, K, out.shape[-1], bias, res, res.shape[-1] if res is not None else N,
                   res_rows, res_cols, int(relu), Bs, _stream())
    return out


_FFN_PACK = {}    # (W1 ptr, W2 ptr, Wpost ptr, F, Npost) -> (image, versions, weakrefs of the owners)


def _owner(t):
    return t._base if t._base is not None else t


def _ffn_pack(W1, W2, Wpost=None, Wpre=None):
    """cached MFMA-fragment image of an FFN's two weight matrices (+ the projection applied behind it), s2d_ffn_pack_f16; rebuilt when
    an owner changes (parameter version; packed copies are replaced by their module, which changes the key)"""
    F, C = W1.shape
    Np = 0 if Wpost is None else Wpost.shape[0]
    ts = (W1, W2) + (() if Wpost is None else (Wpost,)) + (() if Wpre is None else (Wpre,))
    key = tuple(t.data_ptr() for t in ts) + (F, Np, Wpre is not None)
    owners = [_owner(t) for t in ts]
    ver = tuple(o._version + getattr(o, "_s2d_version", 0) for o in owners)
    ent = _FFN_PACK.get(key)
    if ent is None or ent[1] != ver or any(r() is not o for r, o in zip(ent[2], owners)):
        same = ent is not None and all(r() is o for r, o in zip(ent[2], owners))
        img = ent[0] if same else torch.empty((lib().call("s2d_ffn_pack_words", C, F, Np, int(Wpre is not None)),), device=W1.device, dtype=torch.int32)
        lib().call("s2d_ffn_pack_f16", W1, W2, C, F, Wpost, Np, Wpre, img, _stream())
        if not same and len(_FFN_PACK) >= 64:
            for k in [k for k, e in _FFN_PACK.items() if any(r() is None for r in e[2])]:
                del _FFN_PACK[k]
        _FFN_PACK[key] = ent = (img, ver, [weakref.ref(o) for o in owners])
    return ent[0]


def ffn_fusable(W1, W2):
    """the one-launch FFN exists for the split-fp16 arithmetic, model width 256 and hidden widths that are multiples of 32 (<= 2048)"""
    return (_MODE == "f16x3" and not amp_active() and W1.dim() == 2 and W1.shape[1] == 256 and tuple(W2.shape) == (256, W1.shape[0])
            and lib().call("s2d_ffn_pack_words", 256, int(W1.shape[0]), 0, 0) > 0)


_FFN_EPI = 1      # csrc/ffn.hip S2D_FFN_EPI of the loaded library (1: the out_proj form needs no Xn scratch); experiment builds set S2D_FFN_EPI in the environment
import os as _os
if _os.environ.get("S2D_FFN_EPI"):
    _FFN_EPI = int(_os.environ["S2D_FFN_EPI"])


def ffn_fused(x, W1, b1, W2, b2, ln1=None, ln2=None, dropout=None, eps=1e-5, want_xn=False, post=None, pre=None):
    """y = LN2?( xn + drop( W2 . drop( relu( W1 . xn + b1 ) ) + b2 ) ),  xn = LN1?(x)   in one launch (csrc/ffn.hip).
    x [M, 256]; ln1 / ln2 = (gamma, beta) or None; dropout = (p, seed, site_hidden, site_out[, row0]) or None.
    post = (Wpost [Np, 256], bias [Np], pos [S, npos] or None): also out_post[M, Np] = y . Wpost^T + (pos[row % S] on the first npos
    columns -- it carries their bias, `bias` is not read there -- and bias on the others): the next encoder layer's merged
    projection, applied while the row is in registers (needs ln1 and ln2).
    pre = (Wpre [256, 256], bias [256], res [M, 256], site): x is the deformable attention's sampled values and the FFN input becomes
    res + drop( Wpre . x + bias ) (mask site `site` of the same seed): the attention's output projection, dropout1 and residual.
    -> y, (y, xn) with want_xn (requires ln1), with post additionally out_post as the last element."""
    for t in (x, W1, b1, W2, b2) + tuple(ln1 or ()) + tuple(ln2 or ()):
        _chk(t)
    M, C = x.shape
    F = W1.shape[0]
    assert ffn_fusable(W1, W2) and C == 256 and (not want_xn or ln1 is not None)
    y = torch.empty_like(x)
    xn = torch.empty_like(x) if (want_xn or (pre is not None and _FFN_EPI != 1)) else None
    p, seed, site_h, site_o, row0 = 0.0, 0, 0, 0, 0
    if dropout is not None and dropout[0] > 0.0:
        p, seed, site_h, site_o = dropout[:4]
        row0 = dropout[4] if len(dropout) > 4 else 0
    g1, be1 = ln1 if ln1 is not None else (None, None)
    g2, be2 = ln2 if ln2 is not None else (None, None)
    Wp = pb = pp = out_post = None
    Np = S = npos = ldpos = 0
    if post is not None:
        Wp, pb, pp = post
        assert ln1 is not None and ln2 is not None
        for t in (Wp, pb, pp):
            _chk(t)
        Np = Wp.shape[0]
        assert Wp.shape[1] == C and pb.shape == (Np,)
        if pp is not None:
            S, npos = pp.shape
            ldpos = pp.stride(0)
        out_post = torch.empty((M, Np), device=x.device, dtype=torch.float32)
    Wq = qb = qres = None
    site_pre = 0
    if pre is not None:
        Wq, qb, qres, site_pre = pre
        assert ln1 is not None and ln2 is not None and tuple(Wq.shape) == (C, C) and qres.shape == x.shape
        for t in (Wq, qb, qres):
            _chk(t)
    Nq = C if pre is not None else 0
    # counted as its contractions; bytes: input, output(s), weights once
    with _Timed(4.0 * M * F * C + 2.0 * M * (Np + Nq) * C,
                ("ffn", 1, M, F + (Np + Nq) // 2, C, 4.0 * ((2 + (pre is not None)) * M * C + M * Np + 2 * F * C + (Np + Nq) * C))):
        lib().call("s2d_ffn_fused_f32", x, M, C, F, _ffn_pack(W1, W2, Wp, Wq), b1, b2, g1, be1, g2, be2, float(eps), float(p),
                   int(seed) & 0xFFFFFFFFFFFFFFFF, int(site_h), int(site_o), int(row0), xn, y, Np, pb, pp, S, npos, ldpos, out_post, Np,
                   qb, qres, int(site_pre), _stream())
    res = (y,) + ((xn,) if want_xn else ()) + ((out_post,) if post is not None else ())
    return res if len(res) > 1 else y


def dropout_scale(p):
    """the multiplier of a kept element: 256 / (256 - round(256 p)) = 1 / P(keep) with p quantised to 1 / 256 (csrc/dropout.h)"""
    import numpy as np
    t = max(0, min(int(np.float32(p) * np.float32(256.0) + np.float32(0.5)), 255))
    return float(np.float32(256.0) / np.float32(256 - t)) if t else 1.0


def dropout_apply(x, p, seed, site, row0=0, out=None):
    """x [M, N] * mask / P(keep): the mask gemm_nt(dropout=(p, seed, site)) applied (its gradient; the mask itself from ones)"""
    _chk(x)
    M, N = x.shape
    y = torch.empty_like(x) if out is None else out
    lib().call("s2d_dropout_f32", x, M, N, float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(site), int(row0), y, _stream())
    return y


dropout = dropout_apply


_DROP_CALLS = [0]


def next_dropout_seed():
    """a fresh 64-bit Philox key per dropout-carrying forward call: torch's seed (torch.manual_seed makes runs repeatable)
    mixed with a per-process call counter"""
    _DROP_CALLS[0] += 1
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _DROP_CALLS[0] * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


def conv2d_nhwc(x, w, stride=1, pad=0, scale=None, bias=None, res=None, relu=False):
    """x [N,H,W,Cin], w [Cout,KH,KW,Cin] -> [N,Ho,Wo,Cout]."""
    for t in (x, w, scale, bias, res):
        _chk(t)
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    y = torch.empty((N, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    ws = None if amp_active() else _static_split(w, Cout, KH * KW * Cin, KH * KW * Cin)
    with _Timed(2.0 * N * Ho * Wo * Cout * KH * KW * (3 if Cin == 4 else Cin),
                ("conv", KH, N * Ho * Wo, Cout, KH * KW * Cin, 4.0 * (x.numel() + w.numel() + N * Ho * Wo * Cout * (2 if res is not None else 1)))):   # stem: algorithmic Cin is 3
        if amp_active():
            lib().call("s2d_conv2d_nhwc_amp_f32", x, w, y, N, H, W, Cin, Cout, KH, KW, stride, pad, scale, bias, res, int(relu), _stream())
            return y
        lib().call("s2d_conv2d_nhwc_f32", x, w, y, N, H, W, Cin, Cout, KH, KW, stride, pad, scale, bias, res, int(relu), ws,
                   _stream())
    return y


# ----------------------------------------------------------------------------- MSDeformAttn
import numpy as _np


def _host_i64(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return _np.ascontiguousarray(a, dtype=_np.int64)


def msda_forward(value, shapes, level_start, loc, attn_w):
    """value [N,S,M,D], loc [N,Lq,M,L,P,2], attn_w [N,Lq,M,L,P] -> [N,Lq,M*D]."""
    for t in (value, loc, attn_w):
        _chk(t)
    N, S, M, D = value.shape
    Lq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    sh, ls = _host_i64(shapes), _host_i64(level_start)
    out = torch.empty((N, Lq, M * D), device=value.device, dtype=torch.float32)
    lib().call("s2d_msda_forward_f32", value, sh, ls, loc, attn_w, N, S, M, D, L, Lq, P, out, _stream())
    return out


def msda_backward(value, shapes, level_start, loc, attn_w, grad_out, atomics=False):
    """gradients of msda_forward.  Default: the atomic-free, bitwise reproducible form (sampling graph inverted by a stable sort,
    grad_value rows gathered); atomics=True: the reference's scatter with float atomics (s2d_msda_backward_f32, needs no workspace)"""
    for t in (value, loc, attn_w, grad_out):
        _chk(t)
    N, S, M, D = value.shape
    Lq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    sh, ls = _host_i64(shapes), _host_i64(level_start)
    gv, gl, gw = torch.empty_like(value), torch.empty_like(loc), torch.empty_like(attn_w)
    if atomics:
        lib().call("s2d_msda_backward_f32", value, sh, ls, loc, attn_w, grad_out, N, S, M, D, L, Lq, P, gv, gl, gw, _stream())
        return gv, gl, gw
    nb = lib().call("s2d_msda_backward_workspace_bytes", sh, N, M, L, Lq, P)
    ws = torch.empty((nb,), device=value.device, dtype=torch.uint8)
    lib().call("s2d_msda_backward_sorted_f32", value, sh, ls, loc, attn_w, grad_out, N, S, M, D, L, Lq, P, gv, gl, gw, ws, nb, _stream())
    return gv, gl, gw


def _chk_i64_dev(t, shape):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() and tuple(t.shape) == shape):
        raise RuntimeError(f"s2d op needs a contiguous int64 CUDA tensor of shape {shape}")


def msda_forward_dev(value, shapes_dev, level_start_dev, loc, attn_w, want_ws=False):
    """msda_forward with the reference op's argument kinds: spatial shapes [L,2] and level starts [L] are int64 CUDA tensors read
    on the device (ms_deform_attn_cuda.cu:60-75); nothing is copied to the host, cached or synchronised."""
    for t in (value, loc, attn_w):
        _chk(t)
    N, S, M, D = value.shape
    Lq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    _chk_i64_dev(shapes_dev, (L, 2)); _chk_i64_dev(level_start_dev, (L,))
    ws = torch.empty((lib().call("s2d_msda_dev_forward_workspace_bytes"),), device=value.device, dtype=torch.uint8)
    out = torch.empty((N, Lq, M * D), device=value.device, dtype=torch.float32)
    lib().call("s2d_msda_forward_dev_f32", value, shapes_dev, level_start_dev, loc, attn_w, N, S, M, D, L, Lq, P, out, ws, _stream())
    return (out, ws) if want_ws else out


def msda_backward_dev(value, shapes_dev, level_start_dev, loc, attn_w, grad_out, want_ws=False):
    """gradients of msda_forward_dev (atomic-free sorted form), shapes read on the device"""
    for t in (value, loc, attn_w, grad_out):
        _chk(t)
    N, S, M, D = value.shape
    Lq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    _chk_i64_dev(shapes_dev, (L, 2)); _chk_i64_dev(level_start_dev, (L,))
    gv, gl, gw = torch.empty_like(value), torch.empty_like(loc), torch.empty_like(attn_w)
    nb = lib().call("s2d_msda_dev_backward_workspace_bytes", N, S, M, L, Lq, P)
    ws = torch.empty((nb,), device=value.device, dtype=torch.uint8)
    lib().call("s2d_msda_backward_dev_f32", value, shapes_dev, level_start_dev, loc, attn_w, grad_out, N, S, M, D, L, Lq, P, gv, gl, gw,
               ws, nb, _stream())
    return (gv, gl, gw, ws) if want_ws else (gv, gl, gw)


def msda_dev_status(ws):
    """error flag of the *_dev call that used workspace `ws` (0: shapes accepted).  Synchronises the current stream."""
    import ctypes
    err = ctypes.c_int(0)
    lib().call("s2d_msda_dev_status", ws, ctypes.addressof(err), _stream())
    return err.value


def msda_fused_forward(value, shapes, offs_logits, M=8, P=4):
    """value [N,S,M*D]; offs_logits [N,S,>=M*L*P*3] -> [N,S,M*D].  Both may be column slices of one wider row-major
    buffer (row strides ldv / ldoa): the merged [offsets|logits|value] projection writes them side by side."""
    N, S, C = value.shape
    for t in (value, offs_logits):
        if not t.is_cuda or t.dtype != torch.float32 or t.stride(-1) != 1 or t.stride(0) != S * t.stride(1) or t.storage_offset() % 4:
            raise RuntimeError("msda_fused_forward needs f32 CUDA tensors with unit column stride and dense rows")
    sh = _host_i64(shapes)
    L = sh.shape[0]
    out = torch.empty((N, S, C), device=value.device, dtype=torch.float32)
    # algorithmic bytes: value read once + output written once + offsets / logits read once
    with _Timed(0.0, ("msda", N, S, C, L * P, 4.0 * N * (2 * S * C + S * M * L * P * 3))):
        lib().call("s2d_msda_fused_forward_f32", value, value.stride(1), sh, offs_logits, offs_logits.stride(1), N, S, M, C // M, L, P,
                   out, _stream())
    return out


# ----------------------------------------------------------------------------- glue
PIXEL_MEAN = _np.array([123.675, 116.28, 103.53], _np.float32)
PIXEL_STD = _np.array([58.395, 57.12, 57.375], _np.float32)


def normalize_pad(frames_u8, div=32, mean=PIXEL_MEAN, std=PIXEL_STD):
    """frames u8 [F,3,H0,W0] -> f32 [F,Hp,Wp,4]."""
    _chk(frames_u8, torch.uint8)
    F_, _, H0, W0 = frames_u8.shape
    Hp, Wp = (H0 + div - 1) // div * div, (W0 + div - 1) // div * div
    out = torch.empty((F_, Hp, Wp, 4), device=frames_u8.device, dtype=torch.float32)
    lib().call("s2d_normalize_pad_nhwc4_f32", frames_u8, F_, H0, W0, Hp, Wp, _np.ascontiguousarray(mean, _np.float32),
               _np.ascontiguousarray(std, _np.float32), out, _stream())
    return out


def conv2d_nhwc_gate(x, w, gate, stride=1, pad=0, scale=None, gate_scale=1.0):
    """conv2d_nhwc(x, w) * scale where gate > 0, else 0 (gate: a tensor of the output's shape): dgrad with the gate epilogue"""
    for t in (x, w, scale, gate):
        _chk(t)
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    assert tuple(gate.shape) == (N, Ho, Wo, Cout)
    y = torch.empty((N, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    ws = _static_split(w, Cout, KH * KW * Cin, KH * KW * Cin)
    with _Timed(2.0 * N * Ho * Wo * Cout * KH * KW * Cin, ("conv", KH, N * Ho * Wo, Cout, KH * KW * Cin, 4.0 * (x.numel() + w.numel() + 2 * N * Ho * Wo * Cout))):
        lib().call("s2d_conv2d_nhwc_gate_f32", x, w, y, N, H, W, Cin, Cout, KH, KW, stride, pad, scale, gate, float(gate_scale), ws, _stream())
    return y


def maxpool3x3s2(x, want_idx=False):
    """3x3 / stride 2 / pad 1 max pool (NHWC).  want_idx: -> (y, arg-max taps u8 [N,Ho,Wo,C]) for backward.maxpool_backward"""
    _chk(x)
    N, H, W, C = x.shape
    y = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), device=x.device, dtype=torch.float32)
    if want_idx:
        idx = torch.empty(y.shape, device=x.device, dtype=torch.uint8)
        lib().call("s2d_maxpool3x3s2_nhwc_idx_f32", x, N, H, W, C, y, idx, _stream())
        return y, idx
    lib().call("s2d_maxpool3x3s2_nhwc_f32", x, N, H, W, C, y, _stream())
    return y


def groupnorm_nhwc(x, G, gamma, beta, up=None, relu=False, eps=1e-5):
    for t in (x, gamma, beta, up):
        _chk(t)
    N, H, W, C = x.shape
    ws = torch.empty((lib().call("s2d_groupnorm_workspace_doubles", N, H, W, G),), device=x.device, dtype=torch.float64)
    y = torch.empty_like(x)
    hu, wu = (up.shape[1], up.shape[2]) if up is not None else (0, 0)
    lib().call("s2d_groupnorm_nhwc_f32", x, N, H, W, C, G, gamma, beta, float(eps), up, hu, wu, int(relu), ws, y, _stream())
    return y


def layernorm(x, gamma, beta, res=None, eps=1e-5):
    for t in (x, gamma, beta, res):
        _chk(t)
    C = x.shape[-1]
    y = torch.empty_like(x)
    lib().call("s2d_layernorm_f32", x, res, gamma, beta, x.numel() // C, C, float(eps), y, _stream())
    return y


def add_bcast(x, b):
    _chk(x); _chk(b)
    y = torch.empty_like(x)
    lib().call("s2d_add_bcast_f32", x, b, x.numel(), b.numel(), y, _stream())
    return y


def pe_sine(T, H, W, num_pos_feats=128, add_c=None, device="cuda"):
    """-> [max(T,1)*H*W, 2*num_pos_feats] token-major."""
    _chk(add_c)
    out = torch.empty((max(T, 1) * H * W, 2 * num_pos_feats), device=device, dtype=torch.float32)
    lib().call("s2d_pe_sine_f32", T, H, W, num_pos_feats, add_c, out, _stream())
    return out


# ----------------------------------------------------------------------------- masked attention
def attn_mask_bits(mask_logits, B, Q, T, hm, wm, hl, wl, compact=False):
    """mask_logits pixel-major [B, T*hm*wm, ldq] (or, compact, [B, K*4, ldq]: the four source pixels of every key, see
    attn_mask_tap_index) -> (bits int32 [B,K,4], unmasked int32 [B,4])."""
    _chk(mask_logits)
    K = T * hl * wl
    assert mask_logits.shape[1] == (K * 4 if compact else T * hm * wm)
    bits = torch.empty((B, K, 4), device=mask_logits.device, dtype=torch.int32)
    unm = torch.empty((B, 4), device=mask_logits.device, dtype=torch.int32)
    lib().call("s2d_attn_mask_bits", mask_logits, mask_logits.shape[-1], B, Q, T, hm, wm, hl, wl, int(compact), bits, unm, _stream())
    return bits, unm


def attn_mask_tap_index(T, hm, wm, hl, wl, device):
    """int64 [T*hl*wl*4]: for every key of a (hl, wl) level the four pixels of the (hm, wm) map that
    F.interpolate(bilinear, align_corners=False) reads for it, in the order (y0,x0) (y0,x1) (y1,x0) (y1,x1) -- the same
    float32 arithmetic as attn_mask_kernel, so a GEMM restricted to these rows feeds attn_mask_bits(compact=True)."""
    f32 = torch.float32
    ys = torch.arange(hl, dtype=f32, device=device); xs = torch.arange(wl, dtype=f32, device=device)
    sy = torch.clamp((torch.tensor(float(hm), dtype=f32, device=device) / hl) * (ys + 0.5) - 0.5, min=0.0)
    sx = torch.clamp((torch.tensor(float(wm), dtype=f32, device=device) / wl) * (xs + 0.5) - 0.5, min=0.0)
    y0 = sy.to(torch.int64); x0 = sx.to(torch.int64)
    y1 = y0 + (y0 < hm - 1).to(torch.int64); x1 = x0 + (x0 < wm - 1).to(torch.int64)
    rows = torch.stack([y0, y0, y1, y1], -1)                      # [hl, 4]
    cols = torch.stack([x0, x1, x0, x1], -1)                      # [wl, 4]
    pix = rows[:, None, :] * wm + cols[None, :, :]                # [hl, wl, 4]
    t = torch.arange(T, dtype=torch.int64, device=device)[:, None, None, None] * (hm * wm)
    return (t + pix[None]).reshape(-1).contiguous()


def masked_attn(q, k, v, bits=None, unmasked=None, H=8, want_lse=False):
    """q [B,Q,C], k/v [B,K,C] (projected; k and v may be column slices of a wider [B,K,ld] projection output) -> [B,Q,C]
    (and, want_lse, the base-2 log-sum-exp [B,H,128] the backward needs)."""
    _chk(q)
    B, Q, C = q.shape
    K = k.shape[1]
    for t in (k, v):
        if not (t.is_cuda and t.dtype == torch.float32 and t.stride(2) == 1 and t.stride(0) == K * t.stride(1)):
            raise RuntimeError("k / v must be float32 CUDA [B,K,C] with unit column stride and batch stride K * row stride")
    n = lib().call("s2d_attn_workspace_floats", B, H, K)
    ws = torch.empty((n,), device=q.device, dtype=torch.float32)
    out = torch.empty_like(q)
    lse = torch.empty((B, H, 128), device=q.device, dtype=torch.float32) if want_lse else None
    # QK^T + AV: 4 B Q K C flops; K and V rows read once
    with _Timed(4.0 * B * Q * K * C, ("xattn" if bits is not None else "sattn", B, Q, K, C, 4.0 * (2.0 * B * K * C + 2.0 * B * Q * C))):
        lib().call("s2d_masked_attn_f32", q, k, v, k.stride(1), v.stride(1), bits, unmasked, B, Q, K, C, H, ws, out, lse, _stream())
    return (out, lse) if want_lse else out


# ----------------------------------------------------------------------------- matcher / criterion
def matcher_cost(mask_logits, class_logits, tgt, tgt_count, dims, P, weights, coords=None, seed=0):
    """mask_logits [NL,B,T*hm*wm,ldq] pixel-major, class_logits [NL,B,Q,2], tgt u8 [B,Nmax,T,H,W], tgt_count i32 [B].
    dims = (Q, T, hm, wm).  -> C [NL*B, Q, Nmax]"""
    _chk(mask_logits); _chk(class_logits); _chk(tgt, torch.uint8); _chk(tgt_count, torch.int32); _chk(coords)
    NL, B = mask_logits.shape[:2]
    Q, T, hm, wm = dims
    Nmax, H, W = tgt.shape[1], tgt.shape[3], tgt.shape[4]
    n = lib().call("s2d_matcher_workspace_floats", NL, B, T, int(P), H, W)
    ws = torch.empty((n,), device=tgt.device, dtype=torch.float32)
    C = torch.empty((NL * B, Q, Nmax), device=tgt.device, dtype=torch.float32)
    wc, wm_, wd = weights
    lib().call("s2d_matcher_cost_f32", mask_logits, class_logits, tgt, tgt_count, coords, int(seed), NL, B, Q,
               mask_logits.shape[-1], T, hm, wm, H, W, Nmax, P, float(wc), float(wm_), float(wd), ws, C, _stream())
    return C


def lsap(C, tgt_count, B):
    """C [nprob,Q,Nmax] -> idx_q, idx_t int32 [nprob, min(Q,Nmax)], n_match int32 [nprob]"""
    _chk(C); _chk(tgt_count, torch.int32)
    nprob, Q, Nmax = C.shape
    maxm = min(Q, Nmax)
    iq = torch.zeros((nprob, maxm), device=C.device, dtype=torch.int32)
    it = torch.zeros((nprob, maxm), device=C.device, dtype=torch.int32)
    nm = torch.zeros((nprob,), device=C.device, dtype=torch.int32)
    lib().call("s2d_lsap_f32", C, tgt_count, nprob, B, Q, Nmax, iq, it, nm, _stream())
    return iq, it, nm


def kd_targets(t_class_logits, t_mask_logits, dims, H, W, Nmax, thr=0.75, topk=100):
    """teacher class logits [B,Q,2], mask logits pixel-major [B,T*hm*wm,ldq]; dims=(Q,T,hm,wm)."""
    _chk(t_class_logits); _chk(t_mask_logits)
    B = t_class_logits.shape[0]
    Q, T, hm, wm = dims
    dev = t_class_logits.device
    tgt = torch.empty((B, Nmax, T, H, W), device=dev, dtype=torch.uint8)
    count = torch.zeros((B,), device=dev, dtype=torch.int32)
    kept = torch.zeros((B, Nmax), device=dev, dtype=torch.int32)
    nonempty = torch.empty((B, Nmax, T), device=dev, dtype=torch.int32)
    lib().call("s2d_kd_targets_u8", t_class_logits, t_mask_logits, float(thr), int(topk), B, Q, t_mask_logits.shape[-1], T, hm,
               wm, H, W, Nmax, tgt, count, kept, nonempty, _stream())
    return tgt, count, kept, nonempty


def target_nonempty(tgt, count):
    _chk(tgt, torch.uint8); _chk(count, torch.int32)
    B, Nmax, T, H, W = tgt.shape
    out = torch.empty((B, Nmax, T), device=tgt.device, dtype=torch.int32)
    lib().call("s2d_target_nonempty", tgt, count, B, Nmax, T, H, W, out, _stream())
    return out


def point_loss(mask_logits, tgt, tgt_count, nonempty, idx_q, idx_t, n_match, dims, P, oversample=3.0, importance=0.75,
               coords_over=None, coords_rand=None, seed=0, drop_empty=True, world_size=1.0, keep=False):
    """-> losses [NL,2] (loss_mask, loss_dice); keep=True: also the context point_loss_backward needs (the arguments and the
    workspace with the selection state the forward found)"""
    _chk(mask_logits); _chk(tgt, torch.uint8); _chk(coords_over); _chk(coords_rand)
    for t in (tgt_count, nonempty, idx_q, idx_t, n_match):
        _chk(t, torch.int32)
    NL, B = mask_logits.shape[:2]
    Q, T, hm, wm = dims
    Nmax, H, W = tgt.shape[1], tgt.shape[3], tgt.shape[4]
    nbytes = lib().call("s2d_point_loss_workspace_bytes", NL, B, Q, Nmax, T, hm, wm, int(P), float(oversample), float(importance), H, W)
    ws = torch.empty((nbytes,), device=tgt.device, dtype=torch.uint8)
    losses = torch.zeros((NL, 2), device=tgt.device, dtype=torch.float32)
    args = (mask_logits, tgt, tgt_count, nonempty, idx_q, idx_t, n_match, coords_over, coords_rand, int(seed), NL, B, Q,
            mask_logits.shape[-1], T, hm, wm, H, W, Nmax, int(P), float(oversample), float(importance), int(drop_empty), float(world_size), ws)
    lib().call("s2d_point_loss_f32", *args, losses, _stream())
    return (losses, args) if keep else losses


def point_loss_kept_rows(ctx):
    """rows the forward kept, from the context of point_loss(keep=True): ([NL] per layer, total).  The counts sit in the forward's workspace
    (lcount[NL + 1], after six int32 arrays of `rows` entries, csrc/loss.hip loss_setup): a small read back, synchronising."""
    NL, B, Q, T, Nmax = ctx[10], ctx[11], ctx[12], ctx[14], ctx[19]
    rows = NL * B * min(Q, Nmax) * T
    c = ctx[-1].view(torch.int32)[rows * 6:rows * 6 + NL + 1].cpu().tolist()
    return c[:NL], c[NL]


def point_loss_backward(ctx, w_mask, w_dice):
    """ctx from point_loss(keep=True) -> grad rows [NL*B*maxm*T, hm*wm]: d(w_mask*loss_mask + w_dice*loss_dice, summed over
    layers) / d(the matched query's logit map of that (layer, clip, slot, frame))"""
    NL, B, Q, T, hm, wm, Nmax = ctx[10], ctx[11], ctx[12], ctx[14], ctx[15], ctx[16], ctx[19]
    rows = NL * B * min(Q, Nmax) * T
    # the backward walks the rows whose point samples the forward kept (the first min(rows, 4096) ACTIVE rows); the number of
    # active rows sits in the forward's workspace (lcount[NL], after six int32 arrays of `rows` entries): one 4-byte read back
    active = point_loss_kept_rows(ctx)[1]
    if active > min(rows, 4096):
        raise NotImplementedError(f"{active} matched (layer, target, frame) rows in one criterion pass: the gradient path keeps 4096")
    g = torch.empty((rows, hm * wm), device=ctx[0].device, dtype=torch.float32)
    lib().call("s2d_point_loss_backward_f32", *ctx, float(w_mask), float(w_dice), g, None, _stream())
    return g


def class_loss_backward(class_logits, idx_q, n_match, w_ce, eos_coef=0.1):
    """d(w_ce * loss_ce)/d(class_logits [B,Q,2])"""
    _chk(class_logits); _chk(idx_q, torch.int32); _chk(n_match, torch.int32)
    B, Q, _ = class_logits.shape
    out = torch.empty_like(class_logits)
    lib().call("s2d_class_loss_backward_f32", class_logits, idx_q, n_match, B, Q, idx_q.shape[-1], float(eos_coef), float(w_ce), out, _stream())
    return out


def class_loss(class_logits, idx_q, n_match, eos_coef=0.1):
    """class_logits [B,Q,2], idx_q [B,maxm], n_match [B] -> 0-dim loss_ce"""
    _chk(class_logits); _chk(idx_q, torch.int32); _chk(n_match, torch.int32)
    B, Q, _ = class_logits.shape
    out = torch.zeros((1,), device=class_logits.device, dtype=torch.float32)
    lib().call("s2d_class_loss_f32", class_logits, idx_q, n_match, B, Q, idx_q.shape[-1], float(eos_coef), out, _stream())
    return out[0]


# --------------------------------------------------------------------------- eval-side post-processing (infer.hip)
def infer_select(class_logits, K):
    """class_logits [Q,C+1] -> (scores [K] f32, query [K] i32, label [K] i32): softmax[:, :-1], sorted top-K"""
    _chk(class_logits)
    Q, C1 = class_logits.shape
    dev = class_logits.device
    scores = torch.empty((K,), device=dev, dtype=torch.float32)
    query = torch.empty((K,), device=dev, dtype=torch.int32)
    label = torch.empty((K,), device=dev, dtype=torch.int32)
    lib().call("s2d_infer_select_f32", class_logits, Q, C1 - 1, K, scores, query, label, _stream())
    return scores, query, label


def infer_masks(mask_logits, dims, padded, img_size, out_size, query, want_bits=False):
    """mask_logits pixel-major [T*hm*wm, ldq]; dims = (T, hm, wm); -> masks u8 [K,T,oh,ow] (and bit words [K,words])"""
    _chk(mask_logits); _chk(query, torch.int32)
    T, hm, wm = dims
    K = query.shape[0]
    (Hp, Wp), (ih, iw), (oh, ow) = padded, img_size, out_size
    dev = mask_logits.device
    ws = torch.empty((lib().call("s2d_infer_workspace_floats", K, T, hm, wm),), device=dev, dtype=torch.float32)
    masks = torch.empty((K, T, oh, ow), device=dev, dtype=torch.uint8)
    bits = torch.empty((K, lib().call("s2d_mask_bit_words", T, oh, ow)), device=dev, dtype=torch.int32) if want_bits else None
    lib().call("s2d_infer_masks_u8", mask_logits, mask_logits.shape[-1], T, hm, wm, Hp, Wp, ih, iw, oh, ow, query, K, ws, masks,
               bits, _stream())
    return masks, bits


def pack_mask_bits(masks):
    """u8 masks [K, ...] (0 / non-0) -> int32 bit words [K, ceil(n/32)], the layout mask_pair_counts reads"""
    _chk(masks, torch.uint8)
    K = masks.shape[0]
    n = masks[0].numel() if K else 0
    bits = torch.empty((K, (n + 31) // 32), device=masks.device, dtype=torch.int32)
    if K:
        lib().call("s2d_pack_mask_bits_u8", masks, K, n, bits, _stream())
    return bits


def mask_pair_counts(bits):
    """bit-packed masks [K,words] -> int64 [K,K] intersection counts (diagonal = areas)"""
    _chk(bits, torch.int32)
    K, words = bits.shape
    inter = torch.empty((K, K), device=bits.device, dtype=torch.int64)
    lib().call("s2d_mask_pair_counts_u64", bits, K, words, inter, _stream())
    return inter

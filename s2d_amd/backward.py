"""Gradients of the dense layers on the forward kernels (csrc/backward.hip for the helpers).

What `losses.backward()` (engine/train_loop.py:720) does for an nn.Linear / 1x1 convolution y = x W^T + b:
    dx = dy W,   dW = dy^T x,   db = sum_rows dy.
dx is an NT GEMM against W^T.  dW contracts over the M rows (3e5..4e6): the contraction is cut into S slices, the TN kernel
(csrc/gemm_tn.hip) transposes the row-major tiles of dy and x on their way into LDS, and the S partial products are added
in a fixed order, so the result is reproducible (shapes the TN kernel cannot take are transposed once and run as the batch
of one NT-GEMM launch).  Same split-fp16 x3 arithmetic as the forward (fp32-class accuracy)."""
import os
import weakref

import torch

from . import ops
from ._lib import lib


def _st():
    return ops._stream()


_PENDING = None        # deferred "param.grad += g" pairs (begin_deferred_acc .. flush_acc): one multi-tensor launch instead of ~350
_ACC_CHUNK = 1 << 16


def begin_deferred_acc():
    """from here to flush_acc(), acc() on an existing contiguous float32 gradient only records the pair"""
    global _PENDING
    if _PENDING is None:
        _PENDING = ([], set())


def flush_acc(end=False):
    """apply the recorded pairs in one launch (s2d_multi_add_f32); end=True also leaves the deferred mode"""
    global _PENDING
    if _PENDING is None:
        return
    pairs, _ = _PENDING
    _PENDING = None if end else ([], set())
    if not pairs:
        return
    table, ct, co = [], [], []
    for i, (dst, src) in enumerate(pairs):
        n = dst.numel()
        table += [src.data_ptr(), dst.data_ptr(), n]
        for off in range(0, n, _ACC_CHUNK):
            ct.append(i); co.append(off)
    dev = pairs[0][0].device
    # ONE pinned staging buffer (int64: table | chunk offsets | chunk tensors) and one asynchronous copy per flush: a pageable source
    # would be staged by the runtime and drain the host's launch-ahead in the middle of the backward.  A ring of buffers, each reused
    # only after the copy that read it has completed (event), keeps the host from overwriting a table still in flight.
    n_tab, n_ch = len(table), len(ct)
    need = n_tab + n_ch + (n_ch + 1) // 2                 # the int32 chunk -> tensor table rides in the tail, two per word
    slot = _staging(need)
    host = slot["host"]
    host[:n_tab] = torch.tensor(table, dtype=torch.int64)
    host[n_tab:n_tab + n_ch] = torch.tensor(co, dtype=torch.int64)
    host[n_tab + n_ch:need].view(torch.int32)[:n_ch] = torch.tensor(ct, dtype=torch.int32)
    d = torch.empty((need,), dtype=torch.int64, device=dev)
    d.copy_(host[:need], non_blocking=True)
    slot["event"].record(torch.cuda.current_stream(dev))
    lib().call("s2d_multi_add_f32", d[:n_tab], d[n_tab + n_ch:].view(torch.int32), d[n_tab:n_tab + n_ch], n_ch, _ACC_CHUNK, _st())
    # `pairs` (and with it every src) stays referenced until here: later allocations on this stream are ordered behind the launch


_STAGE = {"ring": [], "next": 0}


def _staging(n):
    """next pinned int64 buffer of the ring with room for n words, free again (its last copy has completed)"""
    ring = _STAGE["ring"]
    if not ring:
        for _ in range(4):
            ring.append({"host": torch.empty((max(n, 1 << 14),), dtype=torch.int64).pin_memory(), "event": torch.cuda.Event()})
    slot = ring[_STAGE["next"] % len(ring)]
    _STAGE["next"] += 1
    slot["event"].synchronize()                      # a no-op unless the host is four flushes ahead of the device
    if slot["host"].numel() < n:
        slot["host"] = torch.empty((2 * n,), dtype=torch.int64).pin_memory()
    return slot


def acc(param, g):
    """accumulate a gradient into param.grad (allocating it on first use), as autograd's AccumulateGrad does"""
    g = g.reshape(param.shape)
    if param.grad is None:
        param.grad = g.contiguous().clone() if g.data_ptr() == param.data_ptr() else g.contiguous()
        return
    gr = param.grad
    if not _defer_add(gr, g):
        param.grad += g


def _defer_add(dst, src):
    """dst += src through the pending multi-tensor launch when the deferred mode is on and the pair qualifies -> True"""
    if _PENDING is None or not (dst.is_contiguous() and dst.dtype == torch.float32 and src.dtype == torch.float32 and src.device == dst.device
                                and src.numel() == dst.numel()):
        return False
    lo = dst.data_ptr()
    hi = lo + 4 * dst.numel()
    # one destination per launch: a second contribution to the same gradient -- or to ANY byte range that overlaps a pending one (a
    # whole parameter and a row slice of it, say) -- would be a lost update between blocks of the multi-tensor kernel
    if any(lo < b and a < hi for a, b in _PENDING[1]):
        flush_acc()
    _PENDING[0].append((dst, src.contiguous()))
    _PENDING[1].add((lo, hi))
    return True


def acc_wgrad(param, dy, x):
    """param.grad += dy^T @ x, written by the slice reduction itself when the gradient buffer already exists (one launch and one
    pass over the gradient less than weight_grad + acc); param may be any view-compatible shape of [N, K] (1x1 convolutions)"""
    g = param.grad
    N, K = dy.shape[1], x.shape[1]
    if g is not None and g.is_contiguous() and g.numel() == N * K and g.dtype == torch.float32:
        weight_grad(dy, x, out=g.view(N, K), beta=1.0)
    else:
        acc(param, weight_grad(dy, x))


def acc_wbgrad(wparam, bparam, dy, x):
    """acc_wgrad + acc_bgrad of one linear layer with ONE pass over dy (the bias gradient rides in the weight gradient's kernel)"""
    gw, gb = wparam.grad, bparam.grad
    N, K = dy.shape[1], x.shape[1]
    if (gw is not None and gw.is_contiguous() and gw.numel() == N * K and gw.dtype == torch.float32 and
            gb is not None and gb.is_contiguous() and gb.numel() == N and gb.dtype == torch.float32):
        weight_grad(dy, x, out=gw.view(N, K), beta=1.0, bias_out=gb.view(-1), bias_beta=1.0)
    elif gw is None and gb is None:
        db = torch.empty((N,), device=dy.device, dtype=torch.float32)
        acc(wparam, weight_grad(dy, x, bias_out=db))
        acc(bparam, db)
    else:
        acc_wgrad(wparam, dy, x)
        acc_bgrad(bparam, dy)


def acc_bgrad(param, dy):
    """param.grad += column sums of dy (see acc_wgrad)"""
    g = param.grad
    if g is not None and g.is_contiguous() and g.numel() == dy.shape[1] and g.dtype == torch.float32:
        bias_grad(dy, out=g.view(-1), beta=1.0)
    else:
        acc(param, bias_grad(dy))


def sum_slices(x):
    """x [S, ...] -> sum over the leading dim in a fixed order"""
    ops._chk(x)
    n = x[0].numel()
    out = torch.empty(x.shape[1:], device=x.device, dtype=torch.float32)
    lib().call("s2d_reduce_slices_f32", x, x.shape[0], n, n, 0.0, out, _st())
    return out


def transpose(x, pad_to=None):
    """x [R,C] -> [C, ld] with ld = pad_to or R (columns >= R zero-filled)"""
    ops._chk(x)
    R, C = x.shape
    ld = pad_to or R
    out = torch.empty((C, ld), device=x.device, dtype=torch.float32) if ld == R else torch.zeros((C, ld), device=x.device, dtype=torch.float32)
    lib().call("s2d_transpose_f32", x, R, C, C, out, ld, _st())
    return out


def _slices(M, out_tiles, slots=512):
    """number of contraction slices and rows per slice (a multiple of 32, slices of at least 512 rows).  `slots`: workgroups of the kernel the
    chip holds at once (256 CUs x 2 for the 128-wide TN tile, x 4 for the 64-wide one).  The launch has out_tiles * S workgroups of equal
    length: S is taken near the count that fills the chip once, where out_tiles * S falls just BELOW a multiple of `slots` -- 36 tiles x 15
    slices = 540 workgroups on 512 slots ran as one full round and one of 28 (the 3 x 3 convolutions' weight gradients, round 5)."""
    smax = max(1, min(M // 512 if M >= 512 else 1, 512))
    want = max(1, min((slots + out_tiles - 1) // out_tiles, smax))
    best, best_eff = want, -1.0
    for S in range(max(1, want - want // 4), min(smax, 2 * want + 2) + 1):       # a function of the shape alone: the order of the partial sums stays fixed
        wg = S * out_tiles
        eff = wg / float((wg + slots - 1) // slots * slots)
        if eff > best_eff + 1e-9:
            best, best_eff = S, eff
    S = best
    chunk = ((M + S - 1) // S + 31) // 32 * 32
    S = (M + chunk - 1) // chunk
    return S, chunk


_TN_MAX_OUT = int(os.environ.get("S2D_TN_MAX_OUT", 1 << 40))     # experiments: larger outputs take the transposed-copy path


def _tn_ok(M, N, K):
    return N % 4 == 0 and K % 4 == 0 and M * N * 4 <= 0xFFFFFF00 and M * K * 4 <= 0xFFFFFF00


def weight_grad(dy, x, out=None, beta=0.0, bias_out=None, bias_beta=0.0):
    """dW [N,K] = dy[M,N]^T @ x[M,K]  (out given: out = beta * out + dW).  bias_out [N] given: also bias_out = bias_beta * bias_out +
    column sums of dy -- inside the TN kernel, which has dy in registers anyway, instead of a second pass over it."""
    ops._chk(dy); ops._chk(x)
    M, N = dy.shape
    K = x.shape[1]
    assert x.shape[0] == M
    if N * K <= _TN_MAX_OUT and _tn_ok(M, N, K):                        # no transposed copies: the TN kernel
        S, chunk = _slices(M, ((N + 127) // 128) * ((K + 127) // 128 if K >= 128 else (K + 63) // 64), 512 if K >= 128 else 1024)      # the TN kernel's tiles: 128 x 128 / 128 x 64
        part = torch.empty((S, N, K), device=dy.device, dtype=torch.float32)
        bpart = torch.empty((S, N), device=dy.device, dtype=torch.float32) if bias_out is not None else None
        lib().call("s2d_gemm_tn_f32", dy, x, part, N, K, M, M, N, K, chunk, 0, bpart, _st())
        # one slice (short contractions: the video decoder's 200 query rows): nothing to reduce -- the slice IS the result, and an
        # accumulation into an existing gradient joins the pending multi-tensor add instead of taking a launch of its own
        bias_pending = bias_out is not None and not (S == 1 and bias_beta == 1.0 and _defer_add(bias_out, bpart.view(-1)))
        if out is None:
            if S == 1 and not bias_pending:
                return part.view(N, K)
            out = torch.empty((N, K), device=dy.device, dtype=torch.float32)
            beta = 0.0
        elif S == 1 and beta == 1.0 and not bias_pending and _defer_add(out, part.view(N, K)):
            return out
        if bias_pending:                                  # both reductions in one launch
            lib().call("s2d_reduce_slices_pair_f32", part, N * K, N * K, float(beta), out, bpart, N, N, float(bias_beta), bias_out, S, _st())
        else:
            lib().call("s2d_reduce_slices_f32", part, S, N * K, N * K, float(beta), out, _st())
        return out
    if bias_out is not None:
        bias_grad(dy, out=bias_out, beta=bias_beta)
    S, chunk = _slices(M, ((N + 127) // 128) * ((K + 127) // 128))
    Mp = S * chunk
    dyt, xt = transpose(dy, Mp), transpose(x, Mp)
    part = torch.empty((S, N, K), device=dy.device, dtype=torch.float32)
    # batch b: A = dyt[:, b*chunk : (b+1)*chunk] (row stride Mp), B = xt[:, same], C = part[b]
    lib().call("s2d_gemm_nt_f32", dyt, xt, part, N, K, chunk, Mp, Mp, K, S, chunk, chunk, N * K, None, None, None, K, 0, 0, 0, 0, None, _st())
    if out is None:
        out = torch.empty((N, K), device=dy.device, dtype=torch.float32)
        beta = 0.0
    lib().call("s2d_reduce_slices_f32", part, S, N * K, N * K, float(beta), out, _st())
    return out


def bias_grad(dy, out=None, beta=0.0):
    """db [N] = column sums of dy [M,N]"""
    ops._chk(dy)
    M, N = dy.shape
    rows = max(64, (M + 1023) // 1024)             # up to 1024 slices: four workgroups per CU keep enough rows in flight
    S = (M + rows - 1) // rows
    part = torch.empty((S, N), device=dy.device, dtype=torch.float32)
    lib().call("s2d_colsum_slices_f32", dy, M, N, N, rows, part, _st())
    if out is None:
        out = torch.empty((N,), device=dy.device, dtype=torch.float32)
        beta = 0.0
    lib().call("s2d_reduce_slices_f32", part, S, N, N, float(beta), out, _st())
    return out


_WT = {}     # (data_ptr, shape) -> [W^T, version of the owning tensor, weak reference to the owning tensor]


def _sweep(cache):
    """drop the copies whose owning tensor is gone (called when a new key enters)"""
    for k in [k for k, e in cache.items() if e[2]() is None]:
        del cache[k]


def _transposed_weight(w, Np):
    """W^T [K,Np] of a weight [N,K] (zero columns beyond N), cached per weight version and marked static so that the dense
    kernels also cache its pre-split image: the dgrad GEMMs then run on the same kernels as the forward's"""
    # the owner is identified BEFORE detaching: a detached alias is a fresh object on every call (and does not carry the library's
    # version attribute), so keying on it would rebuild the copy at every call
    base = w._base if w._base is not None else w
    w = w.detach()
    key = (w.data_ptr(), tuple(w.shape), Np)
    ent = _WT.get(key)
    if ent is None or ent[2]() is not base:
        wt = ops.mark_static(transpose(w.contiguous(), Np if Np != w.shape[0] else None))
        wt._s2d_version = 0
        _sweep(_WT)
        ent = _WT[key] = [wt, ops.version_of(base), weakref.ref(base)]     # a WEAK reference: the copy must not keep a replaced weight alive
    elif ent[1] != ops.version_of(base):
        # the weight changed (an optimizer step): transpose into the SAME buffer and bump its version -- a fresh tensor per
        # iteration would enter ops._SPLIT under a new address every time and never leave it (0.5 GB per iteration at c4)
        wc = w.contiguous()
        lib().call("s2d_transpose_f32", wc, wc.shape[0], wc.shape[1], wc.shape[1], ent[0], ent[0].shape[1], _st())
        ent[0]._s2d_version += 1
        ent[1] = ops.version_of(base)
    return ent[0]


_WF = {}


def _flipped_weight(w):
    """the dgrad kernel of a convolution weight [Cout,KH,KW,Cin]: taps flipped, channels swapped -> [Cin,KH,KW,Cout]; cached per
    weight version and marked static (like _transposed_weight), so that the dgrad convolutions run on the forward's kernels with
    a pre-split image"""
    base = w._base if w._base is not None else w          # (see _transposed_weight)
    w = w.detach()
    key = (w.data_ptr(), tuple(w.shape), tuple(w.stride()))
    ent = _WF.get(key)
    if ent is None or ent[2]() is not base:
        wf = ops.mark_static(w.flip(1, 2).permute(3, 1, 2, 0).contiguous())
        wf._s2d_version = 0
        _sweep(_WF)
        ent = _WF[key] = [wf, ops.version_of(base), weakref.ref(base)]
    elif ent[1] != ops.version_of(base):                          # same buffer, new contents (see _transposed_weight)
        ent[0].copy_(w.flip(1, 2).permute(3, 1, 2, 0))
        ent[0]._s2d_version += 1
        ent[1] = ops.version_of(base)
    return ent[0]


_WS2 = {}
_WS2_IDX = {}
_CONV_DGRAD_S2 = os.environ.get("S2D_CONV_DGRAD_S2", "1") != "0"       # 0: stride-2 3 x 3 input gradients through the zero-dilated form


def _stride2_dgrad_weight(w):
    """the 2 x 2 kernel [4 Cin, 2, 2, Cout] of a stride-2 / pad-1 3 x 3 convolution's input gradient: output block (py, px) serves the input
    pixels (2a + py, 2b + px); along one axis an even pixel 2a meets tap 1 of output a, an odd pixel 2a + 1 tap 2 of output a and tap 0 of output
    a + 1 -- with the convolution's pad 1 and the result read at (a + 1, b + 1), tap slot 0 sits on output a and slot 1 on output a + 1.
    Cached per weight version and marked static (like _flipped_weight)."""
    base = w._base if w._base is not None else w
    w = w.detach()
    key = (w.data_ptr(), tuple(w.shape), tuple(w.stride()))
    ent = _WS2.get(key)
    ver = ops.version_of(base)
    if ent is not None and ent[2]() is base and ent[1] == ver:
        return ent[0]
    Co, _, _, Ci = w.shape
    kmap = ((1, 9), (2, 0))                                             # [parity][slot] -> tap along one axis; 9 = no tap (a zero plane)
    idx = _WS2_IDX.get(w.device)
    if idx is None:
        taps = [(kmap[py][jy], kmap[px][jx]) for py in range(2) for px in range(2) for jy in range(2) for jx in range(2)]
        idx = _WS2_IDX[w.device] = torch.tensor([9 if 9 in t else t[0] * 3 + t[1] for t in taps], device=w.device, dtype=torch.long)
    wp = torch.cat([w.permute(3, 1, 2, 0).reshape(Ci, 9, Co), torch.zeros((Ci, 1, Co), device=w.device, dtype=torch.float32)], 1)
    w2 = wp.index_select(1, idx).view(Ci, 4, 2, 2, Co).permute(1, 0, 2, 3, 4).reshape(4 * Ci, 2, 2, Co).contiguous()
    if ent is None or ent[2]() is not base:
        w2 = ops.mark_static(w2)
        w2._s2d_version = 0
        _sweep(_WS2)
        _WS2[key] = [w2, ver, weakref.ref(base)]
        return w2
    ent[0].copy_(w2)                                                    # same buffer, new contents (see _transposed_weight)
    ent[0]._s2d_version += 1
    ent[1] = ver
    return ent[0]


def input_grad(dy, w, res=None, gate=None, gate_scale=1.0, scale=None):
    """dx [M,K] = dy[M,N] @ w[N,K] (+ res: the gradient arriving over a residual connection).  gate [M,K]: the output of the
    ReLU (and dropout) that produced this layer's input -- dx is zeroed where gate <= 0 and multiplied by gate_scale
    elsewhere, in the GEMM's epilogue when the kernels can (one pass over dx less than relu_scale_backward afterwards); scale [K]:
    a per-channel factor applied before res and the gate (the folded BatchNorm of the layer that produced the input)"""
    N = dy.shape[1]
    Np = (N + 3) // 4 * 4                                             # the GEMM wants a contraction length % 4 == 0
    wt = _transposed_weight(w, Np)                                    # [K,Np]: once per weight version
    if Np != N:                                                       # e.g. the 2-way class head: zero-pad the contraction
        dy = torch.nn.functional.pad(dy, (0, Np - N))
    if gate is None:
        return ops.gemm_nt(dy, wt, scale=scale, res=res)
    K = wt.shape[0]
    gate2 = gate.reshape(-1, K)
    if ops.gate_fusable(K) and gate2.stride(-1) == 1 and gate2.stride(0) % 4 == 0:
        return ops.gemm_nt_gate(dy, wt, gate2, gate_scale, res=res, scale=scale)
    dx = ops.gemm_nt(dy, wt, scale=scale, res=res)
    gs = None if gate_scale == 1.0 else torch.full((K,), float(gate_scale), device=dx.device, dtype=torch.float32)
    return relu_scale_backward(dx, gate2, gs)


def linear_backward(x, w, dy, need_dx=True, has_bias=True):
    """gradients of y = x @ w.T + b -> (dx or None, dW, db or None)"""
    return (input_grad(dy, w) if need_dx else None, weight_grad(dy, x), bias_grad(dy) if has_bias else None)


# --------------------------------------------------------------------------- convolutions (NHWC, weights [Cout,KH,KW,Cin])
_CONV_WGRAD_IMPLICIT = os.environ.get("S2D_CONV_WGRAD_IMPLICIT", "1") != "0"      # 0: the round-4 form (padded copies, one launch per tap)


def conv_input_grad(dy, w, stride, pad, in_hw, gate=None, scale=None):
    """dx [N,H,W,Cin] of y = conv2d_nhwc(x, w, stride, pad): a stride-1 convolution of dy (zero-dilated by `stride`) with
    the flipped, channel-transposed kernel and padding KH-1-pad -- the forward implicit-GEMM kernel again."""
    ops._chk(dy)
    N, Ho, Wo, Co = dy.shape
    Co2, KH, KW, Ci = w.shape
    H, W = in_hw
    if stride == 2 and KH == 3 and KW == 3 and pad == 1 and _CONV_DGRAD_S2 and Ci % 4 == 0 and Co % 4 == 0:
        # dx[2a + py][2b + px] only meets the taps of its parity class (1, 2, 2 or 4 of the 9): one stride-1 2 x 2 convolution of dy with a
        # block of Cin output channels per class, then depth-to-space (with the scale / gate of the epilogue forms): 16 products per
        # output position of dy instead of the 36 of the zero-dilated form, no dilated copy
        G = ops.conv2d_nhwc(dy, _stride2_dgrad_weight(w), stride=1, pad=1)             # [N, Ho + 1, Wo + 1, 4 Cin]
        dx = torch.empty((N, H, W, Ci), device=dy.device, dtype=torch.float32)
        if gate is not None:
            ops._chk(gate)
            assert tuple(gate.shape) == (N, H, W, Ci)
        lib().call("s2d_pixel_shuffle2_gate_f32", G, N, Ho + 1, Wo + 1, Ci, H, W, scale, gate, dx, _st())
        return dx
    wf = _flipped_weight(w)                                            # [Cin,KH,KW,Cout], once per weight version, pre-split by the dense kernels
    if KH == 1 and KW == 1 and pad == 0:
        assert gate is None and scale is None
        g = ops.gemm_nt(dy.view(-1, Co), wf.view(Ci, Co)).view(N, Ho, Wo, Ci)
        if stride == 1:
            return g
        dx = torch.zeros((N, H, W, Ci), device=dy.device, dtype=torch.float32)
        dx[:, ::stride, ::stride][:, :Ho, :Wo] = g
        return dx
    if stride > 1:                                                     # zero-dilate: rows/cols of dy land on the stride grid
        Hd, Wd = H + 2 * pad - KH + 1, W + 2 * pad - KW + 1            # size that a stride-1 conv of x would have had
        d = torch.zeros((N, Hd, Wd, Co), device=dy.device, dtype=torch.float32)
        d[:, ::stride, ::stride][:, :Ho, :Wo] = dy
        dy = d
    if gate is not None and ops.gate_fusable(Ci):        # gate [N,H,W,Cin]: the ReLU output that was this convolution's input
        return ops.conv2d_nhwc_gate(dy, wf, gate, stride=1, pad=KH - 1 - pad, scale=scale)
    dx = ops.conv2d_nhwc(dy, wf, stride=1, pad=KH - 1 - pad, scale=scale)
    return dx if gate is None else relu_scale_backward(dx, gate)


def conv_weight_grad(dy, x, KH, KW, stride, pad):
    """dW [Cout,KH,KW,Cin] = sum over positions of dy (x) shifted x.  On the zero-padded input grid a tap is a constant
    offset of the flattened position, so with dy scattered onto that grid (zeros elsewhere) every tap is one contraction
    dYg^T . shift(Xp): the KH*KW taps are contractions on shifted views of the padded input, sliced and reduced in a fixed
    order as in weight_grad (TN kernel for small tap matrices, transposed operands + NT GEMM for large ones)."""
    ops._chk(dy); ops._chk(x)
    N, H, W, Ci = x.shape
    _, Ho, Wo, Co = dy.shape
    if KH == 1 and KW == 1 and pad == 0:
        xs = x if stride == 1 else x[:, ::stride, ::stride][:, :Ho, :Wo].contiguous()
        return weight_grad(dy.view(-1, Co), xs.view(-1, Ci)).view(Co, 1, 1, Ci)
    if Ci * KH * KW <= 256:                                            # few input channels (the stem): explicit im2col, one contraction
        col = torch.empty((N * Ho * Wo, KH * KW * Ci), device=x.device, dtype=torch.float32)
        lib().call("s2d_im2col_nhwc_f32", x, N, H, W, Ci, KH, KW, stride, pad, col, _st())
        return weight_grad(dy.view(-1, Co), col).view(Co, KH, KW, Ci)
    P_out = N * Ho * Wo
    if _CONV_WGRAD_IMPLICIT and Co % 4 == 0 and Ci % 4 == 0 and Ho > 1 and Wo > 1 and _tn_ok(P_out, Co, Ci) and N * H * W * Ci * 4 <= 0xFFFFFF00:
        # every tap in ONE launch of the TN kernel, the input pixel of an (output position, tap) addressed in place: no padded copies,
        # a strided convolution walks its own output positions only; the partial tiles arrive in the weight's layout
        taps = KH * KW
        S, chunk = _slices(P_out, taps * ((Co + 127) // 128) * ((Ci + 127) // 128 if Ci >= 128 else (Ci + 63) // 64), 512 if Ci >= 128 else 1024)
        part = torch.empty((S, Co, KH, KW, Ci), device=x.device, dtype=torch.float32)
        lib().call("s2d_conv_wgrad_tn_f32", dy, x, N, H, W, Ci, Ho, Wo, Co, KH, KW, stride, pad, chunk, part, _st())
        if S == 1:
            return part[0]
        dw = torch.empty((Co, KH, KW, Ci), device=x.device, dtype=torch.float32)
        lib().call("s2d_reduce_slices_f32", part, S, Co * taps * Ci, Co * taps * Ci, 0.0, dw, _st())
        return dw
    Hp, Wp = H + 2 * pad, (W + 2 * pad + 3) // 4 * 4                   # row length % 4: row shifts stay 16-B aligned
    xp = torch.zeros((N, Hp, Wp, Ci), device=x.device, dtype=torch.float32)
    xp[:, pad:pad + H, pad:pad + W] = x
    dg = torch.zeros((N, Hp, Wp, Co), device=x.device, dtype=torch.float32)
    dg[:, 0:Ho * stride:stride, 0:Wo * stride:stride] = dy             # output (y,x) reads input rows y*stride + ky
    P = N * Hp * Wp
    if Co * Ci <= _TN_MAX_OUT and _tn_ok(P, Co, Ci):                   # no transposed copies: the TN kernel
        S, chunk = _slices(P, ((Co + 127) // 128) * ((Ci + 63) // 64))
        taps = KH * KW
        part = torch.empty((S, taps, Co, Ci), device=x.device, dtype=torch.float32)     # the taps' partial tiles interleaved per slice
        for ky in range(KH):
            for kx in range(KW):
                shift = ky * Wp + kx                                   # the tap: B starts `shift` grid positions later
                lib().call("s2d_gemm_tn_f32", dg, xp.data_ptr() + shift * Ci * 4, part.data_ptr() + (ky * KW + kx) * Co * Ci * 4, Co, Ci, P,
                           P - shift, Co, Ci, chunk, taps * Co * Ci, None, _st())
        dw = torch.empty((taps, Co, Ci), device=x.device, dtype=torch.float32)
        lib().call("s2d_reduce_slices_f32", part, S, taps * Co * Ci, taps * Co * Ci, 0.0, dw, _st())    # one reduction for all taps
        return dw.view(KH, KW, Co, Ci).permute(2, 0, 1, 3).contiguous()
    tail = (KH - 1) * Wp + KW                                          # the farthest shift a tap applies
    S, chunk = _slices(P, ((Co + 127) // 128) * ((Ci + 127) // 128))
    Pp = S * chunk
    dgt = transpose(dg.view(P, Co), Pp)                                # [Co, Pp]
    ld = (Pp + tail + 3) // 4 * 4
    xpt = transpose(xp.view(P, Ci), ld)                                # [Ci, ld], zero beyond P
    part = torch.empty((S, Co, Ci), device=x.device, dtype=torch.float32)
    dw = torch.empty((Co, KH, KW, Ci), device=x.device, dtype=torch.float32)
    tmp = torch.empty((Co, Ci), device=x.device, dtype=torch.float32)
    for ky in range(KH):
        for kx in range(KW):
            shifted = xpt[:, ky * Wp + kx:]                            # a view: same row stride, later start
            lib().call("s2d_gemm_nt_f32", dgt, shifted, part, Co, Ci, chunk, Pp, ld, Ci, S, chunk, chunk, Co * Ci, None, None, None, Ci,
                       0, 0, 0, 0, None, _st())
            lib().call("s2d_reduce_slices_f32", part, S, Co * Ci, Co * Ci, 0.0, tmp, _st())
            dw[:, ky, kx] = tmp
    return dw


# --------------------------------------------------------------------------- normalisation / activation
def layernorm_backward(x, dy, gamma, res=None, eps=1e-5):
    """y = LayerNorm(x + res) * gamma + beta -> (dx (also the gradient of res), dgamma, dbeta)"""
    for t in (x, dy, gamma, res):
        ops._chk(t)
    C = x.shape[-1]
    rows = x.numel() // C
    nb = lib().call("s2d_layernorm_backward_blocks", rows)
    part = torch.empty((nb, 2, C), device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x)
    lib().call("s2d_layernorm_backward_f32", x, res, dy, gamma, rows, C, float(eps), dx, part, _st())
    gb = torch.empty((2, C), device=x.device, dtype=torch.float32)
    lib().call("s2d_reduce_slices_f32", part, nb, 2 * C, 2 * C, 0.0, gb, _st())
    return dx, gb[0], gb[1]


def relu_scale_backward(dy, y=None, scale=None, want_res=False):
    """gradient through y = relu(z * scale + bias [+ res]): dz = dy * (y > 0) * scale (channels innermost); want_res: also
    dres = dy * (y > 0), from the same pass"""
    for t in (dy, y, scale):
        ops._chk(t)
    C = dy.shape[-1]
    dz = torch.empty_like(dy)
    dres = torch.empty_like(dy) if want_res else None
    lib().call("s2d_relu_scale_backward_f32", dy, y, scale, dy.numel(), C, dz, dres, _st())
    return (dz, dres) if want_res else dz


def relu_gate_add(a, g, y):
    """a + g * (y > 0) in one pass (a: a gradient that arrives gated already, g: another gradient of the same ReLU output y)"""
    for t in (a, g, y):
        ops._chk(t)
    assert a.shape == g.shape == y.shape
    out = torch.empty_like(a)
    lib().call("s2d_relu_gate_add_f32", a, g, y, a.numel(), out, _st())
    return out


def groupnorm_backward(x, dy, G, gamma, eps=1e-5):
    """y = GroupNorm_G(x) * gamma + beta, NHWC -> (dx, dgamma, dbeta)"""
    for t in (x, dy, gamma):
        ops._chk(t)
    N, H, W, C = x.shape
    ws = torch.empty((2 * lib().call("s2d_groupnorm_workspace_doubles", N, H, W, G),), device=x.device, dtype=torch.float64)
    nb = lib().call("s2d_groupnorm_backward_blocks", N, H, W)
    part = torch.empty((nb, 2, C), device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x)
    lib().call("s2d_groupnorm_backward_f32", x, dy, gamma, N, H, W, C, G, float(eps), ws, dx, part, _st())
    gb = torch.empty((2, C), device=x.device, dtype=torch.float32)
    lib().call("s2d_reduce_slices_f32", part, nb, 2 * C, 2 * C, 0.0, gb, _st())
    return dx, gb[0], gb[1]


def resize_bilinear_backward(dy, hu, wu):
    """adjoint of the bilinear (align_corners=False) resize [N,hu,wu,C] -> dy's [N,H,W,C]"""
    ops._chk(dy)
    N, H, W, C = dy.shape
    dup = torch.empty((N, hu, wu, C), device=dy.device, dtype=torch.float32)
    lib().call("s2d_resize_bilinear_backward_nhwc_f32", dy, N, H, W, C, hu, wu, dup, _st())
    return dup


def groupnorm_up_relu_backward(x, y, dy, G, gamma, up_hw=None, relu=False, eps=1e-5):
    """gradients of ops.groupnorm_nhwc(x, G, gamma, beta, up, relu): -> (dx, dgamma, dbeta, dup or None)"""
    g = relu_scale_backward(dy, y) if relu else dy
    dup = resize_bilinear_backward(g, *up_hw) if up_hw is not None else None
    return groupnorm_backward(x, g, G, gamma, eps) + (dup,)


def maxpool_backward(x, dy, idx=None):
    """gradient of ops.maxpool3x3s2_nhwc.  idx: the arg-max taps of ops.maxpool3x3s2(x, want_idx=True) -- the backward then reads them instead
    of recomputing every window's arg-max from x (x only gives the shape)"""
    ops._chk(dy)
    N, H, W, C = x.shape
    dx = torch.empty((N, H, W, C), device=dy.device, dtype=torch.float32)
    if idx is not None:
        ops._chk(idx, torch.uint8)
        assert tuple(idx.shape) == tuple(dy.shape)
        lib().call("s2d_maxpool3x3s2_backward_idx_nhwc_f32", idx, dy, N, H, W, C, dx, _st())
        return dx
    ops._chk(x)
    lib().call("s2d_maxpool3x3s2_backward_nhwc_f32", x, dy, N, H, W, C, dx, _st())
    return dx


# --------------------------------------------------------------------------- multi-scale deformable attention (fused form)
_MSDA_BWD_REC = os.environ.get("S2D_MSDA_BWD_REC", "1") != "0"


def msda_fused_backward(value, shapes, offs_logits, grad_out, M=8, P=4, merged=False):
    """gradients of ops.msda_fused_forward(value [N,S,C], shapes, offs_logits [N,S,>=288]) -> (d_value [N,S,C],
    d_offs_logits [N,S,M*L*P*3]).  value / offs_logits may be the column slices of the merged projection output (read in place
    through their row stride).  merged=True: both gradients are written into ONE buffer [N,S, M*L*P*3 + C] laid out like the
    merged projection output (offsets | logits | value) and that buffer is returned with them as its column slices --
    -> (d_value view, d_offs_logits view, buffer): the projection's weight / input gradients take it without a concatenation."""
    N, S, C = value.shape
    sh = ops._host_i64(shapes)
    L = sh.shape[0]
    D = C // M
    dev = value.device
    assert value.stride(2) == 1 and value.stride(0) == S * value.stride(1)
    loc = torch.empty((N, S, M, L, P, 2), device=dev, dtype=torch.float32)
    attn = torch.empty((N, S, M, L, P), device=dev, dtype=torch.float32)
    lib().call("s2d_msda_fused_prep_f32", offs_logits, offs_logits.stride(1), sh, N, S, M, L, P, loc, attn, _st())
    lsi = ops._host_i64(torch.cat([torch.zeros(1, dtype=torch.int64), torch.as_tensor(sh).prod(1).cumsum(0)[:-1]]))
    ldd = M * L * P * 3
    if merged:
        buf = torch.empty((N, S, ldd + C), device=dev, dtype=torch.float32)
        doa, gv, ldo, ldg = buf[..., :ldd], buf[..., ldd:], ldd + C, ldd + C
    else:
        buf = None
        doa = torch.empty((N, S, ldd), device=dev, dtype=torch.float32)
        gv = torch.empty((N, S, C), device=dev, dtype=torch.float32)
        ldo, ldg = ldd, C
    go = grad_out.contiguous()
    nb = lib().call("s2d_msda_backward_workspace_bytes", sh, N, M, L, S, P)
    ws = torch.empty((nb,), device=dev, dtype=torch.uint8)
    # the S2D geometry: the query-owned half (d offsets, d logits) in one record-form launch on the raw projection rows, no grad_loc /
    # grad_attn tensors and no chain pass (S2D_MSDA_BWD_REC=0: the two-step form below, which serves every other geometry)
    if _MSDA_BWD_REC and M == 8 and D == 32 and L == 3 and P == 4 and S * value.stride(1) * 4 < 0x7fffffff:
        lib().call("s2d_msda_backward_sorted_strided_f32", value, value.stride(1), sh, lsi, loc, attn, go, N, S, M, D, L, S, P, gv, ldg, None, None,
                   ws, nb, _st())
        lib().call("s2d_msda_fused_backward_query_f32", value, value.stride(1), sh, offs_logits, offs_logits.stride(1), go, N, S, M, D, L, P,
                   doa, ldo, _st())
        return (gv, doa, buf) if merged else (gv, doa)
    gl, ga = torch.empty_like(loc), torch.empty_like(attn)
    lib().call("s2d_msda_backward_sorted_strided_f32", value, value.stride(1), sh, lsi, loc, attn, go, N, S, M, D, L, S, P, gv, ldg, gl, ga,
               ws, nb, _st())
    lib().call("s2d_msda_fused_chain_f32", attn, gl, ga, sh, N, S, M, L, P, doa, ldo, _st())
    return (gv, doa, buf) if merged else (gv, doa)


# --------------------------------------------------------------------------- masked attention
def masked_attn_backward(q, k, v, out, lse, dout, bits=None, unmasked=None, H=8, dk_out=None, dv_out=None):
    """gradients of ops.masked_attn(q, k, v, bits, unmasked) -> (dq [B,Q,C], dk [B,K,C], dv [B,K,C]).  dk_out / dv_out: [B,K,C] views with
    a row stride (column slices of a wider [B,K,n*C] buffer) the key / value gradients are written into instead of fresh tensors"""
    ops._chk(q); ops._chk(out); ops._chk(dout); ops._chk(lse)
    B, Q, C = q.shape
    K = k.shape[1]
    ws = torch.empty((lib().call("s2d_attn_backward_workspace_floats", B, H, K),), device=q.device, dtype=torch.float32)
    dq = torch.empty_like(q)
    dk = dk_out if dk_out is not None else torch.empty((B, K, C), device=q.device, dtype=torch.float32)
    dv = dv_out if dv_out is not None else torch.empty((B, K, C), device=q.device, dtype=torch.float32)
    for t in (dk, dv):
        assert tuple(t.shape) == (B, K, C) and t.stride(2) == 1 and t.stride(0) == K * t.stride(1) and t.dtype == torch.float32
    lib().call("s2d_masked_attn_backward_strided_f32", q, k, v, k.stride(1), v.stride(1), bits, unmasked, out, lse, dout, B, Q, K, C, H, ws, dq,
               dk, dk.stride(1), dv, dv.stride(1), _st())
    return dq, dk, dv

"""`MultiScaleDeformableAttention`: the two functions of the reference's pybind module (ops/src/vision.cpp:18-21,
ops/src/ms_deform_attn.h:25-66), on libs2d_hip.so (s2d_msda_forward_f32 / s2d_msda_backward_f32, csrc/msda.hip).

The two index tensors stay on the GPU, as in the reference: `spatial_shapes` [L,2] and `level_start_index` [L] (int64, built
fresh by the caller on every forward, msdeformattn.py:82-83) go to s2d_msda_forward_dev_f32 / s2d_msda_backward_dev_f32, whose
kernels read them from device memory (ms_deform_attn_cuda.cu:60-75).  Nothing is copied to the host or cached, so a sequence of
clips with different pyramids (MIN_SIZE_TRAIN (360, 480) + random crop) cannot meet stale shapes.

Conventions kept from the CUDA extension (ops/src/cuda/ms_deform_attn_cuda.cu:33-57, :93-116): every tensor must be
contiguous and on the GPU, else RuntimeError; `batch % min(batch, im2col_step) == 0`; outputs are freshly allocated; work is
enqueued on the current stream.  float32 is the path's dtype (the atomic-free kernels); float64 tensors take the extension's other
instantiation (s2d_msda_*_dev_f64: simple kernels for gradient checks in double, ops/test.py).  Difference:
`im2col_step` only takes part in that check -- the kernels need no batch chunking.  There is no torch fallback: a missing
library raises at import."""
import os

import torch

from .. import ops
from .._lib import lib

lib()          # fail at import, loudly, when libs2d_hip.so is absent (ms_deform_attn_func.py:21-29 does the same for the extension)

_DEBUG = bool(int(os.environ.get("S2D_MSDA_CHECK", "0")))    # 1: read the device-side shape check back after every call (syncs)

# Rejected shapes fail loudly WITHOUT a sync: the device-side check of every call also sets this pinned host word (system-scope store,
# s2d_msda_dev_error_word); the host looks at it with a plain memory read on entry to every call and in check(), so a mis-shaped call
# raises at the next call that follows its execution (or at check(), which synchronises) instead of training on zeros.
_WORD = None
_MSG = ("MultiScaleDeformableAttention: a previous call's spatial_shapes / level_start_index did not describe `value` "
        "(H, W > 0, every level inside S, levels not overlapping); that call produced zeros")


def _word():
    global _WORD
    if _WORD is None:
        _WORD = torch.zeros(1, dtype=torch.int32).pin_memory()
        lib().call("s2d_msda_dev_error_word", _WORD)
    return _WORD


def _raise_if_flagged():
    w = _word()
    if int(w[0]) != 0:
        w[0] = 0
        raise RuntimeError(_MSG)


def reset():
    """forget rejected shapes seen so far (other users of the library in this process -- e.g. calls of ops.msda_forward_dev with
    deliberately bad shapes -- set the same process-wide word)"""
    _word()
    torch.cuda.current_stream().synchronize()
    _WORD[0] = 0


def check():
    """synchronise the current stream and raise if any call so far was given shapes that do not describe its `value`"""
    _word()
    torch.cuda.current_stream().synchronize()
    _raise_if_flagged()


def _check(name, t, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} tensor has to be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype} (got {t.dtype})")


def _common(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    dt = value.dtype if isinstance(value, torch.Tensor) and value.dtype == torch.float64 else torch.float32
    _check("value", value, dt); _check("sampling_loc", sampling_loc, dt); _check("attn_weight", attn_weight, dt)
    _check("spatial_shapes", spatial_shapes, torch.int64); _check("level_start_index", level_start_index, torch.int64)
    _raise_if_flagged()
    batch = value.shape[0]
    step = min(batch, int(im2col_step))
    if step <= 0 or batch % step != 0:
        raise RuntimeError(f"batch({batch}) must divide im2col_step({step})")


def _status(ws):
    if _DEBUG and ops.msda_dev_status(ws):
        _word()[0] = 0
        raise RuntimeError("spatial_shapes / level_start_index do not describe `value` (H, W > 0 and every level inside S)")


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """value [N,S,M,D], spatial_shapes [L,2] (H,W), level_start_index [L], sampling_loc [N,Lq,M,L,P,2], attn_weight
    [N,Lq,M,L,P] -> [N,Lq,M*D]"""
    _common(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    if value.dtype == torch.float64:                     # the extension's other instantiation (ops/test.py gradchecks in double)
        N, S, M, D = value.shape
        Lq, L, P = sampling_loc.shape[1], sampling_loc.shape[3], sampling_loc.shape[4]
        out = torch.empty((N, Lq, M * D), device=value.device, dtype=torch.float64)
        lib().call("s2d_msda_forward_dev_f64", value, spatial_shapes, level_start_index, sampling_loc, attn_weight, N, S, M, D, L, Lq, P, out,
                   torch.cuda.current_stream().cuda_stream)
        return out
    out, ws = ops.msda_forward_dev(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, want_ws=True)
    _status(ws)
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight]"""
    _common(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    _check("grad_output", grad_output, value.dtype)
    if value.dtype == torch.float64:
        N, S, M, D = value.shape
        Lq, L, P = sampling_loc.shape[1], sampling_loc.shape[3], sampling_loc.shape[4]
        gv, gl, gw = torch.empty_like(value), torch.empty_like(sampling_loc), torch.empty_like(attn_weight)
        lib().call("s2d_msda_backward_dev_f64", value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, N, S, M, D, L,
                   Lq, P, gv, gl, gw, torch.cuda.current_stream().cuda_stream)
        return [gv, gl, gw]
    gv, gl, gw, ws = ops.msda_backward_dev(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, want_ws=True)
    _status(ws)
    return [gv, gl, gw]

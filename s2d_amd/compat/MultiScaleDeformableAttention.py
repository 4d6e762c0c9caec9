"""`MultiScaleDeformableAttention`: the two functions of the reference's pybind module (ops/src/vision.cpp:18-21,
ops/src/ms_deform_attn.h:25-66), on libs2d_hip.so (s2d_msda_forward_f32 / s2d_msda_backward_f32, csrc/msda.hip).

Conventions kept from the CUDA extension (ops/src/cuda/ms_deform_attn_cuda.cu:33-57, :93-116): every tensor must be
contiguous and on the GPU, else RuntimeError; `batch % min(batch, im2col_step) == 0`; outputs are freshly allocated; work is
enqueued on the current stream.  Differences: float32 only (the path's dtype; the extension also instantiates float64), and
`im2col_step` only takes part in that check -- the kernels need no batch chunking.  There is no torch fallback: a missing
library raises at import."""
import numpy as np
import torch

from .. import ops
from .._lib import lib

lib()          # fail at import, loudly, when libs2d_hip.so is absent (ms_deform_attn_func.py:21-29 does the same for the extension)

_HOST = {}


def _host_i64(t):
    """spatial shapes / level starts are tiny int64 DEVICE tensors in the reference's call; the kernels take them as launch
    parameters.  One device-to-host copy per distinct tensor (keyed by storage, version and shape), not one per call."""
    if not isinstance(t, torch.Tensor):
        return np.ascontiguousarray(t, dtype=np.int64)
    key = (t.data_ptr(), t._version, tuple(t.shape), str(t.device))
    h = _HOST.get(key)
    if h is None:
        if len(_HOST) > 256:
            _HOST.clear()
        h = _HOST[key] = np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.int64)
    return h


def _check(name, t, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} tensor has to be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype} (got {t.dtype})")


def _common(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    _check("value", value); _check("sampling_loc", sampling_loc); _check("attn_weight", attn_weight)
    _check("spatial_shapes", spatial_shapes, None); _check("level_start_index", level_start_index, None)
    batch = value.shape[0]
    step = min(batch, int(im2col_step))
    if step <= 0 or batch % step != 0:
        raise RuntimeError(f"batch({batch}) must divide im2col_step({step})")
    return _host_i64(spatial_shapes), _host_i64(level_start_index)


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """value [N,S,M,D], spatial_shapes [L,2] (H,W), level_start_index [L], sampling_loc [N,Lq,M,L,P,2], attn_weight
    [N,Lq,M,L,P] -> [N,Lq,M*D]"""
    sh, ls = _common(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    return ops.msda_forward(value, sh, ls, sampling_loc, attn_weight)


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight]"""
    sh, ls = _common(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    _check("grad_output", grad_output)
    return list(ops.msda_backward(value, sh, ls, sampling_loc, attn_weight, grad_output))

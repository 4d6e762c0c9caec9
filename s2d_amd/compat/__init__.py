"""Drop-in replacements for the reference's one native FFI: the pybind module
`MultiScaleDeformableAttention` (model_training/mask2former/modeling/pixel_decoder/ops/src/vision.cpp:18-21) and the
autograd function built on it (ops/functions/ms_deform_attn_func.py:32-49), backed by libs2d_hip.so.

    import s2d_amd.compat.MultiScaleDeformableAttention as MSDA          # instead of the CUDA extension
    from s2d_amd.compat.ms_deform_attn_func import MSDeformAttnFunction

`install()` registers the module under the reference's import name, so the reference's own
`ops/functions/ms_deform_attn_func.py` and `ops/modules/ms_deform_attn.py` import and run unmodified."""
import sys


def install():
    from . import MultiScaleDeformableAttention as M
    sys.modules["MultiScaleDeformableAttention"] = M
    return M

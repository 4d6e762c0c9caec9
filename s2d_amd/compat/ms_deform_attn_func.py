"""`MSDeformAttnFunction` with the reference's call signature (ops/functions/ms_deform_attn_func.py:32-49):

    out = MSDeformAttnFunction.apply(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step)

value, sampling_locations and attention_weights receive gradients; the index tensors and im2col_step do not."""
import torch
from torch.autograd.function import once_differentiable

from . import MultiScaleDeformableAttention as MSDA


class MSDeformAttnFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
        return MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                           attention_weights, im2col_step)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, starts, loc, attn = ctx.saved_tensors
        g_value, g_loc, g_attn = MSDA.ms_deform_attn_backward(value, shapes, starts, loc, attn, grad_output.contiguous(),
                                                              ctx.im2col_step)
        return g_value, None, None, g_loc, g_attn, None

"""R50 trunk, host side.  Mirrors detectron2's `build_resnet_backbone` (R-50, STRIDE_IN_1X1 False, FrozenBN,
out_features res2..res5; selected by MODEL.BACKBONE.NAME in
configs/imagenet_video/Base-YouTubeVIS-VideoInstanceSegmentation.yaml:2-16, built at
kd_video_maskformer_model.py:132,135).  detectron2 is not under /root/reference: parameter names follow its public
R-50 checkpoint layout (stem.conv1.*, res2.0.conv1.*, ...shortcut.*, ...norm.{weight,bias,running_mean,running_var})
so `student.0.*` / `backbone.*` checkpoints load unchanged; numerics are "parity unpinned" (DESIGN.md).

Compute: every conv is one launch of the fp32-MFMA implicit-GEMM kernel on NHWC activations with FrozenBN folded
into the epilogue's per-channel scale/shift, ReLU and the residual add fused.
"""
import torch
from torch import nn

from .. import ops

R50_STAGES = (("res2", 3, 64, 256, 1), ("res3", 4, 128, 512, 2), ("res4", 6, 256, 1024, 2), ("res5", 3, 512, 2048, 2))


class FrozenBatchNorm2d(nn.Module):
    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(c))
        self.register_buffer("bias", torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))

    def fold(self):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        return scale.contiguous(), (self.bias - self.running_mean * scale).contiguous()


class ConvBN(nn.Module):
    """detectron2 Conv2d(bias=False, norm=FrozenBN): parameter `weight` [O,C,kh,kw] + submodule `norm`."""

    def __init__(self, cin, cout, k, stride, pad):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")
        self.norm = FrozenBatchNorm2d(cout)
        self.stride, self.pad, self.k = stride, pad, k
        self._packed = None

    def packed(self):
        v = ops.version_of
        key = (v(self.weight), self.weight.device, v(self.norm.weight), v(self.norm.running_var), v(self.norm.bias), v(self.norm.running_mean))
        if self._packed is None or self._packed[0] != key:
            w = self.weight.detach().permute(0, 2, 3, 1)             # [O,kh,kw,C]
            if w.shape[-1] % 4:
                w = torch.nn.functional.pad(w, (0, 4 - w.shape[-1] % 4))  # stem: Cin 3 -> 4 (input is NHWC4)
            nkey = key[1:]                                                # the frozen statistics do not change with the weight:
            if getattr(self, "_fold", None) is None or self._fold[0] != nkey:   # fold them once, not after every optimizer step
                scale, shift = self.norm.fold()
                self._fold = (nkey, scale.float(), shift.float())
            scale, shift = self._fold[1], self._fold[2]
            # own storage (1x1 kernels: permute + contiguous is still a view of the parameter), refreshed in place after an optimizer step
            self._packed = (key, ops.repack(None if self._packed is None else self._packed[1], w.float()), scale, shift)
        return self._packed[1:]

    def packed_scaled(self):
        """the packed weight with the folded BatchNorm scale multiplied into its output channels (diag(scale) . W): the dgrad operand
        when the arriving gradient is d(pre-activation sum), not yet multiplied by the scale"""
        w, scale, _ = self.packed()
        key = self._packed[0]
        ws = getattr(self, "_ws", None)
        if ws is None or ws[0] != key:
            self._ws = ws = (key, ops.repack(None if ws is None else ws[1], w * scale.view(-1, 1, 1, 1)))
        return ws[1]

    def forward(self, x, res=None, relu=True):
        w, scale, shift = self.packed()
        if self.k == 1 and self.stride == 1:
            N, H, W, C = x.shape
            y = ops.gemm_nt(x.view(-1, C), w.view(w.shape[0], -1), scale, shift,
                            None if res is None else res.view(-1, w.shape[0]), relu)
            return y.view(N, H, W, -1)
        return ops.conv2d_nhwc(x, w, self.stride, self.pad, scale, shift, res, relu)

    def backward(self, x, y, dy, relu=True, has_res=False, need_dx=True, dx_res=None, pre_gated=False, in_scale=None, out_gated=False,
                 gate_x=False):
        """gradients of y = act(conv(x) * scale + shift [+ res]): accumulates into self.weight.grad and returns
        (dx [+ dx_res: a gradient arriving at x over another path, added in the dgrad epilogue for 1x1 kernels], dres).
        pre_gated: dy is already d(conv output) (the consumer's dgrad applied this layer's ReLU gate and scale in its epilogue).
        in_scale: x is the ReLU output of a ConvBN with that folded scale; the returned dx is then d(that convolution's output),
        gated by x > 0 and scaled in the dgrad epilogue (one pass over the activation gradient less).
        gate_x: x is a ReLU output (a block output): the returned dx is gated by x > 0 in the dgrad epilogue, not scaled.
        out_gated: dy is d(pre-ReLU sum) already (the consumer of y gated it, gate_x there) but NOT multiplied by this layer's folded
        scale: the scale goes into the dgrad's weight (packed_scaled) and onto the rows of the weight gradient, so the three-pass
        gate / scale kernel over the block output's gradient disappears."""
        from .. import backward as B
        w, scale, _ = self.packed()
        if out_gated:
            dz, dres = dy, (dy if has_res else None)
            w = self.packed_scaled()
        elif pre_gated:
            dz, dres = dy, None
        elif has_res and relu:
            dz, dres = B.relu_scale_backward(dy, y, scale, want_res=True)       # d(conv output), d(residual): one pass
        else:
            dres = dy if has_res else None
            dz = B.relu_scale_backward(dy, y if relu else None, scale)
        dw = B.conv_weight_grad(dz, x, self.k, self.k, self.stride, self.pad)  # [O,kh,kw,C(4)]
        if out_gated:
            dw = dw * scale.view(-1, 1, 1, 1)                                  # dz = dy * scale enters the weight gradient linearly
        dw = dw[..., :self.weight.shape[1]].permute(0, 3, 1, 2)
        if self.weight.grad is None:
            self.weight.grad = dw.contiguous()
        else:
            self.weight.grad += dw
        dx = None
        if need_dx:
            gate = x if (in_scale is not None or gate_x) else None
            if self.k == 1 and self.stride == 1:
                N, H, W, C = x.shape
                dx = B.input_grad(dz.view(-1, dz.shape[-1]), w.view(w.shape[0], -1), None if dx_res is None else dx_res.view(-1, C),
                                  gate=gate, scale=in_scale).view(N, H, W, C)
            else:
                dx = B.conv_input_grad(dz, w, self.stride, self.pad, x.shape[1:3], gate=gate, scale=in_scale)
                if dx_res is not None:
                    dx = dx + dx_res
        return dx, dres


class BasicStem(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = ConvBN(3, 64, 7, 2, 3)

    def forward(self, x, tape=None):
        c = self.conv1(x)
        if tape is not None:
            y, idx = ops.maxpool3x3s2(c, want_idx=True)               # the arg-max taps: the backward routes by them
            tape.append((self, x, c, idx))
            return y
        return ops.maxpool3x3s2(c)

    def backward(self, saved, dy):
        from .. import backward as B
        _, x, c, idx = saved
        self.conv1.backward(x, c, B.maxpool_backward(c, dy, idx), relu=True, need_dx=False)   # the image needs no gradient


class BottleneckBlock(nn.Module):
    def __init__(self, cin, mid, cout, stride):
        super().__init__()
        self.shortcut = ConvBN(cin, cout, 1, stride, 0) if cin != cout else None
        self.conv1 = ConvBN(cin, mid, 1, 1, 0)
        self.conv2 = ConvBN(mid, mid, 3, stride, 1)     # STRIDE_IN_1X1 False: stride on the 3x3
        self.conv3 = ConvBN(mid, cout, 1, 1, 0)

    def forward(self, x, tape=None):
        sc = x if self.shortcut is None else self.shortcut(x, relu=False)
        y1 = self.conv1(x)
        y2 = self.conv2(y1)
        out = self.conv3(y2, res=sc, relu=True)
        if tape is not None:
            tape.append((self, x, sc, y1, y2, out))
        return out

    def backward(self, saved, dout, gated=False, gate_in=False):
        """d(block input) from d(block output); parameter gradients accumulate in the ConvBN weights.
        gated: dout is already multiplied by (out > 0) (the consuming block's conv1 dgrad did it, gate_in there).
        gate_in: x is the previous block's ReLU output: return dx * (x > 0) (in conv1's dgrad epilogue)."""
        _, x, sc, y1, y2, out = saved
        # conv3's and conv2's dgrads gate by the ReLU output they differentiate through and apply that layer's folded scale in
        # their epilogues: d2 / d1 arrive as d(conv2 output) / d(conv1 output)
        d2, dsc = self.conv3.backward(y2, out, dout, relu=True, has_res=True, in_scale=self.conv2.packed()[1], out_gated=gated)
        d1, _ = self.conv2.backward(y1, y2, d2, pre_gated=True, in_scale=self.conv1.packed()[1])
        if self.shortcut is not None:
            dsc, _ = self.shortcut.backward(x, sc, dsc, relu=False, out_gated=True)        # no activation: only the folded scale, taken into the weights
        dx, _ = self.conv1.backward(x, y1, d1, pre_gated=True, dx_res=dsc, gate_x=gate_in)   # both paths meet at x
        return dx


class ResNet50(nn.Module):
    """forward(x NHWC4 float32 [N,H,W,4]) -> {"res2".."res5": NHWC}"""

    size_divisibility = 32

    def __init__(self):
        super().__init__()
        self.stem = BasicStem()
        cin = 64
        for name, nblk, mid, cout, stride in R50_STAGES:
            blocks = []
            for b in range(nblk):
                blocks.append(BottleneckBlock(cin, mid, cout, stride if b == 0 else 1))
                cin = cout
            setattr(self, name, nn.Sequential(*blocks))

    def output_shape(self):
        return {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}

    amp = False      # True: the convolutions take torch.autocast's arithmetic (fp16 operands, f32 accumulate), as the trunk does
                     # under the reference trainer's `with autocast():` (engine/train_loop.py:709); forward / loss only

    def forward(self, x, tape=None):
        """tape: a list that receives what backward() needs (the activations the reference's autograd would keep)"""
        with ops.amp_fp16(self.amp and tape is None):
            y = self.stem(x, tape)
            out = {}
            for name, *_ in R50_STAGES:
                for blk in getattr(self, name):
                    y = blk(y, tape)
                out[name] = y
        return out

    def backward(self, tape, grads):
        """grads: {"res2".."res5": d(loss)/d(output)} (missing = zero).  Walks the tape backwards, accumulating the weight
        gradients of all 53 convolutions (FREEZE_AT 0: the whole trunk trains, Base-YouTubeVIS...yaml:3)."""
        from .. import backward as B
        last = {getattr(self, name)[-1]: name for name, *_ in R50_STAGES}
        first_block = getattr(self, R50_STAGES[0][0])[0]
        d, gated = None, False
        for saved in reversed(tape):
            mod = saved[0]
            if isinstance(mod, BasicStem):
                mod.backward(saved, d)
                continue
            g = grads.get(last.get(mod))
            if g is not None:
                if d is None:
                    d, gated = g, False
                elif gated:
                    d = B.relu_gate_add(d, g, saved[5])               # d arrives gated by this block's output: gate the stage output's own gradient too (one pass)
                else:
                    d = d + g
            # every block but the first hands its input gradient back gated by that input (the previous block's ReLU output), in its conv1
            # dgrad's epilogue: the previous block then needs no gate / scale pass over its 0.2-1 GB output gradient
            gate_in = mod is not first_block
            d = mod.backward(saved, d, gated=gated, gate_in=gate_in)
            gated = gate_in


def build_resnet_backbone(cfg=None, input_shape=None):
    return ResNet50()

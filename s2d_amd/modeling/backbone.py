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
        key = (self.weight._version, self.weight.device, self.norm.weight._version, self.norm.running_var._version,
               self.norm.bias._version, self.norm.running_mean._version)
        if self._packed is None or self._packed[0] != key:
            w = self.weight.detach().permute(0, 2, 3, 1)             # [O,kh,kw,C]
            if w.shape[-1] % 4:
                w = torch.nn.functional.pad(w, (0, 4 - w.shape[-1] % 4))  # stem: Cin 3 -> 4 (input is NHWC4)
            scale, shift = self.norm.fold()
            w = w.contiguous().float()
            if w._base is not None:          # 1x1 kernels: permute + contiguous is still a view of the parameter; own the
                w = w.clone()                # storage so that views of the packed weight resolve to this (marked) tensor
            self._packed = (key, ops.mark_static(w), scale.float(), shift.float())
        return self._packed[1:]

    def forward(self, x, res=None, relu=True):
        w, scale, shift = self.packed()
        if self.k == 1 and self.stride == 1:
            N, H, W, C = x.shape
            y = ops.gemm_nt(x.view(-1, C), w.view(w.shape[0], -1), scale, shift,
                            None if res is None else res.view(-1, w.shape[0]), relu)
            return y.view(N, H, W, -1)
        return ops.conv2d_nhwc(x, w, self.stride, self.pad, scale, shift, res, relu)


class BasicStem(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = ConvBN(3, 64, 7, 2, 3)

    def forward(self, x):
        return ops.maxpool3x3s2(self.conv1(x))


class BottleneckBlock(nn.Module):
    def __init__(self, cin, mid, cout, stride):
        super().__init__()
        self.shortcut = ConvBN(cin, cout, 1, stride, 0) if cin != cout else None
        self.conv1 = ConvBN(cin, mid, 1, 1, 0)
        self.conv2 = ConvBN(mid, mid, 3, stride, 1)     # STRIDE_IN_1X1 False: stride on the 3x3
        self.conv3 = ConvBN(mid, cout, 1, 1, 0)

    def forward(self, x):
        sc = x if self.shortcut is None else self.shortcut(x, relu=False)
        y = self.conv2(self.conv1(x))
        return self.conv3(y, res=sc, relu=True)


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

    def forward(self, x):
        y = self.stem(x)
        out = {}
        for name, *_ in R50_STAGES:
            y = getattr(self, name)(y)
            out[name] = y
        return out


def build_resnet_backbone(cfg=None, input_shape=None):
    return ResNet50()

"""MSDeformAttn pixel decoder, host side.  Mirrors
model_training/mask2former/modeling/pixel_decoder/msdeformattn.py (MSDeformAttnPixelDecoder :164-358,
MSDeformAttnTransformerEncoderOnly :22-89, encoder layer :92-131) and ops/modules/ms_deform_attn.py:34-125, with the
reference's parameter names and shapes so its checkpoints load unchanged.

Activations stay NHWC == the reference's [N, S, C] token layout, so flatten/transpose/split are views.  Per encoder
layer: one GEMM produces sampling offsets + attention logits together (weights concatenated at pack time), one GEMM
the value projection, the fused MSDA kernel does softmax + location arithmetic + bilinear gather, the output
projection GEMM adds the residual in its epilogue, LayerNorm follows; FFN likewise.
"""
import math
import os

import torch
from torch import nn

from .. import ops


class MSDeformAttn(nn.Module):
    """Parameter container + drop-in forward of ops/modules/ms_deform_attn.py:34-125."""

    def __init__(self, d_model=256, n_levels=3, n_heads=8, n_points=4):
        super().__init__()
        self.im2col_step = 128
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()
        self._packed = None

    def _reset_parameters(self):  # ms_deform_attn.py:66-80
        nn.init.constant_(self.sampling_offsets.weight.data, 0.)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid.view(-1))
        nn.init.constant_(self.attention_weights.weight.data, 0.)
        nn.init.constant_(self.attention_weights.bias.data, 0.)
        nn.init.xavier_uniform_(self.value_proj.weight.data)
        nn.init.constant_(self.value_proj.bias.data, 0.)
        nn.init.xavier_uniform_(self.output_proj.weight.data)
        nn.init.constant_(self.output_proj.bias.data, 0.)

    def packed(self):
        """[sampling_offsets | attention_weights | value_proj] as one [288 + C, C] weight (one pass over src), with
        bias [0 | b_value]: the offsets/logits bias travels in the row-periodic pos term of forward_fused."""
        ps = (self.sampling_offsets.weight, self.sampling_offsets.bias, self.attention_weights.weight, self.attention_weights.bias,
              self.value_proj.weight, self.value_proj.bias)
        key = tuple(ops.version_of(p) for p in ps) + (ps[0].device,)
        if self._packed is None or self._packed[0] != key:
            w_oa = torch.cat([ps[0].detach(), ps[2].detach()], 0)
            b_oa = torch.cat([ps[1].detach(), ps[3].detach()], 0)
            w_all = torch.cat([w_oa, ps[4].detach()], 0)
            b_all = torch.cat([torch.zeros_like(b_oa), ps[5].detach()], 0)
            prev = (None,) * 5 if self._packed is None else self._packed            # refreshed in place after an optimizer step (ops.repack)
            self._packed = (key, ops.repack(prev[1], w_all), ops.repack(prev[2], b_all), ops.repack(prev[3], w_oa), ops.repack(prev[4], b_oa))
        return self._packed[1:]

    def projection(self, pos):
        """what the previous layer's fused FFN launch needs to apply this layer's merged projection to its output rows (ops.ffn_fused
        `post`): (w_all [288 + C, C], b_all, pos . W_oa^T + b_oa [S, 288])"""
        S, C = pos.shape[-2:]
        w_all, b_all, w_oa, b_oa = self.packed()
        return w_all, b_all, ops.gemm_nt(pos.view(S, C), w_oa, bias=b_oa)

    def forward_fused(self, src, pos, shapes, res, tape=None, dropout=None, both=None):
        """self-attention over the flattened pyramid with query = src + pos (query positions == value positions;
        pos [S, C] is shared by all N frames).  Returns dropout(output_proj(msda(...))) + res (dropout = (p, seed, site) or
        None: the encoder layer's dropout1, msdeformattn.py:125, fused into the projection's epilogue).
        (src + pos) . W_oa^T = src . W_oa^T + pos . W_oa^T: the second term is one small [S, 288] product per layer and
        enters the merged projection as a row-periodic residual, so src is read once and src + pos is never stored."""
        N, S, C = src.shape
        w_all, b_all, w_oa, b_oa = self.packed()
        n_oa = w_oa.shape[0]
        if both is None:
            pos_oa = ops.gemm_nt(pos.view(S, C), w_oa, bias=b_oa)
            both = ops.gemm_nt(src.view(-1, C), w_all, bias=b_all, res=pos_oa, res_rows=S, res_cols=n_oa)
        both = both.view(N, S, n_oa + C)          # (given: the previous layer's FFN launch already projected its output rows)
        samp = ops.msda_fused_forward(both[..., n_oa:], shapes, both[..., :n_oa], self.n_heads, self.n_points)
        if res is None:                           # the caller applies output_proj + dropout1 + residual itself (the fused FFN launch)
            return samp
        out = ops.gemm_nt(samp.view(-1, C), self.output_proj.weight, bias=self.output_proj.bias, res=res.view(-1, C), dropout=dropout)
        if tape is not None:
            tape.append((self, src, pos, shapes, both, samp, dropout))
        return out.view(N, S, C)

    def backward_fused(self, saved, d_out):
        """d(src) (the attention input, including the residual path res == src) and d(pos) [S,C] from d(output);
        parameter gradients accumulate into .grad.  Mirrors forward_fused step by step."""
        from .. import backward as B
        _, src, pos, shapes, both, samp, drop = saved
        N, S, C = src.shape
        w_all, _, w_oa, _ = self.packed()
        n_oa = w_oa.shape[0]
        n_off = self.sampling_offsets.weight.shape[0]
        d2 = d_out.reshape(-1, C)                                         # d(src + dropout1(proj)): the residual path takes it as is
        dm = d2 if drop is None or drop[0] <= 0.0 else ops.dropout(d2, *drop)   # d(proj): the forward's mask, regenerated
        B.acc_wbgrad(self.output_proj.weight, self.output_proj.bias, dm, samp.view(-1, C))
        d_samp = B.input_grad(dm, self.output_proj.weight).view(N, S, C)
        # both gradients land in one buffer laid out like the merged projection's output: no concatenation, no copy of the value slice
        d_val, d_oa, d_both = B.msda_fused_backward(both[..., n_oa:], shapes, both[..., :n_oa], d_samp, self.n_heads, self.n_points, merged=True)
        d_both = d_both.view(-1, n_oa + C)
        db = torch.empty((n_oa + C,), device=src.device, dtype=torch.float32)
        dw = B.weight_grad(d_both, src.view(-1, C), bias_out=db)          # rows: offsets | logits | value; the bias gradient in the same pass
        d_pos_oa = B.sum_slices(d_both.view(N, S, n_oa + C))[:, :n_oa].contiguous()   # the row-periodic term: same pos row for every frame
        dw_pos = B.weight_grad(d_pos_oa.contiguous(), pos.view(S, C))
        B.acc(self.sampling_offsets.weight, dw[:n_off] + dw_pos[:n_off])
        B.acc(self.attention_weights.weight, dw[n_off:n_oa] + dw_pos[n_off:])
        B.acc(self.value_proj.weight, dw[n_oa:])
        B.acc(self.sampling_offsets.bias, db[:n_off])
        B.acc(self.attention_weights.bias, db[n_off:n_oa])
        B.acc(self.value_proj.bias, db[n_oa:])
        d_src = B.input_grad(d_both, w_all, res=d2).view(N, S, C)         # + the residual path (res == src)
        d_pos = B.input_grad(d_pos_oa.contiguous(), w_oa)
        return d_src, d_pos

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        """The reference's general signature (ms_deform_attn.py:82-125), built from the drop-in MSDA op."""
        N, Lq, C = query.shape
        S = input_flatten.shape[1]
        M, L, P = self.n_heads, self.n_levels, self.n_points
        value = ops.gemm_nt(input_flatten.reshape(-1, C).contiguous(), self.value_proj.weight, bias=self.value_proj.bias)
        if input_padding_mask is not None:
            value = value.view(N, S, C).masked_fill(input_padding_mask[..., None], 0.0)
        value = value.view(N, S, M, C // M)
        q2 = query.reshape(-1, C).contiguous()
        off = ops.gemm_nt(q2, self.sampling_offsets.weight, bias=self.sampling_offsets.bias).view(N, Lq, M, L, P, 2)
        aw = ops.gemm_nt(q2, self.attention_weights.weight, bias=self.attention_weights.bias).view(N, Lq, M, L * P)
        aw = torch.softmax(aw, -1).view(N, Lq, M, L, P)
        shp = torch.as_tensor(input_spatial_shapes, device=query.device)
        norm = torch.stack([shp[..., 1], shp[..., 0]], -1).to(query.dtype)
        loc = reference_points[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
        out = ops.msda_forward(value.contiguous(), input_spatial_shapes, input_level_start_index, loc.contiguous(), aw.contiguous())
        return ops.gemm_nt(out.view(-1, C), self.output_proj.weight, bias=self.output_proj.bias).view(N, Lq, C)


class MSDeformAttnTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.0, n_levels=3, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout_p = dropout

    fuse_ffn = os.environ.get("S2D_FUSE_FFN", "1") != "0"      # False: the two-launch FFN + separate LayerNorms also on the forward-only path (A/B measurements, tests)
    fuse_pre = os.environ.get("S2D_FUSE_PRE", "1") != "0"       # False: output_proj + dropout1 + residual stay the attention's own GEMM launch (A/B runs)
    fuse_next = os.environ.get("S2D_FUSE_NEXT", "1") != "0"     # False: the next layer's merged projection stays its own launch (A/B runs)

    def forward(self, src, pos, shapes, tape=None, both=None, nxt=None):
        """msdeformattn.py:116-131 (post-norm).  src [N,S,C], pos [S,C].
        src = norm1(src + dropout1(attn));  src = norm2(src + dropout3(linear2(dropout2(relu(linear1(src))))))  (:101-125).
        In training mode with p > 0 the three masks are counter-based (csrc/dropout.h: one Philox key per call, sites 0..2)
        and applied in the epilogues of the three GEMMs; the backward regenerates them.  Eval mode: identity, as nn.Dropout."""
        N, S, C = src.shape
        p = float(self.dropout_p) if self.training else 0.0
        seed = ops.next_dropout_seed() if p > 0.0 else 0
        d1, d2, d3 = ((p, seed, 0), (p, seed, 1), (p, seed, 2)) if p > 0.0 else (None, None, None)
        sub = [] if tape is not None else None
        fused = tape is None and self.fuse_ffn and ops.ffn_fusable(self.linear1.weight, self.linear2.weight)
        pre = None
        if fused and self.fuse_pre:
            # ... and the attention's output projection, dropout1 and residual in front of norm1 too: the launch reads the sampled values
            samp = self.self_attn.forward_fused(src, pos, shapes, res=None, both=both)
            pre = (self.self_attn.output_proj.weight, self.self_attn.output_proj.bias, src.view(-1, C), 0)
            x1 = samp
        else:
            x1 = self.self_attn.forward_fused(src, pos, shapes, res=src, tape=sub, dropout=d1, both=both)
        if fused:
            # forward / loss path (nothing kept for a backward): norm1, the whole FFN with both masks, the residual and norm2 in ONE
            # launch (csrc/ffn.hip) -- the 1024-wide hidden activation never reaches memory.  Same masks as the taped path below.
            # nxt (the next layer's attention module): its merged projection of this layer's output rows rides in the same launch;
            # returns (output, that projection) then.
            out = ops.ffn_fused(x1.view(-1, C), self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                                ln1=(self.norm1.weight, self.norm1.bias), ln2=(self.norm2.weight, self.norm2.bias),
                                dropout=(p, seed, 1, 2) if p > 0.0 else None, eps=self.norm1.eps,
                                post=nxt.projection(pos) if nxt is not None else None, pre=pre)
            return (out[0].view(N, S, C), out[1]) if nxt is not None else out.view(N, S, C)
        s1 = ops.layernorm(x1, self.norm1.weight, self.norm1.bias)
        h = ops.gemm_nt(s1.view(-1, C), self.linear1.weight, bias=self.linear1.bias, relu=True, dropout=d2)
        x2 = ops.gemm_nt(h, self.linear2.weight, bias=self.linear2.bias, res=s1.view(-1, C), dropout=d3).view(N, S, C)
        if tape is not None:
            tape.append((self, sub[0], x1, s1, h, x2, (p, seed)))
        return ops.layernorm(x2, self.norm2.weight, self.norm2.bias)

    def backward(self, saved, d_out):
        """-> (d_src, d_pos)"""
        from .. import backward as B
        _, attn_saved, x1, s1, h, x2, (p, seed) = saved
        N, S, C = x1.shape
        d_x2, dg, db = B.layernorm_backward(x2, d_out.contiguous(), self.norm2.weight)
        B.acc(self.norm2.weight, dg); B.acc(self.norm2.bias, db)
        d2 = d_x2.view(-1, C)                                           # d(s1 + dropout3(linear2)): the residual takes it as is
        dm = ops.dropout(d2, p, seed, 2) if p > 0.0 else d2               # d(linear2 output)
        B.acc_wbgrad(self.linear2.weight, self.linear2.bias, dm, h)
        # h = dropout2(relu(z)) = relu(z) * m / P(keep): positive exactly where the unit is active AND kept, so the ReLU
        # gate on h is the combined gate and the dropout factor is a constant per-channel scale
        d_h = B.input_grad(dm, self.linear2.weight, gate=h, gate_scale=ops.dropout_scale(p) if p > 0.0 else 1.0)
        B.acc_wbgrad(self.linear1.weight, self.linear1.bias, d_h, s1.view(-1, C))
        d_s1 = B.input_grad(d_h, self.linear1.weight, res=d2).view(N, S, C)              # + the FFN residual
        d_x1, dg, db = B.layernorm_backward(x1, d_s1, self.norm1.weight)
        B.acc(self.norm1.weight, dg); B.acc(self.norm1.bias, db)
        return self.self_attn.backward_fused(attn_saved, d_x1)


class _Encoder(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)


class MSDeformAttnTransformerEncoderOnly(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, dim_feedforward=1024, dropout=0.0,
                 num_feature_levels=3, enc_n_points=4):
        super().__init__()
        self.encoder = _Encoder([MSDeformAttnTransformerEncoderLayer(d_model, dim_feedforward, dropout, num_feature_levels,
                                                                      nhead, enc_n_points) for _ in range(num_encoder_layers)])
        self.level_embed = nn.Parameter(torch.empty(num_feature_levels, d_model))
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        nn.init.normal_(self.level_embed)


class _ConvGN(nn.Module):
    """detectron2 Conv2d(bias=False, norm=GroupNorm(32)) parameter container: `weight` + `norm.{weight,bias}`."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_uniform_(self.weight, a=1)  # fvcore c2_xavier_fill
        self.norm = nn.GroupNorm(32, cout)
        self._packed = None

    def packed(self):
        key = (ops.version_of(self.weight), self.weight.device)
        if self._packed is None or self._packed[0] != key:
            w = self.weight.detach().permute(0, 2, 3, 1)
            self._packed = (key, ops.repack(None if self._packed is None else self._packed[1], w))
        return self._packed[1]


class MSDeformAttnPixelDecoder(nn.Module):
    """forward_features(features: dict res2..res5 NHWC) ->
    (mask_features [BT,h2,w2,C] NHWC, multi_scale [(tokens [BT,h*w,C], (h,w)) x3: res5,res4,res3 scale])"""

    def __init__(self, conv_dim=256, mask_dim=256, transformer_dropout=0.0, transformer_nheads=8,
                 transformer_dim_feedforward=1024, transformer_enc_layers=6, in_channels=(2048, 1024, 512),
                 res2_channels=256):
        super().__init__()
        self.conv_dim = conv_dim
        self.input_proj = nn.ModuleList([
            nn.Sequential(nn.Conv2d(c, conv_dim, kernel_size=1), nn.GroupNorm(32, conv_dim)) for c in in_channels])
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)
        self.transformer = MSDeformAttnTransformerEncoderOnly(conv_dim, transformer_nheads, transformer_enc_layers,
                                                              transformer_dim_feedforward, transformer_dropout, 3)
        self.mask_features = nn.Conv2d(conv_dim, mask_dim, kernel_size=1)
        nn.init.kaiming_uniform_(self.mask_features.weight, a=1)
        nn.init.constant_(self.mask_features.bias, 0)
        self.adapter_1 = _ConvGN(res2_channels, conv_dim, 1)
        self.layer_1 = _ConvGN(conv_dim, conv_dim, 3)
        self._pos_cache = {}

    @classmethod
    def from_config(cls, cfg, input_shape=None):  # msdeformattn.py:294-312
        return cls(conv_dim=cfg.MODEL.SEM_SEG_HEAD.CONVS_DIM, mask_dim=cfg.MODEL.SEM_SEG_HEAD.MASK_DIM,
                   transformer_dropout=cfg.MODEL.MASK_FORMER.DROPOUT, transformer_nheads=cfg.MODEL.MASK_FORMER.NHEADS,
                   transformer_dim_feedforward=1024,
                   transformer_enc_layers=cfg.MODEL.SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS)

    def _pos(self, shapes, device):
        le = self.transformer.level_embed
        key = (tuple(shapes), ops.version_of(le), device)
        if self._pos_cache.get("key") != key:
            pos = [ops.pe_sine(0, h, w, self.conv_dim // 2, add_c=le[i].detach().contiguous(), device=device)
                   for i, (h, w) in enumerate(shapes)]
            self._pos_cache = {"key": key, "pos": torch.cat(pos, 0).contiguous()}
        return self._pos_cache["pos"]

    @torch.no_grad()
    def forward_features(self, features, tape=None):
        C = self.conv_dim
        srcs, shapes, proj = [], [], []
        for idx, f in enumerate(("res5", "res4", "res3")):   # msdeformattn.py:319-322
            x = features[f]
            N, h, w, cin = x.shape
            conv, gn = self.input_proj[idx][0], self.input_proj[idx][1]
            z = ops.gemm_nt(x.view(-1, cin), conv.weight.view(C, cin), bias=conv.bias).view(N, h, w, C)
            y = ops.groupnorm_nhwc(z, 32, gn.weight, gn.bias, eps=gn.eps)
            proj.append((x, z))
            srcs.append(y.view(N, h * w, C))
            shapes.append((h, w))
        src = torch.cat(srcs, 1).contiguous()
        pos = self._pos(shapes, src.device)
        shp = torch.tensor(shapes, dtype=torch.int64)
        enc = [] if tape is not None else None
        layers = list(self.transformer.encoder.layers)
        both = None
        for i, layer in enumerate(layers):
            # forward-only path: layer i's FFN launch also applies layer i + 1's merged projection to its output rows
            nxt = layers[i + 1].self_attn if (enc is None and i + 1 < len(layers) and layer.fuse_ffn and layer.fuse_next
                                              and ops.ffn_fusable(layer.linear1.weight, layer.linear2.weight)) else None
            out = layer(src, pos, shp, enc, both=both, nxt=nxt)
            src, both = out if nxt is not None else (out, None)
        N = src.shape[0]
        outs, o = [], 0
        for (h, w) in shapes:
            outs.append(src[:, o:o + h * w].contiguous())
            o += h * w
        # extra FPN level on res2 (msdeformattn.py:343-351)
        x = features["res2"]
        _, h2, w2, c2 = x.shape
        cur = ops.gemm_nt(x.view(-1, c2), self.adapter_1.weight.view(C, c2)).view(N, h2, w2, C)
        h3, w3 = shapes[-1]
        y1 = ops.groupnorm_nhwc(cur, 32, self.adapter_1.norm.weight, self.adapter_1.norm.bias, up=outs[-1].view(N, h3, w3, C),
                                eps=self.adapter_1.norm.eps)
        y2 = ops.conv2d_nhwc(y1, self.layer_1.packed(), 1, 1)
        y3 = ops.groupnorm_nhwc(y2, 32, self.layer_1.norm.weight, self.layer_1.norm.bias, relu=True, eps=self.layer_1.norm.eps)
        mf = ops.gemm_nt(y3.view(-1, C), self.mask_features.weight.view(-1, C), bias=self.mask_features.bias).view(N, h2, w2, -1)
        if tape is not None:
            tape.append((self, proj, shapes, enc, x, cur, y1, y2, y3))
        return mf, [(outs[i], shapes[i]) for i in range(3)]

    @torch.no_grad()
    def backward_features(self, saved, d_mf, d_outs):
        """gradients of forward_features: d_mf [N,h2,w2,mask_dim], d_outs: 3 x [N,h*w,C] (None = zero) ->
        {"res2".."res5": d(feature)}; parameter gradients accumulate into .grad (msdeformattn.py:314-358 backwards)."""
        from .. import backward as B
        _, proj, shapes, enc, x2, cur, y1, y2, y3 = saved
        C = self.conv_dim
        N, h2, w2, c2 = x2.shape
        h3, w3 = shapes[-1]
        grads = {}
        # mask_features 1x1 conv <- layer_1 (3x3 conv, GN, ReLU) <- adapter_1 (1x1 conv, GN) + upsampled level-2 tokens
        d = d_mf.reshape(-1, d_mf.shape[-1])
        B.acc_wgrad(self.mask_features.weight, d, y3.view(-1, C))
        B.acc_bgrad(self.mask_features.bias, d)
        d_y3 = B.input_grad(d, self.mask_features.weight.view(-1, C)).view(N, h2, w2, C)
        d_y2, dg, db, _ = B.groupnorm_up_relu_backward(y2, y3, d_y3, 32, self.layer_1.norm.weight, relu=True, eps=self.layer_1.norm.eps)
        B.acc(self.layer_1.norm.weight, dg); B.acc(self.layer_1.norm.bias, db)
        B.acc(self.layer_1.weight, B.conv_weight_grad(d_y2, y1, 3, 3, 1, 1).permute(0, 3, 1, 2))
        d_y1 = B.conv_input_grad(d_y2, self.layer_1.packed(), 1, 1, (h2, w2))
        d_cur, dg, db, d_up = B.groupnorm_up_relu_backward(cur, y1, d_y1, 32, self.adapter_1.norm.weight, up_hw=(h3, w3),
                                                           eps=self.adapter_1.norm.eps)
        B.acc(self.adapter_1.norm.weight, dg); B.acc(self.adapter_1.norm.bias, db)
        dc = d_cur.view(-1, C)
        B.acc_wgrad(self.adapter_1.weight, dc, x2.view(-1, c2))
        grads["res2"] = B.input_grad(dc, self.adapter_1.weight.view(C, c2)).view(N, h2, w2, c2)
        # encoder output tokens: the decoder's gradient per level (+ the FPN's into the finest level)
        parts = []
        for i, (h, w) in enumerate(shapes):
            g = d_outs[i].reshape(N, h * w, C) if d_outs[i] is not None else torch.zeros((N, h * w, C), device=d_mf.device, dtype=torch.float32)
            if i == len(shapes) - 1:
                g = g + d_up.view(N, h * w, C)
            parts.append(g)
        d_src = torch.cat(parts, 1).contiguous()
        d_pos = None
        for layer, saved_l in zip(reversed(self.transformer.encoder.layers), reversed(enc)):
            d_src, dp = layer.backward(saved_l, d_src)
            d_pos = dp if d_pos is None else d_pos + dp
        # pos = sine encoding (constant) + level_embed[level]: the embedding's gradient is the per-level row sum
        o, dle = 0, []
        for (h, w) in shapes:
            dle.append(B.bias_grad(d_pos[o:o + h * w].contiguous()))
            o += h * w
        B.acc(self.transformer.level_embed, torch.stack(dle, 0))
        # input projections: tokens -> GN -> 1x1 conv -> backbone features
        o = 0
        for idx, f in enumerate(("res5", "res4", "res3")):
            h, w = shapes[idx]
            x, z = proj[idx]
            conv, gn = self.input_proj[idx][0], self.input_proj[idx][1]
            d_y = d_src[:, o:o + h * w].contiguous().view(N, h, w, C)
            o += h * w
            d_z, dg, db = B.groupnorm_backward(z, d_y, 32, gn.weight, gn.eps)
            B.acc(gn.weight, dg); B.acc(gn.bias, db)
            dz = d_z.view(-1, C)
            cin = x.shape[-1]
            B.acc_wgrad(conv.weight, dz, x.view(-1, cin))
            B.acc_bgrad(conv.bias, dz)
            grads[f] = B.input_grad(dz, conv.weight.view(C, cin)).view(N, h, w, cin)
        return grads

"""Video masked-attention transformer decoder, host side.  Mirrors
model_training/mask2former_video/modeling/transformer_decoder/video_mask2former_transformer_decoder.py
(VideoMultiScaleMaskedTransformerDecoder :208-480, layers :18-179, MLP :193-205) with the reference's parameter
names/shapes (nn.MultiheadAttention in_proj_weight/in_proj_bias/out_proj.*).

MI355X-first differences (results identical within fp32 rounding):
* keys stay in the pixel decoder's NHWC token layout [B, T*h*w, C] (t-major, == :394-397) -- no permutes;
* mask logits are produced PIXEL-major [NL, B, T*hm*wm, Q] straight from the GEMM (rows = pixels) and kept for all
  10 prediction heads in one buffer: the attention-mask builder, the matcher and the loss read that layout coalesced;
  the reference's [B,Q,T,h,w] view is produced on demand (`MaskOutputs.pred_masks`);
* the [B*8,Q,K] bool mask is never built: a bit matrix + the streamed attention kernel replace it.
"""
import torch
from torch import nn

from .. import ops
from .._lib import lib


class MaskOutputs:
    """All prediction heads of one forward.  class_logits [NL,B,Q,K+1]; mask_logits pixel-major [NL,B,T*hm*wm,ldq]."""

    def __init__(self, class_logits, mask_logits, Q, T, hm, wm):
        self.class_logits, self.mask_logits = class_logits, mask_logits
        self.Q, self.T, self.hm, self.wm = Q, T, hm, wm

    @property
    def dims(self):
        return (self.Q, self.T, self.hm, self.wm)

    def pred_masks(self, layer=-1):
        """reference layout [B,Q,T,hm,wm] (video_...decoder.py:455)"""
        ml = self.mask_logits[layer][..., :self.Q]
        B = ml.shape[0]
        return ml.view(B, self.T, self.hm, self.wm, self.Q).permute(0, 4, 1, 2, 3)

    def as_reference_dict(self):
        """{'pred_logits','pred_masks','aux_outputs'} as the reference returns (:439-446)"""
        NL = self.class_logits.shape[0]
        return {"pred_logits": self.class_logits[-1], "pred_masks": self.pred_masks(-1),
                "aux_outputs": [{"pred_logits": self.class_logits[i], "pred_masks": self.pred_masks(i)} for i in range(NL - 1)]}


class _MHAParams(nn.Module):
    """nn.MultiheadAttention parameter layout."""

    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)


class SelfAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead):
        super().__init__()
        self.self_attn = _MHAParams(d_model)
        self.norm = nn.LayerNorm(d_model)
        self.nhead = nhead

    def forward(self, tgt, query_pos, tape=None):  # forward_post :41-51;  tgt [B,Q,C], query_pos [Q,C]
        B, Q, C = tgt.shape
        W, bi = self.self_attn.in_proj_weight, self.self_attn.in_proj_bias
        qk_in = ops.add_bcast(tgt, query_pos)
        qk = ops.gemm_nt(qk_in.view(-1, C), W[:2 * C], bias=bi[:2 * C]).view(B, Q, 2 * C)
        v = ops.gemm_nt(tgt.view(-1, C), W[2 * C:], bias=bi[2 * C:]).view(B, Q, C)
        q = qk[..., :C].contiguous()
        a = ops.masked_attn(q, qk[..., C:], v, H=self.nhead, want_lse=tape is not None)        # k: a column slice, read in place
        lse = None
        if tape is not None:
            a, lse = a
        x = ops.gemm_nt(a.view(-1, C), self.self_attn.out_proj.weight, bias=self.self_attn.out_proj.bias, res=tgt.view(-1, C))
        if tape is not None:
            tape.append((self, tgt, qk_in, q, qk, v, a, lse, x))
        return ops.layernorm(x.view(B, Q, C), self.norm.weight, self.norm.bias)

    def backward(self, saved, d_out):
        """-> (d_tgt [B,Q,C], d_query_pos [Q,C])"""
        from .. import backward as Bk
        _, tgt, qk_in, q, qk, v, a, lse, x = saved
        B, Q, C = tgt.shape
        m = self.self_attn
        d_x, dg, db = Bk.layernorm_backward(x.view(B, Q, C), d_out.contiguous(), self.norm.weight)
        Bk.acc(self.norm.weight, dg); Bk.acc(self.norm.bias, db)
        d2 = d_x.view(-1, C)
        Bk.acc_wbgrad(m.out_proj.weight, m.out_proj.bias, d2, a.view(-1, C))
        d_a = Bk.input_grad(d2, m.out_proj.weight).view(B, Q, C)
        dq, dk, dv = Bk.masked_attn_backward(q, qk[..., C:], v, a, lse, d_a, H=self.nhead)
        d_qk = torch.cat([dq, dk], -1).view(-1, 2 * C)
        W = m.in_proj_weight
        dW = torch.cat([Bk.weight_grad(d_qk, qk_in.view(-1, C)), Bk.weight_grad(dv.view(-1, C), tgt.view(-1, C))], 0)
        Bk.acc(m.in_proj_weight, dW)
        Bk.acc(m.in_proj_bias, torch.cat([Bk.bias_grad(d_qk), Bk.bias_grad(dv.view(-1, C))], 0))
        d_in = Bk.input_grad(d_qk, W[:2 * C]).view(B, Q, C)                 # d(tgt + query_pos)
        d_tgt = Bk.input_grad(dv.view(-1, C), W[2 * C:], res=d2).view(B, Q, C) + d_in
        return d_tgt, Bk.sum_slices(d_in)


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead):
        super().__init__()
        self.multihead_attn = _MHAParams(d_model)
        self.norm = nn.LayerNorm(d_model)
        self.nhead = nhead

    def forward(self, tgt, k, v, bits, unmasked, query_pos, tape=None):  # forward_post :99-111
        """k, v [B,K,C]: the memory already through this layer's key / value projections (column slices of the per-level
        projections the decoder makes once for the three layers that read a level, `_project_memory`)."""
        B, Q, C = tgt.shape
        W, bi = self.multihead_attn.in_proj_weight, self.multihead_attn.in_proj_bias
        q_in = ops.add_bcast(tgt, query_pos)
        q = ops.gemm_nt(q_in.view(-1, C), W[:C], bias=bi[:C]).view(B, Q, C)
        a = ops.masked_attn(q, k, v, bits, unmasked, H=self.nhead, want_lse=tape is not None)
        lse = None
        if tape is not None:
            a, lse = a
        x = ops.gemm_nt(a.view(-1, C), self.multihead_attn.out_proj.weight, bias=self.multihead_attn.out_proj.bias,
                        res=tgt.view(-1, C))
        if tape is not None:
            tape.append((self, tgt, q_in, q, k, v, bits, unmasked, a, lse, x))
        return ops.layernorm(x.view(B, Q, C), self.norm.weight, self.norm.bias)

    def backward(self, saved, d_out, dk_out=None, dv_out=None):
        """-> (d_tgt, d_query_pos [Q,C], d_k [B,K,C], d_v [B,K,C]); the in-projection's key / value rows get their gradients
        from the decoder's batched memory projection (`_project_memory_backward`).  dk_out / dv_out: strided [B,K,C] views the key / value
        gradients are written into (the decoder's per-level buffers) instead of fresh tensors"""
        from .. import backward as Bk
        _, tgt, q_in, q, k, v, bits, unmasked, a, lse, x = saved
        B, Q, C = tgt.shape
        m = self.multihead_attn
        d_x, dg, db = Bk.layernorm_backward(x.view(B, Q, C), d_out.contiguous(), self.norm.weight)
        Bk.acc(self.norm.weight, dg); Bk.acc(self.norm.bias, db)
        d2 = d_x.view(-1, C)
        Bk.acc_wbgrad(m.out_proj.weight, m.out_proj.bias, d2, a.view(-1, C))
        d_a = Bk.input_grad(d2, m.out_proj.weight).view(B, Q, C)
        dq, dk, dv = Bk.masked_attn_backward(q, k, v, a, lse, d_a, bits, unmasked, H=self.nhead, dk_out=dk_out, dv_out=dv_out)
        dq2 = dq.view(-1, C)
        Wq = m.in_proj_weight[:C]
        self._dWq, self._dbq = Bk.weight_grad(dq2, q_in.view(-1, C)), Bk.bias_grad(dq2)            # merged with the k / v rows later
        d_in = Bk.input_grad(dq2, Wq).view(B, Q, C)
        return d_in + d_x, Bk.sum_slices(d_in), dk, dv


class FFNLayer(nn.Module):
    def __init__(self, d_model, dim_feedforward):
        super().__init__()
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm = nn.LayerNorm(d_model)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, tgt, tape=None):  # forward_post :164-168
        B, Q, C = tgt.shape
        h = ops.gemm_nt(tgt.view(-1, C), self.linear1.weight, bias=self.linear1.bias, relu=True)
        x = ops.gemm_nt(h, self.linear2.weight, bias=self.linear2.bias, res=tgt.view(-1, C))
        if tape is not None:
            tape.append((self, tgt, h, x))
        return ops.layernorm(x.view(B, Q, C), self.norm.weight, self.norm.bias)

    def backward(self, saved, d_out):
        from .. import backward as Bk
        _, tgt, h, x = saved
        B, Q, C = tgt.shape
        d_x, dg, db = Bk.layernorm_backward(x.view(B, Q, C), d_out.contiguous(), self.norm.weight)
        Bk.acc(self.norm.weight, dg); Bk.acc(self.norm.bias, db)
        d2 = d_x.view(-1, C)
        Bk.acc_wbgrad(self.linear2.weight, self.linear2.bias, d2, h)
        d_h = Bk.input_grad(d2, self.linear2.weight, gate=h)
        Bk.acc_wbgrad(self.linear1.weight, self.linear1.bias, d_h, tgt.view(-1, C))
        return Bk.input_grad(d_h, self.linear1.weight, res=d2).view(B, Q, C)


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x, tape=None):
        acts = [x]
        for i, l in enumerate(self.layers):
            x = ops.gemm_nt(x, l.weight, bias=l.bias, relu=i < len(self.layers) - 1)
            acts.append(x)
        if tape is not None:
            tape.append(acts)
        return x

    def backward(self, acts, d):
        from .. import backward as Bk
        for i in reversed(range(len(self.layers))):
            l = self.layers[i]
            Bk.acc_wbgrad(l.weight, l.bias, d, acts[i])
            d = Bk.input_grad(d, l.weight, gate=acts[i] if i > 0 else None)      # acts[i] (i > 0) is the ReLU output feeding layer i
        return d


class VideoMultiScaleMaskedTransformerDecoder(nn.Module):
    def __init__(self, in_channels=256, mask_classification=True, *, num_classes=1, hidden_dim=256, num_queries=100,
                 nheads=8, dim_feedforward=2048, dec_layers=9, pre_norm=False, mask_dim=256, enforce_input_project=False,
                 num_frames=2, detach_cls=False):
        super().__init__()
        assert mask_classification and not pre_norm and in_channels == hidden_dim and not enforce_input_project, \
            "only the shipped S2D configuration (post-norm, identity input_proj) is on the hot path"
        self.num_frames, self.num_heads, self.num_layers, self.num_queries = num_frames, nheads, dec_layers, num_queries
        self.hidden_dim = hidden_dim
        self.transformer_self_attention_layers = nn.ModuleList(SelfAttentionLayer(hidden_dim, nheads) for _ in range(dec_layers))
        self.transformer_cross_attention_layers = nn.ModuleList(CrossAttentionLayer(hidden_dim, nheads) for _ in range(dec_layers))
        self.transformer_ffn_layers = nn.ModuleList(FFNLayer(hidden_dim, dim_feedforward) for _ in range(dec_layers))
        self.decoder_norm = nn.LayerNorm(hidden_dim)
        self.query_feat = nn.Embedding(num_queries, hidden_dim)
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.level_embed = nn.Embedding(3, hidden_dim)
        self.input_proj = nn.ModuleList(nn.Sequential() for _ in range(3))
        self.class_embed = nn.Linear(hidden_dim, num_classes + 1)
        self.mask_embed = MLP(hidden_dim, hidden_dim, mask_dim, 3)
        self._pos_cache = {}
        self._tap_cache = {}
        self._kv_cache = None

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification=True):  # :344-372
        mf = cfg.MODEL.MASK_FORMER
        return cls(in_channels, mask_classification, num_classes=cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES, hidden_dim=mf.HIDDEN_DIM,
                   num_queries=mf.NUM_OBJECT_QUERIES, nheads=mf.NHEADS, dim_feedforward=mf.DIM_FEEDFORWARD,
                   dec_layers=mf.DEC_LAYERS - 1, pre_norm=mf.PRE_NORM, mask_dim=cfg.MODEL.SEM_SEG_HEAD.MASK_DIM,
                   enforce_input_project=mf.ENFORCE_INPUT_PROJ, num_frames=cfg.INPUT.SAMPLING_FRAME_NUM)

    def _pos(self, T, sizes, device):
        """per level: pos3d + level_embed (for keys) and level_embed alone (for values), token-major [T*h*w, C]"""
        le = self.level_embed.weight
        key = (T, tuple(sizes), ops.version_of(le), device)
        if self._pos_cache.get("key") != key:
            self._pos_cache = {"key": key, "posl": [
                ops.pe_sine(T, h, w, self.hidden_dim // 2, add_c=le[i].detach().contiguous(), device=device)
                for i, (h, w) in enumerate(sizes)]}
        return self._pos_cache["posl"]

    def _kv_packed(self):
        """per level: the key and the value in-projection weights / biases of the layers that attend to it (i % 3 == level),
        stacked to [n*C, C] so that the memory of a level is read once for all of them instead of once per layer"""
        C = self.hidden_dim
        mh = [l.multihead_attn for l in self.transformer_cross_attention_layers]
        key = tuple(ops.version_of(m.in_proj_weight) for m in mh) + tuple(ops.version_of(m.in_proj_bias) for m in mh) + (mh[0].in_proj_weight.device,)
        if self._kv_cache is None or self._kv_cache[0] != key:
            packs = []
            prevs = [None] * 3 if self._kv_cache is None else self._kv_cache[1]
            for lvl in range(3):
                ms = [mh[i] for i in range(self.num_layers) if i % 3 == lvl]
                if not ms:                         # fewer than three layers: nobody reads this level
                    packs.append(None)
                    continue
                wk = torch.cat([m.in_proj_weight.detach()[C:2 * C] for m in ms], 0)
                bk = torch.cat([m.in_proj_bias.detach()[C:2 * C] for m in ms], 0)
                wv = torch.cat([m.in_proj_weight.detach()[2 * C:] for m in ms], 0)
                bv = torch.cat([m.in_proj_bias.detach()[2 * C:] for m in ms], 0)
                pv = prevs[lvl] if prevs[lvl] is not None else (None,) * 4                    # refreshed in place after an optimizer step
                packs.append((ops.repack(pv[0], wk), ops.repack(pv[1], bk), ops.repack(pv[2], wv), ops.repack(pv[3], bv)))
            self._kv_cache = (key, packs)
        return self._kv_cache[1]

    def _project_memory(self, multi_scale, B, T, posl, tape=None):
        """keys and values of every layer, per level [B, T*h*w, n*C] (layer i reads columns (i // 3)*C .. of level i % 3).
        K_i = (src + level_embed + pos) . Wk_i^T + bk_i as the reference forms it (:386-394, nn.MultiheadAttention);
        V_i = (src + level_embed) . Wv_i^T + bv_i = src . Wv_i^T + (level_embed . Wv_i^T + bv_i): the bracket is one row
        per level and layer, so src + level_embed is never stored."""
        C = self.hidden_dim
        ks, vs, kins, xs = [], [], [], []
        for lvl, ((tok, (h, w)), pack) in enumerate(zip(multi_scale, self._kv_packed())):
            if pack is None:
                ks.append(None); vs.append(None); kins.append(None); xs.append(None)
                continue
            wk, bk, wv, bv = pack
            x = tok.view(B, T * h * w, C)
            kin = ops.add_bcast(x, posl[lvl])                                               # src + level_embed + pos
            ks.append(ops.gemm_nt(kin.view(-1, C), wk, bias=bk).view(B, T * h * w, -1))
            if ops.amp_active():
                # autocast rounds the value projection's INPUT, src + level_embed, to fp16: the sum is formed, not split over the bias
                xv = ops.add_bcast(x, self.level_embed.weight[lvl:lvl + 1].detach().contiguous())
                vs.append(ops.gemm_nt(xv.view(-1, C), wv, bias=bv).view(B, T * h * w, -1))
            else:
                bvf = ops.gemm_nt(self.level_embed.weight[lvl:lvl + 1].detach().contiguous(), wv, bias=bv).view(-1)
                vs.append(ops.gemm_nt(x.reshape(-1, C), wv, bias=bvf).view(B, T * h * w, -1))
            kins.append(kin); xs.append(x)
        if tape is not None:
            tape.append((kins, xs))
        return ks, vs

    def _project_memory_backward(self, saved, d_ks, d_vs):
        """d_ks / d_vs: per level [B, K, n*C] (zeros where a layer sent nothing).  Accumulates the key / value rows of every
        cross-attention in-projection (together with the query rows the layers left in _dWq / _dbq) and level_embed;
        returns the gradient of the three memory levels [B*T, h*w, C]."""
        from .. import backward as Bk
        kins, xs = saved
        C = self.hidden_dim
        packs = self._kv_packed()
        d_mem, d_le = [], []
        per_layer = {}
        for lvl in range(3):
            if packs[lvl] is None:
                d_mem.append(None)
                d_le.append(torch.zeros((C,), device=self.level_embed.weight.device, dtype=torch.float32))
                continue
            wk, bk, wv, bv = packs[lvl]
            dk2, dv2 = d_ks[lvl].view(-1, d_ks[lvl].shape[-1]), d_vs[lvl].view(-1, d_vs[lvl].shape[-1])
            kin2, x2 = kins[lvl].view(-1, C), xs[lvl].reshape(-1, C)
            le = self.level_embed.weight[lvl:lvl + 1].detach().contiguous()
            dbk = torch.empty((dk2.shape[1],), device=dk2.device, dtype=torch.float32)
            dWk = Bk.weight_grad(dk2, kin2, bias_out=dbk)
            dbv = Bk.bias_grad(dv2)                                                # also d(level_embed . Wv^T + bv)
            dWv = Bk.weight_grad(dv2, x2) + dbv[:, None] * le                      # + outer(d bracket, level_embed)
            d_kin = Bk.input_grad(dk2, wk)
            d_x = Bk.input_grad(dv2, wv, res=d_kin)                                # memory enters keys and values
            d_le.append(Bk.bias_grad(d_kin) + Bk.input_grad(dbv[None].contiguous(), wv).view(-1))
            d_mem.append(d_x)
            layers = [i for i in range(self.num_layers) if i % 3 == lvl]
            for j, i in enumerate(layers):
                per_layer[i] = (dWk[j * C:(j + 1) * C], dbk[j * C:(j + 1) * C], dWv[j * C:(j + 1) * C], dbv[j * C:(j + 1) * C])
        for i, layer in enumerate(self.transformer_cross_attention_layers):
            dWk, dbk, dWv, dbv = per_layer[i]
            Bk.acc(layer.multihead_attn.in_proj_weight, torch.cat([layer._dWq, dWk, dWv], 0))
            Bk.acc(layer.multihead_attn.in_proj_bias, torch.cat([layer._dbq, dbk, dbv], 0))
            layer._dWq = layer._dbq = None
        Bk.acc(self.level_embed.weight, torch.stack(d_le, 0))
        return d_mem

    def _heads(self, layer_slot, output, mf, out_cls, out_ml, target_hw, B, T, hm, wm, ml_slot=None, mf_taps=None, tape=None):
        """forward_prediction_heads :448-467 for all clips; writes slot `layer_slot` of the class buffer and slot `ml_slot`
        of the mask-logit buffer, and returns the attention-mask bits for the next layer.  With `mf_taps` (the mask
        features gathered at the four bilinear source pixels of every key of the next level) only those logits are
        computed: enough for the attention mask, which is all an intermediate prediction of a frozen network feeds."""
        Q, C = self.num_queries, self.hidden_dim
        d = ops.layernorm(output, self.decoder_norm.weight, self.decoder_norm.bias).view(-1, C)
        ops.gemm_nt(d, self.class_embed.weight, bias=self.class_embed.bias, out=out_cls[layer_slot].view(B * Q, -1))
        mlp_tape = [] if tape is not None else None
        e = self.mask_embed(d, mlp_tape).view(B, Q, -1)
        if tape is not None:
            tape.append((layer_slot, output, d, mlp_tape[0], e))
        if mf_taps is not None:
            sub = ops.gemm_nt(mf_taps, e)                                  # [B, K*4, Q]
            return ops.attn_mask_bits(sub, B, Q, T, hm, wm, target_hw[0], target_hw[1], compact=True)
        # einsum "bqc,btchw->bqthw" as a pixel-major GEMM: [T*hm*wm, C] x [Q, C]^T per clip
        ops.gemm_nt(mf, e, out=out_ml[layer_slot if ml_slot is None else ml_slot])
        if target_hw is None:
            return None, None
        return ops.attn_mask_bits(out_ml[layer_slot if ml_slot is None else ml_slot], B, Q, T, hm, wm, target_hw[0], target_hw[1])

    @torch.no_grad()
    def forward(self, multi_scale, mask_features, training=True, aux_masks=True, tape=None):
        """multi_scale: 3 x (tokens [BT,h*w,C], (h,w)) from the pixel decoder (res5, res4, res3 scale);
        mask_features [BT,hm,wm,C] NHWC.  Returns MaskOutputs.
        aux_masks=False (a frozen network whose intermediate predictions are not supervised: the teacher): the mask logits
        of layers 0..L-2 are evaluated only where the next layer's attention mask reads them; `mask_logits` then holds the
        full-map slots only (the last one is the final prediction, as always)."""
        with ops.amp_fp16(self.amp and tape is None):
            return self._forward(multi_scale, mask_features, training, aux_masks, tape)

    amp = False      # True: linear layers and the mask-logit einsum in torch.autocast's arithmetic (fp16 operands, f32 accumulate),
                     # as under the reference trainer's `with autocast():`; LayerNorm / softmax stay f32 as autocast keeps them

    def _forward(self, multi_scale, mask_features, training=True, aux_masks=True, tape=None):
        BT, hm, wm, C = mask_features.shape
        B = BT // self.num_frames if training else 1   # :376
        T = BT // B
        Q, NL = self.num_queries, self.num_layers + 1
        dev = mask_features.device
        sizes = [s for _, s in multi_scale]
        posl = self._pos(T, sizes, dev)
        mem_tape = [] if tape is not None else None
        ks, vs = self._project_memory(multi_scale, B, T, posl, mem_tape)
        head_tape, layer_tape = ([], []) if tape is not None else (None, None)
        mf = mask_features.view(B, T * hm * wm, C)
        ldq = (Q + 3) // 4 * 4
        # slot s (the prediction after layer s-1) feeds the attention mask of layer s at level s % 3; a level whose keys
        # read every pixel of the map (scale <= 2) gains nothing from the tap-gathered form
        def target(s):
            return sizes[s % 3] if s < self.num_layers else None
        def sparse(s):
            t = target(s)
            return (not aux_masks) and t is not None and 4 * t[0] * t[1] < hm * wm
        full = [s for s in range(NL) if not sparse(s)]
        taps = {}
        if not aux_masks:
            for lvl in {s % 3 for s in range(NL) if sparse(s)}:
                idx = self._tap_index(T, hm, wm, sizes[lvl], dev)
                taps[lvl] = mf.index_select(1, idx)                                        # [B, K*4, C]
        out_cls = torch.empty((NL, B, Q, self.class_embed.out_features), device=dev, dtype=torch.float32)
        out_ml = torch.empty((len(full), B, T * hm * wm, ldq), device=dev, dtype=torch.float32)
        if ldq != Q:
            out_ml.zero_()
        qe = self.query_embed.weight.detach()
        output = self.query_feat.weight.detach().unsqueeze(0).repeat(B, 1, 1).contiguous()

        def heads(s, out):
            if sparse(s):
                return self._heads(s, out, mf, out_cls, out_ml, target(s), B, T, hm, wm, mf_taps=taps[s % 3])
            return self._heads(s, out, mf, out_cls, out_ml, target(s), B, T, hm, wm, ml_slot=full.index(s), tape=head_tape)

        # tests only: attention masks held fixed (they are detached, piecewise-constant functions of the parameters; a finite
        # difference of the loss has to keep them constant to see the derivative the backward computes)
        fixed = getattr(self, "_fixed_masks", None)
        bits, unm = heads(0, output)
        for i in range(self.num_layers):
            lvl = i % 3
            if fixed is not None:
                bits, unm = fixed[i]
            c0 = (i // 3) * C
            output = self.transformer_cross_attention_layers[i](output, ks[lvl][..., c0:c0 + C], vs[lvl][..., c0:c0 + C], bits, unm, qe,
                                                                layer_tape)
            output = self.transformer_self_attention_layers[i](output, qe, layer_tape)
            output = self.transformer_ffn_layers[i](output, layer_tape)
            bits, unm = heads(i + 1, output)
        if tape is not None:
            assert aux_masks, "the backward needs every layer's full mask prediction (the supervised network)"
            tape.append((self, mem_tape[0], head_tape, layer_tape, mf, [None if k is None else k.shape for k in ks], (B, T, hm, wm)))
        return MaskOutputs(out_cls, out_ml, Q, T, hm, wm)

    @torch.no_grad()
    def backward(self, saved, d_cls, mask_sources):
        """Gradients of forward (video_mask2former_transformer_decoder.py:374-467 backwards).
        d_cls [NL,B,Q,K+1] or None: d(loss)/d(class logits); mask_sources: list of (rows [NL,B,maxm,T*hm*wm], idx_q [NL*B,maxm])
        -- d(loss)/d(mask logit map) of the matched queries of one criterion pass (ops.point_loss_backward) and the query
        each row belongs to; unmatched slots hold zero rows.  The attention masks are constants (detached, :465).
        Returns (d_mask_features [B*T,hm,wm,C], [d_memory level 0..2 [B*T, h*w, C]]); parameter gradients go to .grad."""
        from .. import backward as Bk
        _, mem_saved, head_tape, layer_tape, mf, kshapes, (B, T, hm, wm) = saved
        Q, C = self.num_queries, self.hidden_dim
        NL = self.num_layers + 1
        dev = mf.device
        npix = T * hm * wm
        d_mf_all = torch.empty((B, npix, C), device=dev, dtype=torch.float32)     # the clips' feature gradients, written in place by their GEMMs
        d_mf = [None] * B
        # every column block of a level's buffer is written by the walk below when each level serves the same number of layers
        # (9 layers over 3 levels: the shipped decoder): no zero fill of ~1.9 GB then
        covered = self.num_layers % 3 == 0
        alloc = torch.empty if covered else torch.zeros
        d_ks = [None if sh is None else alloc(sh, device=dev, dtype=torch.float32) for sh in kshapes]
        d_vs = [None if sh is None else alloc(sh, device=dev, dtype=torch.float32) for sh in kshapes]
        d_qe = torch.zeros((Q, C), device=dev, dtype=torch.float32)

        # d(mask features) does not depend on the layer walk: per clip and criterion pass, the gradient planes of ALL layers
        # are transposed side by side into one [npix, NL*maxm] matrix and contracted with the matching mask embeddings in one
        # GEMM (instead of one read-modify-write pass over the 0.5 GB feature gradient per layer).  The same matrix, contracted over the
        # pixels with the clip's mask features, is d(mask embedding) of the matched queries of all layers: one TN GEMM per clip and
        # pass (round 5; it was one long-K GEMM per layer, clip and pass, each re-reading the clip's 0.47 GB of features).
        # The matched queries' embeddings of all layers are gathered once per pass, and the embedding gradients of all layers scattered back
        # once per pass (unmatched slots carry zero rows and any valid query index: they add exact zeros) -- it was one indexed copy and one
        # index_add per (layer, clip, pass).
        Cm = mf.shape[-1]
        E_all = torch.stack([head_tape[s_][4] for s_ in range(NL)])                 # [NL, B, Q, Cm]
        D_E = torch.zeros_like(E_all)
        for rows, idx_q in mask_sources:
            maxm = rows.shape[2]
            mp = (maxm + 3) // 4 * 4
            ex = idx_q.view(NL, B, maxm).long()[..., None].expand(NL, B, maxm, Cm)
            eg = E_all.gather(2, ex)                                                  # [NL, B, maxm, Cm]
            if mp != maxm:
                eg = torch.nn.functional.pad(eg, (0, 0, 0, mp - maxm))
            de_pass = []
            for b in range(B):
                Dt = torch.zeros((npix, NL * mp), device=dev, dtype=torch.float32) if mp != maxm else \
                    torch.empty((npix, NL * mp), device=dev, dtype=torch.float32)
                for slot in range(NL):
                    lib().call("s2d_transpose_f32", rows[slot, b], maxm, npix, npix, Dt[:, slot * mp:], NL * mp, ops._stream())
                et = eg[:, b].reshape(NL * mp, Cm).t().contiguous()                   # [Cm, NL * mp]
                d_mf[b] = ops.gemm_nt(Dt, et, res=d_mf[b], out=d_mf_all[b])
                de_pass.append(Bk.weight_grad(Dt, mf[b]).view(NL, mp, Cm)[:, :maxm])  # sum_pix D[slot, m, pix] mf[pix, :]
            D_E.scatter_add_(2, ex, torch.stack(de_pass, 1))

        def heads_backward(rec):
            slot, output, d, mlp_acts, e = rec
            d_d = self.mask_embed.backward(mlp_acts, D_E[slot].reshape(B * Q, -1))
            if d_cls is not None:
                dc = d_cls[slot].reshape(B * Q, -1).contiguous()
                Bk.acc_wgrad(self.class_embed.weight, dc, d); Bk.acc_bgrad(self.class_embed.bias, dc)
                d_d = Bk.input_grad(dc, self.class_embed.weight, res=d_d)
            d_out, dg, db = Bk.layernorm_backward(output, d_d.view(B, Q, C), self.decoder_norm.weight)
            Bk.acc(self.decoder_norm.weight, dg); Bk.acc(self.decoder_norm.bias, db)
            return d_out

        d_output = heads_backward(head_tape[NL - 1])
        for i in reversed(range(self.num_layers)):
            lvl, c0 = i % 3, (i // 3) * C
            d_output = self.transformer_ffn_layers[i].backward(layer_tape[3 * i + 2], d_output)
            d_output, dq1 = self.transformer_self_attention_layers[i].backward(layer_tape[3 * i + 1], d_output)
            # the key / value gradients land in their column block of the level's buffer (no copy per layer)
            d_output, dq2, _, _ = self.transformer_cross_attention_layers[i].backward(layer_tape[3 * i], d_output, dk_out=d_ks[lvl][..., c0:c0 + C],
                                                                                      dv_out=d_vs[lvl][..., c0:c0 + C])
            d_qe += dq1 + dq2
            d_output = d_output + heads_backward(head_tape[i])
        Bk.acc(self.query_embed.weight, d_qe)
        Bk.acc(self.query_feat.weight, Bk.sum_slices(d_output.contiguous()))
        d_mem = self._project_memory_backward(mem_saved, d_ks, d_vs)
        if any(d is None for d in d_mf):                                          # no criterion pass carried a mask loss
            d_mf_all.zero_()
        return d_mf_all.view(B * T, hm, wm, C), d_mem

    def _tap_index(self, T, hm, wm, size, device):
        key = (T, hm, wm, tuple(size), device)
        if key not in self._tap_cache:
            self._tap_cache[key] = ops.attn_mask_tap_index(T, hm, wm, size[0], size[1], device)
        return self._tap_cache[key]

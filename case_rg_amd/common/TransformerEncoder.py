"""Encoder layer / stack on the HIP path (reference: common/TransformerEncoder.py:19-77, :82-123).

Layer arithmetic (note the residuals are taken from the *normed* tensors, :66-75):
    s = LN1(x);  s = s + drop(MHA(s));  s = LN2(s);  s = s + drop(W2 drop(act(W1 s)))
realised as: LayerNorm kernel -> QKV GEMM -> attention core -> out-proj GEMM (+dropout +residual in the
epilogue) -> LayerNorm kernel -> FFN GEMM pair (bias+activation+dropout / bias+dropout+residual epilogues).
"""
import copy

import torch
import torch.nn as nn

from .. import config, ops
from .attention import MultiheadAttention, mask_kind


def _check_activation(activation):
    if activation not in ("relu", "gelu"):
        raise RuntimeError("activation should be relu/gelu, not %s." % activation)
    return activation


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu"):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.p = dropout
        self.activation = _check_activation(activation)

    def forward_batch_first(self, x, valid=None, causal=False, normed=False, next_norm=None):
        """x [N, L, E]; valid [N, L] bool (True = token).  ``normed``: x already is LN1(x) (the previous layer applied this layer's
        norm1 behind its feed-forward, ``next_norm``); ``next_norm`` = (gamma, beta, eps) of the LayerNorm that follows this layer."""
        p = config.drop_p(self.p, self.training)
        s = x if normed else ops.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        s = self.self_attn.self_attention(s, valid, causal=causal, residual=s, p_res=self.p,
                                          ln=(self.norm2.weight, self.norm2.bias, self.norm2.eps))
        return ops.ffn(s, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                       self.activation, p_inner=p, p_out=p, residual=s, ln=next_norm)

    def forward_rows(self, x, groups, valids, normed=False, next_norm=None):
        """The same layer over SEVERAL sequence groups at once: x [rows, E] holds the rows of all groups back to back
        (``groups`` = [(first row, sequences, length)]); everything row-local runs once, the attention core once per group."""
        p = config.drop_p(self.p, self.training)
        at = self.self_attn
        s = x if normed else ops.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        if torch.is_grad_enabled() and s.requires_grad:
            qkv, res = ops.linear_carry(s, at.in_proj_weight, at.in_proj_bias)
        else:
            qkv, res = ops.linear(s, at.in_proj_weight, at.in_proj_bias), s
        ctx = ops.attention_groups(qkv, groups, valids, at.num_heads, at.head_dim, p_drop=config.drop_p(at.dropout, self.training))
        s = ops.linear(ctx, at.out_proj.weight, at.out_proj.bias, residual=res, p_drop=config.drop_p(self.p, self.training),
                       ln=(self.norm2.weight, self.norm2.bias, self.norm2.eps))
        return ops.ffn(s, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                       self.activation, p_inner=p, p_out=p, residual=s, ln=next_norm)

    def forward(self, src, src_mask=None, src_key_padding_mask=None):
        """src [L, N, E] (sequence first, as the reference); src_key_padding_mask [N, L] True = pad.  ``src_mask``: None, the causal
        pattern (kernel flag), or any other 2-D additive / bool mask (round 6: reference :66-75 line by line, with the attention through
        MultiheadAttention.forward's additive-mask form -- no caller on the CaSE / Masque path passes one)."""
        kind = mask_kind(src_mask)
        if kind == "general":
            p = config.drop_p(self.p, self.training)
            s = ops.layer_norm(src, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            a, _ = self.self_attn(s, s, s, attn_mask=src_mask, key_padding_mask=src_key_padding_mask)
            s = ops.add(s, ops.dropout(a, self.p, self.training))
            s = ops.layer_norm(s, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            return ops.ffn(s, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, self.activation,
                           p_inner=p, p_out=p, residual=s)
        valid = None if src_key_padding_mask is None else ~src_key_padding_mask
        y = self.forward_batch_first(src.transpose(0, 1).contiguous(), valid, kind == "causal")
        return y.transpose(0, 1)


class TransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def _chained(self, x, valid):
        """Inference form: the row-local half of every layer (out-proj + residual -> LN2 -> FFN -> + residual -> next LN1 -> next QKV) is
        ONE kernel per layer (ops.encoder_chain, csrc/encoder_chain.hip); only the attention core runs between two of them."""
        layers = list(self.layers)
        s, qkv = ops.encoder_chain("head", x, None, None, layers[0])
        for i, layer in enumerate(layers):
            at = layer.self_attn
            ctx = ops.attention(qkv, qkv, qkv, 0, at.embed_dim, 2 * at.embed_dim, at.num_heads, at.head_dim, key_valid=valid)
            if i + 1 < len(layers):
                s, qkv = ops.encoder_chain("full", ctx, s, layer, layers[i + 1])
            else:
                s, qkv = ops.encoder_chain("tail", ctx, s, layer, None)
        return s

    def forward_batch_first(self, x, valid=None, causal=False):
        first = self.layers[0]
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if (not causal and not self.training and all(l.activation == "gelu" for l in self.layers)
                and ops.encoder_chain_supported(x, first.self_attn.embed_dim, first.linear1.out_features, needs_grad)
                and all(l.self_attn.embed_dim == 512 and l.linear1.out_features == 512 and l.self_attn.head_dim == 64 for l in self.layers)):
            x = self._chained(x, valid)
        else:
            return self._layers(x, lambda layer, h, normed, nxt: layer.forward_batch_first(h, valid, causal, normed, nxt))
        if self.norm is not None:
            x = ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x

    def _layers(self, x, run):
        """The layer loop with every LayerNorm that follows a feed-forward pair -- the next layer's norm1, the stack's final norm -- made
        part of that pair's op (ops.ffn(ln=...)): in training its backward then hands the dropout-masked gradient to the pair's GEMMs
        from the LayerNorm kernel instead of a separate pass."""
        layers = list(self.layers)
        tail = None if self.norm is None else (self.norm.weight, self.norm.bias, self.norm.eps)
        for i, layer in enumerate(layers):
            nxt = layers[i + 1].norm1 if i + 1 < len(layers) else None
            x = run(layer, x, i > 0, tail if nxt is None else (nxt.weight, nxt.bias, nxt.eps))
        return x

    def rows_supported(self, dtype, needs_grad):
        at = self.layers[0].self_attn
        return ops.attention_groups_supported(dtype, at.num_heads, at.head_dim, 3 * at.embed_dim, needs_grad)

    def forward_rows(self, x, groups, valids):
        """x [rows, E]: several sequence groups back to back (see TransformerEncoderLayer.forward_rows)."""
        return self._layers(x, lambda layer, h, normed, nxt: layer.forward_rows(h, groups, valids, normed, nxt))

    def forward(self, src, mask=None, src_key_padding_mask=None):
        kind = mask_kind(mask)
        if kind == "general":  # the reference's loop (:19-37): every layer through its own forward with the mask
            out = src
            for layer in self.layers:
                out = layer(out, src_mask=mask, src_key_padding_mask=src_key_padding_mask)
            if self.norm is not None:
                out = ops.layer_norm(out, self.norm.weight, self.norm.bias, self.norm.eps)
            return out
        valid = None if src_key_padding_mask is None else ~src_key_padding_mask
        y = self.forward_batch_first(src.transpose(0, 1).contiguous(), valid, kind == "causal")
        return y.transpose(0, 1)

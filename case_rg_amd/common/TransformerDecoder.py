"""Decoder layers / stack on the HIP path (reference: common/TransformerDecoder.py:21-90, :95-164, :169-218).

    x = LN1(x); x += drop(SelfAttn(x; causal, key pad)); x = LN2(x); x += drop(CrossAttn(x, memory; key pad));
    x = LN3(x); x += drop(W2 drop(act(W1 x)))
The layers return ``(x, None, None)`` by default: the head-averaged attention weights of the reference are discarded by every
caller on the path.  Set ``return_attention = True`` on a layer / stack to get them (reference :77-90, :208-218:
the self-attention and memory-attention weights of the layer, the stack returns its LAST layer's), computed on request by
MultiheadAttention.averaged_weights.
"""
import torch
import torch.nn as nn

from .. import config, ops
from .attention import MultiheadAttention, is_causal_mask, mask_kind
from .Highway import Highway
from .TransformerEncoder import _check_activation, _get_clones


class RawMemory(object):
    """What TransformerDecoder.project_memory hands the greedy loop instead of per-layer K / V projections when the absorbed form (K21,
    ops.attention_decode_mqa) applies: the memory rows themselves, shared by every layer."""

    def __init__(self, rows):
        self.rows = rows


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu"):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.p = dropout
        self.activation = _check_activation(activation)

    def forward_batch_first(self, x, memory, tgt_valid=None, memory_valid=None, causal=True, memory_kv=None, normed=False, next_norm=None):
        """x [N, T, E]; memory [N, S, E]; *_valid bool True = token.  Every LayerNorm that follows an out-projection or the feed-forward pair
        is part of that op (``ln=``: ops.linear / ops.ffn); ``normed`` / ``next_norm`` as in TransformerEncoderLayer.forward_batch_first."""
        p = config.drop_p(self.p, self.training)
        if not normed:
            x = ops.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        x = self.self_attn.self_attention(x, tgt_valid, causal=causal, residual=x, p_res=self.p,
                                          ln=(self.norm2.weight, self.norm2.bias, self.norm2.eps))
        x = self.multihead_attn.cross_attention(x, memory, memory_valid, residual=x, p_res=self.p, kv=memory_kv,
                                                ln=(self.norm3.weight, self.norm3.bias, self.norm3.eps))
        return ops.ffn(x, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                       self.activation, p_inner=p, p_out=p, residual=x, ln=next_norm)

    def step(self, x, t, self_kv, hist_valid, memory_kv, memory_valid):
        """Incremental decoding of position ``t`` (inference): x [N, 1, E]; ``self_kv`` [N, Tmax, 2E] holds the K/V
        projections of positions < t of THIS layer and receives position t; ``hist_valid`` [N, Tmax] marks the positions
        <= t whose token is not PAD (the reference's tgt_key_padding_mask, CaSE/Model.py:105).  Equal to re-running the whole
        prefix (CaSE/Model.py:94-123) because of the causal mask: earlier positions never see later ones."""
        E = self.self_attn.embed_dim
        sa, ca = self.self_attn, self.multihead_attn
        n1, n2, n3 = [(n.weight, n.bias, n.eps) for n in (self.norm1, self.norm2, self.norm3)]
        # round 5: each LayerNorm rides in the prologue of the projection it feeds (ops.ln_linear -> case_gemm_ln) where that applies
        # (bf16, width 512, a multiple of 64 sequences); the normalised rows come back as the residual of the layer's next GEMM
        fused = ops.ln_gemm_supported(x, E) and self.linear1.out_features % 64 == 0 and self.activation in ("gelu", "relu")
        qkv, x = ops.ln_linear(x, n1, sa.in_proj_weight, sa.in_proj_bias)
        if ops.decode_append_supported(qkv, self_kv, sa.num_heads, sa.head_dim):  # the cache append rides in the attention launch
            ctx = ops.attention_decode_append(qkv, self_kv, t, sa.num_heads, sa.head_dim, key_valid=hist_valid)
        else:
            self_kv[:, t] = qkv[:, 0, E:]
            ctx = ops.attention(qkv, self_kv, self_kv, 0, 0, E, sa.num_heads, sa.head_dim, key_valid=hist_valid)
        x = ops.linear(ctx, sa.out_proj.weight, sa.out_proj.bias, residual=x)
        if isinstance(memory_kv, RawMemory):  # K21: attend the raw memory rows with the K / V projections absorbed into q and the output
            x = ca.cross_attention_absorbed(x, memory_kv.rows, memory_valid, ln_in=n2)
        else:
            x = ca.cross_attention(x, None, memory_valid, kv=memory_kv, ln_in=n2)
        if fused:
            h, x = ops.ln_linear(x, n3, self.linear1.weight, self.linear1.bias, act=self.activation)
            return ops.linear(h, self.linear2.weight, self.linear2.bias, residual=x)
        x = ops.layer_norm(x, *n3)
        return ops.ffn(x, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, self.activation,
                       residual=x)

    def _forward_general(self, tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask, memory_key_padding_mask):
        """Reference :76-89 line by line for masks the kernels' flags cannot express (a non-causal ``tgt_mask``, any ``memory_mask``): both
        attentions through MultiheadAttention.forward's additive-mask form.  No caller on the CaSE / Masque path passes such a mask (round 6)."""
        p = config.drop_p(self.p, self.training)
        x = ops.layer_norm(tgt, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        a, _ = self.self_attn(x, x, x, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask)
        x = ops.add(x, ops.dropout(a, self.p, self.training))
        x = ops.layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        a, _ = self.multihead_attn(x, memory, memory, attn_mask=memory_mask, key_padding_mask=memory_key_padding_mask)
        x = ops.add(x, ops.dropout(a, self.p, self.training))
        x = ops.layer_norm(x, self.norm3.weight, self.norm3.bias, self.norm3.eps)
        return ops.ffn(x, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, self.activation,
                       p_inner=p, p_out=p, residual=x)

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None):
        if memory_mask is not None or mask_kind(tgt_mask) == "general":
            if getattr(self, "return_attention", False):
                raise NotImplementedError("return_attention with an arbitrary tgt_mask / memory_mask is not built (no caller on the path)")
            return self._forward_general(tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask, memory_key_padding_mask), None, None
        tv = None if tgt_key_padding_mask is None else ~tgt_key_padding_mask
        mv = None if memory_key_padding_mask is None else ~memory_key_padding_mask
        xb, mb, causal = tgt.transpose(0, 1).contiguous(), memory.transpose(0, 1).contiguous(), is_causal_mask(tgt_mask)
        y = self.forward_batch_first(xb, mb, tv, mv, causal=causal)
        if not getattr(self, "return_attention", False):
            return y.transpose(0, 1), None, None
        # the weights of THIS layer's two attentions, recomputed from its inputs (eval-mode values: no dropout on the probabilities)
        n1 = ops.layer_norm(xb, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        w_self = self.self_attn.averaged_weights(n1, n1, tv, causal)
        with torch.no_grad():
            n2 = ops.layer_norm(self.self_attn.self_attention(n1, tv, causal=causal, residual=n1), self.norm2.weight, self.norm2.bias,
                                self.norm2.eps)
        w_mem = self.multihead_attn.averaged_weights(n2, mb, mv, False)
        return y.transpose(0, 1), w_self, w_mem


class GenericTransformerDecoderLayer(nn.Module):
    """Multi-memory variant that merges each attention through Highway(2E -> E) instead of a residual
    (reference :95-164; not instantiated by CaSE / Masque, kept for API completeness)."""

    def __init__(self, nmemory, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu"):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.self_norm = nn.LayerNorm(d_model)
        self.self_highway = Highway(2 * d_model, d_model)
        self.memory_attns = nn.ModuleList([MultiheadAttention(d_model, nhead, dropout=dropout) for _ in range(nmemory)])
        self.memory_norms = nn.ModuleList([nn.LayerNorm(d_model) for _ in range(nmemory)])
        self.memory_highways = nn.ModuleList([Highway(2 * d_model, d_model) for _ in range(nmemory)])
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.p = dropout
        self.activation = _check_activation(activation)

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None):
        p = config.drop_p(self.p, self.training)
        tv = None if tgt_key_padding_mask is None else ~tgt_key_padding_mask
        x = tgt.transpose(0, 1).contiguous()
        x = ops.layer_norm(x, self.self_norm.weight, self.self_norm.bias, self.self_norm.eps)
        a = self.self_attn.self_attention(x, tv, causal=is_causal_mask(tgt_mask), p_res=self.p)
        x = self.self_highway(torch.cat([x, a], dim=-1))
        for i, mem in enumerate(memory):
            norm = self.memory_norms[i]
            x = ops.layer_norm(x, norm.weight, norm.bias, norm.eps)
            kp = None if memory_key_padding_mask is None else memory_key_padding_mask[i]
            c = self.memory_attns[i].cross_attention(x, mem.transpose(0, 1).contiguous(), None if kp is None else ~kp,
                                                     p_res=self.p)
            x = self.memory_highways[i](torch.cat([x, c], dim=-1))
        y = ops.ffn(x, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, self.activation,
                    p_inner=p, p_out=p, residual=x)
        return y.transpose(0, 1), None, None


class TransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def forward_batch_first(self, x, memory, tgt_valid=None, memory_valid=None, causal=True, memory_kvs=None):
        # one alias of the memory per layer (each layer projects it to K / V): its gradient is then summed in one pass (ops.fanout)
        layers = list(self.layers)
        mems = ops.fanout(memory, len(layers)) if memory_kvs is None else (memory,) * len(layers)
        tail = None if self.norm is None else (self.norm.weight, self.norm.bias, self.norm.eps)
        for i, layer in enumerate(layers):
            nxt = layers[i + 1].norm1 if i + 1 < len(layers) else None
            x = layer.forward_batch_first(x, mems[i], tgt_valid, memory_valid, causal, None if memory_kvs is None else memory_kvs[i],
                                          normed=i > 0, next_norm=tail if nxt is None else (nxt.weight, nxt.bias, nxt.eps))
        return x

    def project_memory(self, memory, absorb=False):
        """Per-layer K/V projections of a memory (constant across greedy steps).  ``absorb`` (the greedy loop): for a long bf16 memory no
        projection is cached at all -- every layer's step attends the raw rows (RawMemory; half the bytes per step, no 2 E-wide caches)."""
        first = self.layers[0].multihead_attn
        if absorb and ops.decode_absorb_supported(memory, first.embed_dim, first.num_heads):
            raw = RawMemory(memory if memory.is_contiguous() else memory.contiguous())
            return [raw for _ in self.layers]
        return [layer.multihead_attn.project_memory(memory) for layer in self.layers]

    def new_self_cache(self, batch, max_len, like):
        E = self.layers[0].self_attn.embed_dim
        return [torch.zeros(batch, max_len, 2 * E, dtype=like.dtype, device=like.device) for _ in self.layers]

    def step(self, x, t, self_kvs, hist_valid, memory_kvs, memory_valid):
        for layer, skv, mkv in zip(self.layers, self_kvs, memory_kvs):
            x = layer.step(x, t, skv, hist_valid, mkv, memory_valid)
        if self.norm is not None:
            x = ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None):
        if memory_mask is not None or mask_kind(tgt_mask) == "general":  # the reference's loop (:189-218): every layer through its own forward
            out = tgt
            for layer in self.layers:
                out = layer(out, memory, tgt_mask=tgt_mask, memory_mask=memory_mask, tgt_key_padding_mask=tgt_key_padding_mask,
                            memory_key_padding_mask=memory_key_padding_mask)[0]
            if self.norm is not None:
                out = ops.layer_norm(out, self.norm.weight, self.norm.bias, self.norm.eps)
            return out, None, None
        tv = None if tgt_key_padding_mask is None else ~tgt_key_padding_mask
        mv = None if memory_key_padding_mask is None else ~memory_key_padding_mask
        if getattr(self, "return_attention", False):  # the reference's loop: every layer through its own forward, the last layer's weights
            out, w_self, w_mem = tgt, None, None
            for layer in self.layers:
                layer.return_attention = True
                try:
                    out, w_self, w_mem = layer(out, memory, tgt_mask=tgt_mask, tgt_key_padding_mask=tgt_key_padding_mask,
                                               memory_key_padding_mask=memory_key_padding_mask)
                finally:
                    layer.return_attention = False
            if self.norm is not None:
                out = ops.layer_norm(out, self.norm.weight, self.norm.bias, self.norm.eps)
            return out, w_self, w_mem
        y = self.forward_batch_first(tgt.transpose(0, 1).contiguous(), memory.transpose(0, 1).contiguous(), tv, mv,
                                     causal=is_causal_mask(tgt_mask))
        return y.transpose(0, 1), None, None

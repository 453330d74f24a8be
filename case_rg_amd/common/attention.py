"""Multi-head attention with torch's parameter schema, computed by the HIP path.

Parameters mirror ``nn.MultiheadAttention`` exactly (``in_proj_weight [3E,E]``, ``in_proj_bias [3E]``,
``out_proj.{weight,bias}``) so reference checkpoints load and ``init_params`` works.  The
arithmetic is: packed QKV projection (one MFMA GEMM), per-head S = QK^T / sqrt(d) with heads addressed by
batch strides inside the packed tensor, masked softmax (+dropout on the probabilities), O = PV written
head-interleaved, output projection with the residual and dropout folded into the GEMM epilogue.
Batch-first [N, L, E] inside; the sequence-first wrappers live in the layer modules.
"""
import torch
import torch.nn as nn

from .. import config, ops


class _OutProj(nn.Module):
    def __init__(self, width):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(width, width))
        self.bias = nn.Parameter(torch.zeros(width))


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        if embed_dim % num_heads:
            raise RuntimeError("embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.dropout = dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = _OutProj(embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.xavier_uniform_(self.out_proj.weight)

    # -- batch-first cores --------------------------------------------------------------------
    def self_attention(self, x, key_valid=None, causal=False, residual=None, p_res=0.0, ln=None):
        """x [N, L, E] -> out_proj(attention(x)) (+ dropout, + residual); ``ln`` = (gamma, beta, eps): the LayerNorm that follows, as part
        of the out-projection op (ops.linear)."""
        E = self.embed_dim
        if residual is x and torch.is_grad_enabled() and x.requires_grad:
            # x + out_proj(attention(in_proj(x))): both gradients of x meet in the in-projection's dX GEMM
            qkv, residual = ops.linear_carry(x, self.in_proj_weight, self.in_proj_bias)
        else:
            qkv = ops.linear(x, self.in_proj_weight, self.in_proj_bias)
        ctx = ops.attention(qkv, qkv, qkv, 0, E, 2 * E, self.num_heads, self.head_dim, key_valid=key_valid, causal=causal,
                            p_drop=config.drop_p(self.dropout, self.training))
        return ops.linear(ctx, self.out_proj.weight, self.out_proj.bias, residual=residual,
                          p_drop=config.drop_p(p_res, self.training), ln=ln)

    def project_memory(self, memory):
        """K/V projection of a memory [N, S, E] -> packed [N, S, 2E] (cacheable across decode steps)."""
        E = self.embed_dim
        return ops.linear(memory, self.in_proj_weight[E:], self.in_proj_bias[E:])

    # -- K21: decode-time cross-attention on the raw memory (absorbed K / V projections) ------------
    def absorbed(self):
        """The layer's projections folded for ops.attention_decode_mqa (inference; rebuilt when a parameter changes):
          wqk [heads E, E] bf16, bqk [heads E] f32:  qp_h = log2(e) / sqrt(d) Wk_h^T (Wq_h x + bq_h)   (q_h . bk_h is constant over the keys)
          wv  [E, E] bf16 (the V rows of in_proj_weight, head h = rows h d .. h d + d - 1: o_h = Wv_h c_h)
          bo  [E] f32 = out_proj.bias + out_proj.weight bv   (the probabilities of a row sum to one)"""
        E, h, d = self.embed_dim, self.num_heads, self.head_dim
        stamp = (ops.PARAM_EPOCH, self.in_proj_weight._version, self.in_proj_weight.data_ptr(), self.in_proj_bias._version,
                 self.out_proj.weight._version, self.out_proj.bias._version, self.out_proj.weight.data_ptr())
        hit = getattr(self, "_absorbed", None)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        with torch.no_grad():
            w, b = self.in_proj_weight.detach(), self.in_proj_bias.detach()
            c = 1.4426950408889634 / (d ** 0.5)
            wqk = torch.empty(h, E, E, dtype=torch.float32, device=w.device)
            # per head: Wk_h^T [E, d] x Wq_h [d, E], exact-f32 MFMA path; A = rows E + h d .. of w read k-major, B = rows h d .. read k-major
            ops.gemm(w, w, wqk, E, E, d, E, E, E, a_off=E * E, a_kmajor=True, b_kmajor=True, batch1=h, sa=(d * E, 0), sb=(d * E, 0),
                     sc=(E * E, 0), alpha=c)
            bqk = (w[E:2 * E].reshape(h, d, E) * b[:E].reshape(h, d, 1)).sum(dim=1).mul_(c).reshape(h * E).contiguous()
            bo = (self.out_proj.bias.detach() + (self.out_proj.weight.detach() * b[2 * E:].reshape(1, E)).sum(dim=1)).contiguous()
            pack = dict(wqk=ops.cast(wqk.reshape(h * E, E), torch.bfloat16), bqk=bqk.float(), bo=bo.float(),
                        wv=ops.cast_param(self.in_proj_weight[2 * E:], torch.bfloat16), wo=ops.cast_param(self.out_proj.weight, torch.bfloat16))
        self._absorbed = (stamp, pack)
        return pack

    def cross_attention_absorbed(self, x, memory, memory_valid=None, residual=None, ln_in=None):
        """x [N, 1, E] (one decode position per sequence), memory [N, S, E] RAW bf16 rows -> out_proj(attention) (+ residual).
        ``ln_in`` = (gamma, beta, eps): x is normalised first (in the prologue of the query projection) and LN(x) is the residual."""
        E, h, d = self.embed_dim, self.num_heads, self.head_dim
        N = x.shape[0]
        pk = self.absorbed()
        if ln_in is not None:
            qp, residual = ops.ln_linear(x.reshape(N, E), ln_in, pk["wqk"], pk["bqk"])
        else:
            qp = ops.linear(x.reshape(N, E), pk["wqk"], pk["bqk"])        # [N, heads E]
        ctx = ops.attention_decode_mqa(qp, memory, memory_valid)          # [N, heads E]: head h's context at columns E h ..
        o = torch.empty(N, E, dtype=ctx.dtype, device=ctx.device)
        # o[:, h d .. h d + d) = ctx[:, E h .. E h + E) Wv_h^T, the eight heads as one batched launch
        ops.gemm(ctx, pk["wv"], o, N, d, E, h * E, E, E, batch1=h, sa=(E, 0), sb=(d * E, 0), sc=(d, 0))
        res2 = None if residual is None else residual.reshape(N, E)
        return ops.linear(o, pk["wo"], pk["bo"], residual=res2).reshape(N, 1, E)

    def cross_attention(self, x, memory, memory_valid=None, residual=None, p_res=0.0, kv=None, ln=None, ln_in=None):
        """x [N, Lq, E], memory [N, S, E] (or a precomputed ``kv``) -> out_proj(attention); ``ln``: see self_attention.  ``ln_in`` (decode
        step, with ``kv``): x is normalised first, in the prologue of the query projection, and LN(x) is the residual."""
        E = self.embed_dim
        if ln_in is not None:
            if kv is None:
                raise RuntimeError("cross_attention(ln_in=...) is the decode-step form: pass the cached projections as kv")
            q, residual = ops.ln_linear(x, ln_in, self.in_proj_weight[:E], self.in_proj_bias[:E])
        elif kv is None:
            # both halves of the packed in-projection are used here: one gradient concatenation instead of two zero-filled slices + an add
            (wq, wkv), (bq, bkv) = ops.split_param_rows(self.in_proj_weight, E), ops.split_param_rows(self.in_proj_bias, E)
            q = ops.linear(x, wq, bq)
            kv = ops.linear(memory, wkv, bkv)
        else:
            q = ops.linear(x, self.in_proj_weight[:E], self.in_proj_bias[:E])
        ctx = ops.attention(q, kv, kv, 0, 0, E, self.num_heads, self.head_dim, key_valid=memory_valid,
                            p_drop=config.drop_p(self.dropout, self.training))
        return ops.linear(ctx, self.out_proj.weight, self.out_proj.bias, residual=residual,
                          p_drop=config.drop_p(p_res, self.training), ln=ln)

    # -- nn.MultiheadAttention-compatible call (sequence-first) ---------------------------------
    def averaged_weights(self, query, key, key_valid=None, causal=False, add_mask=None):
        """Head-averaged attention probabilities [N, Lq, Lk] of batch-first ``query`` [N, Lq, E] over ``key`` [N, Lk, E] -- what
        nn.MultiheadAttention returns as its second value (F.multi_head_attention_forward: softmax weights summed over the heads / heads).
        Computed on request only, by the score GEMM + masked softmax (no dropout: the reference's callers read them in eval mode, if at
        all -- every call site on the CaSE / Masque path discards them); not differentiable."""
        E, h, d = self.embed_dim, self.num_heads, self.head_dim
        with torch.no_grad():
            q = ops.linear(query, self.in_proj_weight[:E], self.in_proj_bias[:E])
            k = ops.linear(key, self.in_proj_weight[E:2 * E], self.in_proj_bias[E:2 * E])
            P, _ = ops.AttentionFn._probabilities(q.contiguous(), k.contiguous(), 0, 0, h, d, ops._u8(key_valid), causal, None, 1.0 / (d ** 0.5), add_mask)
            return P.float().mean(dim=1)

    def forward(self, query, key, value, attn_mask=None, key_padding_mask=None, need_weights=False):
        """[L, N, E] tensors; ``key_padding_mask`` True = pad.  Returns (out [Lq, N, E], weights): ``weights`` is None unless
        ``need_weights`` (then the head-averaged probabilities [N, Lq, Lk], see ``averaged_weights``).  ``key`` and ``value`` may be
        different tensors of the same length (projected with their own rows of the packed in-projection).  ``attn_mask``: the square
        subsequent (causal) mask takes the kernels' causal flag; any other 2-D float / bool mask is added to the scores (split_attn_mask)."""
        valid = None if key_padding_mask is None else ~key_padding_mask
        q = query.transpose(0, 1).contiguous()
        causal, add_mask = split_attn_mask(attn_mask)
        if add_mask is None and key is value and query is key:
            out = self.self_attention(q, valid, causal=causal)
            kb = q
        elif add_mask is None and key is value and not causal:
            kb = key.transpose(0, 1).contiguous()
            out = self.cross_attention(q, kb, valid)
        else:  # the general form: three projections, the attention core on three sources
            E = self.embed_dim
            kb, vb = key.transpose(0, 1).contiguous(), value.transpose(0, 1).contiguous()
            if kb.shape[:2] != vb.shape[:2]:
                raise RuntimeError("key and value must have the same sequence length and batch size")
            w, b = self.in_proj_weight, self.in_proj_bias
            qp, kp, vp = ops.linear(q, w[:E], b[:E]), ops.linear(kb, w[E:2 * E], b[E:2 * E]), ops.linear(vb, w[2 * E:], b[2 * E:])
            ctx = ops.attention(qp, kp, vp, 0, 0, 0, self.num_heads, self.head_dim, key_valid=valid, causal=causal,
                                p_drop=config.drop_p(self.dropout, self.training), add_mask=add_mask)
            out = ops.linear(ctx, self.out_proj.weight, self.out_proj.bias)
        weights = self.averaged_weights(q, kb, valid, causal, add_mask) if need_weights else None
        return out.transpose(0, 1), weights


def split_attn_mask(mask):
    """nn.MultiheadAttention's ``attn_mask`` -> (causal flag, additive f32 [Lq, Lk] mask or None).  The square subsequent mask becomes the
    kernels' causal flag (tagged by generate_square_subsequent_mask, or recognised by one host comparison); ANY other 2-D mask is taken as
    an additive mask on the scores -- float as given, bool with True = masked (-inf) as torch defines it -- and runs the GEMM + softmax + GEMM
    path (round 5; a fully masked row yields NaN, as in torch)."""
    if mask is None:
        return False, None
    if getattr(mask, "_case_causal", False):
        return True, None
    if mask.dim() not in (2, 3):
        raise RuntimeError("attn_mask must be 2-D [Lq, Lk] or 3-D [N * heads, Lq, Lk], as nn.MultiheadAttention takes it")
    n = mask.size(0)
    if mask.dim() == 2 and mask.shape == (n, n) and mask.dtype != torch.bool:
        upper = torch.triu(torch.ones(n, n, dtype=torch.bool, device=mask.device), 1)
        if bool(((mask < -1e9) == upper).all()) and bool((mask.masked_select(~upper) == 0).all()):
            return True, None
    if mask.dtype == torch.bool:
        return False, torch.zeros(mask.shape, dtype=torch.float32, device=mask.device).masked_fill(mask, float("-inf"))
    return False, mask.to(torch.float32)


def mask_kind(mask):
    """"none" | "causal" | "general" for a layer-level ``src_mask`` / ``tgt_mask`` / ``memory_mask`` (round 6): the causal pattern keeps the
    kernels' causal flag; anything else a 2-D mask can say goes through MultiheadAttention.forward's additive form (split_attn_mask)."""
    if mask is None:
        return "none"
    return "causal" if split_attn_mask(mask)[0] else "general"


def is_causal_mask(mask):
    """None -> False; a mask built by ``generate_square_subsequent_mask`` (tagged) -> True; any other float
    mask is checked once on the host against the causal pattern, everything else is rejected."""
    if mask is None:
        return False
    if getattr(mask, "_case_causal", False):
        return True
    n = mask.size(0)
    upper = torch.triu(torch.ones(n, n, dtype=torch.bool, device=mask.device), 1)
    if mask.shape == (n, n) and bool(((mask < -1e9) == upper).all()):
        return True
    raise NotImplementedError("only the square subsequent (causal) attention mask is supported")

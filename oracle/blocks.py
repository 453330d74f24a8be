"""Oracle building blocks (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Plain fp32 PyTorch-CPU restatement of the reference's L1 blocks.  Everything is written as
explicit tensor algebra (no nn.MultiheadAttention / nn.Transformer*), batch-first inside, with the
reference's parameter names so one deterministic filler drives reference, oracle and product.

Dropout: the reference applies dropout in >= 8 places; bit-parity with torch's RNG is impossible,
so the oracle computes the dropout-free function (p treated as 0) -- the same convention the golden
generator uses (it patches dropout to identity in the reference).
"""
import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

NEG_BIG = -1e20  # reference: common/Utils.py:14-21 (neginf for fp32)


# --------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------
def sinusoid_table(max_len, width):
    """pe[p, 2i] = sin(p * w_i), pe[p, 2i+1] = cos(p * w_i), w_i = exp(-2i ln(1e4)/width).
    Reference: common/PositionalEmbedding.py:27-32."""
    pos = torch.arange(max_len, dtype=torch.float32)[:, None]
    freq = torch.exp(torch.arange(0, width, 2, dtype=torch.float32) * (-math.log(10000.0) / width))
    table = torch.zeros(max_len, width)
    table[:, 0::2] = torch.sin(pos * freq)
    table[:, 1::2] = torch.cos(pos * freq)
    return table


def causal_additive_mask(n):
    """[n, n] float mask: 0 on/below the diagonal, -1e20 above (NOT -inf).
    Reference: common/Utils.py:23-28."""
    allowed = torch.tril(torch.ones(n, n, dtype=torch.bool))
    return torch.zeros(n, n).masked_fill(~allowed, NEG_BIG)


def masked_mean(x, valid):
    """sum_l valid*x / sum_l valid.  Reference: common/Utils.py:455-470 (sqrt=False)."""
    w = valid.to(x.dtype)
    return (x * w[..., None]).sum(dim=-2) / w.sum(dim=-1, keepdim=True)


def one_hot_map(ids, vocab):
    """Dense one-hot copy map [B, S, V].  Reference: common/Utils.py:344-355."""
    return F.one_hot(ids, vocab).to(torch.float32)


def _act(name):
    if name == "relu":
        return F.relu
    if name == "gelu":
        return F.gelu  # erf form
    raise RuntimeError("activation should be relu/gelu, not %s." % name)  # TransformerEncoder.py:17


# --------------------------------------------------------------------------------------------
# PositionalEmbedding -- common/PositionalEmbedding.py:22-48
# --------------------------------------------------------------------------------------------
class PositionalEmbedding(nn.Module):
    def __init__(self, embedding_size, dropout=0.1, max_len=5000):
        super().__init__()
        self.embedding_size = embedding_size
        self.register_buffer("pe", sinusoid_table(max_len, embedding_size))

    def forward(self, x):
        length = x.size(-2)
        return x * math.sqrt(self.embedding_size) + self.pe[:length]


# --------------------------------------------------------------------------------------------
# Multi-head attention with torch's parameter schema (torch F.multi_head_attention_forward
# slow path: q scaled by d^-1/2, additive float mask, key padding -> -inf, fp32 softmax)
# --------------------------------------------------------------------------------------------
class _OutProj(nn.Module):
    def __init__(self, width):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(width, width))
        self.bias = nn.Parameter(torch.zeros(width))


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = _OutProj(embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.xavier_uniform_(self.out_proj.weight)

    def forward(self, query, key, value, attn_mask=None, key_padding_mask=None):
        """Sequence-first [L, N, E] in and out (as nn.MultiheadAttention, batch_first=False).
        Returns (out [Lq, N, E], head-averaged weights [N, Lq, Lk])."""
        E, h = self.embed_dim, self.num_heads
        d = E // h
        Lq, N, _ = query.shape
        Lk = key.shape[0]
        wq, wk, wv = self.in_proj_weight.split(E, dim=0)
        bq, bk, bv = self.in_proj_bias.split(E, dim=0)
        q = (query @ wq.t() + bq) * (1.0 / math.sqrt(d))
        k = key @ wk.t() + bk
        v = value @ wv.t() + bv
        # [L, N, h, d] -> [N, h, L, d]
        q = q.reshape(Lq, N, h, d).permute(1, 2, 0, 3)
        k = k.reshape(Lk, N, h, d).permute(1, 2, 0, 3)
        v = v.reshape(Lk, N, h, d).permute(1, 2, 0, 3)
        scores = q @ k.transpose(-1, -2)  # [N, h, Lq, Lk]
        if attn_mask is not None:
            scores = scores + attn_mask
        if key_padding_mask is not None:
            scores = scores.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
        prob = torch.softmax(scores, dim=-1)
        ctx = (prob @ v).permute(2, 0, 1, 3).reshape(Lq, N, E)
        out = ctx @ self.out_proj.weight.t() + self.out_proj.bias
        return out, prob.mean(dim=1)


# --------------------------------------------------------------------------------------------
# Encoder -- common/TransformerEncoder.py:19-77 (layer), :82-123 (stack)
# --------------------------------------------------------------------------------------------
class TransformerEncoderLayer(nn.Module):
    """s = LN1(x); s = s + MHA(s); s = LN2(s); s = s + W2 act(W1 s)  -- the residual is taken from
    the *normed* tensor (TransformerEncoder.py:66-75)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu"):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.activation = _act(activation)

    def forward(self, src, src_mask=None, src_key_padding_mask=None):
        s = self.norm1(src)
        s = s + self.self_attn(s, s, s, attn_mask=src_mask, key_padding_mask=src_key_padding_mask)[0]
        s = self.norm2(s)
        return s + self.linear2(self.activation(self.linear1(s)))


class TransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, src, mask=None, src_key_padding_mask=None):
        out = src
        for layer in self.layers:
            out = layer(out, src_mask=mask, src_key_padding_mask=src_key_padding_mask)
        return self.norm(out) if self.norm is not None else out


# --------------------------------------------------------------------------------------------
# Decoder -- common/TransformerDecoder.py:21-90 (layer), :95-164 (generic/highway layer), :169-218
# --------------------------------------------------------------------------------------------
class TransformerDecoderLayer(nn.Module):
    """x=LN1(x); x+=SelfAttn(x; causal -1e20 + key pad); x=LN2(x); x+=CrossAttn(x, mem; key pad);
    x=LN3(x); x+=W2 act(W1 x).  Returns (x, self weights, memory weights)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu"):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead)
        self.multihead_attn = MultiheadAttention(d_model, nhead)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.activation = _act(activation)

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None,
                tgt_key_padding_mask=None, memory_key_padding_mask=None):
        x = self.norm1(tgt)
        a, w_self = self.self_attn(x, x, x, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask)
        x = self.norm2(x + a)
        c, w_mem = self.multihead_attn(x, memory, memory, attn_mask=memory_mask,
                                       key_padding_mask=memory_key_padding_mask)
        x = self.norm3(x + c)
        return x + self.linear2(self.activation(self.linear1(x))), w_self, w_mem


class Highway(nn.Module):
    """x <- sigma(G x) * f(N x) + (1 - sigma(G x)) * (L x), per layer.  common/Highway.py:5-37."""

    def __init__(self, input_size, output_size, num_layers=1, f=torch.tanh):
        super().__init__()
        self.num_layers = num_layers
        self.nonlinear = nn.ModuleList([nn.Linear(input_size, output_size) for _ in range(num_layers)])
        self.linear = nn.ModuleList([nn.Linear(input_size, output_size) for _ in range(num_layers)])
        self.gate = nn.ModuleList([nn.Linear(input_size, output_size) for _ in range(num_layers)])
        self.f = f

    def forward(self, x):
        for n, l, g in zip(self.nonlinear, self.linear, self.gate):
            t = torch.sigmoid(g(x))
            x = t * self.f(n(x)) + (1.0 - t) * l(x)
        return x


class GenericTransformerDecoderLayer(nn.Module):
    """Multi-memory decoder layer that merges each attention through Highway(2E->E) instead of a
    residual.  common/TransformerDecoder.py:95-164 (never instantiated by CaSE/Masque)."""

    def __init__(self, nmemory, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu"):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead)
        self.self_norm = nn.LayerNorm(d_model)
        self.self_highway = Highway(2 * d_model, d_model)
        self.memory_attns = nn.ModuleList([MultiheadAttention(d_model, nhead) for _ in range(nmemory)])
        self.memory_norms = nn.ModuleList([nn.LayerNorm(d_model) for _ in range(nmemory)])
        self.memory_highways = nn.ModuleList([Highway(2 * d_model, d_model) for _ in range(nmemory)])
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.activation = _act(activation)

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None,
                tgt_key_padding_mask=None, memory_key_padding_mask=None):
        x = self.self_norm(tgt)
        a, w_self = self.self_attn(x, x, x, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask)
        x = self.self_highway(torch.cat([x, a], dim=-1))
        w_mems = []
        for i, mem in enumerate(memory):
            x = self.memory_norms[i](x)
            mm = None if memory_mask is None else memory_mask[i]
            kp = None if memory_key_padding_mask is None else memory_key_padding_mask[i]
            c, w = self.memory_attns[i](x, mem, mem, attn_mask=mm, key_padding_mask=kp)
            w_mems.append(w)
            x = self.memory_highways[i](torch.cat([x, c], dim=-1))
        return x + self.linear2(self.activation(self.linear1(x))), w_self, w_mems


class TransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(decoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None,
                tgt_key_padding_mask=None, memory_key_padding_mask=None):
        out, w_self, w_mem = tgt, None, None
        for layer in self.layers:
            out, w_self, w_mem = layer(out, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                                       tgt_key_padding_mask=tgt_key_padding_mask,
                                       memory_key_padding_mask=memory_key_padding_mask)
        if self.norm is not None:
            out = self.norm(out)
        return out, w_self, w_mem


# --------------------------------------------------------------------------------------------
# TransformerBlock -- common/TransformerBlock.py:7-32
# --------------------------------------------------------------------------------------------
class TransformerBlock(nn.Module):
    """r = x + MHA(LN1(x)) (residual from the UN-normed input); y = W2 act(W1 LN2(r)) with no
    residual; y zeroed at pads.  input [B, N, L, Ein], mask [B, N, L] True=valid."""

    def __init__(self, num_heads, input_hidden_size, output_hidden_size, activation=None):
        super().__init__()
        self.output_hidden_size = output_hidden_size
        self.self_attn = MultiheadAttention(input_hidden_size, num_heads)
        self.norm1 = nn.LayerNorm(input_hidden_size)
        self.norm2 = nn.LayerNorm(input_hidden_size)
        self.linear1 = nn.Linear(input_hidden_size, output_hidden_size)
        self.linear2 = nn.Linear(output_hidden_size, output_hidden_size)
        self.activation = F.relu if activation is None else activation

    def forward(self, input, input_mask):
        B, N, L, E = input.shape
        x = input.reshape(B * N, L, E)
        valid = input_mask.reshape(B * N, L)
        n1 = self.norm1(x).transpose(0, 1)
        a = self.self_attn(n1, n1, n1, key_padding_mask=~valid)[0].transpose(0, 1)
        r = x + a
        y = self.linear2(self.activation(self.linear1(self.norm2(r))))
        y = y.reshape(B, N, L, self.output_hidden_size)
        return y * input_mask[..., None].to(y.dtype)


# --------------------------------------------------------------------------------------------
# BilinearAttention (additive / Bahdanau despite the name) -- common/BilinearAttention.py:5-59
# --------------------------------------------------------------------------------------------
class BilinearAttention(nn.Module):
    def __init__(self, query_size, key_size, hidden_size):
        super().__init__()
        self.linear_key = nn.Linear(key_size, hidden_size, bias=False)
        self.linear_query = nn.Linear(query_size, hidden_size, bias=True)
        self.v = nn.Linear(hidden_size, 1, bias=False)
        self.hidden_size = hidden_size

    def matching(self, query, key, mask=None):
        """s[.., t, j] = v . tanh(Wq q_t + b + Wk k_j); masked -> -inf.  (:24-46)"""
        wq = self.linear_query(query)[..., :, None, :]
        uh = self.linear_key(key)[..., None, :, :]
        s = torch.tanh(wq + uh) @ self.v.weight[0]
        if mask is not None:
            s = s.masked_fill(~mask, float("-inf"))
        return s

    def score(self, query, key, softmax_dim=-1, mask=None):
        """softmax then zero where masked; an all-masked row (NaN) becomes 0.  (:13-21)"""
        s = self.matching(query, key, mask)
        p = torch.softmax(s, dim=softmax_dim)
        if mask is not None:
            p = p.masked_fill(~mask, 0.0)
        return s, p

    def forward(self, query, key, value, mask=None):
        s, p = self.score(query, key, mask=mask)
        ctx = p.reshape(-1, p.size(-2), p.size(-1)) @ value.reshape(-1, value.size(-2), value.size(-1))
        return ctx.reshape(list(value.shape[:-2]) + [p.size(-2), -1]), s, p


# --------------------------------------------------------------------------------------------
# Interaction (BiDAF/DCN dual co-attention) -- common/Interaction.py:5-75, SURVEY A.6
# --------------------------------------------------------------------------------------------
class Interaction(nn.Module):
    """U[n,i,j] = w1.Eq[j] + w2.Ep[i] + (w3*Ep[i]).Eq[j]  (identical to Linear(cat[Eq,Ep,Eq*Ep]),
    Interaction.py:32-36, without materialising [n, Lp, Lq, 3H]); A = softmax_j, Bm = softmax_i, both
    zeroed where masked; A1 = A Eq, B1 = Bm^T Ep, A2 = A B1, B2 = Bm^T A1;
    G_q_p = [Ep, A1, A2, Ep*A1, Ep*A2], G_p_q = [Eq, B1, B2, Eq*B1, Eq*B2], zeroed at pads;
    when num_q == 1 != num_p the query side is max over passages (:73-74)."""

    def __init__(self, hidden_size):
        super().__init__()
        self.hidden_size = hidden_size
        self.dual_att_linear = nn.Linear(3 * hidden_size, 1, bias=False)

    def forward(self, encode_input1, encode_input2, input1_mask, input2_mask):
        B, nq, Lq, H = encode_input1.shape
        _, npass, Lp, _ = encode_input2.shape
        if nq != npass:
            assert nq == 1  # Interaction.py:27
            Eq = encode_input1.expand(-1, npass, -1, -1)
            qmask = input1_mask.expand(-1, npass, -1)
        else:
            Eq, qmask = encode_input1, input1_mask
        Eq = Eq.reshape(B * npass, Lq, H)
        Ep = encode_input2.reshape(B * npass, Lp, H)
        qv = qmask.reshape(B * npass, Lq)
        pv = input2_mask.reshape(B * npass, Lp)
        w1, w2, w3 = self.dual_att_linear.weight[0].split(H)
        U = (Eq @ w1)[:, None, :] + (Ep @ w2)[:, :, None] + (Ep * w3) @ Eq.transpose(1, 2)
        both = pv[:, :, None] & qv[:, None, :]
        U = U.masked_fill(~both, float("-inf"))
        A = torch.softmax(U, dim=2).masked_fill(~both, 0.0)
        Bm = torch.softmax(U, dim=1).masked_fill(~both, 0.0)
        A1 = A @ Eq
        B1 = Bm.transpose(1, 2) @ Ep
        A2 = A @ B1
        B2 = Bm.transpose(1, 2) @ A1
        G_q_p = torch.cat([Ep, A1, A2, Ep * A1, Ep * A2], dim=-1).reshape(B, npass, Lp, 5 * H)
        G_p_q = torch.cat([Eq, B1, B2, Eq * B1, Eq * B2], dim=-1).reshape(B, npass, Lq, 5 * H)
        G_p_q = G_p_q.masked_fill(~qmask.reshape(B, npass, Lq)[..., None], 0.0)
        G_q_p = G_q_p.masked_fill(~input2_mask[..., None], 0.0)
        if nq != npass:
            G_p_q = G_p_q.max(dim=1, keepdim=True)[0]
        return G_p_q, G_q_p


# reference-named aliases for the helper functions (common/Utils.py) so shared test cases can call
# reference, oracle and product through one namespace
generate_square_subsequent_mask = causal_additive_mask


def build_map(b_map, max=None):
    return one_hot_map(b_map, int(b_map.max()) + 1 if max is None else max)


def universal_sentence_embedding(sentences, mask, sqrt=False):
    assert not sqrt
    return masked_mean(sentences, mask)


def topk(gen_output, k=1):
    """k == 1 only: (max value, lowest argmax index), keepdim.  common/Utils.py:156-168."""
    assert k == 1
    return torch.max(gen_output, dim=1, keepdim=True)


# ---- answer post-processing (reference: common/Utils.py:180-217) ----------------------------------------------------------
def to_sentence(batch_indices, id2vocab):
    """common/Utils.py:200-217: per row skip [unused0] (BOS) and [PAD], stop at [unused1] (EOS), empty -> [[UNK]]."""
    out = []
    for row in batch_indices:
        words = []
        for index in row:
            w = id2vocab[int(index)]
            if w in ("[unused0]", "[PAD]"):
                continue
            if w == "[unused1]":
                break
            words.append(w)
        out.append(words if words else ["[UNK]"])
    return out


def remove_duplicate(sents, n=3):
    """common/Utils.py:180-198: repeatedly cut the shortest tail (>= n tokens) whose tokens all occur earlier in the sentence."""
    def once():
        changed = False
        for b in range(len(sents)):
            sent = sents[b]
            if len(sent) <= n:
                continue
            for i in range(len(sent) - n):
                index = len(sent) - i - n
                if all(elem in sent[:index] for elem in sent[index:]):
                    sents[b] = sent[:index]
                    changed = True
                    break
        return changed
    while once():
        pass

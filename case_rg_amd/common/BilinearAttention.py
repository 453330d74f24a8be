"""Additive (Bahdanau) attention -- named BilinearAttention in the reference (common/BilinearAttention.py:5-59).

    s[b,t,j] = v . tanh(Wq q[b,t] + bq + Wk k[b,j]);  p = softmax_j(s | mask), 0 where masked;  ctx = p @ value
The [B, T, S, H] tanh tensor of the reference (10-40 GB at the BASELINE sizes) is never formed: the fused
kernel K7 keeps the H-sum in registers.  Masks on the path are outer products (target valid x memory valid,
CaSE/Model.py:79), which the softmax kernel takes as a row mask and a column mask.
"""
import torch
import torch.nn as nn

from .. import config, ops


class BilinearAttention(nn.Module):
    def __init__(self, query_size, key_size, hidden_size):
        super().__init__()
        self.linear_key = nn.Linear(key_size, hidden_size, bias=False)
        self.linear_query = nn.Linear(query_size, hidden_size, bias=True)
        self.v = nn.Linear(hidden_size, 1, bias=False)
        self.hidden_size = hidden_size

    def project_keys(self, key):
        """uh = Wk k  [B, S, H] (constant across greedy steps)."""
        return ops.linear(key, self.linear_key.weight)

    def project_keys_exp(self, key):
        """e^{2 Wk k} in bf16 [B, S, H] for the fused decode step (K22): the exponential of the key half of tanh(wq + uh) is the same in
        every greedy step, so it is taken once, from the f32 projection."""
        return ops.additive_key_exp(ops.linear(key, self.linear_key.weight, out_dtype=torch.float32))

    def split_query(self, feature, x_width):
        """For a greedy pass whose queries are [x_t | feature] with a ``feature`` [B, 1, F] that does not change over the steps (CaSE/Model.py:77-78):
        (Wx, c) with Wx = the first ``x_width`` columns of the query projection (contiguous, compute dtype) and c = feature Wf^T + b in f32 [B, H] --
        a step then projects x_t alone (K = x_width instead of x_width + F: the small-problem GEMM, no concatenation) and K22 adds c."""
        W, b = self.linear_query.weight.detach(), self.linear_query.bias
        if W.shape[1] != x_width + feature.shape[-1]:
            raise ValueError("split_query: query width %d != %d + %d" % (W.shape[1], x_width, feature.shape[-1]))
        wx = W[:, :x_width].to(feature.dtype).contiguous()
        wf = W[:, x_width:].to(feature.dtype).contiguous()
        c = ops.linear(feature, wf, b, out_dtype=torch.float32)
        return wx, c.reshape(feature.shape[0], -1).contiguous()

    def attend_decode(self, query, value, row_valid, col_valid, eu, prior=None, split=None):
        """One decode position per sequence: query [B, 1, Q] -> (ctx [B, 1, Hv], p or, with ``prior``, p prior / (1e-8 + sum p prior) [B, 1, S]).
        ``split`` = split_query(...): ``query`` is x_t alone."""
        B = query.shape[0]
        if split is not None:
            wq, add = ops.linear(query, split[0], None, out_dtype=torch.float32), split[1]
        else:
            wq, add = ops.linear(query, self.linear_query.weight, self.linear_query.bias, out_dtype=torch.float32), None
        ctx, p, copy = ops.pointer_attend_decode(wq, eu, self.v.weight.detach().reshape(-1).float(), value, col_valid,
                                                 None if row_valid is None else row_valid.reshape(B), prior, wq_add=add)
        return ctx.unsqueeze(1), (p if copy is None else copy).unsqueeze(1)

    def raw_scores(self, query, key=None, uh=None):
        wq = ops.linear(query, self.linear_query.weight, self.linear_query.bias, out_dtype=torch.float32)
        if uh is None:
            uh = self.project_keys(key)
        return ops.additive_scores(wq, uh, self.v.weight.reshape(-1))

    def attend(self, query, key, value, row_valid=None, col_valid=None, uh=None):
        """Hot-path entry: masks as validity vectors.  Returns (ctx [B,T,Hv], p f32 [B,T,S])."""
        B = query.shape[0]
        s = self.raw_scores(query, key, uh)
        p = ops.masked_softmax(s, col_valid, row_valid, outer=B)
        ctx = ops.bmm(ops.cast_to(p, value.dtype), value, b_is_kn=True)
        return ctx, p

    @staticmethod
    def _split_mask(mask):
        """(row valid, column valid) when ``mask`` is their outer product -- what the softmax kernel takes directly -- else None."""
        rv, cv = mask.any(dim=-1), mask.any(dim=-2)
        if not torch.equal(mask, rv.unsqueeze(-1) & cv.unsqueeze(-2)):
            return None
        return rv, cv

    def _probabilities(self, s, mask, softmax_dim=-1):
        """softmax of the raw scores s [B, T, S] under ``mask`` [B, T, S] (True = admissible), 0 where masked and for rows / columns without
        an admissible entry (reference :13-21: softmax, then masked_fill(~mask, 0)).  Outer-product masks -- the only kind on the CaSE /
        Masque path -- ride in the softmax kernel as a row and a column vector; any other mask (round 6) is applied to the scores first.
        ``softmax_dim`` -2 normalises over the query axis (the kernel normalises the last axis of the transposed scores)."""
        B = s.shape[0]
        if softmax_dim in (-2, s.dim() - 2):
            return self._probabilities(s.transpose(1, 2), None if mask is None else mask.transpose(1, 2)).transpose(1, 2)
        if mask is None:
            return ops.masked_softmax(s, None, None, outer=B)
        split = self._split_mask(mask)
        if split is not None:
            return ops.masked_softmax(s, split[1], split[0], outer=B)
        s = s.contiguous().masked_fill(~mask, -float('inf'))
        return ops.masked_softmax(s, None, None, outer=B)  # exp(-inf) = 0; a row of -inf alone gives exact zeros

    @staticmethod
    def _flat(t):
        return t.reshape(-1, t.shape[-2], t.shape[-1])

    def matching(self, query, key, mask=None):
        """[B, *, T, Q] x [B, *, S, K] -> raw scores [B, *, T, S], masked -> -inf (reference :24-46)."""
        lead = query.shape[:-2]
        s = self.raw_scores(self._flat(query), self._flat(key)).reshape(*lead, query.shape[-2], key.shape[-2])
        return s if mask is None else s.masked_fill(~mask, -float('inf'))

    def score(self, query, key, softmax_dim=-1, mask=None):
        lead, nd = query.shape[:-2], query.dim()
        if softmax_dim not in (-1, -2, nd - 1, nd - 2):
            raise NotImplementedError("softmax over the key or the query axis only")
        s = self.raw_scores(self._flat(query), self._flat(key))
        p = self._probabilities(s, None if mask is None else self._flat(mask), -1 if softmax_dim in (-1, nd - 1) else -2)
        s = s.reshape(*lead, *s.shape[-2:])
        return (s if mask is None else s.masked_fill(~mask, -float('inf'))), p.reshape(*lead, *p.shape[-2:])

    def forward(self, query, key, value, mask=None):
        """query [B, *, T, Q], key [B, *, S, K], value [B, *, S, Hv], mask [B, *, T, S] bool -> (ctx [B, *, T, Hv], raw scores, p)."""
        lead = query.shape[:-2]
        s = self.raw_scores(self._flat(query), self._flat(key))
        p = self._probabilities(s, None if mask is None else self._flat(mask))
        v3 = self._flat(value)
        ctx = ops.bmm(ops.cast_to(p, v3.dtype), v3, b_is_kn=True)
        s = s.reshape(*lead, *s.shape[-2:])
        return (ctx.reshape(*lead, ctx.shape[-2], ctx.shape[-1]), (s if mask is None else s.masked_fill(~mask, -float('inf'))),
                p.reshape(*lead, *p.shape[-2:]))

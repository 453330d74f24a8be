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

    def attend_decode(self, query, value, row_valid, col_valid, eu, prior=None):
        """One decode position per sequence: query [B, 1, Q] -> (ctx [B, 1, Hv], p or, with ``prior``, p prior / (1e-8 + sum p prior) [B, 1, S])."""
        B = query.shape[0]
        wq = ops.linear(query, self.linear_query.weight, self.linear_query.bias, out_dtype=torch.float32)
        ctx, p, copy = ops.pointer_attend_decode(wq, eu, self.v.weight.detach().reshape(-1).float(), value, col_valid,
                                                 None if row_valid is None else row_valid.reshape(B), prior)
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
        rv, cv = mask.any(dim=-1), mask.any(dim=-2)
        if not torch.equal(mask, rv.unsqueeze(-1) & cv.unsqueeze(-2)):
            raise NotImplementedError("BilinearAttention on the HIP path takes outer-product masks (row valid x column valid)")
        return rv, cv

    def matching(self, query, key, mask=None):
        s = self.raw_scores(query, key)
        return s if mask is None else s.masked_fill(~mask, -float('inf'))

    def score(self, query, key, softmax_dim=-1, mask=None):
        if softmax_dim not in (-1, query.dim() - 1):
            raise NotImplementedError("softmax over the key axis only")
        s = self.raw_scores(query, key)
        rv, cv = (None, None) if mask is None else self._split_mask(mask)
        p = ops.masked_softmax(s, cv, rv, outer=s.shape[0])
        return (s if mask is None else s.masked_fill(~mask, -float('inf'))), p

    def forward(self, query, key, value, mask=None):
        """query [B, T, Q], key [B, S, K], value [B, S, Hv], mask [B, T, S] bool -> (ctx, raw scores, p)."""
        if query.dim() != 3:
            raise NotImplementedError("3-D inputs only on the HIP path")
        rv, cv = (None, None) if mask is None else self._split_mask(mask)
        s = self.raw_scores(query, key)
        p = ops.masked_softmax(s, cv, rv, outer=s.shape[0])
        ctx = ops.bmm(ops.cast_to(p, value.dtype), value, b_is_kn=True)
        return ctx, (s if mask is None else s.masked_fill(~mask, -float('inf'))), p

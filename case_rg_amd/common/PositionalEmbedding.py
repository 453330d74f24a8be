"""Sinusoidal position injection (reference: common/PositionalEmbedding.py:22-48).

Inside the sequence encoder/decoder the scale-and-add is fused into the embedding gather kernel
(``ops.embed_pos``, K1); this module is the stand-alone form with the reference's constructor and
``pe`` buffer (state_dict key ``...embedding.1.pe``)."""
import math

import torch
import torch.nn as nn

from .. import config, ops


def sinusoid_table(max_len, width):
    """pe[p, 2i] = sin(p w_i), pe[p, 2i+1] = cos(p w_i), w_i = exp(-2i ln(1e4) / width)  (:27-32)."""
    pos = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
    w = torch.exp(torch.arange(0, width, 2, dtype=torch.float32) * (-math.log(10000.0) / width))
    pe = torch.zeros(max_len, width)
    pe[:, 0::2] = torch.sin(pos * w)
    pe[:, 1::2] = torch.cos(pos * w)
    return pe


class PositionalEmbedding(nn.Module):
    def __init__(self, embedding_size, dropout=0.1, max_len=5000):
        super().__init__()
        self.embedding_size = embedding_size
        self.p = dropout
        self.register_buffer('pe', sinusoid_table(max_len, embedding_size))

    def forward(self, x):
        """x [batch, *, L, E] -> x * sqrt(E) + pe[:L], then dropout (training only)."""
        if x.size(-2) > self.pe.size(0):
            raise RuntimeError("sequence length %d exceeds max_len %d" % (x.size(-2), self.pe.size(0)))
        y = ops.scale_add_rows(x, self.pe, math.sqrt(self.embedding_size))
        return ops.dropout(y, self.p, self.training)

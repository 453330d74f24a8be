"""Pieces shared by the CaSE and Masque task models: TransformerBlock stacks, the BCE / NLL losses (K12)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .TransformerBlock import TransformerBlock


def block_stack(num_heads, hidden_size, extra):
    """[TransformerBlock(5H -> H)] + ``extra`` x TransformerBlock(H -> H)  (CaSE/Model.py:137-138,177-178)."""
    return nn.ModuleList([TransformerBlock(num_heads, 5 * hidden_size, hidden_size)] +
                         [TransformerBlock(num_heads, hidden_size, hidden_size) for _ in range(extra)])


def run_blocks(blocks, reps, mask):
    for block in blocks:
        reps = block(reps, mask)
    return reps


def passage_bce(passage_score, passage_label):
    """BCE-with-logits against the one-hot of the gold passage (CaSE/Model.py:281-283).  [B, P] scalars: host-side glue."""
    target = torch.zeros_like(passage_score).scatter_(1, passage_label.unsqueeze(-1), 1.0)
    return F.binary_cross_entropy_with_logits(passage_score.float(), target.float()).unsqueeze(0)


def generation_nll(dist, response):
    """mean over non-pad targets of -log(dist[target] + 1e-8)  (CaSE/Model.py:306, Masque/Model.py:239).
    The gather and its sparse gradient are kernels (K12); the final mean over B*T numbers is glue."""
    V = dist.size(-1)
    rows = ops.nll_rows(dist.reshape(-1, V), response.reshape(-1))
    count = response.ne(0).sum().clamp(min=1)
    return (rows.sum() / count).unsqueeze(0)

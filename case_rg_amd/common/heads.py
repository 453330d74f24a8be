"""Pieces shared by the CaSE and Masque task models: TransformerBlock stacks, the BCE / NLL losses (K12)."""
import os

import torch
import torch.distributed as dist
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


# Training (round 6): the query-side block stack -- a dependent chain of ~70 forward / ~200 backward launches on 2048 rows that leaves the chip almost
# empty -- runs on a second stream beside the passage-side stack (the reference runs the two stacks one after the other, CaSE/Model.py:155-156,
# :196-197; they share no tensor).  An 8-wave GEMM workgroup holds its CU whole, so the side stream gets the tail rounds and the kernel boundaries
# of the passage side: that is enough for most of the chain.  The query ops are created FIRST: in the backward pass autograd then issues them LAST,
# behind the passage side's launches.  cfg 2 step 95.6 -> 94.8 ms (three alternating pairs, one box).  CASE_SIDE_STREAM=0 restores one stream.
SIDE_STREAM = os.environ.get("CASE_SIDE_STREAM", "1") != "0"
# (inference: measured on the greedy pass at B = 256 -- the encode phase gains 0.4 ms of 183, the 63 cached steps behind it LOSE 0.04 ms each
#  (2.5 ms): off unless asked for)
SIDE_STREAM_INFERENCE = os.environ.get("CASE_SIDE_STREAM_INFERENCE", "0") == "1"
SIDE_STREAM_NOT_UNDER_DP = os.environ.get("CASE_SIDE_STREAM_UNDER_DP", "0") != "1"
SIDE_STREAM_MIN_ELEMS = 100_000_000  # elements of the passage-side input [B, P, Lp, 5H]
_side = {}


def run_block_pair(query_blocks, g_pq, query_mask, passage_blocks, g_qp, passage_mask):
    """(query_reps, passage_reps) of two independent block stacks."""
    # (only where the passage side is GPU-bound: at the reference's default geometry -- 16 000 passage rows, a host-bound step -- the second
    #  stream's bookkeeping COSTS 2.3 ms of 19.4; cfg 2 / cfg 5 / Masque B 32 hand over 3.1e8 elements, Masque B 8 7.9e7 (neutral))
    # (not under data parallelism: GradSync keeps 8 CUs free for RCCL during the backward pass, the side stream's launches then start at once
    #  ON THOSE 8 CUs and crawl -- forced one-rank group: 96.8 ms with one stream, 99.3 with two)
    if not (SIDE_STREAM and g_pq.is_cuda and g_qp.numel() >= SIDE_STREAM_MIN_ELEMS and (torch.is_grad_enabled() or SIDE_STREAM_INFERENCE)
            and not torch.cuda.is_current_stream_capturing() and not (SIDE_STREAM_NOT_UNDER_DP and dist.is_available() and dist.is_initialized())):
        return run_blocks(query_blocks, g_pq, query_mask), run_blocks(passage_blocks, g_qp, passage_mask)
    cur = torch.cuda.current_stream()
    side = _side.get(g_pq.device)
    if side is None:
        side = _side[g_pq.device] = torch.cuda.Stream(device=g_pq.device)
    ops.AUX_STREAMS[side.cuda_stream] = side  # (whoever gathers gradients across nodes -- parallel.GradSync -- joins these first)
    ops.AUX_STREAMS[cur.cuda_stream] = cur
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        # tensors of the calling stream's pool that side-stream kernels read, in this pass and (saved) in the backward pass: the allocator must not
        # hand their blocks on while such a kernel is still queued
        g_pq.record_stream(side)
        if torch.is_tensor(query_mask):
            query_mask.record_stream(side)
        query_reps = run_blocks(query_blocks, g_pq, query_mask)
    passage_reps = run_blocks(passage_blocks, g_qp, passage_mask)
    cur.wait_stream(side)
    query_reps.record_stream(cur)
    return query_reps, passage_reps


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

"""Seeded synthetic query/passage batches with the reference's collate_fn schema.

Schema: CaSE/CaSEDataset.py:130-140 (Masque: Masque/MasqueDataset.py:134-144); shapes and the
ragged / filler-passage recipe: SURVEY 8(d).
"""
import numpy as np
import torch

from .fill import CLS, EOS, FIRST_WORD_ID, SEP


def _sequence(rng, length, width, vocab):
    row = np.zeros(width, dtype=np.int64)
    row[0] = CLS
    if length > 2:
        row[1:length - 1] = rng.randint(FIRST_WORD_ID, vocab, size=length - 2)
    row[length - 1] = SEP
    return row


def synth_batch(B, P, Lp, Lq, T, V, seed=123456, ragged=True, filler_passage=True, model="case"):
    """Returns the batch dict (CPU int64/float32 tensors).  ``ragged=False`` gives full-length
    sequences (roofline runs: padded-dense FLOPs == useful FLOPs)."""
    rng = np.random.RandomState(seed)
    query = np.zeros((B, 1, Lq), dtype=np.int64)
    passage = np.zeros((B, P, Lp), dtype=np.int64)
    response = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        lq = rng.randint(max(2, Lq // 2), Lq + 1) if ragged else Lq
        query[b, 0] = _sequence(rng, lq, Lq, V)
        for p in range(P):
            lp = rng.randint(max(2, Lp // 2), Lp + 1) if ragged else Lp
            passage[b, p] = _sequence(rng, lp, Lp, V)
        if ragged and filler_passage and P > 1:
            passage[b, rng.randint(0, P)] = _sequence(rng, 2, Lp, V)  # [CLS][SEP]+PAD, CaSEDataset.py:86-87
        lt = rng.randint(max(2, T // 2), T + 1) if ragged else T
        # half of the answer tokens are copied from the sources so the pointer path carries signal
        pool = np.concatenate([query[b].ravel(), passage[b].ravel()])
        pool = pool[pool >= FIRST_WORD_ID]
        ans = rng.randint(FIRST_WORD_ID, V, size=lt - 1)
        if len(pool):
            take = rng.rand(lt - 1) < 0.5
            ans[take] = pool[rng.randint(0, len(pool), size=int(take.sum()))]
        response[b, :lt - 1] = ans
        response[b, lt - 1] = EOS
    if ragged:  # pad_sequence trims the batch to its longest answer (CaSEDataset.py:135)
        response = response[:, :max(1, int((response != 0).sum(1).max()))]
    batch = {
        "id": torch.arange(B, dtype=torch.long),
        "query": torch.from_numpy(query),
        "passage": torch.from_numpy(passage),
        "response": torch.from_numpy(response),
        "passage_label": torch.from_numpy(rng.randint(0, P, size=B).astype(np.int64)),
        "source_map": torch.from_numpy(np.concatenate([query.reshape(B, -1), passage.reshape(B, -1)], axis=1)),
    }
    if model == "case":
        valid = passage != 0
        batch["token_label"] = torch.from_numpy(((rng.rand(B, P, Lp) < 0.1) & valid).astype(np.float32))
        batch["token_weight"] = torch.ones(B, P, Lp, dtype=torch.float32)
    return batch


def copy_task_batch(B, P, Lp, Lq, T, V, seed):
    """A learnable synthetic task with the same schema ("a synthetic copy task" for the ROUGE acceptance): the answer
    repeats the first T - 1 words of the query, then EOS (Lq >= T + 1); the labelled passage is the one that opens with the
    query's first word and its first T - 1 tokens carry the token labels.  A pointer-generator learns within a few hundred Adam
    steps to walk the query memory position by position; an untrained one scores near zero ROUGE-L."""
    rng = np.random.RandomState(seed)
    query = np.zeros((B, 1, Lq), dtype=np.int64)
    passage = np.zeros((B, P, Lp), dtype=np.int64)
    response = np.zeros((B, T), dtype=np.int64)
    label = np.zeros(B, dtype=np.int64)
    token_label = np.zeros((B, P, Lp), dtype=np.float32)
    span = T - 1
    for b in range(B):
        query[b, 0] = _sequence(rng, Lq, Lq, V)
        for p in range(P):
            passage[b, p] = _sequence(rng, Lp, Lp, V)
        p = label[b] = rng.randint(0, P)
        passage[b, p, 1] = query[b, 0, 1]
        response[b, :span] = query[b, 0, 1:1 + span]
        response[b, span] = EOS
        token_label[b, p, 1:1 + span] = 1.0
    return {
        "id": torch.arange(B, dtype=torch.long), "query": torch.from_numpy(query), "passage": torch.from_numpy(passage),
        "response": torch.from_numpy(response), "passage_label": torch.from_numpy(label),
        "source_map": torch.from_numpy(np.concatenate([query.reshape(B, -1), passage.reshape(B, -1)], axis=1)),
        "token_label": torch.from_numpy(token_label), "token_weight": torch.ones(B, P, Lp, dtype=torch.float32),
    }

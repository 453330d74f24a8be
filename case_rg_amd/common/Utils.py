"""Model-side helpers of the hot path (the on-path subset of the reference's common/Utils.py).

Tokenizers, GloVe loaders, GRU helpers and beam utilities of the reference file are data-prep /
baseline-model code and are out of scope."""
import random

import numpy as np
import torch

from .. import ops
from .Constants import BOS_WORD, EOS_WORD, PAD_WORD, UNK_WORD

NEAR_INF = 1e20
NEAR_INF_FP16 = 65504


def neginf(dtype):
    """A representable finite number near -inf (reference: common/Utils.py:16-21)."""
    return -NEAR_INF_FP16 if dtype is torch.float16 else -NEAR_INF


def _device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def generate_square_subsequent_mask(sz):
    """[sz, sz] float mask, 0 on/below the diagonal and -1e20 above (reference :23-28).  The tensor is
    tagged so the attention modules run their causal kernel path without inspecting it."""
    allowed = torch.tril(torch.ones(sz, sz, dtype=torch.bool, device=_device()))
    mask = torch.zeros(sz, sz, device=_device()).masked_fill(~allowed, neginf(torch.float32))
    mask._case_causal = True
    return mask


def init_seed(seed=None):
    """Seeds numpy / torch / random and the dropout counter RNG of the HIP path (reference :57-65)."""
    import time
    from .. import config
    if seed is None:
        seed = int(time.time())
    np.random.seed(seed)
    torch.manual_seed(seed)
    random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    config.manual_seed(seed)


def new_tensor(array, requires_grad=False):
    return torch.tensor(array, device=_device(), requires_grad=requires_grad)


def build_map(b_map, max=None):
    """Dense one-hot copy map [B, S, V] (reference :344-355).  Kept for API compatibility only: the models
    pass the source *ids* to the pointer scatter kernel (K11) and never build this tensor."""
    B, S = b_map.shape
    if max is None:
        max = int(b_map.max()) + 1
    out = torch.zeros(B, S, max, device=b_map.device)
    out.scatter_(2, b_map.unsqueeze(2), 1.0)
    return out


def universal_sentence_embedding(sentences, mask, sqrt=False):
    """Masked mean over the sequence axis (reference :455-470); ``sqrt=True`` divides the masked sum by sqrt(count) instead
    (= mean * sqrt(count); not used on the CaSE / Masque path)."""
    mean = ops.masked_mean(sentences, mask)
    if sqrt:
        return mean * mask.sum(dim=1).view(-1, 1).to(mean.dtype).sqrt()
    return mean


def topk(gen_output, k=1, PAD=None, BOS=None, UNK=None):
    """(values, indices) of the k best columns per row, keepdim (reference :156-168).  k = 1 -- the greedy pick of the CaSE / Masque
    loop -- is the HIP row-argmax (lowest index on ties, as ``torch.max``); k > 1 is off the hot path and sorts with
    ``torch.topk`` as the reference does.  Unlike the reference the masked ids are zeroed in a copy, not in the caller's tensor."""
    if PAD is not None or BOS is not None or UNK is not None:
        gen_output = gen_output.clone()
        for tok in (PAD, BOS, UNK):
            if tok is not None:
                gen_output[:, tok] = 0
    if k > 1:
        return torch.topk(gen_output, k, dim=1, largest=True, sorted=True)
    idx, val = ops.row_argmax(gen_output)
    return val.unsqueeze(1), idx.unsqueeze(1)


_special_ids = (None, None)  # (vocabulary object, its special ids): the most recent vocabulary only, compared by identity


def _specials(id2vocab):
    """ids of BOS / PAD / EOS in ``id2vocab`` (-1 when absent).  Only the most recent vocabulary is remembered, together with
    the object itself: an ``id()`` key alone would serve stale ids once a freed vocabulary's address is reused by another one."""
    global _special_ids
    if _special_ids[0] is not id2vocab:
        items = id2vocab.items() if hasattr(id2vocab, "items") else enumerate(id2vocab)
        inv = {w: i for i, w in items if w in (BOS_WORD, PAD_WORD, EOS_WORD)}
        _special_ids = (id2vocab, tuple(inv.get(w, -1) for w in (BOS_WORD, PAD_WORD, EOS_WORD)))
    return _special_ids[1]


def to_sentence(batch_indices, id2vocab):
    """ids -> token lists: drop BOS / PAD, stop at EOS, empty -> [UNK] (reference :200-217).  Ids on the GPU are filtered and
    front-packed by one kernel (K13 post-processing) and come back with ONE device->host copy for the whole batch, instead of
    the reference's ``.item()`` per generated token; host ids take the plain loop."""
    if torch.is_tensor(batch_indices) and batch_indices.is_cuda and batch_indices.dim() == 2:
        bos, pad, eos = _specials(id2vocab)
        kept, count = ops.sentence_compact(batch_indices.long(), bos, pad, eos)
        rows, lens = kept.tolist(), count.tolist()
        return [[id2vocab[i] for i in row[:n]] if n else [UNK_WORD] for row, n in zip(rows, lens)]
    rows = batch_indices.tolist() if torch.is_tensor(batch_indices) else batch_indices
    out = []
    for row in rows:
        words = []
        for index in row:
            w = id2vocab[int(index)]
            if w == BOS_WORD or w == PAD_WORD:
                continue
            if w == EOS_WORD:
                break
            words.append(w)
        out.append(words if words else [UNK_WORD])
    return out


def remove_duplicate_once(sents, n=3):
    """One pass of trailing-n-gram de-duplication (reference :170-186)."""
    changed = False
    for b, sent in enumerate(sents):
        if len(sent) <= n:
            continue
        for i in range(len(sent) - n):
            cut = len(sent) - i - n
            if all(tok in sent[:cut] for tok in sent[cut:]):
                sents[b] = sent[:cut]
                changed = True
                break
    return changed


def remove_duplicate(sents, n=3):
    while remove_duplicate_once(sents, n):
        pass

"""Deterministic, name-keyed parameter filler and synthetic vocabulary.

Weights are never stored in fixtures: this filler regenerates them identically for the reference
(in the golden generator), the oracle and the product modules, because all three expose the same
``state_dict`` keys.  Shared tensors (the encoder is reachable under 16 prefixes
in CaSE) are keyed by their lexicographically smallest alias, so the result does not depend on module
registration order.
"""
import math
import zlib

import numpy as np
import torch

# ids mirror bert-base-uncased for the specials the reference names (common/Constants.py:1-7)
SPECIAL_IDS = {"[PAD]": 0, "[unused0]": 1, "[unused1]": 2, "[UNK]": 100, "[CLS]": 101, "[SEP]": 102, "[MASK]": 103}
PAD, BOS, EOS, UNK, CLS, SEP, MASK = 0, 1, 2, 100, 101, 102, 103
FIRST_WORD_ID = 104


def make_vocab(size):
    """Synthetic vocab2id / id2vocab of ``size`` entries with the reference's special tokens."""
    assert size > FIRST_WORD_ID, "vocabulary must hold the special ids (>104)"
    id2vocab = {i: "tok%d" % i for i in range(size)}
    for w, i in SPECIAL_IDS.items():
        id2vocab[i] = w
    vocab2id = {w: i for i, w in id2vocab.items()}
    return vocab2id, id2vocab


def _canonical_names(module):
    groups = {}
    for name, t in module.state_dict(keep_vars=True).items():
        groups.setdefault(t.data_ptr() if t.numel() else id(t), []).append((name, t))
    for aliases in groups.values():
        yield min(n for n, _ in aliases), aliases[0][1]


@torch.no_grad()
def fill_params(module, seed=0, gain=1.0):
    """Fill every parameter of ``module`` from a counter RNG keyed by (seed, canonical name).

    matrices: U(-a, a), a = gain*sqrt(6/(fan_in+fan_out)) (xavier-uniform range, as the reference's
    init_params, common/CumulativeTrainer.py:13-24); LayerNorm-like 1-D ``weight``: 1 + U(-.1, .1);
    1-D ``bias``: U(-.1, .1).  Buffers (the sinusoid tables) are left alone."""
    params = {id(p) for p in module.parameters()}
    for name, t in _canonical_names(module):
        if id(t) not in params:
            continue
        rng = np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
        if t.dim() > 1:
            fan_out, fan_in = t.shape[0], int(np.prod(t.shape[1:]))
            a = gain * math.sqrt(6.0 / (fan_in + fan_out))
            v = rng.uniform(-a, a, size=tuple(t.shape))
        elif name.endswith("weight"):
            v = 1.0 + rng.uniform(-0.1, 0.1, size=tuple(t.shape))
        else:
            v = rng.uniform(-0.1, 0.1, size=tuple(t.shape))
        t.copy_(torch.from_numpy(v.astype(np.float32)))
    return module

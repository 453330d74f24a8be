"""Run-time switches of the MI355X path.

compute dtype
    torch.float32  -- parity mode: f32 activations, exact-f32 MFMA (v_mfma_f32_32x32x2_f32); this is what the
                      1e-3 parity tests run.
    torch.bfloat16 -- throughput mode: bf16 activations / bf16 MFMA with f32 accumulation, f32 statistics,
                      f32 losses and f32 parameter gradients; what bench.py measures.
dropout
    The reference applies dropout at >= 8 sites in ``.train()``.  The product implements them with a counter
    RNG keyed by (seed, running offset, element index): the backward pass regenerates the mask instead of
    storing it, and DP replicas with the same seed/step agree.  Parity tests switch dropout off
    (``set_dropout(False)``) because torch's RNG stream cannot be reproduced; bench.py keeps it on.
"""
import torch

_state = {"dtype": torch.float32, "dropout": True, "seed": 123456, "offset": 0}


def set_compute_dtype(dtype):
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
    _state["dtype"] = dtype


def compute_dtype():
    return _state["dtype"]


def set_dropout(enabled):
    _state["dropout"] = bool(enabled)


def dropout_enabled():
    return _state["dropout"]


def manual_seed(seed):
    _state["seed"] = int(seed) & 0x7FFFFFFFFFFFFFFF
    _state["offset"] = 0


def rng_state():
    """(seed, offset) of the dropout counter stream -- what a resumable checkpoint stores."""
    return _state["seed"], _state["offset"]


def set_rng_state(state):
    _state["seed"], _state["offset"] = int(state[0]), int(state[1])


def next_rng(numel):
    """Reserve ``numel`` counters; returns (seed, offset) for one dropout site."""
    off = _state["offset"]
    _state["offset"] = off + int(numel) + (int(numel) & 1)  # keep offsets even: the kernels hash element pairs
    return _state["seed"], off


def drop_p(p, training):
    """Effective dropout probability of a site."""
    return float(p) if (training and _state["dropout"] and p > 0.0) else 0.0

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

_state = {"dtype": torch.float32, "dropout": True, "seed": 123456, "offset": 0, "base": 0, "device_state": None}


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
    _state["offset"], _state["base"] = 0, 0


def rng_state():
    """(seed, position) of the dropout counter stream -- what a resumable checkpoint stores.  The position is ``base + offset``: with a
    device state the sites of a step are numbered from ``offset`` 0 and ``base`` (the sum of the earlier steps' consumption) is what the
    kernels read from the device; without one ``base`` stays 0."""
    return _state["seed"], _state["base"] + _state["offset"]


def set_rng_state(state):
    _state["seed"] = int(state[0])
    if _state["device_state"] is None:
        _state["offset"], _state["base"] = int(state[1]), 0
    else:
        _state["offset"], _state["base"] = 0, int(state[1])


def begin_step():
    """Device-state mode, once per training step before any dropout site is numbered: fold the previous step's consumption into the
    base and restart the site offsets at 0; returns the base to upload (CaseStepState.rng_base)."""
    _state["base"] += _state["offset"]
    _state["offset"] = 0
    return _state["base"]


def skip_rng(numel):
    """Account for ``numel`` counters consumed without passing through ``next_rng`` (a replayed hipGraph of a captured step)."""
    _state["offset"] += int(numel)


def next_rng(numel):
    """Reserve ``numel`` counters; returns (seed, offset, state) for one dropout site.  ``state`` is None, or the address of the
    CaseStepState (device memory, ABI 600) whose ``rng_base`` every kernel of the site adds to ``offset`` when it RUNS: with a device
    state the sites of one step are numbered from 0 (``begin_step``) and the step's base is data on the device, so a hipGraph-captured
    step draws new masks on every replay (``case_rg_amd.stepstate``)."""
    off = _state["offset"]
    _state["offset"] = off + int(numel) + (int(numel) & 1)  # keep offsets even: the kernels hash element pairs
    return _state["seed"], off, _state["device_state"]


def set_device_state(address):
    """Address of the CaseStepState the dropout sites read their per-step base from (None: arguments only, the round-5 behaviour).
    The stream position is kept: switching modes moves it between ``offset`` and ``base``."""
    pos = _state["base"] + _state["offset"]
    _state["device_state"] = None if not address else int(address)
    if _state["device_state"] is None:
        _state["offset"], _state["base"] = pos, 0
    else:
        _state["offset"], _state["base"] = 0, pos


def device_state():
    return _state["device_state"]


def drop_p(p, training):
    """Effective dropout probability of a site."""
    return float(p) if (training and _state["dropout"] and p > 0.0) else 0.0
